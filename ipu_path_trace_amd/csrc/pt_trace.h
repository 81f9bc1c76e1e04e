// pt_trace.h -- the per-pixel sampling loop as a persistent-threads HIP kernel.
//
// Replaces the reference's GenerateCameraRays / RayTraceKernel / PreProcessEscapedRays codelets
// (src/codelets/codelets.cpp:36-80, :93-227, :312-358) and the poprand noise tensors
// (src/PathTracerApp.cpp:266-299).  One lane carries one path in registers and escaped paths are
// appended to the workgroup's region of the NIF queue (ballot + one LDS atomic per wave), so the
// queue the MFMA kernel reads is dense and no ray state ever goes to HBM.
//
// Round 3: three phases per workgroup.  60 % of the camera rays hit nothing and end after ONE
// intersection test; in the round-2 kernel every loop trip therefore re-generated more than half
// of the wave's lanes -- noise, camera ray, normalisation: the costliest stretch of the kernel --
// with only those lanes active, and bounced the rest beside them (55 % lane utilisation).  Now a
// workgroup first runs ALL its camera rays, 64 new paths per wave and trip, every lane active
// (primary phase: ray generation + first intersection; a miss is final and goes to the queue, a
// hit leaves a 16-byte note -- path index, the half-rounded camera ray, hit distance and object --
// in the workgroup's survivor list: diffuse hits from the front of the list, mirror and glass
// hits from its back), then shades every survivor's first hit, again 64 per wave and trip and,
// thanks to the two-ended list, one material at a time (first-shading phase: Philox block, BSDF,
// 40 bytes of path state into a second list), and only then bounces what is left in the
// persistent loop (secondary phase), where a lane whose path ends is refilled from the state list
// for the price of three loads, so the wave stays full.  Both lists are written and read inside
// the launch and stay in L2.  Same device functions in the same order on the same values:
// bit-identical paths (tests/test_gpu_parity.py).
#pragma once
#include "pt_device_math.h"

#include <utility>

namespace ptd {

enum { MAT_DIFFUSE = 0, MAT_SPECULAR = 1, MAT_REFRACTIVE = 2 };
constexpr int kNumObjects = 6;
constexpr float kEps = 1e-5f;           // INFERRED intersection epsilon (< 1e-4 clear-coat gap)
constexpr float kInf = 3.402823466e+38f;
constexpr float kPi = 3.1415927410125732421875f;

// Scene of src/codelets/codelets.cpp:111-144 as kernel-argument data (wave-uniform -> SGPRs).
struct SceneObject {
  float cx, cy, cz;     // centre
  float radius, r2;     // r2 = radius*radius
  float nx, ny, nz;     // disc normal
  float colr, colg, colb;
  int32_t type;         // MAT_*
  int32_t is_disc;
  // Camera rays start at the origin (codelets.cpp:162): what Sphere / Disc::intersect compute from (origin, object) alone
  // is a constant of the object, formed on the host by the SAME binary32 expressions in the same order (fill_scene):
  float ocx, ocy, ocz;  // sub(o, c) at o = 0
  float c4;             // 4.0f * (dot(oc, oc) - r2)
  float kdisc;          // dot(sub(c, o), n) at o = 0 (disc)
  int32_t same_centre;  // a sphere whose centre equals that of the sphere declared just before it
};

struct TraceParams {
  SceneObject obj[kNumObjects];
  float width_f, height_f;   // image size as float (pixelToRay)
  uint32_t width, height;    // ... and as integers: an item with u >= width or v >= height is worklist padding (not traced)
  float tx, ty;              // tan(fov/2), (h/w) tan(fov/2)
  float aa_scale;            // half-rounded
  float stop_prob;           // half-rounded
  float rr_factor;           // 1 / (1 - stop_prob)
  float ri;                  // half-rounded refractive index
  float azimuth;
  uint32_t seed_lo, seed_hi;
  uint32_t sample_base;      // absolute index of iteration 0 of this batch
  uint32_t max_path_length, roulette_depth;
  int32_t aa_type, samples_half;
  uint32_t n_items, total_paths;  // total_paths = n_items * iterations in this batch
  uint32_t n_waves;          // waves in the grid (chunk stride)
  uint32_t region_cap;       // queue slots owned by one workgroup
  int32_t env_const;
  float env_r, env_g, env_b;
  const uint32_t* pix;       // [n_items] u | v<<16
  float* q_u; float* q_v; float* q_tr; float* q_tg; float* q_tb; uint32_t* q_path;
  uint32_t* region_count;    // [gridDim.x]
  uint8_t* plen;             // [total_paths] length | escaped<<7
  float* rad_r; float* rad_g; float* rad_b;  // [total_paths] (constant-env mode writes here)
  uint4* survivors;          // [gridDim.x][region_cap] primary-phase notes: path index, camera ray (two halves), hit distance, object
  float4* states;            // [3][gridDim.x][region_cap] path states after the first shading: (o, d.x), (d.yz, T.xy), (T.z, idx)
  size_t state_stride;       // gridDim.x * region_cap
  uint32_t div_magic, div_shift;   // idx / n_items = (idx * div_magic) >> div_shift, exact for idx < 2^31 (ptmi_context.h: item_divider)
  unsigned long long* diag;        // profiling build: secondary-phase occupancy counters (OPT bit 4), else nullptr
};

// path index -> (work item, sample iteration).  A 32-bit division costs ~25 vector instructions on this chip and every path
// needs one (survivors two more); n_items is fixed per launch, so the host supplies the round-up reciprocal.
template <bool MAGIC>
__device__ __forceinline__ void split_index(const TraceParams& P, uint32_t idx, uint32_t& item, uint32_t& iter) {
  if constexpr (MAGIC) {
    iter = (uint32_t)(((uint64_t)idx * P.div_magic) >> P.div_shift);
    item = idx - iter * P.n_items;
  } else {
    iter = idx / P.n_items;
    item = idx % P.n_items;
  }
}

struct PathState {
  Vec3 o, d, T;
  uint32_t pixel;    // u | v<<16
  uint32_t sample;
  uint32_t depth;
};

__device__ __forceinline__ float uniform01(uint32_t bits, int half_grid) {
  return half_grid ? (float)(bits >> 21) * 4.8828125e-04f : (float)(bits >> 8) * 5.9604644775390625e-08f;
}

// poprand noise into a half tensor (PathTracerApp.cpp:29-45): Box-Muller on Philox block 0.
__device__ __forceinline__ void aa_noise(const TraceParams& P, uint32_t pixel, uint32_t sample, float& n0, float& n1) {
  uint32_t w[4];
  philox4x32_10(pixel, sample, 0u, 0x5054u, P.seed_lo, P.seed_hi, w);
  if (P.aa_type == 1) {
    float a = (float)(w[0] >> 8) * 5.9604644775390625e-08f;
    float b = (float)(w[1] >> 8) * 5.9604644775390625e-08f;
    n0 = hround(2.0f * a - 1.0f);
    n1 = hround(2.0f * b - 1.0f);
    return;
  }
  float r0 = 0.f, r1 = 0.f;
  for (int attempt = 0; attempt < 2; ++attempt) {
    float u1 = (float)((w[2 * attempt] >> 8) + 1u) * 5.9604644775390625e-08f;
    float u2 = (float)(w[2 * attempt + 1] >> 8) * 5.9604644775390625e-08f;
    float rad = sqrtf(-2.0f * dm_log(u1));
    float s, c;
    dm_sincos2pi(u2, s, c);
    r0 = rad * c;
    r1 = rad * s;
    if (P.aa_type == 2) {
      bool bad = (fabsf(r0) > 3.0f) || (fabsf(r1) > 3.0f);
      if (bad && attempt == 0) continue;
      r0 = fminf(fmaxf(r0, -3.0f), 3.0f);
      r1 = fminf(fmaxf(r1, -3.0f), 3.0f);
    }
    break;
  }
  n0 = hround(r0);
  n1 = hround(r1);
}

// GenerateCameraRays::compute (codelets.cpp:68-75) + Ray ctor (:162-163).
__device__ __forceinline__ void start_path(const TraceParams& P, uint32_t pixel, uint32_t sample, PathState& s,
                                           float& camx, float& camy) {
  float n0, n1;
  aa_noise(P, pixel, sample, n0, n1);
  float c = (float)(pixel & 0xffffu) + hround(P.aa_scale * n0);
  float r = (float)(pixel >> 16) + hround(P.aa_scale * n1);
  float px = ((2.0f * c - P.width_f) / P.width_f) * P.tx;
  float py = -(((2.0f * r - P.height_f) / P.height_f) * P.ty);
  camx = hround(px);
  camy = hround(py);
  s.o = mk(0.f, 0.f, 0.f);
  s.d = normalise(mk(camx, camy, -1.f));
  s.T = mk(1.f, 1.f, 1.f);
  s.pixel = pixel;
  s.sample = sample;
  s.depth = 0;
}

__device__ __forceinline__ float sphere_intersect(Vec3 o, Vec3 d, const SceneObject& ob) {
  Vec3 oc = sub(o, mk(ob.cx, ob.cy, ob.cz));
  float b = 2.0f * dot(oc, d);
  float c_ = dot(oc, oc) - ob.r2;
  float disc = b * b - 4.0f * c_;
  if (disc < 0.0f) return 0.0f;   // (a real branch on purpose: a wave none of whose rays comes near the sphere skips the rest)
  disc = sqrtf(disc);
  float sol1 = -b + disc;
  float sol2 = -b - disc;
  return (sol2 > kEps) ? sol2 * 0.5f : ((sol1 > kEps) ? sol1 * 0.5f : 0.0f);
}

__device__ __forceinline__ float disc_intersect(Vec3 o, Vec3 d, const SceneObject& ob) {
  Vec3 n = mk(ob.nx, ob.ny, ob.nz), c = mk(ob.cx, ob.cy, ob.cz);
  float denom = dot(n, d);
  if (denom == 0.0f) return 0.0f;
  float t = dot(sub(c, o), n) / denom;
  if (!(t > kEps)) return 0.0f;   // (real branches on purpose, as in sphere_intersect)
  Vec3 p = add(o, scale(d, t));
  Vec3 pc = sub(p, c);
  if (dot(pc, pc) > ob.r2) return 0.0f;
  return t;
}

// Scene::intersect (codelets.cpp:183): nearest hit in declaration order, -1 for none.
// The loop stays rolled: one object's constants at a time are fetched from the kernel-argument segment (scalar
// loads, wave-uniform), instead of all of them living in SGPRs.
// A sphere that shares its centre with the object declared before it (the clear-coat pair, codelets.cpp:115-116) reuses that
// object's sub(o, c), dot(oc, d) and dot(oc, oc): the same expressions on the same values (wave-uniform flag from the host).
template <bool PIPE = false>   // PIPE: the next object's constants are requested before this object's arithmetic, awaited after it
__device__ __forceinline__ int nearest_hit(const TraceParams& P, Vec3 o, Vec3 d, float& tbest) {
  int best = -1;
  tbest = kInf;
  float b = 0.f, oc2 = 0.f;
  SceneObject nxt = P.obj[0];
#pragma unroll 1
  for (int i = 0; i < kNumObjects; ++i) {
    SceneObject ob;
    if constexpr (PIPE) { ob = nxt; nxt = P.obj[i + 1 < kNumObjects ? i + 1 : kNumObjects - 1]; }
    else ob = P.obj[i];
    float t;
    if (ob.is_disc) {
      t = disc_intersect(o, d, ob);
    } else {
      if (!ob.same_centre) {
        const Vec3 oc = sub(o, mk(ob.cx, ob.cy, ob.cz));
        b = 2.0f * dot(oc, d);
        oc2 = dot(oc, oc);
      }
      const float c_ = oc2 - ob.r2;                       // sphere_intersect from here on
      float disc = b * b - 4.0f * c_;
      t = 0.0f;
      if (!(disc < 0.0f)) {   // (a real branch on purpose: a wave none of whose rays comes near the sphere skips the rest)
        disc = sqrtf(disc);
        const float sol1 = -b + disc, sol2 = -b - disc;
        t = (sol2 > kEps) ? sol2 * 0.5f : ((sol1 > kEps) ? sol1 * 0.5f : 0.0f);
      }
    }
    if (t > kEps && t < tbest) { tbest = t; best = i; }
  }
  return best;
}

// The same for a camera ray (origin exactly 0): sub(o, c), dot(oc, oc) - r2 and dot(sub(c, o), n) are the object's host-made
// constants, everything that depends on the direction is the expression of sphere_intersect / disc_intersect unchanged, so
// the result is the same float.  (Disc: the hit point add(o, scale(d, t)) is scale(d, t) but for the sign of a zero, which
// sub(p, c) and the squares that follow erase.)
template <bool PIPE = false>
__device__ __forceinline__ int nearest_hit_primary(const TraceParams& P, Vec3 d, float& tbest) {
  int best = -1;
  tbest = kInf;
  float b = 0.f;
  SceneObject nxt = P.obj[0];
#pragma unroll 1
  for (int i = 0; i < kNumObjects; ++i) {
    SceneObject ob;
    if constexpr (PIPE) { ob = nxt; nxt = P.obj[i + 1 < kNumObjects ? i + 1 : kNumObjects - 1]; }
    else ob = P.obj[i];
    float t;
    if (ob.is_disc) {
      const Vec3 n = mk(ob.nx, ob.ny, ob.nz), c = mk(ob.cx, ob.cy, ob.cz);
      const float denom = dot(n, d);
      t = 0.0f;
      if (denom != 0.0f) {
        const float tt = ob.kdisc / denom;
        if (tt > kEps) {
          const Vec3 pc = sub(scale(d, tt), c);
          if (!(dot(pc, pc) > ob.r2)) t = tt;
        }
      }
    } else {
      if (!ob.same_centre) b = 2.0f * dot(mk(ob.ocx, ob.ocy, ob.ocz), d);
      float disc = b * b - ob.c4;
      t = 0.0f;
      if (!(disc < 0.0f)) {
        disc = sqrtf(disc);
        const float sol1 = -b + disc, sol2 = -b - disc;
        t = (sol2 > kEps) ? sol2 * 0.5f : ((sol1 > kEps) ? sol1 * 0.5f : 0.0f);
      }
    }
    if (t > kEps && t < tbest) { tbest = t; best = i; }
  }
  return best;
}


// ---- The scene (src/codelets/codelets.cpp:111-144), ONE table for host and device: fill_scene (ptmi_context.h) copies it into the
// kernel arguments, which is where the kernels read it (a rolled loop, one object's constants at a time in SGPRs).  Round 4
// tried the object loop UNROLLED over this table -- every constant a literal, no scalar load, no s_waitcnt (39 % of the
// kernel's wave-cycles sit in s_waitcnt) -- and lost: 90 VGPRs instead of 60 (five waves per SIMD, and no room beside the NIF
// kernel's waves: the C2 step 2 % slower), 9.36 against 8.75 ms on its own (diag/pt_trace_scene_c.h, profiles/r04_trace_ablation.txt).
struct SceneConst {
  bool disc;
  float cx, cy, cz, radius;
  float nx, ny, nz;
  float colr, colg, colb;
  int type;
};
constexpr float kColourGain = 2.f;                                                                        // :127
__host__ __device__ constexpr SceneConst scene_const(int i) {
  constexpr SceneConst table[kNumObjects] = {
      {false, -1.8575f, -0.98714f, -3.6f, 0.6f, 0.f, 0.f, 0.f, 1.f * kColourGain, .89f * kColourGain, .55f * kColourGain, MAT_DIFFUSE},      // :112,:128,:137
      {false, 0.74795f, -0.55f, -4.3816f, 1.05f, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, MAT_SPECULAR},                                              // :113,:138
      {false, 1.9929f, -1.08666f, (float)-3.23, 0.5f, 0.f, 0.f, 0.f, 0.75f, 0.75f, 0.75f, MAT_REFRACTIVE},                                 // :114,:131,:139
      {false, (float)-0.19931, -1.183f, -2.75f, 0.4f, 0.f, 0.f, 0.f, .8f * kColourGain, .06f * kColourGain, .391f * kColourGain, MAT_DIFFUSE},   // :115,:129,:140
      {false, (float)-0.19931, -1.183f, -2.75f, 0.4001f, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, MAT_REFRACTIVE},                                    // :116,:141
      {true, 0.f, -1.6f, -5.22f, 3.5f, 0.f, 1.f, 0.f, .98f * kColourGain, .76f * kColourGain, .66f * kColourGain, MAT_DIFFUSE},             // :121,:130,:143
  };
  return table[i];
}
__host__ __device__ constexpr bool scene_same_centre(int i) {
  return i > 0 && !scene_const(i).disc && !scene_const(i - 1).disc && scene_const(i).cx == scene_const(i - 1).cx &&
         scene_const(i).cy == scene_const(i - 1).cy && scene_const(i).cz == scene_const(i - 1).cz;
}

#ifdef PTMI_DIAG_BUILD
}  // namespace ptd
#include "diag/pt_trace_scene_c.h"   // round-4 experiment: the object loop unrolled over the compile-time scene (not kept)
namespace ptd {
#endif

enum StepResult { STEP_CONTINUE = 0, STEP_ESCAPED = 1, STEP_DEAD = 2 };

// Per-object data a lane needs once it knows WHICH object it hit, in LDS so that the lane fetches its own object's
// row by index.  (The scene also sits in the kernel arguments; holding all 6 x 13 constants in SGPRs across the
// bounce loop for a select chain overflowed the SGPR file: 600 of the kernel's 2600 instructions were
// v_writelane/v_readlane spill traffic.)
struct HitRow {
  float4 centre;   // cx, cy, cz, -
  float4 normal;   // nx, ny, nz (disc), -
  float4 colour;   // r, g, b, bits of (type | is_disc << 8)
};
__device__ __forceinline__ void fill_hit_table(const TraceParams& P, HitRow* tab) {
#pragma unroll
  for (int i = 0; i < kNumObjects; ++i) {
    tab[i].centre = make_float4(P.obj[i].cx, P.obj[i].cy, P.obj[i].cz, 0.f);
    tab[i].normal = make_float4(P.obj[i].nx, P.obj[i].ny, P.obj[i].nz, 0.f);
    tab[i].colour = make_float4(P.obj[i].colr, P.obj[i].colg, P.obj[i].colb,
                                __uint_as_float((uint32_t)P.obj[i].type | ((uint32_t)P.obj[i].is_disc << 8)));
  }
}

__device__ __forceinline__ int shade_hit(const TraceParams& P, const HitRow* tab, PathState& s, int best, float tbest,
                                         const uint32_t (&w)[4], float rr, uint32_t& length);

// One iteration of the while loop of RayTraceKernel::compute (codelets.cpp:173-216) with the
// AccumulateContributions fold (codelets.cpp:255-292) carried forward as throughput T.
// Returns the path length (contribution-stack size, codelets.cpp:253) through `length` when the
// path ends.
template <bool LEGACY = false, bool SCENE_C = false, bool PIPE = false>
__device__ __forceinline__ int bounce(const TraceParams& P, const HitRow* tab, PathState& s, uint32_t& length);
#ifdef PTMI_DIAG_BUILD
__device__ __forceinline__ int nearest_hit_r3(const TraceParams& P, Vec3 o, Vec3 d, float& tbest);
__device__ __forceinline__ int shade_hit_r3(const TraceParams& P, const HitRow* tab, PathState& s, int best, float tbest,
                                            const uint32_t (&w)[4], float rr, uint32_t& length);
#endif
template <bool LEGACY, bool SCENE_C, bool PIPE>
__device__ __forceinline__ int bounce(const TraceParams& P, const HitRow* tab, PathState& s, uint32_t& length) {
  uint32_t w[4];
  philox4x32_10(s.pixel, s.sample, 1u + s.depth, 0x5054u, P.seed_lo, P.seed_hi, w);
  float rr = 1.0f;
  if (s.depth >= P.roulette_depth) {                          // :176-180
    float u = uniform01(w[0], P.samples_half);
    if (u <= P.stop_prob) {
      length = s.depth ? s.depth : 1u;                        // END overwrites the last record (:219-222)
      return STEP_DEAD;
    }
    rr = P.rr_factor;
  }
  float tbest;
  int best;
#ifdef PTMI_DIAG_BUILD
  if constexpr (LEGACY) best = nearest_hit_r3(P, s.o, s.d, tbest);
  else
#endif
#ifdef PTMI_DIAG_BUILD
  if constexpr (SCENE_C) best = nearest_hit_c(s.o, s.d, tbest, SceneIndices{});
  else
#endif
  best = nearest_hit<PIPE>(P, s.o, s.d, tbest);               // Scene::intersect (:183)
  if (best < 0) {                                             // :184-190 ESCAPED
    s.T = scale(s.T, rr);
    length = s.depth + 1u;
    return STEP_ESCAPED;
  }
#ifdef PTMI_DIAG_BUILD
  if constexpr (LEGACY) return shade_hit_r3(P, tab, s, best, tbest, w, rr, length);
#endif
  return shade_hit(P, tab, s, best, tbest, w, rr, length);
}

// The second half of a loop trip of RayTraceKernel::compute (codelets.cpp:192-216): the ray has hit object `best` at
// distance `tbest`; w = the bounce's Philox block, rr = its roulette weight.
__device__ __forceinline__ int shade_hit(const TraceParams& P, const HitRow* tab, PathState& s, int best, float tbest,
                                         const uint32_t (&w)[4], float rr, uint32_t& length) {
  // the hit object's row, by per-lane index
  const float4 hc = tab[best].centre, hn = tab[best].normal, hcol = tab[best].colour;
  const float cx = hc.x, cy = hc.y, cz = hc.z, nx = hn.x, ny = hn.y, nz = hn.z, cr = hcol.x, cg = hcol.y, cb = hcol.z;
  const uint32_t bits = __float_as_uint(hcol.w);
  const int type = (int)(bits & 0xffu), is_disc = (int)(bits >> 8);
  Vec3 hp = add(s.o, scale(s.d, tbest));
  s.o = hp;
  Vec3 n = is_disc ? mk(nx, ny, nz) : normalise(sub(hp, mk(cx, cy, cz)));
  if (type == MAT_DIFFUSE) {                                  // :199-204, light::diffuse
    float u1 = uniform01(w[1], P.samples_half);
    float u2 = uniform01(w[2], P.samples_half);
    Vec3 rx, ry;
    {   // the branch of light::diffuse's basis as selects: one square root and one division per lane, not two of each per wave
      const bool xmajor = fabsf(n.x) > fabsf(n.y);
      const float m = xmajor ? n.x : n.y;
      const float inv = 1.0f / sqrtf(m * m + n.z * n.z);
      const float a = n.z * inv, b = m * inv;
      rx = xmajor ? mk(-a, 0.0f, b) : mk(0.0f, a, -b);
    }
    ry = cross(n, rx);
    float r = sqrtf(1.0f - u1 * u1);
    float sn, cs;
    dm_sincos2pi(u2, sn, cs);
    Vec3 h = mk(cs * r, sn * r, u1);
    s.d = mk(dot(mk(rx.x, ry.x, n.x), h), dot(mk(rx.y, ry.y, n.y), h), dot(mk(rx.z, ry.z, n.z), h));
    float cost = dot(s.d, n);
    s.T = scale(cwise(s.T, mk(cr, cg, cb)), cost * rr);
  } else {
    // Mirror (:205-207, light::reflect) and glass (:208-213, light::refract) each choose a vector, then share ONE
    // normalisation and ONE throughput update -- a wave that holds lanes of both would otherwise run the square root and the
    // division twice.  Per lane the expressions are those of the two branches: cwise(T, (1, 1, 1)) is T exactly.
    Vec3 v, tint = mk(1.f, 1.f, 1.f);
    float wgt = rr;
    if (type == MAT_SPECULAR) {
      float cost = dot(s.d, n);
      v = sub(s.d, scale(n, cost * 2.0f));
    } else {
      float u = uniform01(w[1], P.samples_half);
      float nn = P.ri;
      float r0 = (1.0f - nn) / (1.0f + nn);
      r0 = r0 * r0;
      if (dot(n, s.d) > 0.0f) { n = scale(n, -1.0f); nn = 1.0f / nn; }
      nn = 1.0f / nn;
      float cost1 = -dot(n, s.d);
      float cost2 = 1.0f - nn * nn * (1.0f - cost1 * cost1);
      float m = 1.0f - cost1;
      float m2 = m * m;
      float rprob = r0 + (1.0f - r0) * (m2 * m2 * m);
      bool refracted = (cost2 > 0.0f && u > rprob);
      // the vector is chosen first (sqrtf of a negative cost2 only feeds the side not taken)
      const Vec3 bent = add(scale(s.d, nn), scale(n, nn * cost1 - sqrtf(cost2)));
      const Vec3 mirrored = add(s.d, scale(n, cost1 * 2.0f));
      v = refracted ? bent : mirrored;
      if (refracted) tint = mk(cr, cg, cb);
      wgt = 1.15f * rr;
    }
    s.d = normalise(v);
    s.T = scale(cwise(s.T, tint), wgt);
  }
  s.depth += 1u;                                              // :215
  if (s.depth >= P.max_path_length) {                         // stack full without an emitter (:173,:219-222)
    length = s.depth;
    return STEP_DEAD;
  }
  return STEP_CONTINUE;
}

// PreProcessEscapedRays::compute (codelets.cpp:333-347).
__device__ __forceinline__ void dir_to_uv(Vec3 d, float azimuth, float& u, float& v) {
  float theta = dm_acos(d.y);
  float phi = dm_atan2(d.z, d.x) + azimuth;
  const float twoPi = 2.f * kPi;
  const float invPi = 1.f / kPi;
  const float inv2Pi = 1.f / twoPi;
  if (phi < 0.f) phi += twoPi;
  else if (phi > twoPi) phi -= twoPi;
  u = theta * invPi;
  v = phi * inv2Pi;
}

constexpr int kTraceBlock = 256;
constexpr uint32_t kRefillThreshold = 16;  // secondary phase: refill once this many lanes are idle (or none is active); 1..24 measured within 1 %

// An escaped path: constant environment -> radiance straight into the per-path result; NIF -> one entry in the
// workgroup's region of the queue (wave ballot + prefix count, ONE LDS atomic per wave, uv of PreProcessEscapedRays).
__device__ __forceinline__ void emit_escaped(const TraceParams& P, bool escaped, const PathState& st, uint32_t idx, uint32_t lane,
                                             uint32_t region_base, uint32_t* wg_count) {
  if (P.env_const) {
    if (escaped) {  // constant environment: total = env (.) T, no NIF
      P.rad_r[idx] = P.env_r * st.T.x;
      P.rad_g[idx] = P.env_g * st.T.y;
      P.rad_b[idx] = P.env_b * st.T.z;
    }
    return;
  }
  const uint64_t esc_mask = __ballot(escaped);
  if (!esc_mask) return;
  uint32_t base = 0;
  if (lane == (uint32_t)__ffsll((long long)esc_mask) - 1u) base = atomicAdd(wg_count, (uint32_t)__popcll(esc_mask));
  base = __shfl(base, __ffsll((long long)esc_mask) - 1, 64);
  if (escaped) {
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(esc_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)esc_mask, 0u));
    const uint32_t q = region_base + base + rank;
    float u, v;
    dir_to_uv(st.d, P.azimuth, u, v);
    P.q_u[q] = u; P.q_v[q] = v;
    P.q_tr[q] = st.T.x; P.q_tg[q] = st.T.y; P.q_tb[q] = st.T.z;
    P.q_path[q] = idx;
  }
}

__device__ __forceinline__ uint32_t pack_half2(float a, float b) {   // both are exact halves (hround): lossless
  union { _Float16 h[2]; uint32_t u; } c;
  c.h[0] = (_Float16)a; c.h[1] = (_Float16)b;
  return c.u;
}

#ifdef PTMI_DIAG_BUILD
}  // namespace ptd
#include "diag/pt_trace_r3fn.h"
#include "diag/pt_trace_rounds.h"   // round-4 experiment: the secondary phase in workgroup-synchronous, material-sorted rounds (not kept)
namespace ptd {
#endif

// OPT (round 4; the product has both, the profiling build keeps the others for the A/B): bit 0 = path index split by a
// reciprocal multiply instead of a division, bit 1 = the camera ray's intersection with the origin folded into constants.
// Profiling build only: bits 2 / 3 timing-only phase cuts, bit 4 occupancy counters of the secondary loop, bit 5 the
// secondary phase in workgroup-synchronous rounds that regroup the paths by material (diag/pt_trace_rounds.h): 14 % fewer
// instructions and 78.7 % lane utilisation, but no faster on its own (the VALU goes from 97 % to 84 % busy behind two
// workgroup barriers per 256 paths) and 2 % slower inside the C2 step -- its 16 KiB of LDS do not fit beside the NIF
// kernel's 157 KiB, so the trace kernel loses its place under the MFMA kernel (profiles/r04_trace_ablation.txt).
constexpr int kTraceOpt = 3;   // (profiling build, bit 7: the object loop unrolled over the compile-time scene, diag/pt_trace_scene_c.h)
template <uint32_t REFILL, int OPT = kTraceOpt>
__device__ __forceinline__ void trace_body(const TraceParams& P) {
  constexpr bool MAGIC = (OPT & 1) != 0, PRIMARY = (OPT & 2) != 0, SCENE_C = (OPT & 128) != 0, PIPE = (OPT & 256) != 0;
  __shared__ uint32_t wg_count;   // escaped paths queued by this workgroup
  __shared__ uint32_t wg_front, wg_back;   // camera rays of this workgroup that hit a diffuse / a mirror or glass object
  __shared__ uint32_t wg_state;            // survivors still alive after their first shading
  __shared__ HitRow hit_table[kNumObjects];
  if (threadIdx.x == 0) { wg_count = 0; wg_front = 0; wg_back = 0; wg_state = 0; fill_hit_table(P, hit_table); }
  __syncthreads();

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wib = threadIdx.x >> 6;                              // wave in the workgroup
  const uint32_t gw = blockIdx.x * (kTraceBlock / 64) + wib;
  const uint32_t region_base = blockIdx.x * P.region_cap;
  uint4* const surv = P.survivors + (size_t)region_base;
  float4* const st0 = P.states + (size_t)region_base;
  float4* const st1 = st0 + P.state_stride;
  float4* const st2 = st1 + P.state_stride;
  // paths this wave owns: chunk j -> idx = (j * n_waves + gw) * 64 + lane
  const uint32_t n_chunks = (P.total_paths + 63u) / 64u;
  const uint32_t my_chunks = (n_chunks > gw) ? (n_chunks - gw + P.n_waves - 1u) / P.n_waves : 0u;

  // ---- primary phase: every lane starts a new path each trip (GenerateCameraRays + the first Scene::intersect)
  for (uint32_t j = 0; j < my_chunks; ++j) {
    const uint32_t idx = (j * P.n_waves + gw) * 64u + lane;
    bool valid = idx < P.total_paths;
    PathState st;
    float camx = 0.f, camy = 0.f, tbest = 0.f;
    int best = -1;
    if (valid) {
      uint32_t item, iter;
      split_index<MAGIC>(P, idx, item, iter);
      const uint32_t pixel = P.pix[item];
      // Worklist padding (LoadBalancer.cpp:66-71: u = v = 65535, "ignored during image accumulation", AccumulatedImage.cpp:66)
      // is not traced: a path record of length 0, no radiance, no NIF evaluation -- the reference's tiles trace these items
      // and throw the result away.
      if ((pixel & 0xffffu) >= P.width || (pixel >> 16) >= P.height) { P.plen[idx] = 0; valid = false; }
      else {
        start_path(P, pixel, P.sample_base + iter, st, camx, camy);
#ifdef PTMI_DIAG_BUILD
        if constexpr (SCENE_C) best = nearest_hit_primary_c(st.d, tbest, SceneIndices{});
        else
#endif
        best = PRIMARY ? nearest_hit_primary<PIPE>(P, st.d, tbest) : nearest_hit<PIPE>(P, st.o, st.d, tbest);
      }
    }
    const bool hit = best >= 0;
    // a miss at depth 0 is final (codelets.cpp:184-190): one record, no roulette below roulette_depth >= 1, so the
    // throughput is (1, 1, 1) x 1 exactly as bounce() would leave it
    const bool escaped = valid && !hit;
    if (escaped) P.plen[idx] = (uint8_t)(1u | 0x80u);
    emit_escaped(P, escaped, st, idx, lane, region_base, &wg_count);
    // survivors: diffuse hits fill the list from the front, mirror / glass hits from the back
    const bool diffuse = hit && (__float_as_uint(hit_table[hit ? best : 0].colour.w) & 0xffu) == (uint32_t)MAT_DIFFUSE;
    const bool other = hit && !diffuse;
    const uint64_t dmask = __ballot(diffuse), omask = __ballot(other);
    uint32_t pos = 0;
    if (dmask) {
      uint32_t base = 0;
      if (lane == (uint32_t)__ffsll((long long)dmask) - 1u) base = atomicAdd(&wg_front, (uint32_t)__popcll(dmask));
      base = __shfl(base, __ffsll((long long)dmask) - 1, 64);
      if (diffuse) pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(dmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)dmask, 0u));
    }
    if (omask) {
      uint32_t base = 0;
      if (lane == (uint32_t)__ffsll((long long)omask) - 1u) base = atomicAdd(&wg_back, (uint32_t)__popcll(omask));
      base = __shfl(base, __ffsll((long long)omask) - 1, 64);
      if (other) pos = P.region_cap - 1u - (base + __builtin_amdgcn_mbcnt_hi((uint32_t)(omask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)omask, 0u)));
    }
    if (hit) surv[pos] = make_uint4(idx, pack_half2(camx, camy), __float_as_uint(tbest), (uint32_t)best);
  }
  __syncthreads();   // the workgroup's survivor list is complete (and visible: the workgroup's own global stores, first read now)
  const uint32_t n_front = wg_front, n_surv = (OPT & 8) ? 0u : n_front + wg_back;   // (bit 3, timing only: primary phase alone)

  // ---- first-shading phase: survivor v of the list (front part, then back part) gets its depth-0 bounce finished -- the half
  // of a loop trip behind the intersection -- 64 survivors per wave and trip, a chunk inside the front part all diffuse
  for (uint32_t c = wib; c * 64u < n_surv; c += 4u) {
    const uint32_t v = c * 64u + lane;
    const bool valid = v < n_surv;
    bool alive = false;
    PathState st;
    uint32_t idx = 0;
    if (valid) {
      const uint4 note = surv[v < n_front ? v : P.region_cap - 1u - (v - n_front)];
      idx = note.x;
      union { uint32_t u; _Float16 h[2]; } cam;
      cam.u = note.y;
      st.o = mk(0.f, 0.f, 0.f);
      st.d = normalise(mk((float)cam.h[0], (float)cam.h[1], -1.f));      // as start_path (codelets.cpp:162-163)
      st.T = mk(1.f, 1.f, 1.f);
      uint32_t item, iter;
      split_index<MAGIC>(P, idx, item, iter);
      st.pixel = P.pix[item];
      st.sample = P.sample_base + iter;
      st.depth = 0;
      uint32_t w[4];
      philox4x32_10(st.pixel, st.sample, 1u, 0x5054u, P.seed_lo, P.seed_hi, w);   // the block of bounce 0; no roulette at depth 0
      uint32_t length = 0;
      int res;
#ifdef PTMI_DIAG_BUILD
      if constexpr ((OPT & 64) != 0) res = shade_hit_r3(P, hit_table, st, (int)note.w, __uint_as_float(note.z), w, 1.0f, length);
      else
#endif
      res = shade_hit(P, hit_table, st, (int)note.w, __uint_as_float(note.z), w, 1.0f, length);
      if (res == STEP_CONTINUE) alive = true;
      else P.plen[idx] = (uint8_t)length;                                  // max_path_length = 1: the stack is full
    }
    const uint64_t amask = __ballot(alive);
    if (amask) {
      uint32_t base = 0;
      if (lane == (uint32_t)__ffsll((long long)amask) - 1u) base = atomicAdd(&wg_state, (uint32_t)__popcll(amask));
      base = __shfl(base, __ffsll((long long)amask) - 1, 64);
      if (alive) {
        const uint32_t e = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(amask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)amask, 0u));
        st0[e] = make_float4(st.o.x, st.o.y, st.o.z, st.d.x);
        st1[e] = make_float4(st.d.y, st.d.z, st.T.x, st.T.y);
        st2[e] = make_float4(st.T.z, __uint_as_float(idx), 0.f, 0.f);
      }
    }
  }
  __syncthreads();
  const uint32_t n_state = (OPT & 4) ? 0u : wg_state;   // (bit 2, timing only: no secondary phase)

#ifdef PTMI_DIAG_BUILD
  if constexpr ((OPT & 32) != 0) {
    secondary_rounds(P, hit_table, n_state, st0, st1, st2, region_base, &wg_count);
    __syncthreads();
    if (threadIdx.x == 0) P.region_count[blockIdx.x] = wg_count;
    return;
  }
#endif
  // ---- secondary phase (round 3, profiling build): the path loop over the shaded survivors, persistent lanes.  Wave w owns entries
  // e(j) = ((j / 64) * 4 + w) * 64 + j % 64 of the state list; an idle lane takes the next one and goes on at depth 1.
  uint32_t cursor = 0;
  const uint32_t s_chunks = (n_state + 63u) / 64u;
  const uint32_t mine = (s_chunks > wib) ? (s_chunks - wib + 3u) / 4u * 64u : 0u;
  PathState st;
  uint32_t idx = 0;
  bool active = false;
  uint32_t dg_trips = 0, dg_active = 0, dg_tail_trips = 0, dg_tail_active = 0;   // (OPT bit 4 only)
  while (true) {
    const uint64_t act_mask = __ballot(active);
    const uint32_t n_active = (uint32_t)__popcll(act_mask);
    const bool more = cursor < mine;
    if (!more && n_active == 0) break;
    if (more && (n_active == 0 || 64u - n_active >= REFILL)) {
      const uint64_t idle = ~act_mask;
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
      if (!active) {
        const uint32_t j = cursor + rank;
        const uint32_t e = ((j >> 6) * 4u + wib) * 64u + (j & 63u);
        if (j < mine && e < n_state) {
          const float4 a = st0[e], b = st1[e], c = st2[e];
          st.o = mk(a.x, a.y, a.z);
          st.d = mk(a.w, b.x, b.y);
          st.T = mk(b.z, b.w, c.x);
          idx = __float_as_uint(c.y);
          uint32_t item, iter;
          split_index<MAGIC>(P, idx, item, iter);
          st.pixel = P.pix[item];
          st.sample = P.sample_base + iter;
          st.depth = 1;
          active = true;
        }
      }
      cursor += 64u - n_active;
    }
    if constexpr ((OPT & 16) != 0) {
      const uint32_t na = (uint32_t)__popcll(__ballot(active));
      dg_trips += 1; dg_active += na;
      if (cursor >= mine) { dg_tail_trips += 1; dg_tail_active += na; }
    }
    int res = STEP_CONTINUE;
    uint32_t length = 0;
    if (active) res = bounce<(OPT & 64) != 0, SCENE_C, PIPE>(P, hit_table, st, length);
    const bool ended = active && res != STEP_CONTINUE;
    const bool escaped = active && res == STEP_ESCAPED;
    if (ended) {
      P.plen[idx] = (uint8_t)(length | (escaped ? 0x80u : 0u));
      active = false;
    }
    emit_escaped(P, escaped, st, idx, lane, region_base, &wg_count);
  }
  if constexpr ((OPT & 16) != 0) {
    if (lane == 0 && P.diag) {
      atomicAdd(&P.diag[0], (unsigned long long)dg_trips); atomicAdd(&P.diag[1], (unsigned long long)dg_active);
      atomicAdd(&P.diag[2], (unsigned long long)dg_tail_trips); atomicAdd(&P.diag[3], (unsigned long long)dg_tail_active);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) P.region_count[blockIdx.x] = wg_count;
}

__global__ __launch_bounds__(kTraceBlock) void trace_kernel(const TraceParams P) { trace_body<kRefillThreshold>(P); }
#ifdef PTMI_DIAG_BUILD
template <int OPT>
__global__ __launch_bounds__(kTraceBlock) void trace_kernel_opt(const TraceParams P) { trace_body<kRefillThreshold, OPT>(P); }   // round-4 A/B (0 = the round-3 kernel; 7, 11: timing-only phase cuts)
#endif

struct PathRecordOut {  // layout of pt_path_record (include/ptmi.h)
  uint32_t length, escaped;
  float dir[3], uv[2], throughput[3], cam[2];
};

// One thread per requested path; same device functions as trace_kernel.
__global__ void trace_paths_kernel(const TraceParams P, const uint16_t* u, const uint16_t* v, const uint32_t* sample,
                                   uint32_t n, PathRecordOut* out) {
  __shared__ HitRow hit_table[kNumObjects];
  if (threadIdx.x == 0) fill_hit_table(P, hit_table);
  __syncthreads();
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  PathState st;
  float camx, camy;
  start_path(P, (uint32_t)u[i] | ((uint32_t)v[i] << 16), sample[i], st, camx, camy);
  uint32_t length = 0;
  int res;
  do { res = bounce(P, hit_table, st, length); } while (res == STEP_CONTINUE);
  PathRecordOut r = {};
  r.length = length;
  r.escaped = (res == STEP_ESCAPED);
  r.cam[0] = camx; r.cam[1] = camy;
  if (r.escaped) {
    r.dir[0] = st.d.x; r.dir[1] = st.d.y; r.dir[2] = st.d.z;
    dir_to_uv(st.d, P.azimuth, r.uv[0], r.uv[1]);
    r.throughput[0] = st.T.x; r.throughput[1] = st.T.y; r.throughput[2] = st.T.z;
  }
  out[i] = r;
}

}  // namespace ptd
