/*
 * pt_oracle.c -- CPU ORACLE (plain C restatement) of the ipu_path_trace hot path.
 * TEST INFRASTRUCTURE ONLY: see pt_oracle.h.  PARITY UNPINNED (no reference fixtures exist).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -mavx2 -mfma -mf16c -fopenmp (oracle/Makefile).
 * -ffp-contract=off matters: every float expression below is evaluated exactly as written
 * (IEEE binary32, round-to-nearest-even, left-to-right), which is the arithmetic contract the
 * HIP kernels reproduce bit for bit.  The only fused operations are the explicit fmaf() calls
 * inside the NIF matmul, whose parity is tolerance-based.
 *
 * Reference files restated (paths relative to /root/reference/src):
 *   codelets/codelets.cpp        GenerateCameraRays :47-79, RayTraceKernel :103-226,
 *                                AccumulateContributions :241-304, PreProcessEscapedRays :319-356,
 *                                PostProcessEscapedRays :366-382, scene constants :111-144
 *   codelets/WrappedArray.hpp    contribution stack :5-68
 *   codelets/TraceRecord.hpp     :7-19
 *   neural_networks/NifModel.cpp encode :185-218, dense stack :295-326, decode :221-245
 *   PathTracerApp.cpp            iteration order :432-458, AA noise :29-45, samples :285-299
 *
 * external/light (absent): every function tagged INFERRED below is reconstructed from the
 * call sites listed in SURVEY.md section 8(c).  The 1.15 refraction gain at codelets.cpp:212,
 * the rouletteWeight/diffuse/reflect/refract call shapes and the colour gain of 2 identify
 * light's lineage as the public "smallpaint" path tracer (K. Zsolnai-Feher), whose published
 * routines (sphere/plane intersection, ons(), hemisphere(), Schlick refraction, Russian
 * roulette) are restated here in single precision.
 */
#include "pt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ bits */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* IEEE binary16 <-> binary32, round-to-nearest-even, subnormals kept (poplar::HALF). */
uint16_t orc_f2h(float f) {
  uint32_t x = f2u(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x200u : 0u));
  if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* >= 65520 rounds to inf */
  if (ax < 0x33000001u) return (uint16_t)sign;              /* <= 2^-25 rounds to zero */
  int32_t e = (int32_t)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u;
  uint32_t shift, hexp;
  if (e < -14) { shift = (uint32_t)(13 + (-14 - e)); hexp = 0; }
  else { shift = 13; hexp = (uint32_t)(e + 15); }
  uint32_t q = m >> shift;
  uint32_t rem = m & ((1u << shift) - 1u);
  uint32_t halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (q & 1u))) q += 1u;
  uint32_t h;
  if (hexp == 0) h = q;                     /* subnormal (q may carry into exponent 1) */
  else h = ((hexp - 1u) << 10) + q;         /* q includes the hidden bit */
  return (uint16_t)(sign | h);
}

float orc_h2f(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 0x1fu;
  uint32_t m = h & 0x3ffu;
  if (e == 0) {
    if (m == 0) return u2f(sign);
    float v = (float)m * 5.9604644775390625e-08f; /* 2^-24, exact */
    return sign ? -v : v;
  }
  if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
  return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

#ifdef ORC_FAST_BUILD
/* TIMING BUILD (oracle/Makefile target libpt_oracle_fast.so: -O3 -march=native -fopenmp, contraction allowed): the same
 * source as the parity checker, compiled the way BASELINE.md section 3 / SURVEY.md 8(d) time a CPU baseline.  Only the
 * cpu_baseline leg of bench.py and tests/test_oracle_fast.py load it; it is never the parity checker.  What changes:
 * binary16 rounding through F16C (same RNE results), the NIF matmul as a register-blocked AVX2 / AVX-512 kernel over
 * batches of 64 samples (per output the same k-ordered FMA chain, so the NIF stays bit-identical to the strict build),
 * and the compiler may contract a*b+c in the trace arithmetic (paths can differ from the strict build in the last ulp). */
#include <immintrin.h>
#if !defined(__F16C__) || !defined(__AVX2__) || !defined(__FMA__)
#error "the timing build needs F16C, AVX2 and FMA (-march=native on any x86-64-v3 host)"
#endif
static inline float hround(float f) { return _cvtsh_ss(_cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)); }
#else
static inline float hround(float f) { return orc_h2f(orc_f2h(f)); }
#endif

/* ------------------------------------------------------------------ RNG
 * The IPU's hardware RNG streams (poprand, PathTracerApp.cpp:29-45,285-299,333-336) cannot be
 * reproduced off-IPU.  Shared, order-independent replacement: Philox4x32-10 (Salmon et al.,
 * SC'11) with key = (seed lo, seed hi) and counter = (u | v<<16, sample index, block, 'PT').
 * block 0 feeds the two anti-alias normals; block 1+depth feeds bounce `depth`:
 * word 0 = roulette, word 1 = diffuse sample1 / refract sample, word 2 = diffuse sample2. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#define ORC_RNG_TAG 0x5054u

static void rng_block(const orc_config* cfg, uint16_t u, uint16_t v, uint32_t sample, uint32_t block,
                      uint32_t out[4]) {
  uint32_t ctr[4] = {(uint32_t)u | ((uint32_t)v << 16), sample, block, ORC_RNG_TAG};
  uint32_t key[2] = {(uint32_t)cfg->seed, (uint32_t)(cfg->seed >> 32)};
  orc_philox4x32_10(ctr, key, out);
}

/* poprand::uniform(HALF, 0, 1) stores U[0,1) as half (PathTracerApp.cpp:297): 2^-11 grid. */
static inline float uniform01(const orc_config* cfg, uint32_t bits) {
  if (cfg->sample_precision == ORC_SAMPLES_HALF) return (float)(bits >> 21) * 4.8828125e-04f;
  return (float)(bits >> 8) * 5.9604644775390625e-08f;
}

/* ------------------------------------------------------------------ deterministic math
 * The reference calls libm (acosf/atan2 codelets.cpp:333-334; tan inside pixelToRay; normal
 * RNG inside poprand).  libm results differ between hosts and GPUs in the last ulp, which would
 * make path-level comparisons fuzzy, so the transcendental functions the trace stage needs are
 * defined here from +,-,*,/,sqrt only (all correctly rounded on both sides).  Each is within
 * 2 ulp of libm (tests/test_oracle_math.py). */

/* natural log for x in (0, 1]; atanh series on m in [sqrt(1/2), sqrt(2)). */
float orc_dm_log(float x) {
  uint32_t ix = f2u(x);
  int32_t e = (int32_t)(ix >> 23) - 127;
  uint32_t mant = ix & 0x7fffffu;
  if (mant > 0x3504f3u) { e += 1; ix = mant | 0x3f000000u; } /* m in [sqrt2/2, 1) */
  else ix = mant | 0x3f800000u;                              /* m in [1, sqrt2] */
  float m = u2f(ix);
  float s = (m - 1.0f) / (m + 1.0f);
  float z = s * s;
  float p = 0.0909090936183929443359375f;            /* 1/11 */
  p = p * z + 0.111111111938953399658203125f;        /* 1/9 */
  p = p * z + 0.142857149243354797363281250f;        /* 1/7 */
  p = p * z + 0.200000002980232238769531250f;        /* 1/5 */
  p = p * z + 0.3333333432674407958984375f;          /* 1/3 */
  float lm = 2.0f * s + (2.0f * s) * (z * p);
  float fe = (float)e;
  return fe * 0.693145751953125f + (lm + fe * 1.42860677e-06f);
}

/* sin(2 pi u), cos(2 pi u) for u in [0,1]. */
void orc_dm_sincos2pi(float u, float* s_out, float* c_out) {
  float t = u * 4.0f;
  int k = (int)(t + 0.5f);
  float r = t - (float)k;                    /* exact, |r| <= 0.5 */
  float x = r * 1.57079637050628662109375f;  /* |x| <= pi/4 */
  float z = x * x;
  float ps = -2.50521083854417187750521e-08f;        /* -1/11! */
  ps = ps * z + 2.75573192239858925109505e-06f;      /*  1/9!  */
  ps = ps * z - 1.98412698412698412698413e-04f;      /* -1/7!  */
  ps = ps * z + 8.33333333333333321768779e-03f;      /*  1/5!  */
  ps = ps * z - 1.66666666666666657414808e-01f;      /* -1/3!  */
  float sn = x + x * (z * ps);
  float pc = 2.08767569878680989792101e-09f;         /*  1/12! */
  pc = pc * z - 2.75573192239858906525573e-07f;      /* -1/10! */
  pc = pc * z + 2.48015873015873015873016e-05f;      /*  1/8!  */
  pc = pc * z - 1.38888888888888894189103e-03f;      /* -1/6!  */
  pc = pc * z + 4.16666666666666643537020e-02f;      /*  1/4!  */
  float cs = (1.0f - 0.5f * z) + (z * z) * pc;
  switch (k & 3) {
    case 0: *s_out = sn;  *c_out = cs;  break;
    case 1: *s_out = cs;  *c_out = -sn; break;
    case 2: *s_out = -sn; *c_out = -cs; break;
    default: *s_out = -cs; *c_out = sn; break;
  }
}

/* atan(t) for t in [0,1]. */
static float dm_atan01(float t) {
  float base = 0.0f;
  if (t > 0.414213567972183227539062f) { /* tan(pi/8) */
    t = (t - 1.0f) / (t + 1.0f);
    base = 0.785398185253143310546875f;  /* pi/4 */
  }
  float z = t * t;
  float p = 0x1.9e0c4cp-5f;
  p = p * z - 0x1.61601ep-4f;
  p = p * z + 0x1.c57fe2p-4f;
  p = p * z - 0x1.248a38p-3f;
  p = p * z + 0x1.99997cp-3f;
  p = p * z - 0x1.555556p-2f;
  return base + (t + t * (z * p));
}

float orc_dm_atan2(float y, float x) {
  const float pi = 3.1415927410125732421875f;
  const float pio2 = 1.57079637050628662109375f;
  float ax = fabsf(x), ay = fabsf(y);
  if (ax == 0.0f && ay == 0.0f) return 0.0f;
  float r;
  if (ay <= ax) r = dm_atan01(ay / ax);
  else r = pio2 - dm_atan01(ax / ay);
  if (x < 0.0f) r = pi - r;
  return (y < 0.0f) ? -r : r;
}

float orc_dm_acos(float x) {
  if (x >= 1.0f) return 0.0f;
  if (x <= -1.0f) return 3.1415927410125732421875f;
  float s = sqrtf((1.0f - x) * (1.0f + x));
  return orc_dm_atan2(s, x);
}

/* ------------------------------------------------------------------ vectors (light::Vector, INFERRED) */
typedef struct { float x, y, z; } vec3;
static inline vec3 V(float x, float y, float z) { vec3 r = {x, y, z}; return r; }
static inline vec3 vadd(vec3 a, vec3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 vsub(vec3 a, vec3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 vscale(vec3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline vec3 vcw(vec3 a, vec3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline float vdot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline vec3 vcross(vec3 a, vec3 b) {
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* smallpaint Vec::norm(): multiply by the reciprocal length. */
static inline vec3 vnorm(vec3 a) { float inv = 1.0f / sqrtf(vdot(a, a)); return vscale(a, inv); }

#define ORC_PI 3.1415927410125732421875f   /* light::Pi (codelets.cpp:335) */
#define ORC_EPS 1e-5f                      /* INFERRED: must be < 1e-4 (clear-coat gap, codelets.cpp:115-116) */
#define ORC_INF 3.402823466e+38f

/* ------------------------------------------------------------------ scene (codelets.cpp:111-144) */
enum { MAT_DIFFUSE = 0, MAT_SPECULAR = 1, MAT_REFRACTIVE = 2 };
typedef struct { int is_disc; vec3 centre; float radius; vec3 normal; vec3 colour; int type; } object_t;
#define NUM_OBJECTS 6
static object_t g_scene[NUM_OBJECTS];
static int g_scene_ready = 0;

static void scene_init(void) {
  if (g_scene_ready) return;
  const float gain = 2.f;                                    /* :127 */
  vec3 sphereColour = vscale(V(1.f, .89f, .55f), gain);      /* :128 */
  vec3 clearCoat = vscale(V(.8f, .06f, .391f), gain);        /* :129 */
  vec3 floorColour = vscale(V(.98f, .76f, .66f), gain);      /* :130 */
  vec3 glassTint = V(0.75f, 0.75f, 0.75f);                   /* :131 */
  vec3 one = V(1.f, 1.f, 1.f);
  vec3 none = V(0, 0, 0);
  /* double literals at :114-116 are converted to float by light::Vector's ctor */
  object_t s[NUM_OBJECTS] = {
    {0, {-1.8575f, -0.98714f, -3.6f}, 0.6f, {0, 0, 0}, sphereColour, MAT_DIFFUSE},               /* :112,:137 */
    {0, {0.74795f, -0.55f, -4.3816f}, 1.05f, {0, 0, 0}, one, MAT_SPECULAR},                       /* :113,:138 */
    {0, {1.9929f, -1.08666f, (float)-3.23}, 0.5f, {0, 0, 0}, glassTint, MAT_REFRACTIVE},          /* :114,:139 */
    {0, {(float)-0.19931, -1.183f, -2.75f}, 0.4f, {0, 0, 0}, clearCoat, MAT_DIFFUSE},             /* :115,:140 */
    {0, {(float)-0.19931, -1.183f, -2.75f}, 0.4001f, {0, 0, 0}, one, MAT_REFRACTIVE},             /* :116,:141 */
    {1, {0.f, -1.6f, -5.22f}, 3.5f, {0.f, 1.f, 0.f}, floorColour, MAT_DIFFUSE},                   /* :121,:143 */
  };
  (void)none;
  memcpy(g_scene, s, sizeof(s));
  g_scene_ready = 1;
}

void orc_scene_object(int i, float centre[3], float* radius, float colour[3], int32_t* type) {
  scene_init();
  centre[0] = g_scene[i].centre.x; centre[1] = g_scene[i].centre.y; centre[2] = g_scene[i].centre.z;
  *radius = g_scene[i].radius;
  colour[0] = g_scene[i].colour.x; colour[1] = g_scene[i].colour.y; colour[2] = g_scene[i].colour.z;
  *type = g_scene[i].type | (g_scene[i].is_disc ? 0x100 : 0);
}

/* light::Sphere::intersect -- INFERRED (smallpaint Sphere::intersect, single precision). */
static float sphere_intersect(vec3 o, vec3 d, vec3 c, float radius) {
  vec3 oc = vsub(o, c);
  float b = 2.0f * vdot(oc, d);
  float c_ = vdot(oc, oc) - radius * radius;
  float disc = b * b - 4.0f * c_;
  if (disc < 0.0f) return 0.0f;
  disc = sqrtf(disc);
  float sol1 = -b + disc;
  float sol2 = -b - disc;
  return (sol2 > ORC_EPS) ? sol2 * 0.5f : ((sol1 > ORC_EPS) ? sol1 * 0.5f : 0.0f);
}

/* light::Disc(normal, centre, radius)::intersect -- INFERRED (plane hit inside the radius). */
static float disc_intersect(vec3 o, vec3 d, vec3 n, vec3 c, float radius) {
  float denom = vdot(n, d);
  if (denom == 0.0f) return 0.0f;
  float t = vdot(vsub(c, o), n) / denom;
  if (!(t > ORC_EPS)) return 0.0f;
  vec3 p = vadd(o, vscale(d, t));
  vec3 pc = vsub(p, c);
  if (vdot(pc, pc) > radius * radius) return 0.0f;
  return t;
}

/* light::Scene<N>::intersect -- INFERRED: nearest hit over the objects in declaration order;
 * "advancing [the ray] to the hit point" (codelets.cpp:182). */
static int scene_intersect(vec3* origin, vec3 d, vec3* normal, float* t_out) {
  scene_init();
  int best = -1;
  float tbest = ORC_INF;
  for (int i = 0; i < NUM_OBJECTS; ++i) {
    const object_t* ob = &g_scene[i];
    float t = ob->is_disc ? disc_intersect(*origin, d, ob->normal, ob->centre, ob->radius)
                          : sphere_intersect(*origin, d, ob->centre, ob->radius);
    if (t > ORC_EPS && t < tbest) { tbest = t; best = i; }
  }
  if (best < 0) return -1;
  vec3 hp = vadd(*origin, vscale(d, tbest));
  *origin = hp;
  *normal = g_scene[best].is_disc ? g_scene[best].normal : vnorm(vsub(hp, g_scene[best].centre));
  *t_out = tbest;
  return best;
}

/* ------------------------------------------------------------------ light:: sampling (INFERRED) */

/* light::pixelToRay(col,row,W,H,fov) -- INFERRED: pinhole, horizontal FOV (PathTracerApp.cpp:808),
 * z = -1 (codelets.cpp:73-75,162); smallpaint camcr() shape with half-angle fov/2 and square
 * pixels (the sphere silhouettes in the reference's images/example.png are consistent with
 * ty = (h/w) tx, not with tan((h/w) fov/2): tests/test_oracle_example_image.py). */
static vec3 pixel_to_ray(float col, float row, uint32_t wi, uint32_t hi, float fov) {
  float w = (float)wi, h = (float)hi;
  float tx = tanf(fov * 0.5f);
  float ty = (h / w) * tx;
  return V(((2.0f * col - w) / w) * tx, -(((2.0f * row - h) / h) * ty), -1.0f);
}

/* light::rouletteWeight(u, stopProb) -> (stop, factor) -- INFERRED (codelets.cpp:178). */
static int roulette(float u, float p, float* factor) {
  if (u <= p) { *factor = 1.0f; return 1; }
  *factor = 1.0f / (1.0f - p);
  return 0;
}

/* smallpaint hemisphere(): uniform hemisphere sample about +z. */
static vec3 hemisphere(float u1, float u2) {
  float r = sqrtf(1.0f - u1 * u1);
  float s, c;
  orc_dm_sincos2pi(u2, &s, &c);
  return V(c * r, s * r, u1);
}

/* smallpaint ons(): orthonormal frame about v1. */
static void ons(vec3 v1, vec3* v2, vec3* v3) {
  if (fabsf(v1.x) > fabsf(v1.y)) {
    float inv = 1.0f / sqrtf(v1.x * v1.x + v1.z * v1.z);
    *v2 = V(-v1.z * inv, 0.0f, v1.x * inv);
  } else {
    float inv = 1.0f / sqrtf(v1.y * v1.y + v1.z * v1.z);
    *v2 = V(0.0f, v1.z * inv, -v1.y * inv);
  }
  *v3 = vcross(v1, *v2);
}

/* light::diffuse(ray, normal, intersection, rrFactor, u1, u2) -- INFERRED (codelets.cpp:200-204):
 * redirect the ray into the hemisphere about the normal; weight = cos(theta) * rrFactor
 * (uniform-hemisphere estimator 2 rho cos(theta): the factor 2 is the scene's colourGain :127). */
static vec3 diffuse_dir(vec3 n, float u1, float u2) {
  vec3 rx, ry;
  ons(n, &rx, &ry);
  vec3 s = hemisphere(u1, u2);
  return V(vdot(V(rx.x, ry.x, n.x), s), vdot(V(rx.y, ry.y, n.y), s), vdot(V(rx.z, ry.z, n.z), s));
}

/* light::reflect(ray, normal) -- INFERRED (codelets.cpp:206). */
static vec3 reflect_dir(vec3 d, vec3 n) {
  float cost = vdot(d, n);
  return vnorm(vsub(d, vscale(n, cost * 2.0f)));
}

/* light::refract(ray, normal, ri, u) -> refracted? -- INFERRED (codelets.cpp:210-211):
 * Schlick Fresnel against u; total internal reflection reflects. */
static int refract_dir(vec3* d, vec3 n, float ri, float u) {
  float nn = ri;
  float r0 = (1.0f - nn) / (1.0f + nn);
  r0 = r0 * r0;
  if (vdot(n, *d) > 0.0f) { n = vscale(n, -1.0f); nn = 1.0f / nn; }
  nn = 1.0f / nn;
  float cost1 = -vdot(n, *d);
  float cost2 = 1.0f - nn * nn * (1.0f - cost1 * cost1);
  float m = 1.0f - cost1;
  float m2 = m * m;
  float rprob = r0 + (1.0f - r0) * (m2 * m2 * m);
  if (cost2 > 0.0f && u > rprob) {
    *d = vnorm(vadd(vscale(*d, nn), vscale(n, nn * cost1 - sqrtf(cost2))));
    return 1;
  }
  *d = vnorm(vadd(*d, vscale(n, cost1 * 2.0f)));
  return 0;
}

/* PreProcessEscapedRays (codelets.cpp:333-347). */
static void dir_to_uv(vec3 d, float azimuth, float* uo, float* vo) {
  float theta = orc_dm_acos(d.y);
  float phi = orc_dm_atan2(d.z, d.x) + azimuth;
  const float twoPi = 2.f * ORC_PI;
  const float invPi = 1.f / ORC_PI;
  const float inv2Pi = 1.f / twoPi;
  if (phi < 0.f) phi += twoPi;
  else if (phi > twoPi) phi -= twoPi;
  *uo = theta * invPi;
  *vo = phi * inv2Pi;
}

/* ------------------------------------------------------------------ AA noise + camera rays */

/* poprand::normal / uniform / truncatedNormal into a HALF tensor (PathTracerApp.cpp:29-45). */
static void aa_noise(const orc_config* cfg, uint16_t u, uint16_t v, uint32_t sample, float out[2]) {
  uint32_t w[4];
  rng_block(cfg, u, v, sample, 0, w);
  if (cfg->aa_noise_type == ORC_AA_UNIFORM) {
    float a = (float)(w[0] >> 8) * 5.9604644775390625e-08f;
    float b = (float)(w[1] >> 8) * 5.9604644775390625e-08f;
    out[0] = hround(2.0f * a - 1.0f);
    out[1] = hround(2.0f * b - 1.0f);
    return;
  }
  /* Box-Muller; u1 in (0,1]. */
  for (int attempt = 0; attempt < 2; ++attempt) {
    float u1 = (float)((w[2 * attempt] >> 8) + 1u) * 5.9604644775390625e-08f;
    float u2 = (float)(w[2 * attempt + 1] >> 8) * 5.9604644775390625e-08f;
    float rad = sqrtf(-2.0f * orc_dm_log(u1));
    float s, c;
    orc_dm_sincos2pi(u2, &s, &c);
    float n0 = rad * c, n1 = rad * s;
    if (cfg->aa_noise_type == ORC_AA_TRUNCATED_NORMAL) {
      /* truncatedNormal(mean 0, std 1, alpha 3): one redraw, then clamp. */
      int bad = (fabsf(n0) > 3.0f) || (fabsf(n1) > 3.0f);
      if (bad && attempt == 0) continue;
      n0 = fminf(fmaxf(n0, -3.0f), 3.0f);
      n1 = fminf(fmaxf(n1, -3.0f), 3.0f);
    }
    out[0] = hround(n0);
    out[1] = hround(n1);
    return;
  }
}

void orc_aa_noise(const orc_config* cfg, uint16_t u, uint16_t v, uint32_t sample, float noise[2]) {
  aa_noise(cfg, u, v, sample, noise);
}

/* GenerateCameraRays::compute (codelets.cpp:68-75): jitter product in half, rays stored as half. */
static void camera_ray(const orc_config* cfg, uint16_t u, uint16_t v, uint32_t sample, float cam[2]) {
  float noise[2];
  aa_noise(cfg, u, v, sample, noise);
  float aa = hround(cfg->aa_noise_scale);
  float fov = hround(cfg->fov_radians);
  float c = (float)u + hround(aa * noise[0]);
  float r = (float)v + hround(aa * noise[1]);
  vec3 p = pixel_to_ray(c, r, cfg->width, cfg->height, fov);
  cam[0] = hround(p.x);
  cam[1] = hround(p.y);
}

/* ------------------------------------------------------------------ RayTraceKernel (codelets.cpp:157-223) */
#define ORC_MAX_DEPTH 64
typedef struct { int type; vec3 clr; float weight; } contribution;

static uint32_t trace_records(const orc_config* cfg, uint16_t u, uint16_t v, uint32_t sample,
                              contribution* stack, float cam_out[2]) {
  scene_init();
  const vec3 zero = V(0, 0, 0), one = V(1, 1, 1);
  uint32_t capacity = cfg->max_path_length;
  if (capacity > ORC_MAX_DEPTH) capacity = ORC_MAX_DEPTH;
  const float stopProb = hround(cfg->stop_prob);          /* Input<half> stopProb :100 */
  const float ri = hround(cfg->refractive_index);         /* Input<half> refractiveIndex :99 */

  float cam[2];
  camera_ray(cfg, u, v, sample, cam);
  if (cam_out) { cam_out[0] = cam[0]; cam_out[1] = cam[1]; }
  vec3 origin = zero;
  vec3 dir = vnorm(V(cam[0], cam[1], -1.f));             /* :162-163, Ray ctor normalises */
  uint32_t depth = 0, size = 0;
  int hitEmitter = 0;

  while (size != capacity) {                              /* :173 */
    uint32_t w[4];
    rng_block(cfg, u, v, sample, 1u + depth, w);
    float rrFactor = 1.f;
    if (depth >= cfg->roulette_depth) {                   /* :176-180 */
      if (roulette(uniform01(cfg, w[0]), stopProb, &rrFactor)) break;
    }
    vec3 normal;
    float t;
    int obj = scene_intersect(&origin, dir, &normal, &t); /* :183 */
    if (obj < 0) {                                        /* :184-190 */
      stack[size].type = ORC_ESCAPED; stack[size].clr = dir; stack[size].weight = rrFactor;
      size++; hitEmitter = 1; break;
    }
    const object_t* ob = &g_scene[obj];                   /* no emitters in the scene (:137-143) */
    if (ob->type == MAT_DIFFUSE) {                        /* :199-204 */
      float s1 = uniform01(cfg, w[1]);
      float s2 = uniform01(cfg, w[2]);
      dir = diffuse_dir(normal, s1, s2);
      float cost = vdot(dir, normal);
      stack[size].type = ORC_DIFFUSE; stack[size].clr = ob->colour; stack[size].weight = cost * rrFactor;
      size++;
    } else if (ob->type == MAT_SPECULAR) {                /* :205-207 */
      dir = reflect_dir(dir, normal);
      stack[size].type = ORC_SPECULAR; stack[size].clr = zero; stack[size].weight = rrFactor;
      size++;
    } else {                                              /* :208-213 */
      int refracted = refract_dir(&dir, normal, ri, uniform01(cfg, w[1]));
      stack[size].type = ORC_REFRACT; stack[size].clr = refracted ? ob->colour : one;
      stack[size].weight = 1.15f * rrFactor;
      size++;
    }
    depth += 1;                                           /* :215 */
  }
  if (!hitEmitter) {                                      /* :219-222 */
    if (size == 0) size = 1; /* reference reads store[-1] here (UB when roulette-depth == 0) */
    stack[size - 1].type = ORC_END; stack[size - 1].clr = zero; stack[size - 1].weight = 0.f;
  }
  return size;
}

/* AccumulateContributions fold (codelets.cpp:255-292); env = radiance for the ESCAPED record. */
static vec3 fold_backward(const contribution* stack, uint32_t size, vec3 env) {
  vec3 total = V(0, 0, 0);
  for (uint32_t i = size; i-- > 0;) {
    const contribution* c = &stack[i];
    switch (c->type) {
      case ORC_DIFFUSE: total = vscale(vcw(total, c->clr), c->weight); break;   /* :265 */
      case ORC_EMIT: total = vadd(total, vscale(c->clr, c->weight)); break;
      case ORC_ESCAPED: total = vadd(total, vscale(env, c->weight)); break;      /* :271 */
      case ORC_REFRACT: total = vscale(vcw(total, c->clr), c->weight); break;   /* :276 */
      case ORC_SPECULAR: total = vscale(total, c->weight); break;                /* :281 */
      default: break;
    }
  }
  return total;
}

/* Forward throughput: T = ((1*c0)*w0 ...)*w_terminal, total = env (.) T. */
static vec3 forward_throughput(const contribution* stack, uint32_t size) {
  vec3 T = V(1, 1, 1);
  for (uint32_t i = 0; i < size; ++i) {
    const contribution* c = &stack[i];
    switch (c->type) {
      case ORC_DIFFUSE: case ORC_REFRACT: T = vscale(vcw(T, c->clr), c->weight); break;
      case ORC_SPECULAR: case ORC_ESCAPED: T = vscale(T, c->weight); break;
      default: break;
    }
  }
  return T;
}

int orc_trace_records(const orc_config* cfg, uint16_t u, uint16_t v, uint32_t sample,
                      int32_t* types, float* clr, float* weight, uint32_t capacity) {
  contribution stack[ORC_MAX_DEPTH];
  uint32_t n = trace_records(cfg, u, v, sample, stack, NULL);
  for (uint32_t i = 0; i < n && i < capacity; ++i) {
    types[i] = stack[i].type;
    clr[3 * i] = stack[i].clr.x; clr[3 * i + 1] = stack[i].clr.y; clr[3 * i + 2] = stack[i].clr.z;
    weight[i] = stack[i].weight;
  }
  return (int)n;
}

int orc_trace_path(const orc_config* cfg, uint16_t u, uint16_t v, uint32_t sample, orc_path* out) {
  contribution stack[ORC_MAX_DEPTH];
  memset(out, 0, sizeof(*out));
  uint32_t n = trace_records(cfg, u, v, sample, stack, out->cam);
  out->length = n;
  out->escaped = (stack[n - 1].type == ORC_ESCAPED);
  if (out->escaped) {
    vec3 d = stack[n - 1].clr;
    out->dir[0] = d.x; out->dir[1] = d.y; out->dir[2] = d.z;
    dir_to_uv(d, cfg->azimuth_radians, &out->uv[0], &out->uv[1]);
    vec3 T = forward_throughput(stack, n);
    out->throughput[0] = T.x; out->throughput[1] = T.y; out->throughput[2] = T.z;
  }
  return 0;
}

/* Object index (0..5, -1 = environment) seen by the central ray of every pixel: the silhouettes Scene::intersect
 * (codelets.cpp:183) and pixelToRay (:73) imply together.  Used to pin the INFERRED camera model and scene geometry
 * against the reference's one rendered image (tests/test_oracle_example_image.py). */
void orc_object_ids(uint32_t w, uint32_t h, float fov, int8_t* ids) {
#pragma omp parallel for schedule(static)
  for (long long r = 0; r < (long long)h; ++r) {
    for (uint32_t c = 0; c < w; ++c) {
      float d[3], o[3] = {0.f, 0.f, 0.f}, t, hp[3], n[3];
      orc_pixel_to_ray((float)c, (float)r, w, h, fov, d);
      float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      d[0] /= len; d[1] /= len; d[2] /= len;
      ids[(size_t)r * w + c] = (int8_t)orc_scene_intersect(o, d, &t, hp, n);
    }
  }
}

/* Object a pixel's central ray finally reaches when it is followed DETERMINISTICALLY through the specular objects: a mirror
 * hit reflects (light::reflect), a glass hit refracts (light::refract with its Fresnel roulette forced to "refract": u = 1;
 * total internal reflection still reflects), a diffuse hit or the environment ends the walk.  ids: 0..5 = the diffuse object
 * reached, -1 = environment, -2 = still inside specular objects after max_bounces; bounces: specular interactions on the way.
 * This is what the reference's images/example.png shows INSIDE its mirror and glass spheres -- the one piece of reference
 * output in which reflect and refract are at work -- so tests/test_oracle_example_image.py can pin both INFERRED routines and
 * the refractive index against it.  `ri` is a parameter for the test's negative controls (1 / 1.5 = eta inverted; 1.33);
 * reflect_variant 1 is a DELIBERATELY WRONG reflection (mirrored about the surface: 2 (d.n) n - d) for the same purpose. */
void orc_specular_ids(uint32_t w, uint32_t h, float fov, float ri, int reflect_variant, int max_bounces, int8_t* ids,
                      uint8_t* bounces) {
  scene_init();
#pragma omp parallel for schedule(dynamic, 8)
  for (long long r = 0; r < (long long)h; ++r) {
    for (uint32_t c = 0; c < w; ++c) {
      vec3 o = V(0.f, 0.f, 0.f);
      vec3 d = vnorm(pixel_to_ray((float)c, (float)r, w, h, fov));
      int id = -2, nb = 0;
      for (int b = 0; b <= max_bounces; ++b) {
        vec3 normal;
        float t;
        const int obj = scene_intersect(&o, d, &normal, &t);
        if (obj < 0) { id = -1; break; }
        const int type = g_scene[obj].type;
        if (type == MAT_DIFFUSE) { id = obj; break; }
        if (b == max_bounces) break;
        if (type == MAT_SPECULAR) {
          if (reflect_variant == 1) { float cost = vdot(d, normal); d = vnorm(vsub(vscale(normal, cost * 2.0f), d)); }
          else d = reflect_dir(d, normal);
        } else {
          (void)refract_dir(&d, normal, ri, 1.0f);
        }
        nb += 1;
      }
      ids[(size_t)r * w + c] = (int8_t)id;
      if (bounces) bounces[(size_t)r * w + c] = (uint8_t)nb;
    }
  }
}

/* ------------------------------------------------------------------ NIF */
struct orc_nif {
  uint32_t n_layers;
  uint32_t emb;
  uint32_t* rows; uint32_t* cols; int32_t* relu; int32_t* has_bias;
  float** kernel;   /* fp16 values widened to float, [rows][cols] */
  float** bias;
  float max; float mean[3]; int32_t log_tonemap;
  uint32_t max_width;
  int32_t f32;                   /* float32 model: layers run in float (NifModel.cpp:314, kernel type = output type) */
  int32_t* lf32;                 /* per layer: this layer's variables are float32 (a mixed model; NULL = all layers as `f32`) */
};

orc_nif* orc_nif_create(const orc_layer* layers, uint32_t n_layers, uint32_t emb, float max,
                        const float mean[3], int32_t log_tonemap) {
  orc_nif* m = (orc_nif*)calloc(1, sizeof(orc_nif));
  m->n_layers = n_layers; m->emb = emb; m->max = max; m->log_tonemap = log_tonemap;
  memcpy(m->mean, mean, 12);
  m->rows = calloc(n_layers, 4); m->cols = calloc(n_layers, 4);
  m->relu = calloc(n_layers, 4); m->has_bias = calloc(n_layers, 4);
  m->kernel = calloc(n_layers, sizeof(float*)); m->bias = calloc(n_layers, sizeof(float*));
  m->max_width = 4 * emb;
  for (uint32_t l = 0; l < n_layers; ++l) {
    uint32_t r = layers[l].rows, c = layers[l].cols;
    m->rows[l] = r; m->cols[l] = c; m->relu[l] = layers[l].relu;
    if (r > m->max_width) m->max_width = r;
    if (c > m->max_width) m->max_width = c;
    m->kernel[l] = (float*)malloc((size_t)r * c * 4);
    for (size_t i = 0; i < (size_t)r * c; ++i) m->kernel[l][i] = orc_h2f(layers[l].kernel[i]);
    m->bias[l] = (float*)calloc(c, 4);
    m->has_bias[l] = layers[l].bias != NULL;
    if (layers[l].bias) for (uint32_t i = 0; i < c; ++i) m->bias[l][i] = orc_h2f(layers[l].bias[i]);
  }
  return m;
}

orc_nif* orc_nif_create_f32(const orc_layer_f32* layers, uint32_t n_layers, uint32_t emb, float max,
                            const float mean[3], int32_t log_tonemap) {
  orc_nif* m = (orc_nif*)calloc(1, sizeof(orc_nif));
  m->n_layers = n_layers; m->emb = emb; m->max = max; m->log_tonemap = log_tonemap; m->f32 = 1;
  memcpy(m->mean, mean, 12);
  m->rows = calloc(n_layers, 4); m->cols = calloc(n_layers, 4);
  m->relu = calloc(n_layers, 4); m->has_bias = calloc(n_layers, 4);
  m->kernel = calloc(n_layers, sizeof(float*)); m->bias = calloc(n_layers, sizeof(float*));
  m->max_width = 4 * emb;
  for (uint32_t l = 0; l < n_layers; ++l) {
    uint32_t r = layers[l].rows, c = layers[l].cols;
    m->rows[l] = r; m->cols[l] = c; m->relu[l] = layers[l].relu;
    if (r > m->max_width) m->max_width = r;
    if (c > m->max_width) m->max_width = c;
    m->kernel[l] = (float*)malloc((size_t)r * c * 4);
    memcpy(m->kernel[l], layers[l].kernel, (size_t)r * c * 4);
    m->bias[l] = (float*)calloc(c, 4);
    m->has_bias[l] = layers[l].bias != NULL;
    if (layers[l].bias) memcpy(m->bias[l], layers[l].bias, (size_t)c * 4);
  }
  return m;
}

/* A model whose layers have their own types: each matmul takes ITS kernel's type (NifModel.cpp:314), the bias add and the
 * ReLU happen in that type (:316-325), and the activations are cast to the next layer's type where it differs (float ->
 * half: round to nearest even; half -> float: exact).  The reference states the per-layer type at :314 but ships no mixed
 * model; the casts between layers are what poplin's "input type = kernel type" needs. */
orc_nif* orc_nif_create_mixed(const orc_layer_any* layers, uint32_t n_layers, uint32_t emb, float max,
                              const float mean[3], int32_t log_tonemap) {
  orc_nif* m = (orc_nif*)calloc(1, sizeof(orc_nif));
  m->n_layers = n_layers; m->emb = emb; m->max = max; m->log_tonemap = log_tonemap;
  memcpy(m->mean, mean, 12);
  m->rows = calloc(n_layers, 4); m->cols = calloc(n_layers, 4);
  m->relu = calloc(n_layers, 4); m->has_bias = calloc(n_layers, 4); m->lf32 = calloc(n_layers, 4);
  m->kernel = calloc(n_layers, sizeof(float*)); m->bias = calloc(n_layers, sizeof(float*));
  m->max_width = 4 * emb;
  for (uint32_t l = 0; l < n_layers; ++l) {
    uint32_t r = layers[l].rows, c = layers[l].cols;
    m->rows[l] = r; m->cols[l] = c; m->relu[l] = layers[l].relu; m->lf32[l] = layers[l].float32 != 0;
    if (r > m->max_width) m->max_width = r;
    if (c > m->max_width) m->max_width = c;
    m->kernel[l] = (float*)malloc((size_t)r * c * 4);
    m->bias[l] = (float*)calloc(c, 4);
    m->has_bias[l] = layers[l].bias != NULL;
    if (m->lf32[l]) {
      memcpy(m->kernel[l], layers[l].kernel, (size_t)r * c * 4);
      if (layers[l].bias) memcpy(m->bias[l], layers[l].bias, (size_t)c * 4);
    } else {
      const uint16_t* k = (const uint16_t*)layers[l].kernel;
      const uint16_t* b = (const uint16_t*)layers[l].bias;
      for (size_t i = 0; i < (size_t)r * c; ++i) m->kernel[l][i] = orc_h2f(k[i]);
      if (b) for (uint32_t i = 0; i < c; ++i) m->bias[l][i] = orc_h2f(b[i]);
    }
  }
  return m;
}

void orc_nif_destroy(orc_nif* m) {
  if (!m) return;
  for (uint32_t l = 0; l < m->n_layers; ++l) { free(m->kernel[l]); free(m->bias[l]); }
  free(m->kernel); free(m->bias); free(m->rows); free(m->cols); free(m->relu); free(m->has_bias); free(m->lf32);
  free(m);
}

uint64_t orc_nif_flops_per_sample(const orc_nif* m) {
  uint64_t f = 0;
  for (uint32_t l = 0; l < m->n_layers; ++l) {
    f += 2ull * m->rows[l] * m->cols[l];
    if (m->has_bias[l]) f += m->cols[l];
  }
  return f;
}

/* NifModel::buildEncodeInput (NifModel.cpp:200-216): uvNorm = (uv-1)*2, times 2^j, cast to half,
 * cos and sin evaluated in half, order [sin u, sin v, cos u, cos v].  Half-precision trig is
 * restated as round_half(libm(float(half arg))). */
void orc_nif_encode(uint32_t emb, float u, float v, float* f) {
  float un = (u - 1.0f) * 2.0f;
  float vn = (v - 1.0f) * 2.0f;
  float p = 1.0f;                       /* makeCoefficients: 2^j (NifModel.cpp:466-472) */
  for (uint32_t j = 0; j < emb; ++j) {
    float au = hround(un * p);
    float av = hround(vn * p);
    f[j] = hround(sinf(au));
    f[j + emb] = hround(sinf(av));
    f[j + 2 * emb] = hround(cosf(au));
    f[j + 3 * emb] = hround(cosf(av));
    p *= 2.0f;
  }
}

#ifdef ORC_FAST_BUILD
#define NIF_BATCH 64
#else
#define NIF_BATCH 16
#endif

#ifdef ORC_FAST_BUILD
/* One dense layer over a batch, timing build: y[b][n] = epilogue(sum_k x[b][k] * Wt[k][n]) with the sum formed as ONE
 * FMA chain in k order per (b, n) from 0 -- exactly the strict build's fmaf loop below -- but with the accumulators of a
 * (samples x 16 outputs) block held in registers and the block's weight panel (K x 16 floats = 20 KB at K = 320) reused
 * from L1 across the batch.  Epilogue as the reference's rounding points: round to half, + bias in half, ReLU. */
#if defined(__AVX512F__)
#define ORC_MB 12
#else
#define ORC_MB 6
#endif
static inline __m256 hround8(__m256 v) { return _mm256_cvtph_ps(_mm256_cvtps_ph(v, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)); }
static void nif_layer_fast(const orc_nif* m, uint32_t l, const float* x, float* y, int B, uint32_t W) {
  const uint32_t K = m->rows[l], N = m->cols[l];
  const float* Wt = m->kernel[l];
  const float* bias = m->bias[l];
  const int f32 = m->lf32 ? m->lf32[l] : m->f32, has_bias = m->has_bias[l], relu = m->relu[l];
  uint32_t n0 = 0;
  for (; n0 + 16 <= N; n0 += 16) {
    for (int b0 = 0; b0 < B; b0 += ORC_MB) {
      const int mb = (B - b0 < ORC_MB) ? B - b0 : ORC_MB;
      const float* xr[ORC_MB];
      for (int i = 0; i < ORC_MB; ++i) xr[i] = x + (size_t)(b0 + (i < mb ? i : 0)) * W;   /* tail rows repeat row 0 (not stored) */
#if defined(__AVX512F__)
      __m512 acc[ORC_MB];
      for (int i = 0; i < ORC_MB; ++i) acc[i] = _mm512_setzero_ps();
      for (uint32_t k = 0; k < K; ++k) {
        const __m512 w = _mm512_loadu_ps(Wt + (size_t)k * N + n0);
        for (int i = 0; i < ORC_MB; ++i) acc[i] = _mm512_fmadd_ps(_mm512_set1_ps(xr[i][k]), w, acc[i]);
      }
      __m256 lo[ORC_MB], hi[ORC_MB];
      for (int i = 0; i < ORC_MB; ++i) { lo[i] = _mm512_castps512_ps256(acc[i]); hi[i] = _mm512_extractf32x8_ps(acc[i], 1); }
#else
      __m256 lo[ORC_MB], hi[ORC_MB];
      for (int i = 0; i < ORC_MB; ++i) { lo[i] = _mm256_setzero_ps(); hi[i] = _mm256_setzero_ps(); }
      for (uint32_t k = 0; k < K; ++k) {
        const __m256 w0 = _mm256_loadu_ps(Wt + (size_t)k * N + n0), w1 = _mm256_loadu_ps(Wt + (size_t)k * N + n0 + 8);
        for (int i = 0; i < ORC_MB; ++i) {
          const __m256 xv = _mm256_broadcast_ss(xr[i] + k);
          lo[i] = _mm256_fmadd_ps(xv, w0, lo[i]);
          hi[i] = _mm256_fmadd_ps(xv, w1, hi[i]);
        }
      }
#endif
      const __m256 b_lo = _mm256_loadu_ps(bias + n0), b_hi = _mm256_loadu_ps(bias + n0 + 8), zero = _mm256_setzero_ps();
      for (int i = 0; i < mb; ++i) {
        __m256 o0 = lo[i], o1 = hi[i];
        if (!f32) { o0 = hround8(o0); o1 = hround8(o1); }                       /* matmul output type = kernel type (:314) */
        if (has_bias) {                                                          /* addInPlace :316-321 */
          o0 = _mm256_add_ps(o0, b_lo); o1 = _mm256_add_ps(o1, b_hi);
          if (!f32) { o0 = hround8(o0); o1 = hround8(o1); }
        }
        if (relu) { o0 = _mm256_max_ps(o0, zero); o1 = _mm256_max_ps(o1, zero); }   /* !(o > 0) -> 0, NaN included */
        _mm256_storeu_ps(y + (size_t)(b0 + i) * W + n0, o0);
        _mm256_storeu_ps(y + (size_t)(b0 + i) * W + n0 + 8, o1);
      }
    }
  }
  for (; n0 < N; ++n0)                                                           /* ragged tail (the 3-wide head) */
    for (int b = 0; b < B; ++b) {
      float a = 0.f;
      for (uint32_t k = 0; k < K; ++k) a = fmaf(x[(size_t)b * W + k], Wt[(size_t)k * N + n0], a);
      float o = f32 ? a : hround(a);
      if (has_bias) o = f32 ? o + bias[n0] : hround(o + bias[n0]);
      if (relu && !(o > 0.f)) o = 0.f;
      y[(size_t)b * W + n0] = o;
    }
}
#endif

/* x: [B][max_width] activations (fp16 values held in float). */
static void nif_forward(const orc_nif* m, const float* u, const float* v, int B, float* bgr,
                        float* bufA, float* bufB, float* input) {
  const uint32_t W = m->max_width + 4 * m->emb;
  const uint32_t in_dim = 4 * m->emb;
  for (int b = 0; b < B; ++b) {
    orc_nif_encode(m->emb, u[b], v[b], input + (size_t)b * in_dim);
    memcpy(bufA + (size_t)b * W, input + (size_t)b * in_dim, in_dim * 4);
  }
  float* x = bufA; float* y = bufB;
  uint32_t xcols = in_dim;
  int x_is_half = 1;                                    /* the Fourier features are half values */
#ifndef ORC_FAST_BUILD
  float acc[NIF_BATCH][1024 + 8];
#endif
  for (uint32_t l = 0; l < m->n_layers; ++l) {
    uint32_t K = m->rows[l], N = m->cols[l];
    if (xcols != K) {                                   /* NifModel.cpp:305-308 */
      for (int b = 0; b < B; ++b) memcpy(x + (size_t)b * W + xcols, input + (size_t)b * in_dim, in_dim * 4);
      xcols += in_dim;
    }
    const int lf32 = m->lf32 ? m->lf32[l] : m->f32;
    if (!lf32 && !x_is_half)                            /* a float16 layer behind a float32 one: its input is cast to half */
      for (int b = 0; b < B; ++b) for (uint32_t k = 0; k < xcols; ++k) x[(size_t)b * W + k] = hround(x[(size_t)b * W + k]);
    x_is_half = !lf32;
#ifdef ORC_FAST_BUILD
    nif_layer_fast(m, l, x, y, B, W);
#else
    const float* Wt = m->kernel[l];
    for (uint32_t n0 = 0; n0 < N; n0 += 1024) {
      uint32_t nn = (N - n0 < 1024) ? N - n0 : 1024;
      for (int b = 0; b < B; ++b) for (uint32_t n = 0; n < nn; ++n) acc[b][n] = 0.f;
      for (uint32_t k = 0; k < K; ++k) {
        const float* wr = Wt + (size_t)k * N + n0;
        for (int b = 0; b < B; ++b) {
          float xv = x[(size_t)b * W + k];
          float* a = acc[b];
          for (uint32_t n = 0; n < nn; ++n) a[n] = fmaf(xv, wr[n], a[n]);  /* poplin::matMul :314 */
        }
      }
      for (int b = 0; b < B; ++b) for (uint32_t n = 0; n < nn; ++n) {
        float o = lf32 ? acc[b][n] : hround(acc[b][n]);   /* matmul output type = kernel type (:314) */
        if (m->has_bias[l]) o = lf32 ? o + m->bias[l][n0 + n] : hround(o + m->bias[l][n0 + n]);   /* addInPlace :316-321 */
        if (m->relu[l] && !(o > 0.f)) o = 0.f;          /* ReLU :323-325 */
        y[(size_t)b * W + n0 + n] = o;
      }
    }
#endif
    float* t = x; x = y; y = t;
    xcols = N;
  }
  for (int b = 0; b < B; ++b) for (int c = 0; c < 3; ++c) {  /* buildDecodeOutput :226-242 */
    float o = x[(size_t)b * W + c] * m->max;
    o = o + m->mean[c];
    if (m->log_tonemap) o = expf(o);
    bgr[3 * b + c] = o;
  }
}

typedef struct { float* a; float* b; float* in; } nif_scratch;
static nif_scratch scratch_alloc(const orc_nif* m) {
  nif_scratch s;
  size_t W = m->max_width + 4 * m->emb;
  s.a = malloc(NIF_BATCH * W * 4); s.b = malloc(NIF_BATCH * W * 4); s.in = malloc(NIF_BATCH * 4 * m->emb * 4);
  return s;
}
static void scratch_free(nif_scratch s) { free(s.a); free(s.b); free(s.in); }

int orc_nif_infer(const orc_nif* m, const float* u, const float* v, size_t n, float* bgr) {
  if (!m || m->cols[m->n_layers - 1] < 3) return -1;
#pragma omp parallel
  {
    nif_scratch s = scratch_alloc(m);
#pragma omp for schedule(dynamic, 8)
    for (long long i = 0; i < (long long)((n + NIF_BATCH - 1) / NIF_BATCH); ++i) {
      size_t o = (size_t)i * NIF_BATCH;
      int B = (int)((n - o < NIF_BATCH) ? n - o : NIF_BATCH);
      nif_forward(m, u + o, v + o, B, bgr + 3 * o, s.a, s.b, s.in);
    }
    scratch_free(s);
  }
  return 0;
}

/* ------------------------------------------------------------------ whole iteration (PathTracerApp.cpp:432-458) */
int orc_render(const orc_config* cfg, const orc_nif* nif, orc_trace_record* rec, size_t n,
               uint32_t sample_base, uint32_t n_samples, orc_stats* stats) {
  if (cfg->env_mode == ORC_ENV_NIF && !nif) return -1;
  if (cfg->max_path_length == 0 || cfg->max_path_length > ORC_MAX_DEPTH) return -2;
  scene_init();
  uint64_t segs = 0, esc = 0;
  const size_t CH = NIF_BATCH;
#pragma omp parallel reduction(+ : segs, esc)
  {
    nif_scratch s = {0, 0, 0};
    if (nif) s = scratch_alloc(nif);
    contribution (*stacks)[ORC_MAX_DEPTH] = malloc(sizeof(contribution) * ORC_MAX_DEPTH * CH);
    uint32_t sizes[NIF_BATCH];
    float us[NIF_BATCH], vs[NIF_BATCH], bgr[3 * NIF_BATCH];
    int slot[NIF_BATCH];
#pragma omp for schedule(dynamic, 4)
    for (long long ci = 0; ci < (long long)((n + CH - 1) / CH); ++ci) {
      size_t base = (size_t)ci * CH;
      size_t cnt = (n - base < CH) ? n - base : CH;
      for (uint32_t si = 0; si < n_samples; ++si) {
        uint32_t sample = sample_base + si;
        int ne = 0;
        for (size_t j = 0; j < cnt; ++j) {
          orc_trace_record* t = &rec[base + j];
          slot[j] = -1;
          /* worklist padding (LoadBalancer.cpp:66-71: u = v = 65535; skipped by the film, AccumulatedImage.cpp:66) is not
           * traced: the reference's tiles trace it and discard the result, this restatement and the HIP path skip it --
           * pathLength + 0, no radiance, not counted in `paths`. */
          if (t->u >= cfg->width || t->v >= cfg->height) { sizes[j] = 0; continue; }
          sizes[j] = trace_records(cfg, t->u, t->v, sample, stacks[j], NULL);
          if (stacks[j][sizes[j] - 1].type == ORC_ESCAPED) {
            if (cfg->env_mode == ORC_ENV_NIF) {
              dir_to_uv(stacks[j][sizes[j] - 1].clr, cfg->azimuth_radians, &us[ne], &vs[ne]);
            }
            slot[j] = ne++;
          }
        }
        if (ne && cfg->env_mode == ORC_ENV_NIF) nif_forward(nif, us, vs, ne, bgr, s.a, s.b, s.in);
        for (size_t j = 0; j < cnt; ++j) {
          orc_trace_record* t = &rec[base + j];
          t->pathLength = (uint16_t)(t->pathLength + sizes[j]);          /* :253 */
          segs += sizes[j];
          if (slot[j] >= 0) {
            esc += 1;
            vec3 env;
            if (cfg->env_mode == ORC_ENV_NIF) {                          /* bgr -> rgb :378 */
              env = V(bgr[3 * slot[j] + 2], bgr[3 * slot[j] + 1], bgr[3 * slot[j] + 0]);
            } else {
              env = V(cfg->env_rgb[0], cfg->env_rgb[1], cfg->env_rgb[2]);
            }
            vec3 total;
            if (cfg->fold == ORC_FOLD_BACKWARD) total = fold_backward(stacks[j], sizes[j], env);
            else total = vcw(env, forward_throughput(stacks[j], sizes[j]));
            t->r += total.x; t->g += total.y; t->b += total.z;            /* :295-297 */
          }
          t->sampleCount = (uint16_t)(t->sampleCount + 1);               /* :300 */
        }
      }
    }
    free(stacks);
    if (nif) scratch_free(s);
  }
  if (stats) {
    uint64_t real = 0;
    for (size_t i = 0; i < n; ++i) real += rec[i].u < cfg->width && rec[i].v < cfg->height;
    stats->paths = real * n_samples; stats->segments = segs; stats->escaped = esc;
  }
  return 0;
}

/* Which build this is: "strict" (the parity checker) or the timing build's compiler flags. */
const char* orc_build_info(void) {
#ifdef ORC_FAST_BUILD
#ifdef ORC_BUILD_FLAGS
  return "timing build: " ORC_BUILD_FLAGS
#else
  return "timing build"
#endif
#if defined(__AVX512F__)
         " [AVX-512 12x16 NIF block]";
#else
         " [AVX2 6x16 NIF block]";
#endif
#else
  return "strict: -O3 -ffp-contract=off -fno-fast-math (parity checker)";
#endif
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------ KAT wrappers */
static vec3 A3(const float* p) { return V(p[0], p[1], p[2]); }
static void S3(float* p, vec3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

void orc_pixel_to_ray(float col, float row, uint32_t w, uint32_t h, float fov, float out[3]) {
  S3(out, pixel_to_ray(col, row, w, h, fov));
}
float orc_intersect_sphere(const float o[3], const float d[3], const float c[3], float radius) {
  return sphere_intersect(A3(o), A3(d), A3(c), radius);
}
float orc_intersect_disc(const float o[3], const float d[3], const float n[3], const float c[3], float radius) {
  return disc_intersect(A3(o), A3(d), A3(n), A3(c), radius);
}
int orc_scene_intersect(const float o[3], const float d[3], float* t, float hp[3], float nrm[3]) {
  vec3 origin = A3(o), normal = V(0, 0, 0);
  float tt = 0.f;
  int obj = scene_intersect(&origin, A3(d), &normal, &tt);
  if (obj >= 0) { *t = tt; S3(hp, origin); S3(nrm, normal); }
  return obj;
}
void orc_reflect(float d[3], const float n[3]) { S3(d, reflect_dir(A3(d), A3(n))); }
int orc_refract(float d[3], const float n[3], float ri, float u) {
  vec3 dd = A3(d);
  int r = refract_dir(&dd, A3(n), ri, u);
  S3(d, dd);
  return r;
}
void orc_hemisphere(float u1, float u2, float out[3]) { S3(out, hemisphere(u1, u2)); }
void orc_diffuse_dir(const float n[3], float u1, float u2, float out[3]) { S3(out, diffuse_dir(A3(n), u1, u2)); }
int orc_roulette(float u, float p, float* factor) { return roulette(u, p, factor); }
void orc_dir_to_uv(const float d[3], float azimuth, float uv[2]) { dir_to_uv(A3(d), azimuth, &uv[0], &uv[1]); }
