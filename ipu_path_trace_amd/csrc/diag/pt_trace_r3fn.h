// diag/pt_trace_r3fn.h -- PROFILING BUILD ONLY: Scene::intersect and the shading half of a bounce as round 3 had them (every
// sphere forms its own sub(o, c) / dot products; mirror and glass each normalise and update the throughput on their own), for
// the same-process A/B of the round-4 forms in pt_trace.h (OPT bit 6).  Included by pt_trace.h.
#pragma once

namespace ptd {

// Scene::intersect (codelets.cpp:183): nearest hit in declaration order, -1 for none.
// The loop stays rolled: one object's constants at a time are fetched from the kernel-argument segment (scalar
// loads, wave-uniform), instead of all of them living in SGPRs.
__device__ __forceinline__ int nearest_hit_r3(const TraceParams& P, Vec3 o, Vec3 d, float& tbest) {
  int best = -1;
  tbest = kInf;
#pragma unroll 1
  for (int i = 0; i < kNumObjects; ++i) {
    const SceneObject ob = P.obj[i];
    float t = ob.is_disc ? disc_intersect(o, d, ob) : sphere_intersect(o, d, ob);
    if (t > kEps && t < tbest) { tbest = t; best = i; }
  }
  return best;
}


// The second half of a loop trip of RayTraceKernel::compute (codelets.cpp:192-216): the ray has hit object `best` at
// distance `tbest`; w = the bounce's Philox block, rr = its roulette weight.
__device__ __forceinline__ int shade_hit_r3(const TraceParams& P, const HitRow* tab, PathState& s, int best, float tbest,
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
  } else if (type == MAT_SPECULAR) {                          // :205-207, light::reflect
    float cost = dot(s.d, n);
    s.d = normalise(sub(s.d, scale(n, cost * 2.0f)));
    s.T = scale(s.T, rr);
  } else {                                                    // :208-213, light::refract
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
    {   // one normalisation per lane: the vector is chosen first (sqrtf of a negative cost2 only feeds the side not taken)
      const Vec3 bent = add(scale(s.d, nn), scale(n, nn * cost1 - sqrtf(cost2)));
      const Vec3 mirrored = add(s.d, scale(n, cost1 * 2.0f));
      s.d = normalise(refracted ? bent : mirrored);
    }
    Vec3 tint = refracted ? mk(cr, cg, cb) : mk(1.f, 1.f, 1.f);
    s.T = scale(cwise(s.T, tint), 1.15f * rr);
  }
  s.depth += 1u;                                              // :215
  if (s.depth >= P.max_path_length) {                         // stack full without an emitter (:173,:219-222)
    length = s.depth;
    return STEP_DEAD;
  }
  return STEP_CONTINUE;
}


}  // namespace ptd
