// diag/pt_trace_scene_c.h -- PROFILING BUILD ONLY (round-4 experiment, not kept): Scene::intersect with the object loop unrolled
// over the compile-time scene table (pt_trace.h::scene_const), every constant a literal.  Included by pt_trace.h.
#pragma once

namespace ptd {

template <int I>
__device__ __forceinline__ void hit_object_c(Vec3 o, Vec3 d, float& b, float& oc2, float& tbest, int& best) {
  constexpr SceneConst S = scene_const(I);
  float t = 0.0f;
  if constexpr (S.disc) {                                   // disc_intersect
    const Vec3 n = mk(S.nx, S.ny, S.nz), c = mk(S.cx, S.cy, S.cz);
    const float denom = dot(n, d);
    if (denom != 0.0f) {
      const float tt = dot(sub(c, o), n) / denom;
      if (tt > kEps) {
        const Vec3 pc = sub(add(o, scale(d, tt)), c);
        constexpr float r2 = S.radius * S.radius;
        if (!(dot(pc, pc) > r2)) t = tt;
      }
    }
  } else {                                                  // sphere_intersect, the clear-coat pair sharing its centre terms
    if constexpr (!scene_same_centre(I)) {
      const Vec3 oc = sub(o, mk(S.cx, S.cy, S.cz));
      b = 2.0f * dot(oc, d);
      oc2 = dot(oc, oc);
    }
    constexpr float r2 = S.radius * S.radius;
    const float c_ = oc2 - r2;
    float disc = b * b - 4.0f * c_;
    if (!(disc < 0.0f)) {   // (a real branch on purpose: a wave none of whose rays comes near the sphere skips the rest)
      disc = sqrtf(disc);
      const float sol1 = -b + disc, sol2 = -b - disc;
      t = (sol2 > kEps) ? sol2 * 0.5f : ((sol1 > kEps) ? sol1 * 0.5f : 0.0f);
    }
  }
  if (t > kEps && t < tbest) { tbest = t; best = I; }
}
template <int... I>
__device__ __forceinline__ int nearest_hit_c(Vec3 o, Vec3 d, float& tbest, std::integer_sequence<int, I...>) {
  int best = -1;
  tbest = kInf;
  float b = 0.f, oc2 = 0.f;
  (hit_object_c<I>(o, d, b, oc2, tbest, best), ...);
  return best;
}

// camera rays (origin exactly 0): the object's constants of nearest_hit_primary, formed at compile time
template <int I>
__device__ __forceinline__ void hit_object_primary_c(Vec3 d, float& b, float& tbest, int& best) {
  constexpr SceneConst S = scene_const(I);
  float t = 0.0f;
  if constexpr (S.disc) {
    constexpr float k0 = (S.cx - 0.f) * S.nx, k1 = (S.cy - 0.f) * S.ny, k2 = (S.cz - 0.f) * S.nz;
    constexpr float kdisc = (k0 + k1) + k2;                 // dot(sub(c, o), n) at o = 0
    const Vec3 n = mk(S.nx, S.ny, S.nz), c = mk(S.cx, S.cy, S.cz);
    const float denom = dot(n, d);
    if (denom != 0.0f) {
      const float tt = kdisc / denom;
      if (tt > kEps) {
        const Vec3 pc = sub(scale(d, tt), c);
        constexpr float r2 = S.radius * S.radius;
        if (!(dot(pc, pc) > r2)) t = tt;
      }
    }
  } else {
    constexpr float ox = 0.f - S.cx, oy = 0.f - S.cy, oz = 0.f - S.cz;                        // sub(o, c) at o = 0
    if constexpr (!scene_same_centre(I)) b = 2.0f * dot(mk(ox, oy, oz), d);
    constexpr float r2 = S.radius * S.radius;
    constexpr float c4 = 4.0f * ((((ox * ox) + (oy * oy)) + (oz * oz)) - r2);                 // 4 (dot(oc, oc) - r2)
    float disc = b * b - c4;
    if (!(disc < 0.0f)) {
      disc = sqrtf(disc);
      const float sol1 = -b + disc, sol2 = -b - disc;
      t = (sol2 > kEps) ? sol2 * 0.5f : ((sol1 > kEps) ? sol1 * 0.5f : 0.0f);
    }
  }
  if (t > kEps && t < tbest) { tbest = t; best = I; }
}
template <int... I>
__device__ __forceinline__ int nearest_hit_primary_c(Vec3 d, float& tbest, std::integer_sequence<int, I...>) {
  int best = -1;
  tbest = kInf;
  float b = 0.f;
  (hit_object_primary_c<I>(d, b, tbest, best), ...);
  return best;
}
using SceneIndices = std::make_integer_sequence<int, kNumObjects>;

}  // namespace ptd
