// pt_device_math.h -- arithmetic of the trace stage on the device.
//
// Contract: every expression here is IEEE binary32 evaluated exactly as written (the file is
// compiled with -ffp-contract=off; '/' and sqrtf are correctly rounded in HIP by default), so a
// path traced on the GPU is bit-identical to the same path traced by the CPU oracle.  The
// transcendental functions the reference takes from libm / poprand (acosf, atan2:
// src/codelets/codelets.cpp:333-334; the normal distribution behind poprand::normal:
// src/PathTracerApp.cpp:34-37) are built from +,-,*,/ so that they cannot differ by platform.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd {

struct Vec3 {
  float x, y, z;
};

__device__ __forceinline__ Vec3 mk(float x, float y, float z) { return Vec3{x, y, z}; }
__device__ __forceinline__ Vec3 add(Vec3 a, Vec3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ Vec3 sub(Vec3 a, Vec3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ Vec3 scale(Vec3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ Vec3 cwise(Vec3 a, Vec3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ Vec3 cross(Vec3 a, Vec3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ Vec3 normalise(Vec3 a) {
  float inv = 1.0f / sqrtf(dot(a, a));
  return scale(a, inv);
}

// binary32 -> binary16 -> binary32, round-to-nearest-even (poplar::HALF storage).
__device__ __forceinline__ float hround(float f) { return (float)(_Float16)f; }

// Philox4x32-10 (Salmon et al., SC'11); constants as Random123.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 / v_mul_lo_u32 pair: the integer
    // multiplies run at quarter rate, and there are forty of them per block in the paired form
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// natural log, x in (0,1]
__device__ __forceinline__ float dm_log(float x) {
  uint32_t ix = __float_as_uint(x);
  int e = (int)(ix >> 23) - 127;
  uint32_t mant = ix & 0x7fffffu;
  if (mant > 0x3504f3u) { e += 1; ix = mant | 0x3f000000u; }
  else ix = mant | 0x3f800000u;
  float m = __uint_as_float(ix);
  float s = (m - 1.0f) / (m + 1.0f);
  float z = s * s;
  float p = 0.0909090936183929443359375f;
  p = p * z + 0.111111111938953399658203125f;
  p = p * z + 0.142857149243354797363281250f;
  p = p * z + 0.200000002980232238769531250f;
  p = p * z + 0.3333333432674407958984375f;
  float lm = 2.0f * s + (2.0f * s) * (z * p);
  float fe = (float)e;
  return fe * 0.693145751953125f + (lm + fe * 1.42860677e-06f);
}

// sin(2 pi u), cos(2 pi u), u in [0,1]
__device__ __forceinline__ void dm_sincos2pi(float u, float& s_out, float& c_out) {
  float t = u * 4.0f;
  int k = (int)(t + 0.5f);
  float r = t - (float)k;
  float x = r * 1.57079637050628662109375f;
  float z = x * x;
  float ps = -2.50521083854417187750521e-08f;
  ps = ps * z + 2.75573192239858925109505e-06f;
  ps = ps * z - 1.98412698412698412698413e-04f;
  ps = ps * z + 8.33333333333333321768779e-03f;
  ps = ps * z - 1.66666666666666657414808e-01f;
  float sn = x + x * (z * ps);
  float pc = 2.08767569878680989792101e-09f;
  pc = pc * z - 2.75573192239858906525573e-07f;
  pc = pc * z + 2.48015873015873015873016e-05f;
  pc = pc * z - 1.38888888888888894189103e-03f;
  pc = pc * z + 4.16666666666666643537020e-02f;
  float cs = (1.0f - 0.5f * z) + (z * z) * pc;
  int q = k & 3;
  float s = (q & 1) ? cs : sn;
  float c = (q & 1) ? sn : cs;
  s_out = (q & 2) ? -s : s;
  c_out = (q == 1 || q == 2) ? -c : c;
}

// (Written with selects instead of branches: on a 64-lane wave both sides of a data-dependent branch run one after the
// other whenever the lanes disagree -- for atan2 that was two divisions and two polynomials per call.  Every lane still
// evaluates exactly the expressions of the branchy form the oracle has: the value a lane does not take is computed and
// dropped, so the results are bit-identical.)
__device__ __forceinline__ float dm_atan01(float t) {
  const bool big = t > 0.414213567972183227539062f;
  const float reduced = (t - 1.0f) / (t + 1.0f);
  t = big ? reduced : t;
  const float base = big ? 0.785398185253143310546875f : 0.0f;
  float z = t * t;
  float p = 0x1.9e0c4cp-5f;
  p = p * z - 0x1.61601ep-4f;
  p = p * z + 0x1.c57fe2p-4f;
  p = p * z - 0x1.248a38p-3f;
  p = p * z + 0x1.99997cp-3f;
  p = p * z - 0x1.555556p-2f;
  return base + (t + t * (z * p));
}

__device__ __forceinline__ float dm_atan2(float y, float x) {
  const float pi = 3.1415927410125732421875f;
  const float pio2 = 1.57079637050628662109375f;
  float ax = fabsf(x), ay = fabsf(y);
  const bool zero = (ax == 0.0f && ay == 0.0f);
  const bool steep = !(ay <= ax);
  const float a = dm_atan01((steep ? ax : ay) / (steep ? ay : ax));   // ay / ax, or ax / ay for the steep half
  float r = steep ? pio2 - a : a;
  if (x < 0.0f) r = pi - r;
  r = (y < 0.0f) ? -r : r;
  return zero ? 0.0f : r;
}

__device__ __forceinline__ float dm_acos(float x) {
  float s = sqrtf((1.0f - x) * (1.0f + x));
  const float r = dm_atan2(s, x);
  return (x >= 1.0f) ? 0.0f : ((x <= -1.0f) ? 3.1415927410125732421875f : r);
}

}  // namespace ptd
