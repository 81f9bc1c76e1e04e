// pt_trace_v1.h -- PROFILING BUILD ONLY: the round-2 trace kernel (one phase: a lane whose path ends is re-generated
// in place, with only the idle lanes active), kept as the A/B baseline of the two-phase kernel in ../pt_trace.h
// (PTMI_TRACE_KERNEL=v1 selects it per launch; scripts/ab_trace.py).
#pragma once
#include "pt_trace.h"

namespace ptd {

constexpr uint32_t kRefillThresholdV1 = 20;  // refill once this many lanes are idle (or none is active)

__global__ __launch_bounds__(kTraceBlock) void trace_kernel_v1(const TraceParams P) {
  __shared__ uint32_t wg_count;
  __shared__ HitRow hit_table[kNumObjects];
  if (threadIdx.x == 0) { wg_count = 0; fill_hit_table(P, hit_table); }
  __syncthreads();

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t gw = blockIdx.x * (kTraceBlock / 64) + (threadIdx.x >> 6);
  const uint32_t region_base = blockIdx.x * P.region_cap;
  uint32_t cursor = 0;  // next unassigned j of this wave's strided path sequence
  // paths this wave owns: j -> idx = ((j / 64) * n_waves + gw) * 64 + (j % 64)
  const uint32_t n_chunks = (P.total_paths + 63u) / 64u;
  const uint32_t my_chunks = (n_chunks > gw) ? (n_chunks - gw + P.n_waves - 1u) / P.n_waves : 0u;
  const uint32_t my_paths = my_chunks * 64u;

  PathState st;
  uint32_t idx = 0;
  bool active = false;

  while (true) {
    const uint64_t act_mask = __ballot(active);
    const uint32_t n_active = (uint32_t)__popcll(act_mask);
    const bool more = cursor < my_paths;
    if (!more && n_active == 0) break;
    if (more && (n_active == 0 || 64u - n_active >= kRefillThresholdV1)) {
      // hand the next paths of the wave's sequence to the idle lanes (ballot + prefix count)
      const uint64_t idle = ~act_mask;
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
      if (!active) {
        const uint32_t j = cursor + rank;
        const uint32_t cand = ((j >> 6) * P.n_waves + gw) * 64u + (j & 63u);
        if (j < my_paths && cand < P.total_paths) {
          idx = cand;
          const uint32_t item = cand % P.n_items;
          const uint32_t iter = cand / P.n_items;
          float camx, camy;
          start_path(P, P.pix[item], P.sample_base + iter, st, camx, camy);
          active = true;
        }
      }
      cursor += 64u - n_active;
    }
    int res = STEP_CONTINUE;
    uint32_t length = 0;
    if (active) res = bounce(P, hit_table, st, length);
    const bool ended = active && res != STEP_CONTINUE;
    const bool escaped = active && res == STEP_ESCAPED;
    if (ended) {
      P.plen[idx] = (uint8_t)(length | (escaped ? 0x80u : 0u));
      active = false;
    }
    if (P.env_const) {
      if (escaped) {  // constant environment: total = env (.) T, no NIF
        P.rad_r[idx] = P.env_r * st.T.x;
        P.rad_g[idx] = P.env_g * st.T.y;
        P.rad_b[idx] = P.env_b * st.T.z;
      }
    } else {
      const uint64_t esc_mask = __ballot(escaped);
      if (esc_mask) {
        uint32_t base = 0;
        if (lane == (uint32_t)__ffsll((long long)esc_mask) - 1u) base = atomicAdd(&wg_count, (uint32_t)__popcll(esc_mask));
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
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) P.region_count[blockIdx.x] = wg_count;
}

}  // namespace ptd
