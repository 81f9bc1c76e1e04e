// diag/pt_trace_rounds.h -- PROFILING BUILD ONLY (round-4 experiment, not kept): the trace kernel's secondary phase in
// workgroup-synchronous rounds that regroup the paths by material through an LDS queue.  Included by pt_trace.h.
#pragma once

namespace ptd {

// ---- secondary phase in workgroup-synchronous rounds.
// The per-wave persistent loop kept its lanes 98 % occupied while the state list lasted, but (i) a trip ran the code of all
// three materials and of the escape one after the other, each for its few lanes (lane utilisation of the phase 45 %), (ii)
// a wave's last paths ran alone: 15 % of the trips at 25 % occupancy (depth 16: 20 % at 19 %, scripts/count_trace.py), and
// (iii) every trip paid a Philox block although below the roulette depth only diffuse and glass hits use one.
// Here a round is: every thread without a path takes the next entry of the state list; ALL paths are intersected (the one
// stretch that is the same for every path); a miss below the roulette depth is final; the rest is pushed into a 256-entry
// queue in LDS -- diffuse hits and the misses that still owe a roulette draw from the front, mirror and glass hits from
// the back -- and after a barrier thread i takes entry i, so that a wave shades ONE material (but for the two waves at the
// seams), draws its Philox block only where the bounce uses one, and the paths that go on stay in the registers of threads
// 0 .. n-1: the tail shrinks towards wave 0 instead of thinning out in every wave.  The roulette draw depends on
// (pixel, sample, depth) alone, so taking it after the intersection instead of before (codelets.cpp:176-183) changes no
// path; same device functions on the same values: bit-identical.
__device__ __forceinline__ uint32_t wave_slot(bool want, uint32_t lane, uint32_t* counter) {
  const uint64_t m = __ballot(want);
  if (!m) return 0u;
  uint32_t base = 0;
  const int first = __ffsll((long long)m) - 1;
  if ((int)lane == first) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = __shfl(base, first, 64);
  return base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ void secondary_rounds(const TraceParams& P, const HitRow* hit_table, uint32_t n_state, const float4* st0,
                                                 const float4* st1, const float4* st2, uint32_t region_base, uint32_t* wg_count) {
  __shared__ float4 q_lds[4][kTraceBlock];     // queue entries: (o, d.x) (d.yz, T.xy) (T.z, idx, depth | object << 8, hit distance) (pixel, sample, -, -)
  __shared__ uint32_t q_count[2][2];           // [round parity][front, back]
  __shared__ uint32_t q_cursor;                // next entry of the state list
  const uint32_t lane = threadIdx.x & 63u;
  if (threadIdx.x == 0) { q_count[0][0] = q_count[0][1] = q_count[1][0] = q_count[1][1] = 0; q_cursor = 0; }
  __syncthreads();
  PathState st;
  uint32_t idx = 0;
  bool have = false;
  for (uint32_t round = 0;; ++round) {
    uint32_t* const cnt = q_count[round & 1u];
    // ---- fill: idle threads take the next entries of the state list (depth 1: the first bounce was shaded in phase two)
    if (q_cursor < n_state) {                  // (stale reads only cost an atomic: the cursor never goes back)
      const uint32_t e = wave_slot(!have, lane, &q_cursor);
      if (!have && e < n_state) {
        const float4 a = st0[e], b = st1[e], c = st2[e];
        st.o = mk(a.x, a.y, a.z);
        st.d = mk(a.w, b.x, b.y);
        st.T = mk(b.z, b.w, c.x);
        idx = __float_as_uint(c.y);
        uint32_t item, iter;
        split_index<true>(P, idx, item, iter);
        st.pixel = P.pix[item];
        st.sample = P.sample_base + iter;
        st.depth = 1;
        have = true;
      }
    }
    // ---- every path: Scene::intersect (codelets.cpp:183)
    float tbest = 0.f;
    int best = -1;
    if (have) best = nearest_hit(P, st.o, st.d, tbest);
    const bool owes_roulette = st.depth >= P.roulette_depth;
    const bool escaped_now = have && best < 0 && !owes_roulette;      // :184-190 with the weight 1 of a bounce below the roulette depth
    if (escaped_now) P.plen[idx] = (uint8_t)((st.depth + 1u) | 0x80u);
    emit_escaped(P, escaped_now, st, idx, lane, region_base, wg_count);
    const bool queued = have && !escaped_now;
    const bool diffuse = best >= 0 && (__float_as_uint(hit_table[best >= 0 ? best : 0].colour.w) & 0xffu) == (uint32_t)MAT_DIFFUSE;
    const bool front = queued && (best < 0 || diffuse);
    const bool back = queued && !front;
    uint32_t pos = wave_slot(front, lane, &cnt[0]);
    const uint32_t pos_b = wave_slot(back, lane, &cnt[1]);
    if (back) pos = (uint32_t)kTraceBlock - 1u - pos_b;
    if (queued) {
      q_lds[0][pos] = make_float4(st.o.x, st.o.y, st.o.z, st.d.x);
      q_lds[1][pos] = make_float4(st.d.y, st.d.z, st.T.x, st.T.y);
      q_lds[2][pos] = make_float4(st.T.z, __uint_as_float(idx), __uint_as_float(st.depth | ((uint32_t)(best + 1) << 8)), tbest);
      q_lds[3][pos] = make_float4(__uint_as_float(st.pixel), __uint_as_float(st.sample), 0.f, 0.f);
    }
    have = false;
    __syncthreads();
    const uint32_t n_front = cnt[0], n_queue = n_front + cnt[1];
    if (n_queue == 0 && q_cursor >= n_state) break;                   // uniform: nothing queued, nothing left to take
    // ---- thread i takes entry i: front part (diffuse hits, misses that owe a roulette draw), then back part (mirror, glass)
    best = -1;
    if (threadIdx.x < n_queue) {
      const uint32_t at = threadIdx.x < n_front ? threadIdx.x : (uint32_t)kTraceBlock - 1u - (threadIdx.x - n_front);
      const float4 a = q_lds[0][at], b = q_lds[1][at], c = q_lds[2][at], d = q_lds[3][at];
      st.o = mk(a.x, a.y, a.z);
      st.d = mk(a.w, b.x, b.y);
      st.T = mk(b.z, b.w, c.x);
      idx = __float_as_uint(c.y);
      const uint32_t packed = __float_as_uint(c.z);
      st.depth = packed & 0xffu;
      best = (int)(packed >> 8) - 1;
      tbest = c.w;
      st.pixel = __float_as_uint(d.x);
      st.sample = __float_as_uint(d.y);
      have = true;
    }
    __syncthreads();                                                   // every entry is in registers: the queue may be refilled
    if (threadIdx.x == 0) { q_count[(round + 1u) & 1u][0] = 0; q_count[(round + 1u) & 1u][1] = 0; }   // last read before this round's first barrier
    // ---- the rest of the loop trip (codelets.cpp:176-180, 192-216): Philox block of the bounce where it is used
    bool escaped = false;
    if (have) {
      const bool roul = st.depth >= P.roulette_depth;
      const int type = best >= 0 ? (int)(__float_as_uint(hit_table[best].colour.w) & 0xffu) : -1;
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      if (roul || type == MAT_DIFFUSE || type == MAT_REFRACTIVE)
        philox4x32_10(st.pixel, st.sample, 1u + st.depth, 0x5054u, P.seed_lo, P.seed_hi, w);
      float rr = 1.0f;
      bool dead = false;
      if (roul) {                                                      // :176-180
        if (uniform01(w[0], P.samples_half) <= P.stop_prob) dead = true;
        rr = P.rr_factor;
      }
      uint32_t length = 0;
      int res;
      if (dead) { length = st.depth ? st.depth : 1u; res = STEP_DEAD; }
      else if (best < 0) { st.T = scale(st.T, rr); length = st.depth + 1u; res = STEP_ESCAPED; }   // :184-190
      else res = shade_hit(P, hit_table, st, best, tbest, w, rr, length);
      if (res != STEP_CONTINUE) {
        escaped = res == STEP_ESCAPED;
        P.plen[idx] = (uint8_t)(length | (escaped ? 0x80u : 0u));
        have = false;
      }
    }
    emit_escaped(P, escaped, st, idx, lane, region_base, wg_count);
  }
}

}  // namespace ptd
