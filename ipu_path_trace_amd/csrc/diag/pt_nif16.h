// pt_nif16.h -- nif_kernel_v4: nif_kernel_v3's pipeline on v_mfma_f32_16x16x32_f16.
//
// Same ring, same riders, same rounding points as v3 (pt_nif.h); only the MFMA shape and with it the operand
// layouts change.  A wave still owns 32 samples, now as two 16-sample B tiles (b = 0, 1), and a 32-feature output
// tile is two 16-row A tiles (ft = 0, 1): one k-step of 32 is 2 A pieces x 2 B fragments = four independent
// 16-cycle MFMAs, the same 64 pipe cycles as v3's pair of 32x32x16.  The smaller shape draws less power per
// FLOP, so the sustained clock under load is higher (MI355X_MICROARCH.md, DVFS).
//
// Layouts (lane = 16 qg + c):
//   A piece   lane holds W^T[row(c)][k(qg, e = 0..7)]                         (packed by pack_nif16, ptmi.hip)
//   B frag    lane holds X^T[k(qg, e)][sample c]
//   D tile    lane holds rows 4 qg + i (i = 0..3) of column c
// so after the epilogue of output tile j the lane holds features 32 j + 16 ft + 4 qg + i, which is taken as
// k-step j of the next layer with slot (qg, e) = feature 32 j + (e < 4 ? 4 qg + e : 16 + 4 qg + e - 4).
// Fourier features: k-step s', slot (qg, e): coordinate qg & 1 (u, v), frequency 4 (2 s' + (qg >> 1)) + (e & 3),
// sin for e < 4 and cos for e >= 4; frequency blocks past E are zero on both operands (48 features in 64 slots).
#pragma once
#include "pt_nif.h"

namespace ptd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int H, int E, int WAVES, int TPS>
struct NifV4Geometry {
  static constexpr int KS = H / 32;            // k-steps of 32 over a hidden activation vector
  static constexpr int NT = H / 32;            // 32-feature output tiles of a hidden layer
  static constexpr int FB = E / 4;             // frequency blocks of four
  static constexpr int IS = (FB + 1) / 2;      // k-steps over the (padded) Fourier features
  static constexpr int R = 3;
  static constexpr int SLAB_PIECES = ((TPS * 2 * (KS + IS) + WAVES - 1) / WAVES) * WAVES;
  static constexpr int PW = SLAB_PIECES / WAVES;
  static constexpr int T0 = (SLAB_PIECES / (2 * IS)) < NT ? (SLAB_PIECES / (2 * IS)) : NT;   // layer-0 tiles per slab
  static constexpr int T0_LAST = (NT % T0) ? (NT % T0) : T0;
  static constexpr int min3(int a, int b, int c) { return a < b ? (a < c ? a : c) : (b < c ? b : c); }
  // fewest pieces any wave issues for any slab (head: KS pieces of one 16-row tile; last layer-0 slab; hidden)
  static constexpr int MINP = min3(KS / WAVES, (T0_LAST * 2 * IS) / WAVES, (TPS * 2 * KS) / WAVES);
  static constexpr int SLOT_BYTES = SLAB_PIECES * 1024;
  static constexpr int SCAN_BYTES = ((kMaxRegions + 1) * 4 + 511) / 512 * 512;
  static constexpr int MAX_LAYERS = 8;
  static constexpr int BIAS_BYTES = (((MAX_LAYERS - 1) * NT + 1) * 64 + 511) / 512 * 512;
  static constexpr int LDS_BYTES = SCAN_BYTES + BIAS_BYTES + R * SLOT_BYTES;
  static_assert(H % 32 == 0 && E % 4 == 0, "unsupported NIF shape");
  static_assert(NT % TPS == 0, "tiles per stage must divide the tile count");
  static_assert(LDS_BYTES <= 160 * 1024, "ring does not fit the LDS");
};

template <int H, int E, int WAVES, int TPS>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void nif_kernel_v4(const NifParams P) {
  using G = NifV4Geometry<H, E, WAVES, TPS>;
  constexpr int KS = G::KS, NT = G::NT, IS = G::IS, R = G::R, PW = G::PW, T0 = G::T0;
  constexpr int TS = 32;
  constexpr int THREADS = 64 * WAVES;
  constexpr int NG = KS;                      // one k-step (2 A pieces, 4 MFMAs) per group
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* tile_start = reinterpret_cast<uint32_t*>(smem);
  char* bias_lds = smem + G::SCAN_BYTES;
  char* ring = bias_lds + G::BIAS_BYTES;
  const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15;
  const int qg = lane >> 4;

  {  // exclusive scan of per-region wave-tile counts (partials live in the not-yet-used ring)
    uint32_t* partial = reinterpret_cast<uint32_t*>(ring);
    const uint32_t per = (P.n_regions + THREADS - 1u) / THREADS;
    uint32_t sum = 0;
    for (uint32_t i = 0; i < per; ++i) {
      uint32_t r = threadIdx.x * per + i;
      if (r < P.n_regions) sum += (P.region_count[r] + TS - 1u) / TS;
    }
    partial[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t run = 0;
      for (int i = 0; i < THREADS; ++i) { uint32_t t = partial[i]; partial[i] = run; run += t; }
      tile_start[P.n_regions] = run;
    }
    __syncthreads();
    uint32_t run = partial[threadIdx.x];
    for (uint32_t i = 0; i < per; ++i) {
      uint32_t r = threadIdx.x * per + i;
      if (r < P.n_regions) { tile_start[r] = run; run += (P.region_count[r] + TS - 1u) / TS; }
    }
    const uint32_t n_btiles = P.bias_base[P.n_layers - 1] + 1u;
    for (uint32_t i = threadIdx.x; i < n_btiles * 4u; i += THREADS)
      reinterpret_cast<uint4*>(bias_lds)[i] = P.bpack[i];
    __syncthreads();
  }
  const uint32_t total_tiles = tile_start[P.n_regions];
  const uint32_t wg_tiles = (total_tiles + WAVES - 1u) / WAVES;
  if (blockIdx.x >= wg_tiles) return;   // whole workgroup leaves together

  // Slab stream of one pass: layer 0 in groups of T0 tiles, hidden layers in groups of TPS, head alone.
  // ppt = pieces per output tile: 2 per k-step (ft = 0, 1); the head has a single 16-row tile (1 per k-step).
  const uint32_t n_layers = P.n_layers;
  uint32_t pf_l = 0, pf_j = 0, pf_q = 0;
  uint32_t pf_first = 0, pf_cnt = 1, pf_slot = 0, pf_ntile = 0;
  auto slab_begin = [&]() {
    const bool head = pf_l + 1 == n_layers;
    const uint32_t ksteps = (pf_l == 0) ? (uint32_t)IS : KS + (((P.concat_mask >> pf_l) & 1u) ? IS : 0);
    const uint32_t ppt = head ? ksteps : 2u * ksteps;
    const uint32_t tiles_l = head ? 1u : (uint32_t)NT;
    const uint32_t group = (pf_l == 0) ? (uint32_t)T0 : (uint32_t)TPS;
    pf_ntile = (tiles_l - pf_j < group) ? tiles_l - pf_j : group;
    pf_cnt = pf_ntile * ppt;
    pf_first = P.piece_base[pf_l] + pf_j * ppt;
    pf_slot = ring_lds + (pf_q % R) * G::SLOT_BYTES;
  };
  auto slab_piece = [&](int i) {
    const uint32_t piece = (uint32_t)wave + (uint32_t)WAVES * i;
    if (piece >= pf_cnt) return;
    const char* src = reinterpret_cast<const char*>(P.wpack) + ((size_t)(pf_first + piece) * 1024 + lane * 16);
    glds16(src, pf_slot + piece * 1024u);
  };
  auto slab_end = [&]() {
    const uint32_t tiles_l = (pf_l + 1 == n_layers) ? 1u : (uint32_t)NT;
    pf_q += 1;
    pf_j += pf_ntile;
    if (pf_j == tiles_l) { pf_j = 0; pf_l = (pf_l + 1 == n_layers) ? 0u : pf_l + 1; }
  };
#pragma unroll
  for (int i = 0; i < R - 1; ++i) {
    slab_begin();
#pragma unroll
    for (int k = 0; k < PW; ++k) slab_piece(k);
    slab_end();
  }
  unsigned long long t_cycles = 0, t_real = 0;   // in-kernel clock stamps of workgroup 0 (this kernel exists in the profiling build only)
  if (blockIdx.x == 0 && threadIdx.x == 0) { t_cycles = __builtin_amdgcn_s_memtime(); t_real = __builtin_amdgcn_s_memrealtime(); }
  uint32_t q = 0;
  int pf_next = PW;

  auto stage_sync = [&]() -> const uint4* {
#pragma unroll
    for (int k = 0; k < PW; ++k) if (k >= pf_next) slab_piece(k);
    if (pf_next != PW + 1) slab_end();
    // counted wait: every wave issues at least MINP pieces per slab (see v3)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::MINP * (R - 2)) : "memory");
    asm volatile("s_barrier" ::: "memory");
    slab_begin();
    pf_next = 0;
    const uint4* slot = reinterpret_cast<const uint4*>(ring + (q % R) * G::SLOT_BYTES) + lane;
    q += 1;
    return slot;
  };
  auto dma_slot = [&]() {
    if (pf_next < PW) { slab_piece(pf_next); pf_next += 1; }
  };
  pf_next = PW + 1;

  for (uint32_t g = blockIdx.x; g < wg_tiles; g += gridDim.x) {
    const uint32_t wt = (uint32_t)WAVES * g + wave;
    const bool tile_valid = wt < total_tiles;
    uint32_t lo = 0, hi = P.n_regions;
    const uint32_t wts = tile_valid ? wt : 0u;
    while (hi - lo > 1u) {
      uint32_t mid = (lo + hi) >> 1;
      if (tile_start[mid] <= wts) lo = mid; else hi = mid;
    }
    const uint32_t region = lo;
    const uint32_t local = (wts - tile_start[region]) * TS;
    const uint32_t count = tile_valid ? P.region_count[region] : 0u;
    const uint32_t qi0 = region * P.region_cap + local + c;   // sample of B tile 0; tile 1 is 16 further

    half8 in[2][IS];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float coord = 0.5f;
      if (local + 16 * b + c < count) coord = (qg & 1) ? P.q_v[qi0 + 16 * b] : P.q_u[qi0 + 16 * b];
      const float x = (coord - 1.0f) * 2.0f;
#pragma unroll
      for (int s = 0; s < IS; ++s) {
        const int fb = 2 * s + (qg >> 1);
        half8 f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a = (float)(_Float16)(x * (float)(1u << (4 * fb + k)));
          float sn, cs;
          fast_sincos(a, sn, cs);
          f[k] = (_Float16)sn;
          f[4 + k] = (_Float16)cs;
        }
        if (fb >= G::FB) f = (half8)((_Float16)0.0f);   // padding slots: zero on both operands
        in[b][s] = f;
      }
    }

    half8 cur[2][KS], nxt[2][KS];

    // Epilogue of one output tile in four chunks, chunk ch = 2 b + ft: fp32 -> fp16 (RNE), + bias in fp16,
    // ReLU (NifModel.cpp:314-325); fills elements 4 ft .. 4 ft + 3 of the next layer's fragment [b][j].
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    struct Pending {
      f32x4 acc[2][2];    // [ft][b]
      const char* bias;   // this lane's 16 bytes of packed bias: 4 halves for ft = 0, 4 for ft = 1
      uint32_t floor;     // packed fp16 pair: 0 (ReLU) or -inf (linear)
    };
    auto epi_begin = [&](Pending& p, const f32x4 (&acc)[2][2], uint32_t layer, int j) __attribute__((always_inline)) {
      p.acc[0][0] = acc[0][0]; p.acc[0][1] = acc[0][1]; p.acc[1][0] = acc[1][0]; p.acc[1][1] = acc[1][1];
      p.bias = bias_lds + ((size_t)(P.bias_base[layer] + j) * 4 + qg) * 16;
      p.floor = ((P.relu_mask >> layer) & 1u) ? 0u : 0xfc00fc00u;
    };
    auto epi_chunk = [&](const Pending& p, auto chc, half8& o0, half8& o1, uint2 bias_now) __attribute__((always_inline)) {
      constexpr int ch = decltype(chc)::value;
      constexpr int b = ch >> 1, ft = ch & 1;
      half8& o = b ? o1 : o0;
      const f32x4 a = p.acc[ft][b];
      half2v x0 = {(_Float16)a[0], (_Float16)a[1]};
      half2v x1 = {(_Float16)a[2], (_Float16)a[3]};
      {
        union { uint2 u; half2v hh[2]; } bb;   // layers without a bias carry packed zeros: x + 0 = x
        bb.u = bias_now;
        x0 = x0 + bb.hh[0];
        x1 = x1 + bb.hh[1];
      }
      asm("v_pk_max_f16 %0, %1, %2" : "=v"(x0) : "v"(x0), "s"(p.floor));
      asm("v_pk_max_f16 %0, %1, %2" : "=v"(x1) : "v"(x1), "s"(p.floor));
      o[4 * ft + 0] = x0[0]; o[4 * ft + 1] = x0[1];
      o[4 * ft + 2] = x1[0]; o[4 * ft + 3] = x1[1];
    };
    auto bias_at = [&](const Pending& p, int ch) -> uint2 { return *reinterpret_cast<const uint2*>(p.bias + (ch & 1) * 8); };
    auto epi_all = [&](const Pending& p, half8& o0, half8& o1) __attribute__((always_inline)) {
      epi_chunk(p, IC<0>{}, o0, o1, bias_at(p, 0));
      epi_chunk(p, IC<1>{}, o0, o1, bias_at(p, 1));
      epi_chunk(p, IC<2>{}, o0, o1, bias_at(p, 2));
      epi_chunk(p, IC<3>{}, o0, o1, bias_at(p, 3));
    };

    // One 32-feature output tile: per k-step two A pieces from LDS (two k-steps in flight, pinned by the empty
    // asm as in v3) and four independent MFMAs.
    auto tile_mma = [&](const uint4* wj, half8 (&src)[2][KS], bool concat, f32x4 (&acc)[2][2], auto&& after_group)
                        __attribute__((always_inline)) {
      acc[0][0] = (f32x4)(0.0f); acc[0][1] = (f32x4)(0.0f); acc[1][0] = (f32x4)(0.0f); acc[1][1] = (f32x4)(0.0f);
      half8 A[2][2];
      A[0][0] = as_half8(wj[0]);
      A[0][1] = as_half8(wj[64]);
      auto group = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g2 = decltype(gc)::value;
        if constexpr (g2 + 1 < NG) {
          A[(g2 + 1) & 1][0] = as_half8(wj[((g2 + 1) * 2 + 0) * 64]);
          A[(g2 + 1) & 1][1] = as_half8(wj[((g2 + 1) * 2 + 1) * 64]);
        }
        asm volatile("" : "+v"(A[g2 & 1][0]), "+v"(A[g2 & 1][1])::"memory");
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[g2 & 1][0], src[0][g2], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[g2 & 1][0], src[1][g2], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[g2 & 1][1], src[0][g2], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[g2 & 1][1], src[1][g2], acc[1][1], 0, 0, 0);
        after_group(gc);
      };
      for_each_index(std::make_integer_sequence<int, NG>{}, group);
      if (concat) {
        half8 t[IS][2];
#pragma unroll
        for (int s = 0; s < IS; ++s) {
          t[s][0] = as_half8(wj[((KS + s) * 2 + 0) * 64]);
          t[s][1] = as_half8(wj[((KS + s) * 2 + 1) * 64]);
        }
#pragma unroll
        for (int s = 0; s < IS; ++s) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t[s][0], in[0][s], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t[s][0], in[1][s], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t[s][1], in[0][s], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(t[s][1], in[1][s], acc[1][1], 0, 0, 0);
        }
      }
    };

    // ---- layer 0: Fourier features -> H, slabs of T0 tiles
    {
      const uint4* slot = nullptr;
      Pending pend;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (j % T0 == 0) slot = stage_sync();
        half8 a0[IS][2];
#pragma unroll
        for (int s = 0; s < IS; ++s) {
          a0[s][0] = as_half8(slot[(((j % T0) * IS + s) * 2 + 0) * 64]);
          a0[s][1] = as_half8(slot[(((j % T0) * IS + s) * 2 + 1) * 64]);
        }
        f32x4 acc[2][2];
        acc[0][0] = (f32x4)(0.0f); acc[0][1] = (f32x4)(0.0f); acc[1][0] = (f32x4)(0.0f); acc[1][1] = (f32x4)(0.0f);
#pragma unroll
        for (int s = 0; s < IS; ++s) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[s][0], in[0][s], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[s][0], in[1][s], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[s][1], in[0][s], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[s][1], in[1][s], acc[1][1], 0, 0, 0);
        }
        if (j > 0) epi_all(pend, cur[0][j - 1], cur[1][j - 1]);
        dma_slot();
        epi_begin(pend, acc, 0, j);
      }
      epi_all(pend, cur[0][NT - 1], cur[1][NT - 1]);
    }

    // ---- hidden layers; src/dst register sets alternate so nothing is copied
    auto hidden = [&](half8 (&src)[2][KS], half8 (&dst)[2][KS], uint32_t l) __attribute__((always_inline)) {
      const bool concat = (P.concat_mask >> l) & 1u;
      const uint32_t ppt = 2u * (KS + (concat ? IS : 0));
      Pending pend;
      uint2 b_next = {0u, 0u};
      const uint4* slot = nullptr;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (j % TPS == 0) slot = stage_sync();
        f32x4 acc[2][2];
        tile_mma(slot + (size_t)(j % TPS) * ppt * 64, src, concat, acc, [&](auto gc) __attribute__((always_inline)) {
          constexpr int g2 = decltype(gc)::value;
          if constexpr (g2 < 4) {
            if (j > 0) {
              const uint2 b_now = b_next;
              if constexpr (g2 < 3 && g2 + 1 < NG) b_next = bias_at(pend, g2 + 1);
              epi_chunk(pend, gc, dst[0][j - 1], dst[1][j - 1], b_now);
            }
          }
          if constexpr (NG >= 8) { if constexpr (g2 >= 4 && g2 < 4 + (PW + TPS - 1) / TPS) dma_slot(); }
          else dma_slot();
        });
        if (j > 0) {   // narrow networks: fewer than four groups per tile, finish the leftover chunks here
          if constexpr (NG < 2) epi_chunk(pend, IC<1>{}, dst[0][j - 1], dst[1][j - 1], bias_at(pend, 1));
          if constexpr (NG < 3) epi_chunk(pend, IC<2>{}, dst[0][j - 1], dst[1][j - 1], bias_at(pend, 2));
          if constexpr (NG < 4) epi_chunk(pend, IC<3>{}, dst[0][j - 1], dst[1][j - 1], bias_at(pend, 3));
        }
        epi_begin(pend, acc, l, j);
        b_next = bias_at(pend, 0);
      }
      epi_all(pend, dst[0][NT - 1], dst[1][NT - 1]);
    };
    {
      uint32_t l = 1;
      for (; l + 2 < n_layers; l += 2) {
        hidden(cur, nxt, l);
        hidden(nxt, cur, l + 1);
      }
      if (l + 1 < n_layers) {
        hidden(cur, nxt, l);
#pragma unroll
        for (int s = 0; s < KS; ++s) { cur[0][s] = nxt[0][s]; cur[1][s] = nxt[1][s]; }
      }
    }

    // ---- head: one 16-row tile (3 outputs), one A piece and two MFMAs per k-step
    {
      const uint32_t l = n_layers - 1;
      const bool concat = (P.concat_mask >> l) & 1u;
      const uint4* slot = stage_sync();
      f32x4 hacc[2];
      hacc[0] = (f32x4)(0.0f); hacc[1] = (f32x4)(0.0f);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const half8 a = as_half8(slot[s * 64]);
        hacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, cur[0][s], hacc[0], 0, 0, 0);
        hacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, cur[1][s], hacc[1], 0, 0, 0);
        if (s < PW) dma_slot();
      }
      if (concat) {
#pragma unroll
        for (int s = 0; s < IS; ++s) {
          const half8 a = as_half8(slot[(KS + s) * 64]);
          hacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, in[0][s], hacc[0], 0, 0, 0);
          hacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, in[1][s], hacc[1], 0, 0, 0);
        }
      }
      Pending pend;
      {
        f32x4 four[2][2];
        four[0][0] = hacc[0]; four[0][1] = hacc[1]; four[1][0] = hacc[0]; four[1][1] = hacc[1];
        epi_begin(pend, four, l, 0);
      }
      half8 o0, o1;
      epi_chunk(pend, IC<0>{}, o0, o1, bias_at(pend, 0));   // ft 0, b 0
      epi_chunk(pend, IC<2>{}, o0, o1, bias_at(pend, 0));   // ft 0, b 1
      const float mean[3] = {P.mean0, P.mean1, P.mean2};
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (qg == 0 && local + 16 * b + c < count) {
          const half8& ob = b ? o1 : o0;
          const uint32_t qi = qi0 + 16 * b;
          float bgr[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            float o = (float)ob[k] * P.max;
            o = o + mean[k];
            bgr[k] = P.log_tonemap ? __expf(o) : o;
          }
          if (P.out_bgr) {
            P.out_bgr[3 * (size_t)qi + 0] = bgr[0];
            P.out_bgr[3 * (size_t)qi + 1] = bgr[1];
            P.out_bgr[3 * (size_t)qi + 2] = bgr[2];
          } else {
            const uint32_t path = P.q_path[qi];
            P.rad_r[path] = bgr[2] * P.q_tr[qi];
            P.rad_g[path] = bgr[1] * P.q_tg[qi];
            P.rad_b[path] = bgr[0] * P.q_tb[qi];
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead DMA before the wave ends
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_nif_clock[0] = __builtin_amdgcn_s_memtime() - t_cycles;
    g_nif_clock[1] = __builtin_amdgcn_s_memrealtime() - t_real;
  }
}

}  // namespace ptd
