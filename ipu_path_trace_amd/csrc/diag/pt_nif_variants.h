// pt_nif_variants.h -- earlier generations of the NIF kernel, kept for A/B measurements only.
//
// Compiled only into the profiling build (-DPTMI_DIAG_BUILD; ptmi.hip: PTMI_NIF_VARIANT=1, PTMI_NIF_WIDE=fused).
//  * nif_kernel       v1: weights streamed from L2 into registers per k-step, 4 waves x 64 samples.
//  * nif_wide_kernel  hidden 512/1024 fused in one kernel with the activations of a 64-sample tile in LDS; bound by
//                     the weight stream at ~240 TFLOP/s, replaced by the layer-by-layer path of pt_nif_gemm.h.
// Same packing, rounding points and arithmetic as the product kernels in pt_nif.h.
#pragma once
#include "pt_nif.h"
#include "pt_nif_gemm.h"

namespace ptd {

template <int H, int E, int NB>
__global__ __launch_bounds__(256, 1) void nif_kernel(const NifParams P) {
  constexpr int KS = H / 16;   // k-steps over a hidden activation vector
  constexpr int NT = H / 32;   // 32-feature output tiles of a hidden layer
  constexpr int IS = E / 4;    // k-steps over the 4E Fourier features
  constexpr int TS = 32 * NB;  // samples per wave tile
  static_assert(H % 32 == 0 && E % 4 == 0, "unsupported NIF shape");

  __shared__ uint32_t tile_start[kMaxRegions + 1];
  __shared__ uint32_t partial[256];
  {
    // exclusive scan of per-region wave-tile counts (every workgroup redundantly; <= 2048 regions)
    const uint32_t per = (P.n_regions + 255u) / 256u;
    uint32_t sum = 0;
    for (uint32_t i = 0; i < per; ++i) {
      uint32_t r = threadIdx.x * per + i;
      if (r < P.n_regions) sum += (P.region_count[r] + TS - 1u) / TS;
    }
    partial[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t run = 0;
      for (int i = 0; i < 256; ++i) { uint32_t t = partial[i]; partial[i] = run; run += t; }
      tile_start[P.n_regions] = run;
    }
    __syncthreads();
    uint32_t run = partial[threadIdx.x];
    for (uint32_t i = 0; i < per; ++i) {
      uint32_t r = threadIdx.x * per + i;
      if (r < P.n_regions) { tile_start[r] = run; run += (P.region_count[r] + TS - 1u) / TS; }
    }
    __syncthreads();
  }
  const uint32_t total_tiles = tile_start[P.n_regions];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = lane & 31;
  const int h = lane >> 5;

  for (uint32_t wt = blockIdx.x * 4u + wave; wt < total_tiles; wt += gridDim.x * 4u) {
    // region holding wave tile wt (wave-uniform binary search in LDS)
    uint32_t lo = 0, hi = P.n_regions;
    while (hi - lo > 1u) {
      uint32_t mid = (lo + hi) >> 1;
      if (tile_start[mid] <= wt) lo = mid; else hi = mid;
    }
    const uint32_t region = lo;
    const uint32_t local = (wt - tile_start[region]) * TS;
    const uint32_t count = P.region_count[region];
    const uint32_t qbase = region * P.region_cap + local;

    // ---- encode (NifModel.cpp:200-216): lane half 0 makes the u features, half 1 the v features
    half8 in[NB][IS];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const uint32_t off = local + 32u * b + c;
      const uint32_t q = qbase + 32u * b + c;
      float coord = 0.5f;
      if (off < count) coord = h ? P.q_v[q] : P.q_u[q];
      const float x = (coord - 1.0f) * 2.0f;
#pragma unroll
      for (int s = 0; s < IS; ++s) {
        half8 f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a = (float)(_Float16)(x * (float)(1u << (4 * s + k)));
          float sn, cs;
          fast_sincos(a, sn, cs);
          f[k] = (_Float16)sn;
          f[4 + k] = (_Float16)cs;
        }
        in[b][s] = f;
      }
    }

    half8 cur[NB][KS], nxt[NB][KS];

    auto epilogue = [&](const f32x16& acc, half8& o0, half8& o1, uint32_t layer, int j) {
      half8 l8, h8;
#pragma unroll
      for (int i = 0; i < 8; ++i) { l8[i] = (_Float16)acc[i]; h8[i] = (_Float16)acc[8 + i]; }
      if (P.bias_mask & (1u << layer)) {
        const uint4* bp = P.bpack + ((size_t)(P.bias_base[layer] + j) * 2 + h) * 2;
        l8 = l8 + as_half8(bp[0]);
        h8 = h8 + as_half8(bp[1]);
      }
      if (P.relu_mask & (1u << layer)) {
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        l8 = __builtin_elementwise_max(l8, z);
        h8 = __builtin_elementwise_max(h8, z);
      }
      o0 = l8;
      o1 = h8;
    };

    // ---- layer 0: 4E -> H
    {
      const uint4* wp = P.wpack + (size_t)P.piece_base[0] * 64 + lane;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x16 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = (f32x16)(0.0f);
#pragma unroll
        for (int s = 0; s < IS; ++s) {
          const half8 a = as_half8(wp[(size_t)(j * IS + s) * 64]);
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, in[b][s], acc[b], 0, 0, 0);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) epilogue(acc[b], cur[b][2 * j], cur[b][2 * j + 1], 0, j);
      }
    }

    // ---- hidden layers 1 .. n_layers-2: H (+4E) -> H
    for (uint32_t l = 1; l + 1 < P.n_layers; ++l) {
      const bool concat = (P.concat_mask >> l) & 1u;
      const uint32_t ksteps = KS + (concat ? IS : 0);
      const uint4* wp = P.wpack + (size_t)P.piece_base[l] * 64 + lane;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x16 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = (f32x16)(0.0f);
        const uint4* wj = wp + (size_t)j * ksteps * 64;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const half8 a = as_half8(wj[(size_t)s * 64]);
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, cur[b][s], acc[b], 0, 0, 0);
        }
        if (concat) {
#pragma unroll
          for (int s = 0; s < IS; ++s) {
            const half8 a = as_half8(wj[(size_t)(KS + s) * 64]);
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, in[b][s], acc[b], 0, 0, 0);
          }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) epilogue(acc[b], nxt[b][2 * j], nxt[b][2 * j + 1], l, j);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int s = 0; s < KS; ++s) cur[b][s] = nxt[b][s];
    }

    // ---- head: H (+4E) -> 3 (one 32-row tile, rows 0..2 used), decode, apply to the path
    {
      const uint32_t l = P.n_layers - 1;
      const bool concat = (P.concat_mask >> l) & 1u;
      const uint4* wj = P.wpack + (size_t)P.piece_base[l] * 64 + lane;
      f32x16 acc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = (f32x16)(0.0f);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const half8 a = as_half8(wj[(size_t)s * 64]);
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, cur[b][s], acc[b], 0, 0, 0);
      }
      if (concat) {
#pragma unroll
        for (int s = 0; s < IS; ++s) {
          const half8 a = as_half8(wj[(size_t)(KS + s) * 64]);
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, in[b][s], acc[b], 0, 0, 0);
        }
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        half8 o0, o1;
        epilogue(acc[b], o0, o1, l, 0);
        const uint32_t off = local + 32u * b + c;
        if (h == 0 && off < count) {  // rows 0..3 of the tile live in lane half 0, registers 0..3
          // buildDecodeOutput (NifModel.cpp:226-242): cast f32, * max, + (mean - eps), exp
          float bgr[3];
          const float mean[3] = {P.mean0, P.mean1, P.mean2};
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            float o = (float)o0[k] * P.max;
            o = o + mean[k];
            bgr[k] = P.log_tonemap ? __expf(o) : o;
          }
          const uint32_t q = qbase + 32u * b + c;
          if (P.out_bgr) {
            P.out_bgr[3 * (size_t)q + 0] = bgr[0];
            P.out_bgr[3 * (size_t)q + 1] = bgr[1];
            P.out_bgr[3 * (size_t)q + 2] = bgr[2];
          } else {
            // PostProcessEscapedRays (codelets.cpp:378): clr = (bgr[2], bgr[1], bgr[0]); then the
            // forward form of the AccumulateContributions fold: total = env (.) T.
            const uint32_t path = P.q_path[q];
            P.rad_r[path] = bgr[2] * P.q_tr[q];
            P.rad_g[path] = bgr[1] * P.q_tg[q];
            P.rad_b[path] = bgr[0] * P.q_tb[q];
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------- wide layers (hidden 512 / 1024: BASELINE config C5)
//
// A 1024-wide activation vector does not fit a wave's registers, so here the four waves of a workgroup share one
// tile of 64 samples whose activations live in LDS as ready-made B fragments ([k-step][b][lane] 16 B, 128 KiB at
// hidden 1024).  Each wave owns every fourth 32-feature output tile of a layer, streams that tile's weight pieces
// straight from L2 (no other wave needs them), keeps its results in registers until every wave has finished
// reading the layer's input, then writes them back into the same LDS image for the next layer.  Same packing,
// same rounding points and same arithmetic as the register-resident kernels.
template <int H, int E>
__global__ __launch_bounds__(256, 1) void nif_wide_kernel(const NifParams P) {
  constexpr int KS = H / 16, NT = H / 32, IS = E / 4, NB = 2, TS = 32 * NB, NTW = NT / 4;
  static_assert(NT % 4 == 0, "output tiles are dealt to four waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* act = reinterpret_cast<uint4*>(smem);                               // [KS][NB][64] uint4
  uint32_t* tile_start = reinterpret_cast<uint32_t*>(smem + (size_t)KS * NB * 1024);
  uint32_t* partial = tile_start + kMaxRegions + 1;
  {
    const uint32_t per = (P.n_regions + 255u) / 256u;
    uint32_t sum = 0;
    for (uint32_t i = 0; i < per; ++i) {
      uint32_t r = threadIdx.x * per + i;
      if (r < P.n_regions) sum += (P.region_count[r] + TS - 1u) / TS;
    }
    partial[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t run = 0;
      for (int i = 0; i < 256; ++i) { uint32_t t = partial[i]; partial[i] = run; run += t; }
      tile_start[P.n_regions] = run;
    }
    __syncthreads();
    uint32_t run = partial[threadIdx.x];
    for (uint32_t i = 0; i < per; ++i) {
      uint32_t r = threadIdx.x * per + i;
      if (r < P.n_regions) { tile_start[r] = run; run += (P.region_count[r] + TS - 1u) / TS; }
    }
    __syncthreads();
  }
  const uint32_t total_tiles = tile_start[P.n_regions];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;

  for (uint32_t wt = blockIdx.x; wt < total_tiles; wt += gridDim.x) {
    uint32_t lo = 0, hi = P.n_regions;
    while (hi - lo > 1u) {
      uint32_t mid = (lo + hi) >> 1;
      if (tile_start[mid] <= wt) lo = mid; else hi = mid;
    }
    const uint32_t local = (wt - tile_start[lo]) * TS;
    const uint32_t count = P.region_count[lo];
    const uint32_t qbase = lo * P.region_cap + local;

    half8 in[NB][IS];   // every wave encodes the tile's 64 samples for itself
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const uint32_t off = local + 32u * b + c;
      float coord = 0.5f;
      if (off < count) coord = h ? P.q_v[qbase + 32u * b + c] : P.q_u[qbase + 32u * b + c];
      const float x = (coord - 1.0f) * 2.0f;
#pragma unroll
      for (int s = 0; s < IS; ++s) {
        half8 f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a = (float)(_Float16)(x * (float)(1u << (4 * s + k)));
          float sn, cs;
          fast_sincos(a, sn, cs);
          f[k] = (_Float16)sn;
          f[4 + k] = (_Float16)cs;
        }
        in[b][s] = f;
      }
    }

    auto epilogue = [&](const f32x16& acc, half8& o0, half8& o1, uint32_t layer, uint32_t j) {
      half8 l8, h8;
#pragma unroll
      for (int i = 0; i < 8; ++i) { l8[i] = (_Float16)acc[i]; h8[i] = (_Float16)acc[8 + i]; }
      if (P.bias_mask & (1u << layer)) {
        const uint4* bp = P.bpack + ((size_t)(P.bias_base[layer] + j) * 2 + h) * 2;
        l8 = l8 + as_half8(bp[0]);
        h8 = h8 + as_half8(bp[1]);
      }
      if (P.relu_mask & (1u << layer)) {
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        l8 = __builtin_elementwise_max(l8, z);
        h8 = __builtin_elementwise_max(h8, z);
      }
      o0 = l8;
      o1 = h8;
    };
    // k-steps over the LDS-resident activations, weights from L2, four pieces in flight per group
    auto mma_act = [&](const uint4* wj, f32x16 (&acc)[NB]) __attribute__((always_inline)) {
      // weight pieces come from L2 with ~1 us latency: two groups of GW pieces in flight (register double buffer)
      constexpr int GW = 8;
      static_assert(KS % (2 * GW) == 0, "k-steps are consumed in pairs of groups");
      uint4 w0[GW], w1[GW];
#pragma unroll
      for (int i = 0; i < GW; ++i) w0[i] = wj[(size_t)i * 64];
      for (int s0 = 0; s0 < KS; s0 += 2 * GW) {
#pragma unroll
        for (int i = 0; i < GW; ++i) w1[i] = wj[(size_t)(s0 + GW + i) * 64];
#pragma unroll
        for (int i = 0; i < GW; ++i) {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const half8 bf = as_half8(act[((size_t)(s0 + i) * NB + b) * 64 + lane]);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(w0[i]), bf, acc[b], 0, 0, 0);
          }
        }
        if (s0 + 2 * GW < KS) {
#pragma unroll
          for (int i = 0; i < GW; ++i) w0[i] = wj[(size_t)(s0 + 2 * GW + i) * 64];
        }
#pragma unroll
        for (int i = 0; i < GW; ++i) {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const half8 bf = as_half8(act[((size_t)(s0 + GW + i) * NB + b) * 64 + lane]);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_half8(w1[i]), bf, acc[b], 0, 0, 0);
          }
        }
      }
    };
    auto mma_in = [&](const uint4* wj, f32x16 (&acc)[NB]) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < IS; ++s) {
        const half8 a = as_half8(wj[(size_t)s * 64]);
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, in[b][s], acc[b], 0, 0, 0);
      }
    };

    half8 out[NTW][2][NB];
    auto write_back = [&]() {   // results become the next layer's B fragments: k-steps 2j, 2j+1
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int j = wave + 4 * t;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            union { half8 hh; uint4 u; } cv;
            cv.hh = out[t][sub][b];
            act[((size_t)(2 * j + sub) * NB + b) * 64 + lane] = cv.u;
          }
      }
    };

    // ---- layer 0 (features in registers)
    {
      const uint4* wp = P.wpack + (size_t)P.piece_base[0] * 64 + lane;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int j = wave + 4 * t;
        f32x16 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = (f32x16)(0.0f);
        mma_in(wp + (size_t)j * IS * 64, acc);
#pragma unroll
        for (int b = 0; b < NB; ++b) epilogue(acc[b], out[t][0][b], out[t][1][b], 0, j);
      }
      __syncthreads();   // previous tile's readers are done with the LDS image
      write_back();
      __syncthreads();
    }
    // ---- hidden layers
    for (uint32_t l = 1; l + 1 < P.n_layers; ++l) {
      const bool concat = (P.concat_mask >> l) & 1u;
      const uint32_t ksteps = KS + (concat ? IS : 0);
      const uint4* wp = P.wpack + (size_t)P.piece_base[l] * 64 + lane;
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int j = wave + 4 * t;
        f32x16 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = (f32x16)(0.0f);
        const uint4* wj = wp + (size_t)j * ksteps * 64;
        mma_act(wj, acc);
        if (concat) mma_in(wj + (size_t)KS * 64, acc);
#pragma unroll
        for (int b = 0; b < NB; ++b) epilogue(acc[b], out[t][0][b], out[t][1][b], l, j);
      }
      __syncthreads();   // every wave has read this layer's input
      write_back();
      __syncthreads();
    }
    // ---- head (one 32-row tile, rows 0..2 used): wave 0
    if (wave == 0) {
      const uint32_t l = P.n_layers - 1;
      const bool concat = (P.concat_mask >> l) & 1u;
      const uint4* wj = P.wpack + (size_t)P.piece_base[l] * 64 + lane;
      f32x16 acc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = (f32x16)(0.0f);
      mma_act(wj, acc);
      if (concat) mma_in(wj + (size_t)KS * 64, acc);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        half8 o0, o1;
        epilogue(acc[b], o0, o1, l, 0);
        const uint32_t off = local + 32u * b + c;
        if (h == 0 && off < count) {
          float bgr[3];
          const float mean[3] = {P.mean0, P.mean1, P.mean2};
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            float o = (float)o0[k] * P.max;
            o = o + mean[k];
            bgr[k] = P.log_tonemap ? __expf(o) : o;
          }
          const uint32_t q = qbase + 32u * b + c;
          if (P.out_bgr) {
            P.out_bgr[3 * (size_t)q + 0] = bgr[0];
            P.out_bgr[3 * (size_t)q + 1] = bgr[1];
            P.out_bgr[3 * (size_t)q + 2] = bgr[2];
          } else {
            const uint32_t path = P.q_path[q];
            P.rad_r[path] = bgr[2] * P.q_tr[q];
            P.rad_g[path] = bgr[1] * P.q_tg[q];
            P.rad_b[path] = bgr[0] * P.q_tb[q];
          }
        }
      }
    }
  }
}

// ================================================================ wide-NIF layer kernel variants (profiling build only)
// nifg_layer_v1_kernel: the round-1 layer kernel (every wave interleaves its own loads, reads and MFMAs), kept as the A/B
// baseline and for its ablation bits.  nifg_layer_ld_kernel: the ping-pong kernel with four dedicated loader waves --
// measured 2.5 % SLOWER than ping-pong alone (profiles/r02_c5_ablation.txt).
// DIAG (timing-only builds, results invalid): bit 0 = no loads into the ring, bit 1 = no LDS reads of fragments,
// bit 2 = no barrier, bit 3 = every load from one L2-hot piece, bit 4 = activation loads from one L2-hot piece.
// (Tried and dropped: `nt` on the activation stream, -2.5 %; profiles/r01_g_c5_ablation.txt.)
// One dense layer over a chunk: D[256 features x 256 samples] per workgroup pass, 8 waves of 128 x 64
// (4 x 2 accumulator tiles of 32 x 32).  Weights and activations arrive by LDS-DMA into a ring of four stages of
// two k-steps (8 + 8 pieces each); every wave issues exactly four pieces per stage (two paired loads), so one counted s_waitcnt
// covers the ring (see nif_kernel_v3), and the loader's cursor runs ahead across output blocks, so a block's
// epilogue stores overlap the next block's first loads.
//
// Block order: workgroup g sits on XCD g % 8; the n_ftiles / 8 feature blocks of one sample block run at the
// same time on the same XCD, so the sample block's activation pieces are fetched from HBM / Infinity Cache once
// and hit that XCD's L2 for the other feature blocks; the layer's weights (<= 2 MiB) stay in every L2.
template <int DIAG>
__global__ __launch_bounds__(512, 2) void nifg_layer_v1_kernel(const NifGemmParams P) {
  constexpr int R = kGemmStages;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* bias_lds = smem;
  char* ring = smem + kGemmBiasBytes;
  const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int h = lane >> 5;

  const uint32_t ntiles = chunk_tile_count(P.total_tiles, P.tile0, P.chunk_tiles);
  const uint32_t nsb = (ntiles + 7u) / 8u;                 // sample blocks of 8 tiles
  const uint32_t FB = P.n_ftiles / 8u;                     // feature blocks of 8 tiles
  const uint32_t xcd = blockIdx.x & 7u, cidx = blockIdx.x >> 3, cpx = gridDim.x >> 3;
  const uint32_t fb = cidx % FB, sbi0 = cidx / FB, spx = cpx / FB;
  if (xcd + 8u * sbi0 >= nsb) return;                      // nothing for this workgroup (uniform)
  const uint32_t nks = P.ks_act + P.ks_in;
  const uint32_t nst = (nks + kGemmKps - 1u) / kGemmKps;

  for (uint32_t i = threadIdx.x; i < P.n_ftiles * 4u; i += 512u)
    reinterpret_cast<uint4*>(bias_lds)[i] = P.bpack[(size_t)P.bias_base * 4u + i];
  __syncthreads();

  // ---- loader: stage (pf_it, pf_st) -> ring slot pf_q % R.  Wave w loads weight tile w and sample tile w, both
  // k-steps of each with one M0 set-up and one uniform base address (SGPR pair) + a constant per-lane offset: a
  // tile's k-steps are contiguous in memory and in the slot ([A: tile][k][1 KiB] | [B: tile][k][1 KiB]), so the
  // instruction's immediate offset addresses the second piece on both sides.  (With an odd k-step count the last
  // stage's second piece is whatever follows in memory -- in bounds, never multiplied.)
  const uint32_t lane16 = (uint32_t)lane * 16u;
  uint32_t pf_it = 0, pf_st = 0, pf_q = 0;
  auto issue_pair = [&](int which) {   // 0: weights, 1: activations
    const uint32_t s0 = kGemmKps * pf_st;
    const char* base;
    if (which == 0) {
      const uint32_t j = fb * 8u + (uint32_t)wave;
      base = reinterpret_cast<const char*>(P.wpack) + ((size_t)(P.piece_base + j * nks + s0) << 10);
    } else {
      const uint32_t t = (xcd + 8u * (sbi0 + spx * pf_it)) * 8u + (uint32_t)wave;
      base = (s0 < P.ks_act) ? reinterpret_cast<const char*>(P.act_in) + (((size_t)t * P.act_stride + s0) << 10)
                             : reinterpret_cast<const char*>(P.feat) + (((size_t)t * P.feat_stride + (s0 - P.ks_act)) << 10);
    }
    if constexpr (DIAG & 8) base = reinterpret_cast<const char*>(P.wpack) + ((size_t)wave << 11);                    // every load L2-hot
    if constexpr (DIAG & 16) { if (which) base = reinterpret_cast<const char*>(P.wpack) + ((size_t)wave << 11); }   // B loads L2-hot
    const uint32_t dst = ring_lds + (pf_q % R) * kGemmStageBytes + (uint32_t)which * 16384u + ((uint32_t)wave << 11);
    if constexpr (!(DIAG & 1)) {
      uint32_t keep;
      asm volatile(
          "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
          "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(lane16), "s"(reinterpret_cast<uint64_t>(base)), "s"(dst)
          : "memory");
    }
  };
  auto stage_issued = [&]() {
    pf_q += 1;
    pf_st += 1;
    if (pf_st == nst) {
      // past the last block the cursor stays on it: those loads land in slots nobody reads (uniform load count)
      if (xcd + 8u * (sbi0 + spx * (pf_it + 1u)) < nsb) { pf_it += 1; pf_st = 0; }
      else pf_st = nst - 1u;
    }
  };
#pragma unroll
  for (int k = 0; k < R - 1; ++k) {
    issue_pair(0);
    issue_pair(1);
    stage_issued();
  }

  // Consumer.  A stage's fragments are read one half-stage ahead of their MFMAs, across the barrier: the barrier sits
  // in the MIDDLE of stage q (between its two k-steps) and certifies that stage q + 1 has landed, so the first
  // k-step's fragments of stage q + 1 are fetched under the second k-step's MFMAs of stage q, and the second
  // k-step's fragments under the first k-step's MFMAs.  Neither the barrier skew nor the burst of 8 waves x 6 LDS
  // reads behind it is then followed by an MFMA that waits for it.
  uint32_t q = 0;            // consumer stage
  uint32_t since_store = 2;  // stages since the last epilogue's 16 stores entered the vmcnt queue
  half8 A0[4], B0[2], A1[4], B1[2];
  {
    if (nst < 4u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    const uint4* slot = reinterpret_cast<const uint4*>(ring) + lane;
#pragma unroll
    for (int a = 0; a < 4; ++a) A0[a] = as_half8(slot[(4 * wm + a) * 128]);
#pragma unroll
    for (int b = 0; b < 2; ++b) B0[b] = as_half8(slot[1024 + (2 * wn + b) * 128]);
  }
  for (uint32_t it = 0;; ++it) {
    const uint32_t sb = xcd + 8u * (sbi0 + spx * it);
    if (sb >= nsb) break;
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) { acc[a][0] = (f32x16)(0.0f); acc[a][1] = (f32x16)(0.0f); }

    for (uint32_t st = 0; st < nst; ++st) {
      const uint4* slot = reinterpret_cast<const uint4*>(ring + (q % R) * kGemmStageBytes) + lane;
      const uint4* next = reinterpret_cast<const uint4*>(ring + ((q + 1u) % R) * kGemmStageBytes) + lane;
      q += 1;
      const bool two = kGemmKps * st + 1u < nks;
      // (with an odd k-step count the last stage's second half holds a copy of the first: read, not multiplied)
      if constexpr (!(DIAG & 2)) {
#pragma unroll
        for (int a = 0; a < 4; ++a) A1[a] = as_half8(slot[(4 * wm + a) * 128 + 64]);
#pragma unroll
        for (int b = 0; b < 2; ++b) B1[b] = as_half8(slot[1024 + (2 * wn + b) * 128 + 64]);
      } else {
#pragma unroll
        for (int a = 0; a < 4; ++a) A1[a] = A0[a];
        B1[0] = B0[1]; B1[1] = B0[0];
      }
      asm volatile("" : "+v"(A0[0]), "+v"(A0[1]), "+v"(A0[2]), "+v"(A0[3]), "+v"(B0[0]), "+v"(B0[1])::"memory");
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[a], B0[0], acc[a][0], 0, 0, 0);
        acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[a], B0[1], acc[a][1], 0, 0, 0);
      }
      issue_pair(0);
#pragma unroll
      for (int a = 2; a < 4; ++a) {
        acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[a], B0[0], acc[a][0], 0, 0, 0);
        acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[a], B0[1], acc[a][1], 0, 0, 0);
      }
      issue_pair(1);
      stage_issued();
      // My pieces of stage q + 1 have landed when at most the younger operations are outstanding: two stages of
      // four loads, plus the previous block's 16 stores while they are younger than the stage awaited.  All of
      // this stage's fragments are in registers (lgkmcnt(0)), so behind the barrier its slot is free as well.
      if (nst < 4u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (since_store < 2u) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if constexpr (!(DIAG & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      since_store += 1;
      if constexpr (!(DIAG & 2)) {
#pragma unroll
        for (int a = 0; a < 4; ++a) A0[a] = as_half8(next[(4 * wm + a) * 128]);
#pragma unroll
        for (int b = 0; b < 2; ++b) B0[b] = as_half8(next[1024 + (2 * wn + b) * 128]);
      }
      if (two) {
        asm volatile("" : "+v"(A1[0]), "+v"(A1[1]), "+v"(A1[2]), "+v"(A1[3]), "+v"(B1[0]), "+v"(B1[1])::"memory");
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1[a], B1[0], acc[a][0], 0, 0, 0);
          acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1[a], B1[1], acc[a][1], 0, 0, 0);
        }
      }
    }

    // ---- epilogue: fp32 -> fp16 (RNE), + bias in fp16, ReLU; 16 whole pieces per wave, always stored (tiles past
    // the end of the queue land in the buffer's padding), so the store count the waits above assume is exact
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const uint32_t j = fb * 8u + 4u * wm + a;
      const uint4* bp = reinterpret_cast<const uint4*>(bias_lds) + ((size_t)j * 2 + h) * 2;
      const half8 b_lo = as_half8(bp[0]), b_hi = as_half8(bp[1]);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        half8 l8, h8;
#pragma unroll
        for (int i = 0; i < 8; ++i) { l8[i] = (_Float16)acc[a][b][i]; h8[i] = (_Float16)acc[a][b][8 + i]; }
        l8 = l8 + b_lo;
        h8 = h8 + b_hi;
        if (P.relu) {
          const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          l8 = __builtin_elementwise_max(l8, z);
          h8 = __builtin_elementwise_max(h8, z);
        }
        const uint32_t t = sb * 8u + 2u * wn + b;
        uint4* out = P.act_out + ((size_t)t * P.act_stride + 2u * j) * 64 + lane;
        union { half8 hh; uint4 u; } c0, c1;
        c0.hh = l8;
        c1.hh = h8;
        out[0] = c0.u;
        out[64] = c1.u;
      }
    }
    since_store = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the run-ahead loads before the wave ends
}

// ---------------------------------------------------------------- ping-pong + dedicated loader waves
//
// What holds nifg_layer_kernel back is the ISSUE of its LDS-DMA loads: with the loads removed (timing-only build) it
// runs 30 % faster, with the LDS fragment reads removed 2 % (profiles/r02_c5_ablation.txt).  An LDS-DMA instruction costs
// a wave that is also reading LDS and feeding the matrix pipe 100-185 cycles, a wave that does nothing else ~22
// (MI355X_MICROARCH.md, rows "LDS-DMA piece issue cost" and "ldsdma-fill").  This kernel's 166 VGPRs leave room for a
// third wave per SIMD, so four extra waves (8-11, one per SIMD) do ALL the loading -- stage S + 3 into the ring, a
// counted wait for their share of stage S + 1, and the same four phase barriers per stage -- and the eight compute waves
// of nifg_layer_kernel issue no vector-memory instruction in the loop at all.  (The register-resident kernel of the
// 320-wide NIF cannot do this: its waves need 216 VGPRs, two per SIMD.)
template <int DIAG>
__global__ __launch_bounds__(768, 3) void nifg_layer_ld_kernel(const NifGemmParams P) {
  constexpr int R = kGemmStages;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* bias_lds = smem;
  char* ring = smem + kGemmBiasBytes;
  const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  const uint32_t ntiles = chunk_tile_count(P.total_tiles, P.tile0, P.chunk_tiles);
  const uint32_t nsb = (ntiles + 7u) / 8u;                 // sample blocks of 8 tiles
  const uint32_t FB = P.n_ftiles / 8u;                     // feature blocks of 8 tiles
  const uint32_t xcd = blockIdx.x & 7u, cidx = blockIdx.x >> 3, cpx = gridDim.x >> 3;
  const uint32_t fb = cidx % FB, sbi0 = cidx / FB, spx = cpx / FB;
  const uint32_t sb0 = xcd + 8u * sbi0, sb_step = 8u * spx;
  if (sb0 >= nsb) return;                                  // nothing for this workgroup (uniform)
  const uint32_t n_blocks = (nsb - sb0 + sb_step - 1u) / sb_step;
  const uint32_t nks = P.ks_act + P.ks_in;
  const uint32_t nst = (nks + kGemmKps - 1u) / kGemmKps;
  const uint32_t total_stages = n_blocks * nst;

  for (uint32_t i = threadIdx.x; i < P.n_ftiles * 4u; i += 768u)
    reinterpret_cast<uint4*>(bias_lds)[i] = P.bpack[(size_t)P.bias_base * 4u + i];
  __syncthreads();

  auto phase_end = [&]() __attribute__((always_inline)) {
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);   // nothing, MFMAs included, moves across a phase boundary
  };

  if (wave >= 8) {
    // ---- loader wave l: weight tiles 2l, 2l+1 and sample tiles 2l, 2l+1 of every stage, both k-steps of a tile with
    // one M0 set-up, a uniform base (SGPR pair) + constant per-lane offset, and the immediate offset for the second piece
    const uint32_t l = (uint32_t)wave - 8u;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    uint32_t pf_it = 0, pf_st = 0, pf_q = 0;
    auto issue_stage = [&]() {
      const uint32_t s0 = kGemmKps * pf_st;
      const uint32_t slot = ring_lds + (pf_q % R) * kGemmStageBytes;
#pragma unroll
      for (int which = 0; which < 2; ++which) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint32_t tile = 2u * l + (uint32_t)i;
          const char* base;
          if (which == 0) {
            const uint32_t j = fb * 8u + tile;
            base = reinterpret_cast<const char*>(P.wpack) + ((size_t)(P.piece_base + j * nks + s0) << 10);
          } else {
            const uint32_t t = (sb0 + sb_step * pf_it) * 8u + tile;
            base = (s0 < P.ks_act) ? reinterpret_cast<const char*>(P.act_in) + (((size_t)t * P.act_stride + s0) << 10)
                                   : reinterpret_cast<const char*>(P.feat) + (((size_t)t * P.feat_stride + (s0 - P.ks_act)) << 10);
          }
          const uint32_t dst = slot + (uint32_t)which * 16384u + (tile << 11);
          if constexpr (!(DIAG & 1)) {
            uint32_t keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(lane16), "s"(reinterpret_cast<uint64_t>(base)), "s"(dst)
                : "memory");
          }
        }
      }
      pf_q += 1;
      pf_st += 1;
      if (pf_st == nst) {
        // past the last block the cursor stays on it: those loads land in slots nobody reads (uniform load count)
        if (pf_it + 1u < n_blocks) { pf_it += 1; pf_st = 0; }
        else pf_st = nst - 1u;
      }
    };
#pragma unroll
    for (int k = 0; k < R - 1; ++k) issue_stage();
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // stage 0 (8 loads per stage and loader wave)
    phase_end();                                          // certifies stage 0
    for (uint32_t s = 0; s < total_stages; ++s) {
      issue_stage();                                      // stage s + 3 -> the slot of stage s - 1 (readers done in phase 4s - 1)
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // my share of stage s + 1 has landed
      phase_end();                                        // end of phase 4s: stage s + 1 certified (first reader: phase 4s + 3)
      phase_end();
      phase_end();
      phase_end();
    }
    phase_end();                                          // the compute halves' one-phase offset
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the run-ahead loads before the wave ends
    return;
  }

  // ---- compute waves 0-7: the phase sequence of nifg_layer_kernel without any vector-memory issue in the loop
  const int wm = wave & 1, wn = wave >> 1;
  const int h = lane >> 5;
  const bool second = wave >= 4;
  half8 FA[4], FBv[2];
  auto read_frags = [&](const uint4* slot, int kk) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < 4; ++a) FA[a] = as_half8(slot[(4 * wm + a) * 128 + 64 * kk]);
#pragma unroll
    for (int b = 0; b < 2; ++b) FBv[b] = as_half8(slot[1024 + (2 * wn + b) * 128 + 64 * kk]);
  };
  uint32_t q = 0;
  phase_end();                                            // stage 0 certified by the loaders
  read_frags(reinterpret_cast<const uint4*>(ring) + lane, 0);
  if (second) phase_end();

  for (uint32_t it = 0; it < n_blocks; ++it) {
    const uint32_t sb = sb0 + sb_step * it;
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) { acc[a][0] = (f32x16)(0.0f); acc[a][1] = (f32x16)(0.0f); }

    for (uint32_t st = 0; st < nst; ++st) {
      const uint4* slot = reinterpret_cast<const uint4*>(ring + (q % R) * kGemmStageBytes) + lane;
      const uint4* next = reinterpret_cast<const uint4*>(ring + ((q + 1u) % R) * kGemmStageBytes) + lane;
      q += 1;
      const bool two = kGemmKps * st + 1u < nks;   // (an odd k-step count: the last stage has one k-step)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[a], FBv[0], acc[a][0], 0, 0, 0);
        acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[a], FBv[1], acc[a][1], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      phase_end();
      read_frags(slot, 1);
      phase_end();
      if (two) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[a], FBv[0], acc[a][0], 0, 0, 0);
          acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[a], FBv[1], acc[a][1], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
      }
      phase_end();
      read_frags(next, 0);
      phase_end();
    }

    // ---- epilogue: fp32 -> fp16 (RNE), + bias in fp16, ReLU; 16 whole pieces per wave (tiles past the end of the
    // queue land in the buffer's padding)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const uint32_t j = fb * 8u + 4u * wm + a;
      const uint4* bp = reinterpret_cast<const uint4*>(bias_lds) + ((size_t)j * 2 + h) * 2;
      const half8 b_lo = as_half8(bp[0]), b_hi = as_half8(bp[1]);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        half8 l8, h8;
#pragma unroll
        for (int i = 0; i < 8; ++i) { l8[i] = (_Float16)acc[a][b][i]; h8[i] = (_Float16)acc[a][b][8 + i]; }
        l8 = l8 + b_lo;
        h8 = h8 + b_hi;
        if (P.relu) {
          const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
          l8 = __builtin_elementwise_max(l8, z);
          h8 = __builtin_elementwise_max(h8, z);
        }
        const uint32_t t = sb * 8u + 2u * wn + b;
        uint4* out = P.act_out + ((size_t)t * P.act_stride + 2u * j) * 64 + lane;
        union { half8 hh; uint4 u; } c0, c1;
        c0.hh = l8;
        c1.hh = h8;
        out[0] = c0.u;
        out[64] = c1.u;
      }
    }
  }
  if (!second) phase_end();
}

}  // namespace ptd
