// pt_nif_variants.h -- earlier generations of the NIF kernel, kept for A/B measurements only.
//
// Compiled only into the profiling build (-DPTMI_DIAG_BUILD; ptmi.hip: PTMI_NIF_VARIANT=1, PTMI_NIF_WIDE=fused).
//  * nif_kernel       v1: weights streamed from L2 into registers per k-step, 4 waves x 64 samples.
//  * nif_wide_kernel  hidden 512/1024 fused in one kernel with the activations of a 64-sample tile in LDS; bound by
//                     the weight stream at ~240 TFLOP/s, replaced by the layer-by-layer path of pt_nif_gemm.h.
// Same packing, rounding points and arithmetic as the product kernels in pt_nif.h.
#pragma once
#include "pt_nif.h"

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

}  // namespace ptd
