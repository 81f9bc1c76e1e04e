// pt_nif_f32.h -- NIFs stored as float32 (Hdf5Model.cpp:109-133 accepts float32 variables): the layers run in float.
//
// The reference gives a matmul the type of its kernel (NifModel.cpp:314: poplin::matMul(..., l.kernel.type, ...)), adds the
// bias and applies ReLU in that type (:316-325), and casts the half-precision Fourier features to it (:211-216, :299-302).
// Rounds 1-2 rounded float32 weights to binary16 on upload and ran the fp16 kernels; this path keeps them in float:
// v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain in k order (cdna_hip_programming.md section 3, "FP32-input MFMA"), so a
// layer's output is the sequential fmaf sum the oracle forms.  It runs at the fp32 matrix rate (157 TFLOP/s, 1/16 of the fp16
// MFMA rate): a fidelity path for the models that need it, not a fast one -- the shipped NIFs are fp16
// (nif_models/urban_alley_01_4k_fp16_yuv).
//
// Layer by layer over chunks of the queue, activations in two HBM buffers in the layout the MFMA hands them over in
// ("packed", below); one wave computes 64 samples x 64 features, both operands loaded straight into the lane order the
// instruction takes (no LDS, no barrier).
//
// Packed activations.  A layer is computed transposed, D^T = W^T x in^T: operand A = a 32-feature x 2-input tile of W^T
// (lane (m, kk) holds W[k0 + kk][f0 + m]: two 128-byte rows of the row-major kernel per load), operand B = 2 inputs x 32
// samples (lane (n, kk) holds in[sample n][k0 + kk]).  A lane of the result then owns, for ITS sample n, the features
// 8 q + 4 kk + (0..3), q = 0..3, of a 32-feature tile: registers 4 q .. 4 q + 3 are four consecutive inputs of the next
// layer.  They are stored as one float4 per lane,
//     packed[((tile * (width / 8) + group) * 64 + lane) * 4 + c]  =  x[sample 32 tile + (lane & 31)][8 group + 4 (lane >> 5) + c],
// so a wave's store and the next layer's load are each ONE contiguous kilobyte, and two v_permlane32_swap turn the loaded
// (x, y, z, w) = ([k0|k4], [k1|k5], [k2|k6], [k3|k7]) (low | high half of the wave) into the B operands [k0|k1], [k2|k3],
// [k4|k5], [k6|k7]: eight inputs per 16-byte load, consumed in k order.  (A row-major layout costs 4 x the L2 -> L1 bytes:
// a lane's 16 bytes of a 128-byte line per load, the line gone from L1 before its next use -- measured 54 % MFMA busy
// at full clock; this layout: see DESIGN.md 4.2c.)
#pragma once
#include "pt_nif_gemm.h"

namespace ptd {

struct NifF32Params {
  const float* w;            // this layer's kernel, row-major [k_act + k_in][ldw]
  const float* bias;         // [ldw] (zeros where the layer has none)
  uint32_t ldw;              // padded output width (multiple of 32)
  uint32_t k_act, k_in;      // inputs taken from the previous activations / from the Fourier features (multiples of 16)
  uint32_t relu;
  uint32_t half_out;         // this layer's variables are binary16 (a mixed model): the matmul output is rounded to half and the
                             // bias is added in half (NifModel.cpp:314-321, a matmul takes its kernel's type)
  uint32_t cast_half;        // the NEXT layer is binary16 and this one is not: the activations it reads are cast to half
  const float* act_in;       // packed, width lda
  const float* feat;         // packed, width ldf
  float* act_out;            // packed, width ldw
  uint32_t lda, ldf;
  const uint32_t* total_tiles;
  uint32_t tile0, chunk_tiles;
};

__device__ __forceinline__ size_t nif32_packed(uint32_t tile, uint32_t width, uint32_t group, uint32_t lane) {
  return (((size_t)tile * (width >> 3) + group) * 64u + lane) * 4u;
}

// Fourier features of a chunk, packed floats of width 4 E: [sin u, sin v, cos u, cos v] x E, the half-precision trig of
// NifModel.cpp:200-216 cast to float (exact).  One wave per queue tile of 32 samples: lane = sample + 32 * coordinate.
template <int E>
__global__ __launch_bounds__(256) void nif32_encode_kernel(const NifParams P, const uint32_t* tile_start, uint32_t tile0,
                                                            uint32_t chunk_tiles, float* feat) {
  __shared__ uint32_t ts[kMaxRegions + 1];
  const uint32_t ntiles = chunk_tile_count(tile_start + P.n_regions, tile0, chunk_tiles);
  if (blockIdx.x * 4u >= ntiles) return;
  for (uint32_t i = threadIdx.x; i <= P.n_regions; i += 256u) ts[i] = tile_start[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  for (uint32_t lt = blockIdx.x * 4u + wave; lt < ntiles; lt += gridDim.x * 4u) {
    const TileRef r = find_tile(ts, P.n_regions, P, tile0 + lt);
    float coord = 0.5f;
    if (r.local + c < r.count) coord = h ? P.q_v[r.qbase + c] : P.q_u[r.qbase + c];
#pragma unroll
    for (int g = 0; g < E / 4; ++g) {
      const half8 f = fourier_group(coord, g, P.n_freq);
      const int fs = h * E + 4 * g, fc = 2 * E + fs;        // features fs..fs+3 = sin, fc..fc+3 = cos of this coordinate
      *reinterpret_cast<float4*>(feat + nif32_packed(lt, 4 * E, fs >> 3, c + 32 * ((fs >> 2) & 1))) =
          make_float4((float)f[0], (float)f[1], (float)f[2], (float)f[3]);
      *reinterpret_cast<float4*>(feat + nif32_packed(lt, 4 * E, fc >> 3, c + 32 * ((fc >> 2) & 1))) =
          make_float4((float)f[4], (float)f[5], (float)f[6], (float)f[7]);
    }
  }
}

// One dense layer in float over a chunk: out[s][f] = act(sum_k in[s][k] W[k][f] + b[f]), k over the previous activations and
// then (first layer, concat layers) over the features -- the order of the reference's concat(x, input) (NifModel.cpp:305-308).
//
// A workgroup = four waves = 256 samples x 64 features; a wave = 64 samples x 64 features (2 x 2 accumulator tiles).  The
// INPUTS go from memory straight into the B operand's lane order (one float4 per lane and tile, above).  The WEIGHTS are
// the same for every sample block, and every wave of the chip walks them in step: loaded per wave they hit the same two or
// three L2 channels at once (first version: 54 % MFMA busy at full clock).  So a workgroup brings each 16-input x 64-feature
// slice in once, through LDS (4 KB, double-buffered, one barrier per slice = 32 MFMAs per wave), and the four waves read
// their A operands from there (lanes read consecutive words: conflict-free).  A stage is 8 inputs = 16 MFMAs of 64 cycles;
// the next stage's inputs and the next slice's weights are in flight while this one's MFMAs run.  The feature blocks of one
// sample block are consecutive workgroups on ONE XCD (the block order below): they share the input tiles in that XCD's L2.
struct NifF32Stage { float4 x[2]; };

__device__ __forceinline__ void nif32_load_inputs(NifF32Stage& S, const float* in0, size_t tile_stride) {
  S.x[0] = *reinterpret_cast<const float4*>(in0);
  S.x[1] = *reinterpret_cast<const float4*>(in0 + tile_stride);
}

template <bool TWO>
__device__ __forceinline__ void nif32_mfma_stage(const NifF32Stage& S, const float* ws, f32x16 (&acc)[2][2]) {
  float x[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const auto xy = __builtin_amdgcn_permlane32_swap(__float_as_uint(S.x[t].x), __float_as_uint(S.x[t].y), false, false);
    const auto zw = __builtin_amdgcn_permlane32_swap(__float_as_uint(S.x[t].z), __float_as_uint(S.x[t].w), false, false);
    x[t][0] = __uint_as_float(xy[0]); x[t][1] = __uint_as_float(zw[0]);
    x[t][2] = __uint_as_float(xy[1]); x[t][3] = __uint_as_float(zw[1]);
  }
  float w[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s) {   // ws = &slice[stage rows][kk][m]: row 2 s + kk, features m and 32 + m
    w[s][0] = ws[2 * s * 64];
    w[s][1] = TWO ? ws[2 * s * 64 + 32] : 0.f;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {   // A = W^T (row = feature), B = inputs (column = sample): D[feature][sample]
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s][0], x[0][s], acc[0][0], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s][0], x[1][s], acc[1][0], 0, 0, 0);
    if (TWO) {
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s][1], x[0][s], acc[0][1], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s][1], x[1][s], acc[1][1], 0, 0, 0);
    }
  }
}

// Keeps the loads of a stage where they are written, a stage ahead of their use: the memory clobber stops the IR passes
// from sinking them to the first use, the scheduling barrier stops the machine scheduler.
#define NIF32_PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

__device__ __forceinline__ float nif32_hround(float x) { return (float)(_Float16)x; }   // RNE, subnormals kept: v_cvt_f16_f32 / v_cvt_f32_f16

// Bias, activation and the packed store of a block's accumulators.  MODE (mixed models only): 1 = this layer is binary16
// (round the sum to half, add the bias in half), 2 = cast the result to half for a binary16 layer that follows a float one.
template <bool TWO, bool RELU, int MODE = 0>
__device__ __forceinline__ void nif32_store(const NifF32Params& P, const f32x16 (&acc)[2][2], uint32_t tl, uint32_t f0, uint32_t lane) {
  float4 b[2][4];              // all bias loads first: a load issued between the stores would wait for every store before it
#pragma unroll
  for (int u = 0; u < (TWO ? 2 : 1); ++u)
#pragma unroll
    for (int q = 0; q < 4; ++q)   // this lane's four features of register group q: f0 + 32 u + 8 q + 4 kk + (0..3)
      b[u][q] = *reinterpret_cast<const float4*>(P.bias + f0 + 32u * u + 8u * q + 4u * (lane >> 5));
  NIF32_PIN();
#pragma unroll
  for (int u = 0; u < (TWO ? 2 : 1); ++u) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t f = f0 + 32u * u + 8u * q + 4u * (lane >> 5);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float4 o;
        if constexpr (MODE == 1) {   // matmul output in half, + bias in half (zero where the layer has none: x + 0 rounds to x)
          o = make_float4(nif32_hround(nif32_hround(acc[t][u][4 * q]) + b[u][q].x), nif32_hround(nif32_hround(acc[t][u][4 * q + 1]) + b[u][q].y),
                          nif32_hround(nif32_hround(acc[t][u][4 * q + 2]) + b[u][q].z), nif32_hround(nif32_hround(acc[t][u][4 * q + 3]) + b[u][q].w));
        } else {
          o = make_float4(acc[t][u][4 * q] + b[u][q].x, acc[t][u][4 * q + 1] + b[u][q].y, acc[t][u][4 * q + 2] + b[u][q].z,
                          acc[t][u][4 * q + 3] + b[u][q].w);      // addInPlace (:316-321)
        }
        if (RELU) { o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f; o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f; }   // (:323-325)
        if constexpr (MODE == 2) o = make_float4(nif32_hround(o.x), nif32_hround(o.y), nif32_hround(o.z), nif32_hround(o.w));
        *reinterpret_cast<float4*>(P.act_out + nif32_packed(tl + t, P.ldw, f >> 3, lane)) = o;
      }
    }
  }
}

template <bool TWO, int MODE>   // TWO: both 32-feature tiles of the block exist (false: the last block of a 32 (mod 64) wide layer)
__device__ __forceinline__ void nif32_block(const NifF32Params& P, float (*slice)[16][64], uint32_t tl, uint32_t f0) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t m = lane & 31u, kk = lane >> 5;
  f32x16 acc[2][2];            // [sample tile][feature tile]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[t][u] = (f32x16)(0.0f);
  const uint32_t K = P.k_act + P.k_in;                      // a multiple of 16 (pack_nif_f32)
  auto inputs = [&](NifF32Stage& S, uint32_t k0) {
    const bool act = k0 < P.k_act;
    const float* base = act ? P.act_in : P.feat;
    const uint32_t ld = act ? P.lda : P.ldf, kb = act ? k0 : k0 - P.k_act;
    nif32_load_inputs(S, base + nif32_packed(tl, ld, kb >> 3, lane), (size_t)(ld >> 3) * 256u);
  };
  // the workgroup's weight slice of 16 inputs: thread i brings features 4 (i & 15) .. +3 of input row i >> 4
  const uint32_t wr = threadIdx.x >> 4, wc = 4u * (threadIdx.x & 15u);
  const bool wvalid = f0 + wc < P.ldw;                      // (ldw is a multiple of 32: a float4 is inside or outside)
  const float* wsrc = P.w + (size_t)wr * P.ldw + (wvalid ? f0 + wc : 0u);
  const float wkeep = wvalid ? 1.f : 0.f;                   // (a branch-free zero for the columns past the layer's width)
  auto weights = [&](uint32_t k0) {
    const float4 v = *reinterpret_cast<const float4*>(wsrc + (size_t)k0 * P.ldw);
    return TWO ? v : make_float4(v.x * wkeep, v.y * wkeep, v.z * wkeep, v.w * wkeep);
  };
  NifF32Stage S0, S1;
  inputs(S0, 0);
  *reinterpret_cast<float4*>(&slice[0][wr][wc]) = weights(0);
  __syncthreads();
  for (uint32_t k0 = 0, it = 0; k0 < K; k0 += 16u, ++it) {
    const float* ws = &slice[it & 1u][kk][m];
    const float4 wn = weights(min(k0 + 16u, K - 16u));      // unconditional (the last one re-reads a slice): no branch, so the
    inputs(S1, k0 + 8u);                                    // wait counts stay exact
    NIF32_PIN();
    nif32_mfma_stage<TWO>(S0, ws, acc);
    NIF32_PIN();
    inputs(S0, min(k0 + 16u, K - 8u));
    NIF32_PIN();
    nif32_mfma_stage<TWO>(S1, ws + 8 * 64, acc);
    NIF32_PIN();
    *reinterpret_cast<float4*>(&slice[(it + 1u) & 1u][wr][wc]) = wn;   // read last in iteration it - 1, before its barrier
    __syncthreads();
  }
  if (P.relu) nif32_store<TWO, true, MODE>(P, acc, tl, f0, lane);
  else nif32_store<TWO, false, MODE>(P, acc, tl, f0, lane);
}

// MODE: 0 = a float32 layer; mixed models only: 1 = a binary16 layer (P.half_out), 2 = a float32 layer whose successor is
// binary16 (P.cast_half).  Separate instantiations: the all-float32 kernel keeps its 104 VGPRs (four waves per SIMD).
template <int MODE>
__global__ __launch_bounds__(256, 2) void nif32_layer_kernel(const NifF32Params P) {
  __shared__ float slice[2][16][64];
  const uint32_t ntiles = chunk_tile_count(P.total_tiles, P.tile0, P.chunk_tiles);
  // gridDim.x is a multiple of 8: workgroup b runs on XCD b % 8; give every XCD a contiguous run of output blocks
  const uint32_t blk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint32_t ncol = (P.ldw + 63u) / 64u;
  const uint32_t sb = blk / ncol, col = blk - sb * ncol;   // sample block of 256 = 8 queue tiles (the chunk holds a multiple)
  if (8u * sb >= ntiles) return;                            // the whole workgroup: no barrier is left waiting
  const uint32_t tl = 8u * sb + 2u * (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t f0 = col * 64u;
  if (f0 + 32u < P.ldw) nif32_block<true, MODE>(P, slice, tl, f0);
  else nif32_block<false, MODE>(P, slice, tl, f0);
}

// Head (3 outputs) in float, decode (NifModel.cpp:221-245) and scatter (codelets.cpp:366-382).  One thread per sample, the
// inputs in k order through fmaf: the sum the oracle forms.  Lane c + 32 t of a wave is sample c of the wave's tile t; the
// packed layout makes its loads 512-byte runs.
struct NifF32Head {
  const float* w;            // [k_act + k_in][4]
  float bias0, bias1, bias2;
  uint32_t k_act, k_in, relu;
  uint32_t half_out;         // the head's variables are binary16 (a mixed model): sum rounded to half, bias added in half
  const float* act_in; const float* feat;
  uint32_t lda, ldf;
  uint32_t tile0, chunk_tiles;
};
__global__ __launch_bounds__(256) void nif32_head_kernel(const NifParams P, const NifF32Head Hd, const uint32_t* tile_start) {
  __shared__ uint32_t ts[kMaxRegions + 1];
  const uint32_t ntiles = chunk_tile_count(tile_start + P.n_regions, Hd.tile0, Hd.chunk_tiles);
  if (blockIdx.x * 8u >= ntiles) return;
  for (uint32_t i = threadIdx.x; i <= P.n_regions; i += 256u) ts[i] = tile_start[i];
  __syncthreads();
  const uint32_t c = threadIdx.x & 31u;
  for (uint32_t lt = blockIdx.x * 8u + (threadIdx.x >> 5); lt < ntiles; lt += gridDim.x * 8u) {
    const TileRef r = find_tile(ts, P.n_regions, P, Hd.tile0 + lt);
    if (r.local + c >= r.count) continue;
    const uint32_t qi = r.qbase + c;
    float acc[3] = {0.f, 0.f, 0.f};
    const float4* wk = reinterpret_cast<const float4*>(Hd.w);
    auto source = [&](const float* base, uint32_t ld, uint32_t kcount) {
      for (uint32_t j = 0; j < (kcount >> 3); ++j) {
        const float4 lo = *reinterpret_cast<const float4*>(base + nif32_packed(lt, ld, j, c));
        const float4 hi = *reinterpret_cast<const float4*>(base + nif32_packed(lt, ld, j, c + 32u));
        const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float4 wv = wk[k];
          acc[0] = fmaf(x[k], wv.x, acc[0]); acc[1] = fmaf(x[k], wv.y, acc[1]); acc[2] = fmaf(x[k], wv.z, acc[2]);
        }
        wk += 8;
      }
    };
    source(Hd.act_in, Hd.lda, Hd.k_act);
    source(Hd.feat, Hd.ldf, Hd.k_in);
    const float bias[3] = {Hd.bias0, Hd.bias1, Hd.bias2};
    const float mean[3] = {P.mean0, P.mean1, P.mean2};
    float bgr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float o = Hd.half_out ? nif32_hround(nif32_hround(acc[k]) + bias[k]) : acc[k] + bias[k];
      if (Hd.relu) o = o > 0.f ? o : 0.f;
      o = o * P.max;
      o = o + mean[k];
      bgr[k] = P.log_tonemap ? decode_exp(o) : o;
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

#undef NIF32_PIN

}  // namespace ptd
