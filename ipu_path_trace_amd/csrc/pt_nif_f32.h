// pt_nif_f32.h -- NIFs stored as float32 (Hdf5Model.cpp:109-133 accepts float32 variables): the layers run in float.
//
// The reference gives a matmul the type of its kernel (NifModel.cpp:314: poplin::matMul(..., l.kernel.type, ...)), adds the
// bias and applies ReLU in that type (:316-325), and casts the half-precision Fourier features to it (:211-216, :299-302).
// Rounds 1-2 rounded float32 weights to binary16 on upload and ran the fp16 kernels; this path keeps them in float:
// v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain in k order (cdna_hip_programming.md section 3, "FP32-input MFMA"), so a
// layer's output is bit for bit the sequential fmaf sum the oracle forms.  It runs at the fp32 vector rate (1/16 of the fp16
// MFMA rate): a fidelity path for the models that need it, not a fast one -- the shipped NIFs are fp16
// (nif_models/urban_alley_01_4k_fp16_yuv).
//
// Layer by layer over chunks of the queue, activations row-major [sample][feature] in two HBM buffers; one workgroup
// computes 128 samples x 32 features (four waves of one 32 x 32 accumulator tile), operands staged through LDS in
// k-chunks of 32.
#pragma once
#include "pt_nif_gemm.h"

namespace ptd {

struct NifF32Params {
  const float* w;            // this layer's kernel, row-major [k_act + k_in][ldw]
  const float* bias;         // [ldw] (zeros where the layer has none)
  uint32_t ldw;              // padded output width (multiple of 32)
  uint32_t k_act, k_in;      // inputs taken from the previous activations / from the Fourier features
  uint32_t relu;
  const float* act_in;       // [chunk samples][lda]
  const float* feat;         // [chunk samples][ldf]
  float* act_out;            // [chunk samples][ldw]
  uint32_t lda, ldf;
  const uint32_t* total_tiles;
  uint32_t tile0, chunk_tiles;
};

// Fourier features of a chunk, row-major floats [sample][4 E]: [sin u, sin v, cos u, cos v] x E, the half-precision trig of
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
    float* row = feat + ((size_t)lt * 32u + c) * (4 * E);
#pragma unroll
    for (int g = 0; g < E / 4; ++g) {
      const half8 f = fourier_group(coord, g, P.n_freq);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        row[h * E + 4 * g + k] = (float)f[k];                 // sin u | sin v
        row[2 * E + h * E + 4 * g + k] = (float)f[4 + k];     // cos u | cos v
      }
    }
  }
}

// One dense layer in float over a chunk: out[s][f] = act(sum_k in[s][k] W[k][f] + b[f]), k over the previous activations and
// then (first layer, concat layers) over the features -- the order of the reference's concat(x, input) (NifModel.cpp:305-308).
__global__ __launch_bounds__(256) void nif32_layer_kernel(const NifF32Params P) {
  __shared__ float As[128][33];     // [sample][k], padded: lanes of a half-wave read different rows
  __shared__ float Bs[32][32];      // [k][feature]
  const uint32_t ntiles = chunk_tile_count(P.total_tiles, P.tile0, P.chunk_tiles);
  const uint32_t s0 = blockIdx.x * 128u;
  if (s0 >= ntiles * 32u) return;
  const uint32_t f0 = blockIdx.y * 32u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, kk = lane >> 5;
  f32x16 acc = (f32x16)(0.0f);
  const uint32_t K = P.k_act + P.k_in;
  for (uint32_t k0 = 0; k0 < K; k0 += 32u) {
    // stage 128 x 32 inputs and 32 x 32 weights (k_act and k_in are multiples of 4; rows past K are zero-filled)
    for (uint32_t i = threadIdx.x; i < 128u * 32u; i += 256u) {
      const uint32_t row = i >> 5, k = k0 + (i & 31u);
      float v = 0.f;
      if (k < P.k_act) v = P.act_in[(size_t)(s0 + row) * P.lda + k];
      else if (k < K) v = P.feat[(size_t)(s0 + row) * P.ldf + (k - P.k_act)];
      As[row][i & 31u] = v;
    }
    for (uint32_t i = threadIdx.x; i < 32u * 32u; i += 256u) {
      const uint32_t k = k0 + (i >> 5);
      Bs[i >> 5][i & 31u] = (k < K) ? P.w[(size_t)k * P.ldw + f0 + (i & 31u)] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; ++s)   // A = inputs (row = sample), B = weights (column = feature): D[sample][feature]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[wave * 32 + r][2 * s + kk], Bs[2 * s + kk][r], acc, 0, 0, 0);
    __syncthreads();
  }
  const float b = P.bias[f0 + r];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t row = (uint32_t)((i & 3) + 8 * (i >> 2) + 4 * kk);
    float o = acc[i] + b;                                           // addInPlace (:316-321)
    if (P.relu) o = o > 0.f ? o : 0.f;                              // ReLU (:323-325)
    P.act_out[(size_t)(s0 + wave * 32 + row) * P.ldw + f0 + r] = o;
  }
}

// Head (3 outputs) in float, decode (NifModel.cpp:221-245) and scatter (codelets.cpp:366-382).  One thread per sample, the
// inputs in k order through fmaf: the sum the oracle forms.
struct NifF32Head {
  const float* w;            // [k_act + k_in][4]
  float bias0, bias1, bias2;
  uint32_t k_act, k_in, relu;
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
    const size_t sample = (size_t)lt * 32u + c;
    float acc[3] = {0.f, 0.f, 0.f};
    const float* x = Hd.act_in + sample * Hd.lda;
    for (uint32_t k = 0; k < Hd.k_act; ++k) {
      const float xv = x[k];
      const float4 wv = reinterpret_cast<const float4*>(Hd.w)[k];
      acc[0] = fmaf(xv, wv.x, acc[0]); acc[1] = fmaf(xv, wv.y, acc[1]); acc[2] = fmaf(xv, wv.z, acc[2]);
    }
    const float* ft = Hd.feat + sample * Hd.ldf;
    for (uint32_t k = 0; k < Hd.k_in; ++k) {
      const float xv = ft[k];
      const float4 wv = reinterpret_cast<const float4*>(Hd.w)[Hd.k_act + k];
      acc[0] = fmaf(xv, wv.x, acc[0]); acc[1] = fmaf(xv, wv.y, acc[1]); acc[2] = fmaf(xv, wv.z, acc[2]);
    }
    const float bias[3] = {Hd.bias0, Hd.bias1, Hd.bias2};
    const float mean[3] = {P.mean0, P.mean1, P.mean2};
    float bgr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float o = acc[k] + bias[k];
      if (Hd.relu) o = o > 0.f ? o : 0.f;
      o = o * P.max;
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

}  // namespace ptd
