// pt_nif.h -- the NIF environment-light MLP as a fused MFMA kernel.
//
// Replaces NifModel::buildEncodeInput / buildInference / buildDecodeOutput
// (src/neural_networks/NifModel.cpp:185-218, :295-326, :221-245), the batch serialisation of
// PathTracerApp::buildEnvironmentNif (src/PathTracerApp.cpp:147-198) and PostProcessEscapedRays
// (src/codelets/codelets.cpp:366-382).  It runs only on the compacted queue of escaped paths.
//
// Layout: the network is evaluated transposed, H_{l+1}^T[out x batch] = W_l^T[out x in] H_l^T,
// with v_mfma_f32_32x32x16_f16.  A = a 32x16 tile of W_l^T (pre-packed lane-linear on upload),
// B = 16 features x 32 samples of activations.  The 32x32 f32 result has the sample on the lane
// and the feature in the register index, which is exactly the B-operand layout of the next
// layer's k-steps (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's
// operand"), so activations never leave registers: fp32 accumulator -> fp16 (RNE) -> +bias
// (fp16) -> ReLU -> next layer's B fragments.  The k permutation that trick implies is folded
// into the weight packing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMaxLayers = 16;
constexpr int kMaxRegions = 2048;

struct NifParams {
  const uint4* wpack;              // 1 KiB pieces: [piece][lane] 16 B
  const uint4* bpack;              // [(ntile_index * 2 + h) * 2 + {0,1}] 16 B
  uint32_t n_layers;               // dense layers incl. the head
  uint32_t piece_base[kMaxLayers]; // first piece of layer l
  uint32_t bias_base[kMaxLayers];  // first n-tile of layer l in bpack
  uint32_t concat_mask;            // bit l: layer l takes concat(x, input) (NifModel.cpp:305-308)
  uint32_t relu_mask, bias_mask;
  float max, mean0, mean1, mean2;
  int32_t log_tonemap;
  // queue of escaped paths, one region per trace workgroup
  const float* q_u; const float* q_v; const float* q_tr; const float* q_tg; const float* q_tb;
  const uint32_t* q_path;
  const uint32_t* region_count;
  uint32_t n_regions, region_cap;
  float* rad_r; float* rad_g; float* rad_b;   // per path: env(rgb) * throughput
  float* out_bgr;                             // standalone inference: decoded BGR [n][3]
};

__device__ __forceinline__ half8 as_half8(uint4 v) {
  union { uint4 u; half8 h; } c;
  c.u = v;
  return c.h;
}

// sin and cos of a (|a| <= 8192) via two-constant reduction + v_sin/v_cos (revolutions).
__device__ __forceinline__ void fast_sincos(float a, float& s, float& c) {
  float n = rintf(a * 0.15915494309189535f);
  float r = fmaf(-n, 6.28125f, a);
  r = fmaf(-n, 0.0019353071795864769f, r);
  float t = r * 0.15915494309189535f;
  s = __builtin_amdgcn_sinf(t);
  c = __builtin_amdgcn_cosf(t);
}

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

// ---------------------------------------------------------------- worklist <-> device SoA

struct TraceRecordDev {  // include/ptmi.h pt_trace_record, 20 bytes
  uint16_t u, v;
  float r, g, b;
  uint16_t sampleCount, pathLength;
};

struct Accum {
  uint32_t* pix;       // u | v<<16
  float* r; float* g; float* b;
  uint32_t* count;     // sampleCount (u16 semantics applied when packing)
  uint32_t* length;    // pathLength
};

__global__ void unpack_records_kernel(const TraceRecordDev* rec, uint32_t n, Accum A) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  TraceRecordDev t = rec[i];
  A.pix[i] = (uint32_t)t.u | ((uint32_t)t.v << 16);
  A.r[i] = t.r; A.g[i] = t.g; A.b[i] = t.b;
  A.count[i] = t.sampleCount;
  A.length[i] = t.pathLength;
}

__global__ void pack_records_kernel(TraceRecordDev* rec, uint32_t n, Accum A) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  TraceRecordDev t;
  uint32_t p = A.pix[i];
  t.u = (uint16_t)(p & 0xffffu); t.v = (uint16_t)(p >> 16);
  t.r = A.r[i]; t.g = A.g[i]; t.b = A.b[i];
  t.sampleCount = (uint16_t)A.count[i];   // uint16 wrap-around as in TraceRecord.hpp:10-11
  t.pathLength = (uint16_t)A.length[i];
  rec[i] = t;
}

__global__ void clear_accum_kernel(uint32_t n, Accum A) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  A.r[i] = 0.f; A.g[i] = 0.f; A.b[i] = 0.f;
  A.count[i] = 0; A.length[i] = 0;
}

// AccumulateContributions::compute (codelets.cpp:249-301) for the k iterations of one batch, in
// iteration order so the fp32 sums match the reference's (and the oracle's) order exactly.
__global__ void accumulate_kernel(uint32_t n, uint32_t iters, const uint8_t* plen, const float* rad_r,
                                  const float* rad_g, const float* rad_b, Accum A, unsigned long long* counters) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t segs = 0, esc = 0;
  if (i < n) {
    float r = A.r[i], g = A.g[i], b = A.b[i];
    for (uint32_t k = 0; k < iters; ++k) {
      const size_t p = (size_t)k * n + i;
      const uint32_t pl = plen[p];
      segs += pl & 0x7fu;
      if (pl & 0x80u) {
        r += rad_r[p]; g += rad_g[p]; b += rad_b[p];   // :295-297
        esc += 1;
      }
    }
    A.r[i] = r; A.g[i] = g; A.b[i] = b;
    A.count[i] += iters;                                // :300
    A.length[i] += segs;                                // :253
  }
  // block reduction of the two counters -> one 64-bit atomic pair per workgroup
  __shared__ uint32_t ssegs[256], sesc[256];
  ssegs[threadIdx.x] = segs; sesc[threadIdx.x] = esc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { ssegs[threadIdx.x] += ssegs[threadIdx.x + s]; sesc[threadIdx.x] += sesc[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(&counters[0], (unsigned long long)ssegs[0]);
    atomicAdd(&counters[1], (unsigned long long)sesc[0]);
  }
}

// (b, g, r) / sampleCount per work item: the value AccumulatedImage::accumulate adds
// (src/AccumulatedImage.cpp:69-71), for the multi-GPU HDR gather.
__global__ void export_hdr_kernel(uint32_t n, Accum A, float* bgr) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float scale = 1.f / (float)(uint16_t)A.count[i];
  bgr[3 * (size_t)i + 0] = A.b[i] * scale;
  bgr[3 * (size_t)i + 1] = A.g[i] * scale;
  bgr[3 * (size_t)i + 2] = A.r[i] * scale;
}

}  // namespace ptd
