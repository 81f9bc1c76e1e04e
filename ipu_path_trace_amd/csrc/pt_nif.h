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


// ---------------------------------------------------------------- v2: weights through an LDS ring
//
// Same arithmetic as nif_kernel, but the WAVES waves of a workgroup walk the network in lockstep,
// each on its own tile of 32*NB samples, and share every weight piece: layer 0 and all biases stay
// resident in LDS, layers >= 1 stream through a ring of R slots (one slot = the pieces of one
// 32-feature output tile) filled by global_load_lds (1 KiB per wave-instruction, L2 -> LDS, no
// registers).  Each wave issues exactly PW pieces per slab, so the wait before a slab's barrier is
// the counted s_waitcnt vmcnt(PW*(R-2)): the next R-2 slabs stay in flight across the barrier.
// WAVES = 8 (two waves per SIMD, 256 registers each) lets one wave's LDS reads and epilogue VALU
// run under its partner's MFMAs.
template <int H, int E, int WAVES>
struct NifV2Geometry {
  static constexpr int KS = H / 16;
  static constexpr int NT = H / 32;
  static constexpr int IS = E / 4;
  static constexpr int R = 4;
  static constexpr int SLAB_PIECES = ((KS + IS + WAVES - 1) / WAVES) * WAVES;  // room for a concat layer
  static constexpr int PW = SLAB_PIECES / WAVES;                               // pieces each wave loads per slab
  static constexpr int SLOT_BYTES = SLAB_PIECES * 1024;
  static constexpr int SCAN_BYTES = ((kMaxRegions + 1) * 4 + 511) / 512 * 512;
  static constexpr int BIAS_BYTES = (kMaxLayers * NT * 64 + 511) / 512 * 512;
  static constexpr int W0_BYTES = NT * IS * 1024;
  static constexpr int LDS_BYTES = SCAN_BYTES + BIAS_BYTES + W0_BYTES + R * SLOT_BYTES;
};

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to lds_byte_addr + 16*lane.
// Inline asm on purpose: with the builtin hipcc waits vmcnt(0) before every ds_read it cannot
// prove disjoint from the DMA destination, which serialises the ring.  hipcc does not count these
// loads, so every wait for them is the hand-placed counted s_waitcnt below
// (cdna_hip_programming.md section 5.7; M0 is written in the same statement that reads it).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_byte_addr) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_byte_addr)
      : "memory");
}

template <int H, int E, int NB, int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void nif_kernel_v2(const NifParams P) {
  using G = NifV2Geometry<H, E, WAVES>;
  constexpr int KS = G::KS, NT = G::NT, IS = G::IS, R = G::R, PW = G::PW;
  constexpr int TS = 32 * NB;
  constexpr int THREADS = 64 * WAVES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* tile_start = reinterpret_cast<uint32_t*>(smem);
  char* bias_lds = smem + G::SCAN_BYTES;
  char* w0_lds = bias_lds + G::BIAS_BYTES;
  char* ring = w0_lds + G::W0_BYTES;
  const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31;
  const int h = lane >> 5;

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
    // resident data: every bias tile and the layer-0 pieces
    const uint32_t n_btiles = P.bias_base[P.n_layers - 1] + 1u;
    for (uint32_t i = threadIdx.x; i < n_btiles * 4u; i += THREADS)
      reinterpret_cast<uint4*>(bias_lds)[i] = P.bpack[i];
    const uint4* w0 = P.wpack + (size_t)P.piece_base[0] * 64;
    for (uint32_t i = threadIdx.x; i < (uint32_t)(NT * IS * 64); i += THREADS)
      reinterpret_cast<uint4*>(w0_lds)[i] = w0[i];
    __syncthreads();
  }
  const uint32_t total_tiles = tile_start[P.n_regions];
  const uint32_t wg_tiles = (total_tiles + WAVES - 1u) / WAVES;
  if (blockIdx.x >= wg_tiles) return;   // whole workgroup leaves together

  // slab stream of one pass over the network: (layer l >= 1, n-tile j); the head has one tile
  const uint32_t n_layers = P.n_layers;
  uint32_t pf_l = 1, pf_j = 0, pf_q = 0;   // prefetch cursor
  auto issue_slab = [&]() {
    const uint32_t cnt = KS + (((P.concat_mask >> pf_l) & 1u) ? IS : 0);
    const uint32_t first = P.piece_base[pf_l] + pf_j * cnt;
    const uint32_t slot = ring_lds + (pf_q % R) * G::SLOT_BYTES;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      uint32_t piece = (uint32_t)wave + (uint32_t)WAVES * i;
      if (piece >= cnt) piece = cnt - 1u;            // uniform load count: re-load the last piece
      const char* src = reinterpret_cast<const char*>(P.wpack) + ((size_t)(first + piece) * 1024 + lane * 16);
      glds16(src, slot + piece * 1024u);
    }
    pf_q += 1;
    pf_j += 1;
    const uint32_t tiles_l = (pf_l + 1 == n_layers) ? 1u : (uint32_t)NT;
    if (pf_j == tiles_l) { pf_j = 0; pf_l = (pf_l + 1 == n_layers) ? 1u : pf_l + 1; }
  };
#pragma unroll
  for (int i = 0; i < R - 1; ++i) issue_slab();
  uint32_t q = 0;  // consumer stage

  auto stage_sync = [&]() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (R - 2)) : "memory");
    asm volatile("s_barrier" ::: "memory");
    issue_slab();
  };

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
    const uint32_t qbase = region * P.region_cap + local;

    half8 in[NB][IS];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const uint32_t off = local + 32u * b + c;
      const uint32_t qi = qbase + 32u * b + c;
      float coord = 0.5f;
      if (off < count) coord = h ? P.q_v[qi] : P.q_u[qi];
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
        const uint4* bp = reinterpret_cast<const uint4*>(bias_lds) + ((size_t)(P.bias_base[layer] + j) * 2 + h) * 2;
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

    // One 32-feature output tile: KS (+IS) k-steps, A fragment of step s+1 read while step s multiplies.
    auto tile_mma = [&](const uint4* wj, half8 (&src)[NB][KS], bool concat, f32x16 (&acc)[NB]) __attribute__((always_inline)) {
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[b] = (f32x16)(0.0f);
      half8 a_next = as_half8(wj[0]);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const half8 a = a_next;
        if (s + 1 < KS) a_next = as_half8(wj[(s + 1) * 64]);
        else if (concat) a_next = as_half8(wj[KS * 64]);
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, src[b][s], acc[b], 0, 0, 0);
      }
      if (concat) {
#pragma unroll
        for (int s = 0; s < IS; ++s) {
          const half8 a = a_next;
          if (s + 1 < IS) a_next = as_half8(wj[(KS + s + 1) * 64]);
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, in[b][s], acc[b], 0, 0, 0);
        }
      }
    };

    // ---- layer 0 from the resident copy
    {
      const uint4* wp = reinterpret_cast<const uint4*>(w0_lds) + lane;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x16 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = (f32x16)(0.0f);
#pragma unroll
        for (int s = 0; s < IS; ++s) {
          const half8 a = as_half8(wp[(j * IS + s) * 64]);
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, in[b][s], acc[b], 0, 0, 0);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) epilogue(acc[b], cur[b][2 * j], cur[b][2 * j + 1], 0, j);
      }
    }

    // ---- hidden layers through the ring; src/dst register sets alternate so nothing is copied
    auto hidden = [&](half8 (&src)[NB][KS], half8 (&dst)[NB][KS], uint32_t l) __attribute__((always_inline)) {
      const bool concat = (P.concat_mask >> l) & 1u;
      f32x16 accp[NB];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        stage_sync();
        const uint4* wj = reinterpret_cast<const uint4*>(ring + (q % R) * G::SLOT_BYTES) + lane;
        q += 1;
        f32x16 acc[NB];
        tile_mma(wj, src, concat, acc);
        if (j > 0) {  // epilogue of the previous tile, free to overlap this tile's MFMAs
#pragma unroll
          for (int b = 0; b < NB; ++b) epilogue(accp[b], dst[b][2 * (j - 1)], dst[b][2 * (j - 1) + 1], l, j - 1);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) accp[b] = acc[b];
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) epilogue(accp[b], dst[b][2 * (NT - 1)], dst[b][2 * (NT - 1) + 1], l, NT - 1);
    };
    {
      uint32_t l = 1;
      for (; l + 2 < n_layers; l += 2) {
        hidden(cur, nxt, l);
        hidden(nxt, cur, l + 1);
      }
      if (l + 1 < n_layers) {  // odd number of hidden layers behind layer 0
        hidden(cur, nxt, l);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int s = 0; s < KS; ++s) cur[b][s] = nxt[b][s];
      }
    }

    // ---- head
    {
      const uint32_t l = n_layers - 1;
      const bool concat = (P.concat_mask >> l) & 1u;
      stage_sync();
      const uint4* wj = reinterpret_cast<const uint4*>(ring + (q % R) * G::SLOT_BYTES) + lane;
      q += 1;
      f32x16 acc[NB];
      tile_mma(wj, cur, concat, acc);
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
          const uint32_t qi = qbase + 32u * b + c;
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
