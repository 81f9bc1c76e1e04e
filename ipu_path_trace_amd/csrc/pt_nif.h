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

#include <type_traits>
#include <utility>

namespace ptd {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMaxLayers = 16;
constexpr int kMaxRegions = 2048;

struct NifParams {
  const uint4* wpack;              // 1 KiB pieces: [piece][lane] 16 B
  const uint4* bpack;              // [(ntile_index * 2 + h) * 2 + {0,1}] 16 B
  uint32_t n_layers;               // dense layers incl. the head
  uint32_t n_freq;                 // true frequencies per coordinate; slots n_freq..E-1 (E padded to 4 | E) are zero features
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

// exp() of the decode (NifModel.cpp:239-241: popops::exp in float), full range: +inf above 88.72, SUBNORMAL results
// between -87.3 and -103.3 (the fast intrinsic's v_exp_f32 flushes those to 0), NaN for NaN.  Three calls per sample
// against a MFLOP of MLP: the accurate function costs nothing measurable.
__device__ __forceinline__ float decode_exp(float x) { return expf(x); }

// A LINEAR layer's "ReLU floor" for v_pk_max_f16: a quiet NaN.  max(x, qNaN) = x for every x, NaN included (with a floor
// of -inf a NaN activation came out as -inf: the instruction returns the operand that is not a NaN) -- NifModel.cpp:323-325
// applies no non-linearity at all to such a layer.
constexpr uint32_t kLinearFloor = 0x7e007e00u;

// sin and cos of a (|a| <= 8192) via two-constant reduction + v_sin/v_cos (revolutions).
__device__ __forceinline__ void fast_sincos(float a, float& s, float& c) {
  float n = rintf(a * 0.15915494309189535f);
  float r = fmaf(-n, 6.28125f, a);
  r = fmaf(-n, 0.0019353071795864769f, r);
  float t = r * 0.15915494309189535f;
  s = __builtin_amdgcn_sinf(t);
  c = __builtin_amdgcn_cosf(t);
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

// Two consecutive pieces (2 KiB of memory -> 2 KiB of LDS) with one M0 set-up: uniform base address in an SGPR pair,
// constant per-lane offset, and the instruction's immediate offset, which applies to the global AND the LDS side.
__device__ __forceinline__ void glds16x2(const void* sbase, uint32_t lane_off, uint32_t lds_byte_addr) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_off), "s"(reinterpret_cast<uint64_t>(sbase)), "s"(lds_byte_addr)
      : "memory");
}

// The same with the non-temporal bit on both loads (profiling build, the wide-NIF activation stream: a line marked for
// early replacement in L2 should not push out the layer's weights).
__device__ __forceinline__ void glds16x2_nt(const void* sbase, uint32_t lane_off, uint32_t lds_byte_addr) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 nt\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024 nt\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_off), "s"(reinterpret_cast<uint64_t>(sbase)), "s"(lds_byte_addr)
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
          if (s == IS - 1 && k > 0 && (uint32_t)(4 * s + k) >= P.n_freq) { sn = 0.f; cs = 0.f; }   // padded frequency slot (E rounded up to 4 | E)
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
            bgr[k] = P.log_tonemap ? decode_exp(o) : o;
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

// ---------------------------------------------------------------- v3: every layer through the ring
//
// As nif_kernel_v2 with NB = 1, but a ring stage carries TPS output tiles (slab = TPS * ksteps
// pieces), layer 0 streams through the ring too, and R = 3 slots of 48 KiB fill the 160 KiB LDS.
// Everything that is not an MFMA is issued in the shadow of one: a tile's k-steps run in groups of
// four MFMAs, and behind each group the wave issues (a) the LDS reads of the group after next,
// (b) one chunk of the PREVIOUS tile's epilogue (cvt/bias/ReLU on four accumulator registers) and
// (c) one piece of the weight DMA for the stage after next.  Measured by ablation
// (profiles/r01_c_nif_ablation.txt): un-overlapped, the epilogue cost 32 % and the DMA issue burst
// plus barrier 23 % of the kernel.
template <int V>
using IC = std::integral_constant<int, V>;
template <int... I, class F>
__device__ __forceinline__ void for_each_index(std::integer_sequence<int, I...>, F&& f) {
  (f(IC<I>{}), ...);
}

template <int H, int E, int WAVES, int TPS>
struct NifV3Geometry {
  static constexpr int KS = H / 16;
  static constexpr int NT = H / 32;
  static constexpr int IS = E / 4;
  static constexpr int R = 3;
  // pieces are loaded in pairs (one M0 set-up, one uniform base, immediate offset for the second piece)
  static constexpr int SLAB_PIECES = ((TPS * (KS + IS) + 2 * WAVES - 1) / (2 * WAVES)) * (2 * WAVES);
  static constexpr int PW = SLAB_PIECES / (2 * WAVES);                           // PAIRS each wave loads per slab
  static constexpr int SLOT_BYTES = SLAB_PIECES * 1024;
  static constexpr int T0 = (SLAB_PIECES / IS) < NT ? (SLAB_PIECES / IS) : NT;   // layer-0 tiles per slab
  static constexpr int T0_LAST = (NT % T0) ? (NT % T0) : T0;
  static constexpr int pairs_per_wave(int pieces) { return ((pieces + 1) / 2) / WAVES; }
  static constexpr int min2(int a, int b) { return a < b ? a : b; }
  // fewest load instructions any wave issues for any slab (head: KS pieces; hidden: TPS x KS; layer 0: T0 / T0_LAST x IS)
  static constexpr int MINP = 2 * min2(min2(pairs_per_wave(KS), pairs_per_wave(TPS * KS)),
                                       min2(pairs_per_wave(T0 * IS), pairs_per_wave(T0_LAST * IS)));
  static constexpr int SCAN_BYTES = ((kMaxRegions + 1) * 4 + 511) / 512 * 512;
  static constexpr int MAX_LAYERS = 8;                                           // bias tiles resident in LDS
  static constexpr int BIAS_BYTES = (((MAX_LAYERS - 1) * NT + 1) * 64 + 511) / 512 * 512;
  static constexpr int LDS_BYTES = SCAN_BYTES + BIAS_BYTES + R * SLOT_BYTES;
  static_assert(NT % TPS == 0, "tiles per stage must divide the tile count");
  static_assert(LDS_BYTES <= 160 * 1024, "ring does not fit the LDS");
};

// DIAG (timing-only builds, results invalid): bit 0 = no ring sync / DMA, bit 1 = no LDS reads of A,
// bit 2 = no epilogue arithmetic, bit 3 = no encode, bit 6 = HALF the LDS reads of A in the hidden layers (one fragment
// per pair of MFMAs, used twice: exactly the LDS traffic of an NB = 2 kernel -- two 32-sample tiles per wave sharing every
// weight fragment -- with nothing else changed: the upper bound of what NB = 2 could buy, profiles/r05_nif_nb2.txt).  Bit 5 (valid results, profiling build): workgroup 0 stamps
// s_memtime / s_memrealtime around its whole tile loop -> the in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6).
#ifdef PTMI_DIAG_BUILD
__device__ unsigned long long g_nif_clock[2];   // shader cycles, 100 MHz ticks of the last stamped launch's workgroup 0
#endif
template <int H, int E, int WAVES, int TPS, int DIAG = 0>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void nif_kernel_v3(const NifParams P) {
  using G = NifV3Geometry<H, E, WAVES, TPS>;
  constexpr int KS = G::KS, NT = G::NT, IS = G::IS, R = G::R, PW = G::PW, T0 = G::T0;
  constexpr int TS = 32;
  constexpr int THREADS = 64 * WAVES;
  constexpr int GR = 2;                       // MFMAs per group (A double buffer = 4 fragments)
  constexpr int NG = KS / GR;                 // groups per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* tile_start = reinterpret_cast<uint32_t*>(smem);
  char* bias_lds = smem + G::SCAN_BYTES;
  char* ring = bias_lds + G::BIAS_BYTES;
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
    // bias tiles resident in LDS; layers without a bias carry packed zeros (x + 0 = x in fp16)
    const uint32_t n_btiles = P.bias_base[P.n_layers - 1] + 1u;
    for (uint32_t i = threadIdx.x; i < n_btiles * 4u; i += THREADS)
      reinterpret_cast<uint4*>(bias_lds)[i] = P.bpack[i];
    __syncthreads();
  }
  const uint32_t total_tiles = tile_start[P.n_regions];
  const uint32_t wg_tiles = (total_tiles + WAVES - 1u) / WAVES;
  if (blockIdx.x >= wg_tiles) return;   // whole workgroup leaves together
  // DIAG bit 4 (16): static priority for the second-dispatched half of the workgroup, the arbitration loser of every SIMD
  // pair (MI355X_MICROARCH.md, "Two waves per SIMD", item 4)
  if constexpr (DIAG & 16) { if (wave >= WAVES / 2) __builtin_amdgcn_s_setprio(1); }
  // DIAG bit 7 (128, valid results): every wave of the kernel at raised priority for its whole life, so that the trace kernel's
  // waves co-resident on the SIMD (priority 0) only get the issue slots the MFMA waves leave (round-5 A/B: profiles/r05_nif_nb2.txt, section 4)
  if constexpr (DIAG & 128) __builtin_amdgcn_s_setprio(2);

  // Slab stream of one pass: layer 0 in groups of T0 tiles, hidden layers in groups of TPS, head alone.
  const uint32_t n_layers = P.n_layers;
  uint32_t pf_l = 0, pf_j = 0, pf_q = 0;   // prefetch cursor: layer, first tile of the slab, stage number
  uint32_t pf_first = 0, pf_cnt = 1, pf_slot = 0, pf_ntile = 0;
  auto slab_begin = [&]() {
    const uint32_t ksteps = (pf_l == 0) ? (uint32_t)IS : KS + (((P.concat_mask >> pf_l) & 1u) ? IS : 0);
    const uint32_t tiles_l = (pf_l + 1 == n_layers) ? 1u : (uint32_t)NT;
    const uint32_t group = (pf_l == 0) ? (uint32_t)T0 : (uint32_t)TPS;
    pf_ntile = (tiles_l - pf_j < group) ? tiles_l - pf_j : group;
    pf_cnt = pf_ntile * ksteps;
    pf_first = P.piece_base[pf_l] + pf_j * ksteps;
    pf_slot = ring_lds + (pf_q % R) * G::SLOT_BYTES;
  };
  const uint32_t lane16 = (uint32_t)lane * 16u;
  auto slab_piece = [&](int i) {   // pair i of this wave: pieces 2 (wave + WAVES i), + 1
    if constexpr (DIAG & 1) return;
    const uint32_t piece = 2u * ((uint32_t)wave + (uint32_t)WAVES * i);
    if (piece >= pf_cnt) return;                     // exact counts: a wave issues ceil((pairs - wave) / WAVES) pairs
    // (an odd slab's last pair also copies the piece that follows it in memory into an unused part of the slot)
    const char* base = reinterpret_cast<const char*>(P.wpack) + ((size_t)(pf_first + piece) << 10);
    glds16x2(base, lane16, pf_slot + piece * 1024u);
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
  uint32_t q = 0;   // consumer stage
  int pf_next = PW; // next piece of the slab being prefetched during this stage (PW = none pending)
#ifdef PTMI_DIAG_BUILD
  unsigned long long t_cycles = 0, t_real = 0;
  if constexpr ((DIAG & 32) != 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { t_cycles = __builtin_amdgcn_s_memtime(); t_real = __builtin_amdgcn_s_memrealtime(); }
  }
#endif

  // Start of a ring stage: my pieces of this stage's slab have landed (the slab after it may still be in
  // flight), everyone's have after the barrier, and the slot read two stages ago is free for the next slab.
  auto stage_sync = [&]() -> const uint4* {
    if constexpr (!(DIAG & 1)) {
#pragma unroll
      for (int k = 0; k < PW; ++k) if (k >= pf_next) slab_piece(k);   // pieces a short stage had no room for
      if (pf_next != PW + 1) slab_end();
      // Counted wait.  Piece counts are exact, so waves and slabs differ in how many pieces they have in flight;
      // every wave issues at least MINP pieces for every slab, so leaving MINP in flight guarantees that all of
      // THIS stage's pieces (older than the next slab's) have landed.  DMA latency is hidden either way
      // (profiles/r01_c_nif_ablation.txt, DIAG 16).
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::MINP * (R - 2)) : "memory");
      asm volatile("s_barrier" ::: "memory");
      slab_begin();
      pf_next = 0;
    }
    const uint4* slot = reinterpret_cast<const uint4*>(ring + (q % R) * G::SLOT_BYTES) + lane;
    q += 1;
    return slot;
  };
  auto dma_slot = [&]() {   // one DMA piece behind a group of MFMAs
    if constexpr (!(DIAG & 1)) {
      if (pf_next < PW) { slab_piece(pf_next); pf_next += 1; }
    }
  };
  pf_next = PW + 1;   // the prologue already issued and closed its slabs

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
    const uint32_t qi = region * P.region_cap + local + c;
    const bool sample_valid = local + c < count;

    half8 in[IS];
    {
      float coord = 0.5f;
      if (sample_valid) coord = h ? P.q_v[qi] : P.q_u[qi];
      const float x = (coord - 1.0f) * 2.0f;
#pragma unroll
      for (int s = 0; s < IS; ++s) {
        half8 f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a = (float)(_Float16)(x * (float)(1u << (4 * s + k)));
          float sn, cs;
          if constexpr (DIAG & 8) { sn = a; cs = a + 1.0f; }
          else fast_sincos(a, sn, cs);
          if (s == IS - 1 && k > 0 && (uint32_t)(4 * s + k) >= P.n_freq) { sn = 0.f; cs = 0.f; }   // padded frequency slot (E rounded up to 4 | E)
          f[k] = (_Float16)sn;
          f[4 + k] = (_Float16)cs;
        }
        in[s] = f;
      }
    }

    half8 cur[KS], nxt[KS];

    // Epilogue of one tile in four chunks of four accumulator registers: fp32 -> fp16 (RNE, v_cvt_pk),
    // + bias in fp16 (v_pk_add_f16), ReLU (v_pk_max_f16): NifModel.cpp:314-325.  Branch-free: a linear
    // layer uses a floor that makes the max an identity (kLinearFloor).  Chunk ch fills 32-bit words 2*(ch&1), 2*(ch&1)+1 of o0 (ch < 2) or o1.
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    struct Pending {
      f32x16 acc;
      const char* bias;   // this lane half's 32 bytes of packed bias
      uint32_t floor;     // packed fp16 pair: 0 (ReLU) or a quiet NaN (linear: kLinearFloor)
    };
    // Per-layer constants are fetched once per layer (layer_consts), not per tile: an s_load of a kernel argument
    // indexed by the layer can only be awaited with lgkmcnt(0), which would also drain the LDS reads in flight.
    struct LayerConsts {
      const char* bias;   // this lane half's bias bytes of the layer's tile 0
      uint32_t floor;
    };
    auto layer_consts = [&](uint32_t layer) -> LayerConsts {
      LayerConsts c;
      c.bias = bias_lds + ((size_t)P.bias_base[layer] * 2 + h) * 32;
      c.floor = ((P.relu_mask >> layer) & 1u) ? 0u : kLinearFloor;
      return c;
    };
    auto epi_begin = [&](Pending& p, const f32x16& acc, const LayerConsts& c, int j) __attribute__((always_inline)) {
      p.acc = acc;
      p.bias = c.bias + j * 64;
      p.floor = c.floor;
    };
    auto epi_chunk = [&](const Pending& p, auto chc, half8& o0, half8& o1, uint2 bias_now) __attribute__((always_inline)) {
      constexpr int ch = decltype(chc)::value;
      half8& o = (ch < 2) ? o0 : o1;
      const int e0 = (ch & 1) * 4;
      if constexpr (DIAG & 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[e0 + i] = (_Float16)1.0f;
        asm volatile("" ::"v"(p.acc[4 * ch]));
        return;
      }
      // uniform branches on purpose: they split the block, which keeps hipcc's register pressure in budget
      // (the branch-free form spills ~200 VGPRs), and in-order issue overlaps the chunk with the MFMAs anyway
      half2v x0 = {(_Float16)p.acc[4 * ch + 0], (_Float16)p.acc[4 * ch + 1]};
      half2v x1 = {(_Float16)p.acc[4 * ch + 2], (_Float16)p.acc[4 * ch + 3]};
      {
        union { uint2 u; half2v hh[2]; } bb;   // layers without a bias carry packed zeros: x + 0 = x
        bb.u = bias_now;
        x0 = x0 + bb.hh[0];
        x1 = x1 + bb.hh[1];
      }
      // ReLU as one v_pk_max_f16 against a uniform floor (0: max(NaN, 0) = 0 as the oracle's !(x > 0) -> 0; a linear layer: kLinearFloor);
      // the builtin max costs a canonicalising max and a select on top
      asm("v_pk_max_f16 %0, %1, %2" : "=v"(x0) : "v"(x0), "s"(p.floor));
      asm("v_pk_max_f16 %0, %1, %2" : "=v"(x1) : "v"(x1), "s"(p.floor));
      o[e0 + 0] = x0[0]; o[e0 + 1] = x0[1];
      o[e0 + 2] = x1[0]; o[e0 + 3] = x1[1];
    };
    auto bias_at = [&](const Pending& p, int ch) -> uint2 { return *reinterpret_cast<const uint2*>(p.bias + ch * 8); };
    auto epi_all = [&](const Pending& p, half8& o0, half8& o1) __attribute__((always_inline)) {
      epi_chunk(p, IC<0>{}, o0, o1, bias_at(p, 0));
      epi_chunk(p, IC<1>{}, o0, o1, bias_at(p, 1));
      epi_chunk(p, IC<2>{}, o0, o1, bias_at(p, 2));
      epi_chunk(p, IC<3>{}, o0, o1, bias_at(p, 3));
    };

    // One 32-feature output tile.  A fragments come from LDS in groups of GR, two groups in flight; the
    // empty asm pins that order (hipcc otherwise sinks every ds_read next to its MFMA and waits
    // lgkmcnt(0) per k-step) and makes the compiler wait for exactly the group about to multiply.
    // `A` outlives the tile: when the next tile sits in the same ring stage (already landed), its first group is
    // fetched behind this tile's last group (`next`), so only the first tile after a barrier waits for LDS.
    auto tile_mma = [&](const uint4* wj, half8 (&src)[KS], bool concat, f32x16& acc, half8 (&A)[2][GR], bool preloaded,
                        const uint4* next, auto&& after_group) __attribute__((always_inline)) {
      acc = (f32x16)(0.0f);
      constexpr int RD = (DIAG & 64) ? 1 : GR;   // fragments really read per group
      if (!preloaded) {
#pragma unroll
        for (int i = 0; i < RD; ++i) A[0][i] = (DIAG & 2) ? in[i % IS] : as_half8(wj[i * 64]);
      }
      auto group = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g2 = decltype(gc)::value;
        if constexpr (g2 + 1 < NG) {
#pragma unroll
          for (int i = 0; i < RD; ++i)
            A[(g2 + 1) & 1][i] = (DIAG & 2) ? in[(g2 + i) % IS] : as_half8(wj[((g2 + 1) * GR + i) * 64]);
        } else if constexpr (NG % 2 == 0 && !(DIAG & 2)) {
          if (next) {
#pragma unroll
            for (int i = 0; i < RD; ++i) A[0][i] = as_half8(next[i * 64]);
          }
        }
        if constexpr (GR == 4)
          asm volatile("" : "+v"(A[g2 & 1][0]), "+v"(A[g2 & 1][1]), "+v"(A[g2 & 1][2]), "+v"(A[g2 & 1][3])::"memory");
        else if constexpr (RD == 1)
          asm volatile("" : "+v"(A[g2 & 1][0])::"memory");
        else
          asm volatile("" : "+v"(A[g2 & 1][0]), "+v"(A[g2 & 1][1])::"memory");
#pragma unroll
        for (int i = 0; i < GR; ++i)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g2 & 1][i % RD], src[g2 * GR + i], acc, 0, 0, 0);
        // keep the scheduler from sinking this group's MFMAs below the last group's fragment wait
        if constexpr (g2 + 2 == NG) __builtin_amdgcn_sched_barrier(0);
        after_group(gc);
      };
      for_each_index(std::make_integer_sequence<int, NG>{}, group);
      if (concat) {
        half8 t[IS];
#pragma unroll
        for (int s = 0; s < IS; ++s) t[s] = as_half8(wj[(KS + s) * 64]);
#pragma unroll
        for (int s = 0; s < IS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(t[s], in[s], acc, 0, 0, 0);
      }
    };

    // ---- layer 0: 4E -> H, slabs of T0 tiles (3 MFMAs per tile: epilogue and DMA ride behind each tile)
    {
      const uint4* slot = nullptr;
      Pending pend;
      const LayerConsts lc = layer_consts(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (j % T0 == 0) slot = stage_sync();
        half8 a0[IS];
#pragma unroll
        for (int s = 0; s < IS; ++s) a0[s] = as_half8(slot[((j % T0) * IS + s) * 64]);
        f32x16 acc = (f32x16)(0.0f);
#pragma unroll
        for (int s = 0; s < IS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[s], in[s], acc, 0, 0, 0);
        if (j > 0) epi_all(pend, cur[2 * (j - 1)], cur[2 * (j - 1) + 1]);
        dma_slot();
        epi_begin(pend, acc, lc, j);
      }
      epi_all(pend, cur[2 * (NT - 1)], cur[2 * (NT - 1) + 1]);
    }

    // ---- hidden layers; src/dst register sets alternate so nothing is copied
    auto hidden = [&](half8 (&src)[KS], half8 (&dst)[KS], uint32_t l) __attribute__((always_inline)) {
      const bool concat = (P.concat_mask >> l) & 1u;
      const uint32_t ksteps = KS + (concat ? IS : 0);
      Pending pend;
      uint2 b_next = {0u, 0u};
      const uint4* slot = nullptr;
      const LayerConsts lc = layer_consts(l);
      half8 A[2][GR];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (j % TPS == 0) slot = stage_sync();
        f32x16 acc;
        const bool chained = NG % 2 == 0 && !(DIAG & 2);
        const uint4* wj = slot + (size_t)(j % TPS) * ksteps * 64;
        tile_mma(wj, src, concat, acc, A, chained && (j % TPS) != 0, (chained && (j % TPS) + 1 < TPS) ? wj + (size_t)ksteps * 64 : nullptr,
                 [&](auto gc) __attribute__((always_inline)) {
          constexpr int g2 = decltype(gc)::value;
          if constexpr (g2 < 4) {
            if (j > 0) {
              const uint2 b_now = b_next;
              if constexpr (g2 < 3 && g2 + 1 < NG) b_next = bias_at(pend, g2 + 1);   // lands behind the next group's MFMAs
              epi_chunk(pend, gc, dst[2 * (j - 1)], dst[2 * (j - 1) + 1], b_now);
            }
          }
          // DMA riders behind groups 4.. of each tile, away from the epilogue chunks of groups 0..3
          if constexpr (NG >= 8) { if constexpr (g2 >= 4 && g2 < 4 + (PW + TPS - 1) / TPS) dma_slot(); }
          else dma_slot();
        });
        if (j > 0) {   // narrow networks: fewer than four groups per tile, finish the leftover chunks here
          if constexpr (NG < 2) epi_chunk(pend, IC<1>{}, dst[2 * (j - 1)], dst[2 * (j - 1) + 1], bias_at(pend, 1));
          if constexpr (NG < 3) epi_chunk(pend, IC<2>{}, dst[2 * (j - 1)], dst[2 * (j - 1) + 1], bias_at(pend, 2));
          if constexpr (NG < 4) epi_chunk(pend, IC<3>{}, dst[2 * (j - 1)], dst[2 * (j - 1) + 1], bias_at(pend, 3));
        }
        // The next tile's first fragments were requested before this tile's last MFMAs; retiring them here (a real
        // s_waitcnt the compiler's counter model sees: lgkmcnt(0), vmcnt/expcnt untouched) keeps it from waiting
        // for the following reads as well at the control-flow join behind the concat tail.
        if (chained && (j % TPS) + 1 < TPS) __builtin_amdgcn_s_waitcnt(0xC07F);
        epi_begin(pend, acc, lc, j);
        b_next = bias_at(pend, 0);   // read one tile ahead of chunk 0
      }
      epi_all(pend, dst[2 * (NT - 1)], dst[2 * (NT - 1) + 1]);
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
        for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
      }
    }

    // ---- head
    {
      const uint32_t l = n_layers - 1;
      const bool concat = (P.concat_mask >> l) & 1u;
      const uint4* slot = stage_sync();
      f32x16 acc;
      half8 A[2][GR];
      tile_mma(slot, cur, concat, acc, A, false, nullptr, [&](auto) __attribute__((always_inline)) { dma_slot(); });
      Pending pend;
      epi_begin(pend, acc, layer_consts(l), 0);
      half8 o0, o1;
      epi_chunk(pend, IC<0>{}, o0, o1, bias_at(pend, 0));
      if (h == 0 && sample_valid) {
        float bgr[3];
        const float mean[3] = {P.mean0, P.mean1, P.mean2};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          float o = (float)o0[k] * P.max;
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
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead DMA before the wave ends
#ifdef PTMI_DIAG_BUILD
  if constexpr ((DIAG & 32) != 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      g_nif_clock[0] = __builtin_amdgcn_s_memtime() - t_cycles;
      g_nif_clock[1] = __builtin_amdgcn_s_memrealtime() - t_real;
    }
  }
#endif
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

// Also counts the items that are NOT padding (u < width and v < height, the test of AccumulatedImage.cpp:66) into *n_real:
// padding items are not traced, and pt_stats.paths counts real paths only.
__global__ void unpack_records_kernel(const TraceRecordDev* rec, uint32_t n, Accum A, uint32_t width, uint32_t height,
                                      unsigned long long* n_real) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool real = false;
  if (i < n) {
    TraceRecordDev t = rec[i];
    A.pix[i] = (uint32_t)t.u | ((uint32_t)t.v << 16);
    A.r[i] = t.r; A.g[i] = t.g; A.b[i] = t.b;
    A.count[i] = t.sampleCount;
    A.length[i] = t.pathLength;
    real = t.u < width && t.v < height;
  }
  const uint64_t m = __ballot(real);
  if ((threadIdx.x & 63u) == 0 && m) atomicAdd(n_real, (unsigned long long)__popcll(m));
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
// Block reduction of the two step counters -> one 64-bit atomic pair per workgroup.
__device__ __forceinline__ void accumulate_counters(uint32_t segs, uint32_t esc, unsigned long long* counters) {
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

// One work item per thread: worklists whose size is not a multiple of four.
__global__ __launch_bounds__(256) void accumulate_kernel(uint32_t n, uint32_t iters, const uint8_t* plen, const float* rad_r,
                                                         const float* rad_g, const float* rad_b, Accum A, unsigned long long* counters) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t segs = 0, esc = 0;
  if (i < n) {
    float r = A.r[i], g = A.g[i], b = A.b[i];
    // Eight iterations' loads are issued before the first add (a path's radiance is loaded whether or not it escaped: what a
    // dead path left there is never added), so that the pass waits for memory once per eight iterations, not twice per one;
    // the adds themselves stay in iteration order.
    constexpr uint32_t U = 8;
    for (uint32_t k0 = 0; k0 < iters; k0 += U) {
      uint32_t pl[U];
      float vr[U], vg[U], vb[U];
#pragma unroll
      for (uint32_t j = 0; j < U; ++j) {
        const uint32_t k = (k0 + j < iters) ? k0 + j : iters - 1u;
        const size_t p = (size_t)k * n + i;
        pl[j] = plen[p];
        vr[j] = rad_r[p]; vg[j] = rad_g[p]; vb[j] = rad_b[p];
      }
#pragma unroll
      for (uint32_t j = 0; j < U; ++j) {
        if (k0 + j < iters) {
          segs += pl[j] & 0x7fu;
          if (pl[j] & 0x80u) {
            r += vr[j]; g += vg[j]; b += vb[j];      // :295-297
            esc += 1;
          }
        }
      }
    }
    A.r[i] = r; A.g[i] = g; A.b[i] = b;
    A.count[i] += iters;                                // :300
    A.length[i] += segs;                                // :253
  }
  accumulate_counters(segs, esc, counters);
}

// Four CONSECUTIVE work items per thread (n % 4 == 0, so path index k n + i is a multiple of four for every iteration k):
// one 4-byte load of the four path records and three 16-byte loads of their radiance per iteration instead of four 1-byte
// and twelve 4-byte ones -- a wave's load instruction covers 256 B / 1 KiB of consecutive memory, not 64 B / 256 B.  Each
// pixel's adds are the same adds in the same (iteration) order: bit-identical to the kernel above.
__global__ __launch_bounds__(256) void accumulate4_kernel(uint32_t n, uint32_t iters, const uint8_t* plen, const float* rad_r,
                                                          const float* rad_g, const float* rad_b, Accum A, unsigned long long* counters) {
  const uint32_t i = 4u * (blockIdx.x * blockDim.x + threadIdx.x);
  uint32_t segs = 0, esc = 0;
  if (i < n) {
    float4 r = *reinterpret_cast<const float4*>(A.r + i), g = *reinterpret_cast<const float4*>(A.g + i),
           b = *reinterpret_cast<const float4*>(A.b + i);
    uint32_t len[4] = {0, 0, 0, 0};
    constexpr uint32_t U = 4;   // iterations in flight per thread: 4 x 52 B
    for (uint32_t k0 = 0; k0 < iters; k0 += U) {
      uint32_t pl[U];
      float4 vr[U], vg[U], vb[U];
#pragma unroll
      for (uint32_t j = 0; j < U; ++j) {
        const uint32_t k = (k0 + j < iters) ? k0 + j : iters - 1u;
        const size_t p = (size_t)k * n + i;
        pl[j] = *reinterpret_cast<const uint32_t*>(plen + p);
        vr[j] = *reinterpret_cast<const float4*>(rad_r + p);
        vg[j] = *reinterpret_cast<const float4*>(rad_g + p);
        vb[j] = *reinterpret_cast<const float4*>(rad_b + p);
      }
#pragma unroll
      for (uint32_t j = 0; j < U; ++j) {
        if (k0 + j < iters) {
          const uint32_t w = pl[j];
          len[0] += w & 0x7fu; len[1] += (w >> 8) & 0x7fu; len[2] += (w >> 16) & 0x7fu; len[3] += (w >> 24) & 0x7fu;
          if (w & 0x80u) { r.x += vr[j].x; g.x += vg[j].x; b.x += vb[j].x; }               // :295-297
          if (w & 0x8000u) { r.y += vr[j].y; g.y += vg[j].y; b.y += vb[j].y; }
          if (w & 0x800000u) { r.z += vr[j].z; g.z += vg[j].z; b.z += vb[j].z; }
          if (w & 0x80000000u) { r.w += vr[j].w; g.w += vg[j].w; b.w += vb[j].w; }
          esc += (uint32_t)__popc(w & 0x80808080u);
        }
      }
    }
    *reinterpret_cast<float4*>(A.r + i) = r;
    *reinterpret_cast<float4*>(A.g + i) = g;
    *reinterpret_cast<float4*>(A.b + i) = b;
    uint4 cnt = *reinterpret_cast<const uint4*>(A.count + i), ln = *reinterpret_cast<const uint4*>(A.length + i);
    cnt.x += iters; cnt.y += iters; cnt.z += iters; cnt.w += iters;                        // :300
    ln.x += len[0]; ln.y += len[1]; ln.z += len[2]; ln.w += len[3];                        // :253
    *reinterpret_cast<uint4*>(A.count + i) = cnt;
    *reinterpret_cast<uint4*>(A.length + i) = ln;
    segs = len[0] + len[1] + len[2] + len[3];
  }
  accumulate_counters(segs, esc, counters);
}

// (b, g, r) / sampleCount per work item: the value AccumulatedImage::accumulate adds
// (src/AccumulatedImage.cpp:69-71), for the multi-GPU HDR gather.  The device keeps sampleCount in 32 bits, so a film
// that stays resident over many steps (count > 65535) still divides by the true count; only the TraceRecord wire
// format wraps at 16 bits (pack_records_kernel).
__global__ void export_hdr_kernel(uint32_t n, Accum A, float* bgr) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // an item that has no samples yet (after pt_setup / pt_clear_accumulators) exports 0, not 0/0
  const uint32_t cnt = A.count[i];
  const float scale = cnt ? 1.f / (float)cnt : 0.f;
  bgr[3 * (size_t)i + 0] = A.b[i] * scale;
  bgr[3 * (size_t)i + 1] = A.g[i] * scale;
  bgr[3 * (size_t)i + 2] = A.r[i] * scale;
}

// Resident film: AccumulatedImage::accumulate (src/AccumulatedImage.cpp:59-74: hdr += (b, g, r) * (1 / sampleCount))
// followed by LoadBalancer::clearInactiveAccumulators (src/LoadBalancer.cpp:198-213) for one work item, on the device.
// Same fp32 expressions as the host code (no contraction), so a film kept here equals the host film bit for bit.
// Per-tile path-length bookkeeping for the balancer (pt_tile_costs): tile of a work item, or ~0 for padding items.
struct TileGrid {
  uint32_t tile_w, tile_h, tiles_x, n_tiles;   // n_tiles = 0: bookkeeping off
  unsigned long long* cost;                     // [n_tiles] sums of pathLength
};
__device__ __forceinline__ uint32_t tile_of(uint32_t pix, const TileGrid& T) {
  const uint32_t u = pix & 0xffffu, v = pix >> 16;
  const uint32_t t = (v / T.tile_h) * T.tiles_x + u / T.tile_w;
  return (u / T.tile_w < T.tiles_x && t < T.n_tiles) ? t : ~0u;
}
__global__ void tile_cost_kernel(uint32_t n, Accum A, TileGrid T) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t len = A.length[i], t = tile_of(A.pix[i], T);
  if (len && t != ~0u) atomicAdd(&T.cost[t], (unsigned long long)len);
}

__global__ void film_accumulate_kernel(uint32_t n, Accum A, float* film, TileGrid T) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (T.n_tiles) {   // the path lengths about to be cleared go into the per-tile sums first
    const uint32_t len = A.length[i], t = tile_of(A.pix[i], T);
    if (len && t != ~0u) atomicAdd(&T.cost[t], (unsigned long long)len);
  }
  const uint32_t cnt = (uint16_t)A.count[i];     // the host divides by the uint16 wire field
  if (cnt) {
    const float scale = 1.f / (float)cnt;
    film[3 * (size_t)i + 0] += A.b[i] * scale;
    film[3 * (size_t)i + 1] += A.g[i] * scale;
    film[3 * (size_t)i + 2] += A.r[i] * scale;
  }
  A.r[i] = 0.f; A.g[i] = 0.f; A.b[i] = 0.f;
  A.count[i] = 0; A.length[i] = 0;
}

}  // namespace ptd
