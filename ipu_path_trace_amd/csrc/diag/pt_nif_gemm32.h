// pt_nif_gemm32.h -- PROFILING BUILD ONLY: the round-2 layer-by-layer kernels on v_mfma_f32_32x32x16_f16 (32-sample
// pieces of 16 inputs), kept as the A/B baseline of the 16x16x32 kernels in ../pt_nif_gemm.h (PTMI_GEMM_SHAPE=32 at
// pt_upload_nif selects them; scripts/ab_c5.py).  Shared declarations (NifGemmParams, ring constants, scan, find_tile)
// come from ../pt_nif_gemm.h.
#pragma once
#include "pt_nif_gemm.h"

namespace ptd {

constexpr int kGemmKps = 2;                     // k-steps (of 16) per stage

// One dense layer over a chunk: D[256 features x 256 samples] per workgroup pass, 8 waves of 128 x 64
// (4 x 2 accumulator tiles of 32 x 32).  Weights and activations arrive by LDS-DMA into a ring of four stages of
// two k-steps (8 + 8 pieces each); wave w loads weight tile w and sample tile w, both k-steps of a tile with one M0
// set-up, a uniform base address (SGPR pair) + constant per-lane offset and the instruction's immediate offset for
// the second piece (a tile's k-steps are contiguous in memory and in the slot: [A: tile][k][1 KiB] | [B: tile][k][1 KiB];
// with an odd k-step count the last stage's second piece is whatever follows in memory -- in bounds, never multiplied).
// Every wave issues exactly four loads per stage, so one counted s_waitcnt covers the ring, and the loader's cursor runs
// ahead across output blocks, so a block's epilogue stores overlap the next block's first loads.
//
// Block order: workgroup g sits on XCD g % 8; the n_ftiles / 8 feature blocks of one sample block run at the
// same time on the same XCD, so the sample block's activation pieces are fetched from HBM / Infinity Cache once
// and hit that XCD's L2 for the other feature blocks; the layer's weights (<= 2 MiB) stay in every L2.
//
// Ping-pong.  The round-1 kernel (nifg_layer_v1_kernel in pt_nif_variants.h) let every wave interleave its own LDS-DMA
// issue, fragment reads and MFMAs; the two waves of a SIMD (w and w + 4) then ran the same schedule and stalled at the
// same points (MFMA pipe 53 % busy, profiles/r02b_c5_pmc.json).  Here the workgroup's halves alternate ROLES phase by
// phase, separated by a workgroup barrier (the arrangement of the guide's 256^2 8-phase GEMM, cdna_hip_programming.md
// section 5): in every phase one wave of each SIMD issues its eight MFMAs of a k-step back to back at raised priority
// while its partner issues everything else -- one paired LDS-DMA load for the stage three ahead, the counted wait, the
// six fragment reads for ITS next k-step -- so the matrix pipe always has an issuer.  Fragments are read one phase
// before they are multiplied (single-buffered: 24 registers instead of 48).  Both halves run the same code; waves 4-7
// run it one phase late.  +3.7 % end to end at C5 (profiles/r02_c5_ablation.txt, which also records what did NOT help:
// dedicated loader waves, deferred stores, and that the barriers cost nothing).
//
//   global phase                   4S              4S+1            4S+2            4S+3            4S+4
//   waves 0-3                      MFMA k0(S)      A-pair, wait,   MFMA k1(S)      B-pair, read    MFMA k0(S+1)
//                                                  read k1(S)                      k0(S+1)
//   waves 4-7 (one phase late)     B-pair, read    MFMA k0(S)      A-pair, wait,   MFMA k1(S)      B-pair, read
//                                  k0(S)                           read k1(S)                      k0(S+1)
//
// "wait" = my pieces of stage S + 1 have landed (counted vmcnt); the whole of stage S + 1
// is certified (both halves have waited by the end of phase 4S+2) before its first reader (waves 0-3 in phase 4S+3).
// The slot of stage S + 3 is that of stage S - 1, whose last readers finished in phase 4S-1.  Same arithmetic, rounding points and store layout
// as the fused kernels.  DIAG (timing-only builds, results invalid): bit 0 = no loads into the ring, bit 1 = no LDS reads
// of fragments, bit 2 = no barrier, bit 3 = every load from one L2-hot piece, bit 4 = activation loads from one L2-hot piece.
template <int DIAG>
__global__ __launch_bounds__(512, 2) void nifg_layer_kernel(const NifGemmParams P) {
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
  const bool second = wave >= 4;   // the SIMD partners of waves 0-3 (MI355X_MICROARCH.md, "Two waves per SIMD")

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

  // ---- loader: wave w loads weight tile w and sample tile w of stage (pf_it, pf_st) -> ring slot pf_q % R
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
    if constexpr (DIAG & 8) base = reinterpret_cast<const char*>(P.wpack) + ((size_t)wave << 11);
    if constexpr (DIAG & 16) { if (which) base = reinterpret_cast<const char*>(P.wpack) + ((size_t)wave << 11); }
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
      if (xcd + 8u * (sbi0 + spx * (pf_it + 1u)) < nsb) { pf_it += 1; pf_st = 0; }
      else pf_st = nst - 1u;
    }
  };
  // Phases per stage.  Four (product): MFMA k0 | load | MFMA k1 | load.  Two (DIAG bit 7, valid results): one MFMA phase of
  // 16 MFMAs and one load phase per stage with the loader four stages ahead -- measured equal (1084 vs 1087 TFLOP/s at C5,
  // profiles/r02_c5_ablation.txt) at 190 instead of 166 VGPRs, so the shorter phases stay.
  constexpr bool kTwoPhase = (DIAG & 128) != 0;
  constexpr int kAhead = kTwoPhase ? R : R - 1;   // stages the loader runs ahead of the stage being multiplied
#pragma unroll
  for (int k = 0; k < kAhead; ++k) {
    issue_pair(0);
    issue_pair(1);
    stage_issued();
  }

  half8 FA[kTwoPhase ? 2 : 1][4], FBv[kTwoPhase ? 2 : 1][2];
  auto read_frags = [&](const uint4* slot, int kk, auto setc) __attribute__((always_inline)) {
    constexpr int set = decltype(setc)::value;
    if constexpr (!(DIAG & 2)) {
#pragma unroll
      for (int a = 0; a < 4; ++a) FA[set][a] = as_half8(slot[(4 * wm + a) * 128 + 64 * kk]);
#pragma unroll
      for (int b = 0; b < 2; ++b) FBv[set][b] = as_half8(slot[1024 + (2 * wn + b) * 128 + 64 * kk]);
    }
  };
  // DIAG bit 6 (64): s_memtime stamps around every barrier of block 1's first 16 stages, waves 0 and 4 of workgroup 0,
  // parked in the unused half of the bias area and copied out at the end (stamps[wave >> 2][stage][8]: entry and exit of the
  // stage's phase barriers).  Valid results; the stamps cost a few per cent.
  uint32_t stamp_it = 0, stamp_st = 0;
  auto stamp = [&](int idx) __attribute__((always_inline)) {
    if constexpr (DIAG & 64) {
      if (blockIdx.x == 0 && (wave & 3) == 0 && stamp_it == 1u && stamp_st < 16u) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) reinterpret_cast<unsigned long long*>(bias_lds + 2048)[(((uint32_t)wave >> 2) * 16u + stamp_st) * 8u + (uint32_t)idx] = t;
      }
    }
  };
  int phase_no = 0;
  auto phase_end = [&]() __attribute__((always_inline)) {
    stamp(2 * (phase_no & 3));
    if constexpr (!(DIAG & 4)) asm volatile("s_barrier" ::: "memory");
    stamp(2 * (phase_no & 3) + 1);
    phase_no += 1;
    __builtin_amdgcn_sched_barrier(0);   // nothing, MFMAs included, moves across a phase boundary
  };
  uint32_t q = 0;            // consumer stage
  uint32_t since_store = 2;  // stages since the last epilogue's 16 stores entered the vmcnt queue
  // Counted wait of a load phase (the previous block's 16 stores count while they are younger than the stage awaited).
  // Four phases: my pieces of stage q + 1 have landed when at most the pair just issued for stage q + 3 and the four loads
  // of stage q + 2 are outstanding.  Two phases: the fragments of stage q + 1 are read in THIS phase, so that stage was
  // certified a phase ago and the wait is for stage q + 2: outstanding at most the four loads just issued for q + 4 and
  // the four of q + 3.
  auto wait_ahead = [&]() __attribute__((always_inline)) {
    if (nst < 4u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (kTwoPhase) {
      if (since_store < 2u) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      if (since_store < 2u) asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
  };

  {  // prologue: stage 0 (two phases: and stage 1) certified here, the next one by the first stage's own wait
    if (nst < 4u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    phase_end();
    read_frags(reinterpret_cast<const uint4*>(ring) + lane, 0, IC<0>{});
    if constexpr (kTwoPhase) read_frags(reinterpret_cast<const uint4*>(ring) + lane, 1, IC<1>{});
  }
  if constexpr (DIAG & 2) {
#pragma unroll
    for (int a = 0; a < 4; ++a) FA[0][a] = as_half8(reinterpret_cast<const uint4*>(bias_lds)[a * 64 + lane]);
    FBv[0][0] = FA[0][1]; FBv[0][1] = FA[0][2];
    if constexpr (kTwoPhase) {
#pragma unroll
      for (int a = 0; a < 4; ++a) FA[1][a] = FA[0][a];
      FBv[1][0] = FA[0][2]; FBv[1][1] = FA[0][1];
    }
  }
  // Both halves run the SAME phase sequence; waves 4-7 run it one phase late, which is what makes the roles alternate.
  // They pay the offset with one barrier here, waves 0-3 with one at the very end.
  if (second) phase_end();

  for (uint32_t it = 0;; ++it) {
    const uint32_t sb = xcd + 8u * (sbi0 + spx * it);
    if (sb >= nsb) break;
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) { acc[a][0] = (f32x16)(0.0f); acc[a][1] = (f32x16)(0.0f); }
    auto multiply = [&](auto setc) __attribute__((always_inline)) {
      constexpr int set = decltype(setc)::value;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[set][a], FBv[set][0], acc[a][0], 0, 0, 0);
        acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FA[set][a], FBv[set][1], acc[a][1], 0, 0, 0);
      }
    };

    for (uint32_t st = 0; st < nst; ++st) {
      const uint4* slot = reinterpret_cast<const uint4*>(ring + (q % R) * kGemmStageBytes) + lane;
      const uint4* next = reinterpret_cast<const uint4*>(ring + ((q + 1u) % R) * kGemmStageBytes) + lane;
      q += 1;
      const bool two = kGemmKps * st + 1u < nks;   // (an odd k-step count: the last stage has one k-step)
      stamp_it = it; stamp_st = st; phase_no = 0;
      if constexpr (kTwoPhase) {
        // ---- MFMA phase: both k-steps of the stage
        __builtin_amdgcn_s_setprio(1);
        multiply(IC<0>{});
        if (two) multiply(IC<1>{});
        __builtin_amdgcn_s_setprio(0);
        phase_end();
        // ---- load phase: both pairs of the stage four ahead (into the slot of the stage just multiplied: its fragments are
        // in registers, and the partner half drained its reads of it before the barrier that started this phase), certify
        // my share of the stage two ahead, all twelve fragments of the next stage; the reads are drained before the
        // phase ends, so the next phase's loads may overwrite their slot
        read_frags(next, 0, IC<0>{});      // fragment reads first: their latency passes under the LDS-DMA issue and wait below
        read_frags(next, 1, IC<1>{});
        issue_pair(0);
        issue_pair(1);
        stage_issued();
        wait_ahead();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        phase_end();
      } else {
        // ---- MFMA phase, k-step 0
        __builtin_amdgcn_s_setprio(1);
        multiply(IC<0>{});
        __builtin_amdgcn_s_setprio(0);
        phase_end();
        // ---- load phase: weights of the stage three ahead, certify my share of the next stage, fragments of k-step 1
        // (the fragment reads go first: issued right before the barrier, their LDS latency would open the next MFMA phase;
        // here it passes under the LDS-DMA issue and the counted wait)
        read_frags(slot, 1, IC<0>{});
        issue_pair(0);
        wait_ahead();
        phase_end();
        // ---- MFMA phase, k-step 1
        if (two) {
          __builtin_amdgcn_s_setprio(1);
          multiply(IC<0>{});
          __builtin_amdgcn_s_setprio(0);
        }
        phase_end();
        // ---- load phase: activations of the stage three ahead, fragments of the next stage's k-step 0 (certified by the
        // barrier that ended this wave's previous load phase at the latest)
        read_frags(next, 0, IC<0>{});
        issue_pair(1);
        stage_issued();
        phase_end();
      }
      since_store += 1;
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
  stamp_st = 99;
  if (!second) phase_end();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the run-ahead loads before the wave ends
  if constexpr (DIAG & 64) {
    if (blockIdx.x == 0 && (wave & 3) == 0 && P.stamps && nst >= 16u) {
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(bias_lds + 2048) + ((uint32_t)wave >> 2) * 128u;
      for (int i = lane; i < 128; i += 64) P.stamps[((uint32_t)wave >> 2) * 128u + i] = src[i];
    }
  }
}

// Fourier features of a chunk as B pieces (NifModel.cpp:185-218): feat[tile][E / 4].  One wave per tile.
template <int E>
__global__ __launch_bounds__(256) void nifg_encode_kernel(const NifParams P, const uint32_t* tile_start, uint32_t tile0,
                                                           uint32_t chunk_tiles, uint4* feat) {
  constexpr int IS = E / 4;
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
    const float x = (coord - 1.0f) * 2.0f;
#pragma unroll
    for (int s = 0; s < IS; ++s) {
      union { half8 hh; uint4 u; } f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a = (float)(_Float16)(x * (float)(1u << (4 * s + k)));
        float sn, cs;
        fast_sincos(a, sn, cs);
        if (s == IS - 1 && k > 0 && (uint32_t)(4 * s + k) >= P.n_freq) { sn = 0.f; cs = 0.f; }   // padded frequency slot (E rounded up to 4 | E)
        f.hh[k] = (_Float16)sn;
        f.hh[4 + k] = (_Float16)cs;
      }
      feat[((size_t)lt * IS + s) * 64 + lane] = f.u;
    }
  }
}

// Head (3 outputs = rows 0..2 of one 32-row tile), decode (NifModel.cpp:221-245) and scatter
// (codelets.cpp:366-382).  A wave takes four sample tiles so a weight piece is fetched once per 128 samples.
__global__ __launch_bounds__(256) void nifg_head_kernel(const NifParams P, const NifGemmParams G, const uint32_t* tile_start) {
  constexpr int NB = 4;
  __shared__ uint32_t ts[kMaxRegions + 1];
  const uint32_t ntiles = chunk_tile_count(tile_start + P.n_regions, G.tile0, G.chunk_tiles);
  if (blockIdx.x * 4u * NB >= ntiles) return;
  for (uint32_t i = threadIdx.x; i <= P.n_regions; i += 256u) ts[i] = tile_start[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, h = lane >> 5;
  const uint4* wj = G.wpack + (size_t)G.piece_base * 64 + lane;
  for (uint32_t lt0 = (blockIdx.x * 4u + wave) * NB; lt0 < ntiles; lt0 += gridDim.x * 4u * NB) {
    f32x16 acc[NB];
    const uint4* xb[NB];
    const uint4* fbp[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      acc[b] = (f32x16)(0.0f);
      const uint32_t lt = (lt0 + b < ntiles) ? lt0 + b : ntiles - 1u;   // clamp: recomputes a valid tile, not stored
      xb[b] = G.act_in + (size_t)lt * G.act_stride * 64 + lane;
      fbp[b] = G.feat + (size_t)lt * G.feat_stride * 64 + lane;
    }
    for (uint32_t s = 0; s < G.ks_act; ++s) {
      const half8 a = as_half8(wj[(size_t)s * 64]);
#pragma unroll
      for (int b = 0; b < NB; ++b)
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, as_half8(xb[b][(size_t)s * 64]), acc[b], 0, 0, 0);
    }
    for (uint32_t s = 0; s < G.ks_in; ++s) {
      const half8 a = as_half8(wj[(size_t)(G.ks_act + s) * 64]);
#pragma unroll
      for (int b = 0; b < NB; ++b)
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, as_half8(fbp[b][(size_t)s * 64]), acc[b], 0, 0, 0);
    }
    const uint4* bp = G.bpack + ((size_t)G.bias_base * 2 + h) * 2;
    const half8 b_lo = as_half8(bp[0]);
    const float mean[3] = {P.mean0, P.mean1, P.mean2};
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (lt0 + b >= ntiles) continue;
      const TileRef r = find_tile(ts, P.n_regions, P, G.tile0 + lt0 + b);
      if (h == 0 && r.local + c < r.count) {
        float bgr[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          _Float16 o16 = (_Float16)acc[b][k];
          o16 = o16 + b_lo[k];
          if (G.relu) o16 = o16 > (_Float16)0.0f ? o16 : (_Float16)0.0f;
          float o = (float)o16 * P.max;
          o = o + mean[k];
          bgr[k] = P.log_tonemap ? __expf(o) : o;
        }
        const uint32_t qi = r.qbase + c;
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

}  // namespace ptd
