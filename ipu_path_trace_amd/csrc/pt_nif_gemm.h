// pt_nif_gemm.h -- wide NIFs (hidden > 320, BASELINE config C5: 8 x 1024) layer by layer, v_mfma_f32_16x16x32_f16.
//
// A 1024-wide activation vector fits neither a wave's registers nor, for more than 64 samples, a CU's LDS, and a
// 64-sample tile re-streams 2 MiB of weights per layer for 134 MFLOP (64 FLOP per weight byte).  At this width a layer
// is a large enough GEMM (2 K N = 2.1 MFLOP against 4 KiB of activation traffic per sample) to run on its own: the queue
// is cut into chunks of a few thousand 32-sample tiles whose activations ping-pong between two HBM buffers, and each
// layer (NifModel.cpp:295-326: matMul, + bias, ReLU) is one launch of nifg16_layer_kernel over the chunk.
//
// Round 3: the 16x16x32 MFMA shape.  On this chip an MFMA-dense loop is clock-limited by power, and the 16x16x32 shape
// holds a higher clock than 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md, "DVFS give-back" item 7); the
// round-2 kernel (32x32x16; removed in round 5, numbers in profiles/r03_c5_ablation.txt) sat on that power line.
// The phase structure is unchanged; what changed is the fragment contract:
//
//   weights      piece (s, f)   = A operand of k-step s (32 inputs), feature tile f (16 outputs): lane (r = lane & 15,
//                                 q = lane >> 4) holds W^T[16 f + r][k(q, 0..7)]                     (pack_nif_g16, ptmi.hip)
//   activations  piece (T, s, h) = B operand of sample tile 2 T + h (16 samples; T = the queue's 32-sample tile), k-step s:
//                                 lane (c = lane & 15, q) holds sample c, inputs k(q, 0..7)
//   k(q, e) = 32 s + (e < 4 ? 4 q + e : 16 + 4 q + (e - 4))
//
// so that an accumulator tile pair -- features 32 j .. 32 j + 15 and 32 j + 16 .. 32 j + 31 of 16 samples: the C layout
// has the sample on lane & 15 and features 4 q .. 4 q + 3 in the four registers -- rounded to fp16 with bias and ReLU
// applied IS piece (T, j, h) of the next layer: the epilogue stores whole pieces and the next layer's loader copies
// pieces global -> LDS by DMA with no transposition and no bank conflict.  Memory order of the activation pieces is
// [T][s][h], of the weight pieces [s][f]: the two pieces a wave loads per operand and stage are adjacent (one M0 set-up,
// one uniform base, the instruction's immediate offset for the second piece).
// Rounding points are those of NifModel.cpp:295-326, identical to the fused kernels (pt_nif.h).
//
// The 3-wide head (NifModel.cpp:295-326 once more, then buildDecodeOutput :221-245) is fused into the LAST hidden
// layer's epilogue: the fp16 output pieces, still in registers, are multiplied with the head's weight pieces (16 more
// MFMAs per wave and block against 1024), the three partial sums per sample -- one per (feature block, wave row) --
// go to a small fp32 buffer, and nifg16_finish_kernel adds them in a fixed order, adds the head's own Fourier-feature
// inputs where it has any, rounds to fp16, adds the bias, decodes and scatters.  The last layer's 256 MiB of
// activations per chunk are neither written nor read back (round 2: a 61-64 us head launch per chunk).
#pragma once
#include "pt_nif.h"

namespace ptd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct NifGemmParams {
  const uint4* wpack;        // all weight pieces
  const uint4* bpack;        // all bias tiles (64 B each)
  uint32_t piece_base;       // first piece of this layer
  uint32_t bias_base;        // first bias tile of this layer
  uint32_t ks_act, ks_in;    // k-steps taken from activations / from the Fourier-feature pieces
  uint32_t relu;
  uint32_t n_ftiles;         // 32-feature output groups of this layer (multiple of 8)
  const uint4* act_in;       // activation pieces of the previous layer
  const uint4* feat;         // Fourier-feature pieces
  uint4* act_out;            // activation pieces of this layer
  uint32_t act_stride, feat_stride;   // k-steps per sample tile in act_* / feat
  const uint32_t* total_tiles;   // device scalar: 32-sample tiles in the queue
  uint32_t tile0, chunk_tiles;   // this launch covers queue tiles [tile0, tile0 + chunk_tiles)
  unsigned long long* stamps;    // profiling build only: time stamps, else nullptr
  // fused head (last hidden layer only)
  uint32_t head_piece_base;      // first piece of the head: [k-step] pieces of one 16-row tile (rows 0..2 = B, G, R)
  float4* head_partial;          // [slice = 2 fb + wm][chunk sample] partial sums (x, y, z = B, G, R)
  uint32_t partial_stride;       // samples per slice (= chunk_tiles * 32)
};

constexpr int kGemmStages = 4;                  // (32-shape kernels of the profiling build: ring slots of A + B)
constexpr int kGemmStageBytes = 32 * 1024;      // (32-shape kernels: one stage, 16 A pieces + 16 B pieces)
constexpr int kGemmASlots = 4;                  // ring of weight half-stages (16 pieces = 16 KiB each)
constexpr int kGemmBSlots = 4;                  // ring of activation half-stages (5 = "B leads", measured below: no gain)
constexpr int kGemmHalfBytes = 16 * 1024;
constexpr int kGemmBiasBytes = 4096;            // up to 64 output groups of 32 features
constexpr int kGemmHeadBytes = 8192;            // the head's 8 pieces of one 256-feature block
constexpr int kGemmLdsBytes = kGemmBiasBytes + kGemmHeadBytes + (kGemmASlots + 5) * kGemmHalfBytes;   // 156 KiB of 160 (room for the B-leads build)

// Tiles of the queue in this chunk (0 if the queue ends before it).
__device__ __forceinline__ uint32_t chunk_tile_count(const uint32_t* total_tiles, uint32_t tile0, uint32_t chunk_tiles) {
  const uint32_t total = *total_tiles;
  if (total <= tile0) return 0u;
  return (total - tile0 < chunk_tiles) ? total - tile0 : chunk_tiles;
}

// One dense layer over a chunk: D[256 features x 256 samples] per workgroup pass, 8 waves of 128 x 64 (8 x 4 accumulator
// tiles of 16 x 16 = 128 registers).  Weights and activations arrive by LDS-DMA into a ring of four stages of one k-step
// (16 + 16 pieces, 32 KiB); wave w loads weight tiles 2 w, 2 w + 1 and the 32-sample tile w (= sample tiles 2 w, 2 w + 1),
// each pair with one M0 set-up.  Every wave issues exactly four loads per stage, so one counted s_waitcnt covers the ring,
// and the loader's cursor runs ahead across output blocks, so a block's epilogue stores overlap the next block's first loads.
//
// Block order: workgroup g sits on XCD g % 8; the n_ftiles / 8 feature blocks of one sample block run at the same time
// on the same XCD, so the sample block's activation pieces are fetched from HBM / Infinity Cache once and hit that
// XCD's L2 for the other feature blocks; the layer's weights (<= 2 MiB) stay in every L2.
//
// Ping-pong (round 2, cdna_hip_programming.md section 5, the 256^2 8-phase GEMM): the workgroup's halves alternate ROLES
// phase by phase, separated by a workgroup barrier: in every phase one wave of each SIMD issues its sixteen MFMAs (one
// 64 x 64 quadrant of its tile over the stage's K = 32) back to back at raised priority while its partner issues
// everything else -- one paired LDS-DMA load for the stage three ahead, the counted wait, the fragment reads for ITS next
// MFMA phase -- so the matrix pipe always has an issuer.  Both halves run the same code; waves 4-7 run it one phase late.
//
//   global phase                   4S              4S+1            4S+2            4S+3            4S+4
//   waves 0-3                      MFMA q0(S)      A-pair, wait,   MFMA q1(S)      B-pair, read    MFMA q0(S+1)
//                                                  read A'(S)                      A, B (S+1)
//   waves 4-7 (one phase late)     B-pair, read    MFMA q0(S)      A-pair, wait,   MFMA q1(S)      B-pair, read
//                                  A, B (S)                        read A'(S)                      A, B (S+1)
//
// Round 4: the streams no longer push the weights out of L2.  One XCD runs 8 sample blocks x 4 feature blocks at a time; between
// two uses of a weight line (one block, ~25 us) its L2 (4 MiB) saw 8 x (512 KiB of activations in + 512 KiB out), so every
// sample block fetched the layer's 2 MiB of weights again (FETCH + WRITE 671.6 MB per hidden layer = 1.25 x algorithmic).  The
// activation loads now carry `nt` and the output stores `sc1` (write-through, line dropped): 563.7 MB = 1.05 x algorithmic, +1.2 %
// (profiles/r04_c5_ablation.txt; either hint alone: +0.4 % / see there).  DIAG bits 13 / 10 / 11 switch them back for the A/B.
//
// Where the time goes (round 3, profiles/r03_c5_ablation.txt: in-kernel s_memtime / s_memrealtime stamps, timing-only
// builds).  With every load redirected to an L2-hot piece the kernel needs exactly the cycles it needs with no loads at all
// (MFMA pipe 78 % busy inside a block) -- the LDS-DMA issue, the ring and the barriers cost no cycles -- but runs at
// 1.85 GHz instead of 2.18: moving 32 KiB per stage L2 -> LDS -> registers is paid in clock (power), not in stalls.  Fed
// from memory, the activation stream adds 13 % cycles (the weights, L2-resident, add none) while the clock recovers to
// 2.0-2.1 GHz: the chip is power-limited either way.  Tried against the activation stalls, all within +-0.5 %: an
// activation ring one slot deeper than the weight ring ("B leads": RB = 5, DIAG bit 9 selects it; a wave's vector-memory
// operations retire in order, so the counted wait for stage S + 1 covers every older load and B can lead A by one stage
// at most), write-through (sc1) or non-temporal output stores ALONE (round 4: together with nt loads they are worth
// +1.2 %), chunks small enough for both activation buffers to stay in the Infinity Cache.  One workgroup taking the four feature blocks of its sample block in turn
// instead of four neighbours sharing them through L2 (bit 12): 16 % slower.
//
// q0 = feature tiles 0-3 of the wave x its four sample tiles, q1 = feature tiles 4-7 x the same four (the B fragments
// stay, the A fragments are re-read: A').  "wait" = my pieces of stage S + 1 have landed (counted vmcnt); the whole of
// stage S + 1 is certified (both halves have waited by the end of phase 4S+2) before its first reader (waves 0-3 in
// phase 4S+3).  The slot of stage S + 3 is that of stage S - 1, whose last readers finished in phase 4S-1.
// DIAG (timing-only builds, results invalid): bit 0 = no loads into the ring, bit 1 = no LDS reads of fragments,
// bit 3 = every load from one L2-hot piece, bit 4 = activation loads from one L2-hot piece, bit 6 / 7 = no weight / no activation loads, bit 8 = the activation pair as plain
// loads into (unused) registers instead of LDS-DMA.  Bit 5 (valid results): in-kernel clock stamps of workgroup 0.
template <int FUSE_HEAD, int DIAG>
__global__ __launch_bounds__(512, 2) void nifg16_layer_kernel(const NifGemmParams P) {
  constexpr int RA = kGemmASlots;
  constexpr int RB = (DIAG & 512) ? 5 : kGemmBSlots;   // DIAG bit 9 (valid results): the activation ring one slot deeper, its loads one stage further ahead
  constexpr int LEAD = RB - RA;
  constexpr int kStores = FUSE_HEAD ? 4 : 16;   // vector-memory stores per wave and block (they share the vmcnt queue)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* bias_lds = smem;
  char* head_lds = smem + kGemmBiasBytes;
  char* ring_a = head_lds + kGemmHeadBytes;
  char* ring_b = ring_a + RA * kGemmHalfBytes;
  const uint32_t ring_a_lds = __builtin_amdgcn_readfirstlane(
      (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring_a);
  const uint32_t ring_b_lds = ring_a_lds + RA * kGemmHalfBytes;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int q4 = lane >> 4;
  const bool second = wave >= 4;   // the SIMD partners of waves 0-3 (MI355X_MICROARCH.md, "Two waves per SIMD")

  const uint32_t ntiles = chunk_tile_count(P.total_tiles, P.tile0, P.chunk_tiles);
  const uint32_t nsb = (ntiles + 7u) / 8u;                 // sample blocks of 8 queue tiles (256 samples)
  const uint32_t FB = P.n_ftiles / 8u;                     // feature blocks of 256
  const uint32_t NF16 = P.n_ftiles * 2u;                   // 16-feature tiles of the layer
  const uint32_t xcd = blockIdx.x & 7u, cidx = blockIdx.x >> 3, cpx = gridDim.x >> 3;
  const uint32_t fb0 = cidx % FB, sbi0 = cidx / FB, spx = cpx / FB;
  // Block it of this workgroup.  Product: the FB feature blocks of a sample block are FB neighbouring workgroups (same XCD,
  // running at the same time).  DIAG bit 12 (valid results, FUSE_HEAD = 0 only): one workgroup takes the FB feature blocks
  // of its sample block one after the other.
  constexpr bool kSeqFb = (DIAG & 4096) != 0 && !FUSE_HEAD;
  auto blk_fb = [&](uint32_t it) -> uint32_t { return kSeqFb ? it % FB : fb0; };
  auto blk_sb = [&](uint32_t it) -> uint32_t { return kSeqFb ? xcd + 8u * (cidx + cpx * (it / FB)) : xcd + 8u * (sbi0 + spx * it); };
  if (blk_sb(0) >= nsb) return;                            // nothing for this workgroup (uniform)
  const uint32_t nst = P.ks_act + P.ks_in;                 // stages = k-steps

  // the output stores go through a buffer descriptor, for their cache-policy bits
  const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(P.act_out ? P.act_out : P.act_in), 0, 0x7fffffff, 0x00020000);
  const uint32_t relu_floor = P.relu ? 0u : kLinearFloor;   // packed fp16 pair (pt_nif.h: a quiet NaN makes the max an identity)
  unsigned long long t_cycles = 0, t_real = 0;
  if constexpr (DIAG & 32) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { t_cycles = __builtin_amdgcn_s_memtime(); t_real = __builtin_amdgcn_s_memrealtime(); }
  }

  for (uint32_t i = threadIdx.x; i < P.n_ftiles * 4u; i += 512u)
    reinterpret_cast<uint4*>(bias_lds)[i] = P.bpack[(size_t)P.bias_base * 4u + i];
  if constexpr (FUSE_HEAD) {   // the head's pieces of this feature block: k-steps 8 fb .. 8 fb + 7
    reinterpret_cast<uint4*>(head_lds)[threadIdx.x] = P.wpack[((size_t)P.head_piece_base + 8u * fb0) * 64u + threadIdx.x];
  }
  __syncthreads();

  // ---- loader: wave w loads weight tiles 2w, 2w+1 (cursor pa) and sample tiles 2w, 2w+1 (cursor pb) of a stage into the
  // wave's 2 KiB of that operand's ring slot
  const uint32_t lane16 = (uint32_t)lane * 16u;
  struct Cursor { uint32_t it, st, slot; };
  Cursor pa{0, 0, 0}, pb{0, 0, 0};
  auto advance = [&](Cursor& c, uint32_t slots) {
    c.slot = (c.slot + 1u == slots) ? 0u : c.slot + 1u;
    c.st += 1;
    if (c.st == nst) {
      if (blk_sb(c.it + 1u) < nsb) { c.it += 1; c.st = 0; }
      else c.st = nst - 1u;   // past the last block: the same stage again (in bounds, never multiplied)
    }
  };
  auto issue_a = [&]() {
    const char* base = reinterpret_cast<const char*>(P.wpack) + ((size_t)(P.piece_base + pa.st * NF16 + blk_fb(pa.it) * 16u + 2u * (uint32_t)wave) << 10);
    if constexpr (DIAG & 8) base = reinterpret_cast<const char*>(P.wpack) + ((size_t)wave << 11);
    const uint32_t dst = ring_a_lds + pa.slot * kGemmHalfBytes + ((uint32_t)wave << 11);
    if constexpr (!(DIAG & 1) && !(DIAG & 64)) glds16x2(base, lane16, dst);
    advance(pa, RA);
  };
  auto issue_b = [&]() {
    const uint32_t t = blk_sb(pb.it) * 8u + (uint32_t)wave;
    const char* base = (pb.st < P.ks_act) ? reinterpret_cast<const char*>(P.act_in) + (((size_t)t * P.act_stride + pb.st) << 11)
                                          : reinterpret_cast<const char*>(P.feat) + (((size_t)t * P.feat_stride + (pb.st - P.ks_act)) << 11);
    if constexpr ((DIAG & 8) || (DIAG & 16)) base = reinterpret_cast<const char*>(P.wpack) + ((size_t)wave << 11);
    const uint32_t dst = ring_b_lds + pb.slot * kGemmHalfBytes + ((uint32_t)wave << 11);
    if constexpr ((DIAG & 256) != 0) {                             // timing only: the activation pair as plain loads to registers
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      u32x4 d0, d1;
      asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                   : "=&v"(d0), "=&v"(d1) : "v"(lane16), "s"(reinterpret_cast<uint64_t>(base)) : "memory");
      asm volatile("" ::"v"(d0), "v"(d1));
    } else if constexpr (!(DIAG & 1) && !(DIAG & 128)) {
      // the activation stream is read once: non-temporal, so that its lines are the first to leave L2 (bit 13: plain loads)
      if constexpr ((DIAG & 8192) == 0) glds16x2_nt(base, lane16, dst);
      else glds16x2(base, lane16, dst);
    }
    advance(pb, RB);
  };
  // prologue: A(0), B(0), ..., A(RA - 2), B(RA - 2), then the stages B leads by
#pragma unroll
  for (int k = 0; k < RA - 1; ++k) { issue_a(); issue_b(); }
#pragma unroll
  for (int k = 0; k < LEAD; ++k) issue_b();

  half8 FA[4], FBv[4];
  auto read_a = [&](uint32_t slot, int half) __attribute__((always_inline)) {
    if constexpr (!(DIAG & 2)) {
      const uint4* p = reinterpret_cast<const uint4*>(ring_a + slot * kGemmHalfBytes) + lane;
#pragma unroll
      for (int a = 0; a < 4; ++a) FA[a] = as_half8(p[(8 * wm + 4 * half + a) * 64]);
    }
  };
  auto read_b = [&](uint32_t slot) __attribute__((always_inline)) {
    if constexpr (!(DIAG & 2)) {
      const uint4* p = reinterpret_cast<const uint4*>(ring_b + slot * kGemmHalfBytes) + lane;
#pragma unroll
      for (int b = 0; b < 4; ++b) FBv[b] = as_half8(p[(4 * wn + b) * 64]);
    }
  };
  auto phase_end = [&]() __attribute__((always_inline)) {
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);   // nothing, MFMAs included, moves across a phase boundary
  };
  uint32_t qa = 0, qb = 0;   // ring slots of the consumer's stage
  uint32_t since_store = 2;  // stages since the last epilogue's stores entered the vmcnt queue
  // Counted wait of load phase A, issued right after A(q + 3): my pieces of stage q + 1 have landed when at most the loads
  // younger than A(q + 1) are outstanding -- A(q + 2), A(q + 3) and B(q + 2 + LEAD), B(q + 3 + LEAD)... i.e. two A pairs and
  // two B pairs whatever the lead (B(q + 1 + LEAD) was issued just before A(q + 2)'s stage; with the lead it is B(q + 2) and
  // B(q + 3) that follow A(q + 1), without it B(q + 1) is older and B(q + 2) younger: one B pair less) -- plus the previous
  // block's stores while they are younger than the stage awaited.
  constexpr int kYoung = 2 * (2 + 1 + LEAD);   // loads issued after A(q + 1), up to and including A(q + 3)
  auto wait_ahead = [&]() __attribute__((always_inline)) {
    if (nst < 6u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (since_store < 2u) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kYoung + kStores) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kYoung) : "memory");
  };

  {  // prologue: stage 0 certified here (everything issued after B(0) may be outstanding), the next one by the first stage's wait
    if (nst < 6u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (2 * (RA - 2) + LEAD)) : "memory");
    phase_end();
    read_a(0, 0);
    read_b(0);
  }
  if constexpr (DIAG & 2) {
#pragma unroll
    for (int a = 0; a < 4; ++a) { FA[a] = as_half8(reinterpret_cast<const uint4*>(bias_lds)[a * 64 + lane]); FBv[a] = FA[a]; }
  }
  // Both halves run the SAME phase sequence; waves 4-7 run it one phase late, which is what makes the roles alternate.
  // They pay the offset with one barrier here, waves 0-3 with one at the very end.
  if (second) phase_end();

  for (uint32_t it = 0;; ++it) {
    const uint32_t sb = blk_sb(it), fb = blk_fb(it);
    if (sb >= nsb) break;
    f32x4 acc[8][4];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4)(0.0f);
    auto multiply = [&](auto halfc) __attribute__((always_inline)) {
      constexpr int half = decltype(halfc)::value;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[4 * half + a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FA[a], FBv[b], acc[4 * half + a][b], 0, 0, 0);
    };

    for (uint32_t st = 0; st < nst; ++st) {
      const uint32_t slot_a = qa;
      qa = (qa + 1u == RA) ? 0u : qa + 1u;
      qb = (qb + 1u == RB) ? 0u : qb + 1u;
      // ---- MFMA phase, quadrant 0
      __builtin_amdgcn_s_setprio(1);
      multiply(IC<0>{});
      __builtin_amdgcn_s_setprio(0);
      phase_end();
      // ---- load phase A: the A fragments of quadrant 1 (read first: their LDS latency passes under the LDS-DMA issue and
      // the counted wait), weights of the stage three ahead, certify my share of the next stage
      read_a(slot_a, 1);
      issue_a();
      wait_ahead();
      phase_end();
      // ---- MFMA phase, quadrant 1
      __builtin_amdgcn_s_setprio(1);
      multiply(IC<1>{});
      __builtin_amdgcn_s_setprio(0);
      phase_end();
      // ---- load phase B: fragments of the next stage's quadrant 0 (certified by the barrier that ended this wave's
      // previous load phase at the latest), activations of the stage three ahead
      read_a(qa, 0);
      read_b(qb);
      issue_b();
      phase_end();
      since_store += 1;
    }

    // ---- epilogue: fp32 -> fp16 (RNE), + bias in fp16, ReLU (NifModel.cpp:314-325).  Piece (T, j, h) of the next layer =
    // accumulator tiles (2 s, b) | (2 s + 1, b); always stored (tiles past the end of the queue land in the buffer's
    // padding), so the store count the waits above assume is exact.
    f32x4 hacc[4];
    if constexpr (FUSE_HEAD) {
#pragma unroll
      for (int b = 0; b < 4; ++b) hacc[b] = (f32x4)(0.0f);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint32_t j = fb * 8u + 4u * wm + s;   // 32-feature group = k-step of the next layer
      const half8 bias = as_half8(reinterpret_cast<const uint4*>(bias_lds)[j * 4u + q4]);
      half8 hw;
      if constexpr (FUSE_HEAD) hw = as_half8(reinterpret_cast<const uint4*>(head_lds)[(4 * wm + s) * 64 + lane]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        half8 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = (_Float16)acc[2 * s][b][i]; o[4 + i] = (_Float16)acc[2 * s + 1][b][i]; }
        o = o + bias;
        {   // ReLU as one v_pk_max_f16 per register against a uniform floor: 0, or kLinearFloor for a linear layer (max(x, qNaN) = x, NaN included)
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          union { half8 hh; u32x4 u; } c;
          c.hh = o;
#pragma unroll
          for (int i = 0; i < 4; ++i) asm("v_pk_max_f16 %0, %1, %2" : "=v"(c.u[i]) : "v"(c.u[i]), "s"(relu_floor));
          o = c.hh;
        }
        if constexpr (FUSE_HEAD) {
          hacc[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hw, o, hacc[b], 0, 0, 0);
        } else {
          const uint32_t t16 = sb * 16u + 4u * wn + b;
          {   // sc1 = write-through: the line does not stay in L2 (the next layer reads it in another launch, long after it
              // would have been evicted anyway).  Bit 10: plain stores, bit 11: nt stores.
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            union { half8 hh; u32x4 u; } c1;
            c1.hh = o;
            const uint32_t off = ((((t16 >> 1) * P.act_stride + j) * 2u + (t16 & 1u)) * 64u + (uint32_t)lane) * 16u;
            __builtin_amdgcn_raw_buffer_store_b128(c1.u, out_rsrc, off, 0, (DIAG & 1024) ? 0 : (DIAG & 2048) ? 2 : 16);
          }
        }
      }
    }
    if constexpr (FUSE_HEAD) {
      // rows 0..2 of the head tile live in registers 0..2 of lanes 0-15 (q4 = 0); every wave issues the four stores
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t sample = (sb * 16u + 4u * wn + b) * 16u + (lane & 15);
        float4* dst = P.head_partial + (size_t)(fb * 2u + wm) * P.partial_stride + sample;
        if (q4 == 0) *dst = make_float4(hacc[b][0], hacc[b][1], hacc[b][2], 0.f);
      }
    }
    since_store = 0;
  }
  if (!second) phase_end();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the run-ahead loads before the wave ends
  if constexpr (DIAG & 32) {
    // in-kernel clock (MI355X_MICROARCH.md, "DVFS give-back" item 6): shader cycles / 100 MHz ticks over the whole loop
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.stamps) {
      P.stamps[0] = __builtin_amdgcn_s_memtime() - t_cycles;
      P.stamps[1] = __builtin_amdgcn_s_memrealtime() - t_real;
    }
  }
}

// tile_start[r] = first 32-sample tile of queue region r; tile_start[n_regions] = total.  One workgroup.
__global__ __launch_bounds__(256) void nifg_scan_kernel(const uint32_t* region_count, uint32_t n_regions, uint32_t* tile_start) {
  __shared__ uint32_t partial[256];
  const uint32_t per = (n_regions + 255u) / 256u;
  uint32_t sum = 0;
  for (uint32_t i = 0; i < per; ++i) {
    const uint32_t r = threadIdx.x * per + i;
    if (r < n_regions) sum += (region_count[r] + 31u) / 32u;
  }
  partial[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int i = 0; i < 256; ++i) { const uint32_t t = partial[i]; partial[i] = run; run += t; }
    tile_start[n_regions] = run;
  }
  __syncthreads();
  uint32_t run = partial[threadIdx.x];
  for (uint32_t i = 0; i < per; ++i) {
    const uint32_t r = threadIdx.x * per + i;
    if (r < n_regions) { tile_start[r] = run; run += (region_count[r] + 31u) / 32u; }
  }
}

// Region and offset of queue tile wt (binary search over the scan in LDS).
struct TileRef {
  uint32_t qbase, local, count;
};
__device__ __forceinline__ TileRef find_tile(const uint32_t* ts_lds, uint32_t n_regions, const NifParams& P, uint32_t wt) {
  uint32_t lo = 0, hi = n_regions;
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (ts_lds[mid] <= wt) lo = mid; else hi = mid;
  }
  TileRef r;
  r.local = (wt - ts_lds[lo]) * 32u;
  r.count = P.region_count[lo];
  r.qbase = lo * P.region_cap + r.local;
  return r;
}

// Fourier features of one coordinate (NifModel.cpp:185-218): frequencies 4 g .. 4 g + 3 as [sin x4 | cos x4], rounded to
// fp16 as the reference's half tensors are; slots at or beyond n_freq (E padded up) are zero.
__device__ __forceinline__ half8 fourier_group(float coord, int g, uint32_t n_freq) {
  const float x = (coord - 1.0f) * 2.0f;
  half8 f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a = (float)(_Float16)(x * (float)(1u << (4 * g + k)));
    float sn, cs;
    fast_sincos(a, sn, cs);
    if ((uint32_t)(4 * g + k) >= n_freq) { sn = 0.f; cs = 0.f; }
    f[k] = (_Float16)sn;
    f[4 + k] = (_Float16)cs;
  }
  return f;
}

// Fourier features of a chunk as B pieces: feat[T][s][h], s < IS32 = ceil(E / 8).  Lane (c = lane & 15, q = lane >> 4) of
// piece (T, s, h) holds sample 16 h + c, coordinate q & 1 (0 = u), frequency group 2 s + (q >> 1) -- the k order the
// input k-steps of pack_nif_g16 are packed for.  One wave per queue tile.
template <int E>
__global__ __launch_bounds__(256) void nifg16_encode_kernel(const NifParams P, const uint32_t* tile_start, uint32_t tile0,
                                                             uint32_t chunk_tiles, uint4* feat) {
  constexpr int IS32 = (E + 7) / 8;
  __shared__ uint32_t ts[kMaxRegions + 1];
  const uint32_t ntiles = chunk_tile_count(tile_start + P.n_regions, tile0, chunk_tiles);
  if (blockIdx.x * 4u >= ntiles) return;
  for (uint32_t i = threadIdx.x; i <= P.n_regions; i += 256u) ts[i] = tile_start[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, q4 = lane >> 4;
  for (uint32_t lt = blockIdx.x * 4u + wave; lt < ntiles; lt += gridDim.x * 4u) {
    const TileRef r = find_tile(ts, P.n_regions, P, tile0 + lt);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t off = 16u * h + c;
      float coord = 0.5f;
      if (r.local + off < r.count) coord = (q4 & 1) ? P.q_v[r.qbase + off] : P.q_u[r.qbase + off];
#pragma unroll
      for (int s = 0; s < IS32; ++s) {
        union { half8 hh; uint4 u; } f;
        f.hh = fourier_group(coord, 2 * s + (q4 >> 1), P.n_freq);
        feat[(((size_t)lt * IS32 + s) * 2 + h) * 64 + lane] = f.u;
      }
    }
  }
}

// Head, second half (the first is the MFMAs in the last hidden layer's epilogue): per sample, the partial sums of the
// 2 FB slices in slice order, + the head's own Fourier-feature inputs if it has any (x = concat(x, input),
// NifModel.cpp:305-308; fp32 FMAs over the fp16-rounded features), rounded to fp16, + bias in fp16, activation, decode
// (NifModel.cpp:221-245: exp(x max + mean)), BGR -> RGB x throughput and scatter (codelets.cpp:366-382).
struct NifHeadParams {
  const float4* partial;       // [slices][partial_stride]
  uint32_t slices, partial_stride;
  const float4* in_weights;    // [4 groups (sin u, sin v, cos u, cos v)][E] head weights of the feature inputs, or nullptr
  uint32_t n_in;               // E (padded) when in_weights is given
  float bias0, bias1, bias2;   // fp16 values
  uint32_t relu;
  uint32_t tile0, chunk_tiles;
};
__global__ __launch_bounds__(256) void nifg16_finish_kernel(const NifParams P, const NifHeadParams Hd, const uint32_t* tile_start) {
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
    for (uint32_t sl = 0; sl < Hd.slices; ++sl) {
      const float4 p = Hd.partial[(size_t)sl * Hd.partial_stride + sample];
      acc[0] += p.x; acc[1] += p.y; acc[2] += p.z;
    }
    if (Hd.in_weights) {
      const float cu = P.q_u[qi], cv = P.q_v[qi];
      for (uint32_t g = 0; g * 4u < Hd.n_in; ++g) {
        const half8 fu = fourier_group(cu, (int)g, P.n_freq), fv = fourier_group(cv, (int)g, P.n_freq);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t f = 4u * g + k;
          if (f >= Hd.n_in) break;
          const float4 wsu = Hd.in_weights[0 * Hd.n_in + f], wsv = Hd.in_weights[1 * Hd.n_in + f];
          const float4 wcu = Hd.in_weights[2 * Hd.n_in + f], wcv = Hd.in_weights[3 * Hd.n_in + f];
          const float su = (float)fu[k], sv = (float)fv[k], cu_ = (float)fu[4 + k], cv_ = (float)fv[4 + k];
          acc[0] += wsu.x * su + wsv.x * sv + wcu.x * cu_ + wcv.x * cv_;
          acc[1] += wsu.y * su + wsv.y * sv + wcu.y * cu_ + wcv.y * cv_;
          acc[2] += wsu.z * su + wsv.z * sv + wcu.z * cu_ + wcv.z * cv_;
        }
      }
    }
    const float bias[3] = {Hd.bias0, Hd.bias1, Hd.bias2};
    const float mean[3] = {P.mean0, P.mean1, P.mean2};
    float bgr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      _Float16 o16 = (_Float16)acc[k];
      o16 = o16 + (_Float16)bias[k];
      if (Hd.relu) o16 = o16 > (_Float16)0.0f ? o16 : (_Float16)0.0f;
      float o = (float)o16 * P.max;
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

}  // namespace ptd
