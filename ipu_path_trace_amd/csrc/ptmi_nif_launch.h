// ptmi_nif_launch.h -- launchers of the NIF kernels: fused (pt_nif.h), layer by layer (pt_nif_gemm.h), float32 (pt_nif_f32.h)
// Part of the one translation unit ptmi.hip (host side of include/ptmi.h); included there, in this order:
// ptmi_context.h, ptmi_nif_pack.h, ptmi_nif_launch.h, [the entry points in ptmi.hip], ptmi_film_comm.h.
#pragma once

namespace {

// Dynamic-LDS opt-in of a kernel, once per device (one bit per device: the host app drives devices from threads).
int set_dynamic_lds(pt_handle h, const void* fn, int bytes, std::atomic<unsigned long long>& done) {
  const unsigned long long bit = 1ull << (h->cfg.device & 63);
  if (done.load(std::memory_order_acquire) & bit) return PT_OK;
  PT_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.fetch_or(bit, std::memory_order_release);
  return PT_OK;
}

template <int HID, int E, int NB, int WAVES>
int launch_nif_v2(pt_handle h, const ptd::NifParams& N, int blocks) {
  using G = ptd::NifV2Geometry<HID, E, WAVES>;
  static std::atomic<unsigned long long> attr_set{0};
  if (int rc = set_dynamic_lds(h, reinterpret_cast<const void*>(&ptd::nif_kernel_v2<HID, E, NB, WAVES>), G::LDS_BYTES, attr_set)) return rc;
  static const std::string name = "nif_kernel_v2<" + std::to_string(HID) + ", " + std::to_string(E) + ", " + std::to_string(NB) + ", " + std::to_string(WAVES) + ">";
  h->nif_kernel = name;
  hipLaunchKernelGGL((ptd::nif_kernel_v2<HID, E, NB, WAVES>), dim3(blocks), dim3(64 * WAVES), G::LDS_BYTES, h->stream, N);
  PT_HIP(hipGetLastError());
  return PT_OK;
}

template <int HID, int E, int WAVES, int TPS, int DIAG = 0>
int launch_nif_v3(pt_handle h, const ptd::NifParams& N, int blocks) {
  using G = ptd::NifV3Geometry<HID, E, WAVES, TPS>;
  static std::atomic<unsigned long long> attr_set{0};
  if (int rc = set_dynamic_lds(h, reinterpret_cast<const void*>(&ptd::nif_kernel_v3<HID, E, WAVES, TPS, DIAG>), G::LDS_BYTES, attr_set)) return rc;
  static const std::string name = "nif_kernel_v3<" + std::to_string(HID) + ", " + std::to_string(E) + ", " + std::to_string(WAVES) + ", " + std::to_string(TPS) + ", " + std::to_string(DIAG) + ">";
  h->nif_kernel = name;
  hipLaunchKernelGGL((ptd::nif_kernel_v3<HID, E, WAVES, TPS, DIAG>), dim3(blocks), dim3(64 * WAVES), G::LDS_BYTES, h->stream, N);
  PT_HIP(hipGetLastError());
  return PT_OK;
}

#ifdef PTMI_DIAG_BUILD
// Timing-only ablations of the headline kernel (results are garbage): see nif_kernel_v3's DIAG bits.
template <int HID, int E>
bool launch_nif_diag(pt_handle h, const ptd::NifParams& N, int blocks) {
  const int diag = getenv("PTMI_NIF_DIAG") ? atoi(getenv("PTMI_NIF_DIAG")) : 0;   // read per launch: A/B rounds interleave in one process
  if constexpr (HID == 320 && E == 12) {
    switch (diag) {
      case 1: launch_nif_v3<HID, E, 8, 2, 1>(h, N, blocks); return true;
      case 2: launch_nif_v3<HID, E, 8, 2, 2>(h, N, blocks); return true;
      case 3: launch_nif_v3<HID, E, 8, 2, 3>(h, N, blocks); return true;
      case 4: launch_nif_v3<HID, E, 8, 2, 4>(h, N, blocks); return true;
      case 7: launch_nif_v3<HID, E, 8, 2, 7>(h, N, blocks); return true;
      case 15: launch_nif_v3<HID, E, 8, 2, 15>(h, N, blocks); return true;
      case 16: launch_nif_v3<HID, E, 8, 2, 16>(h, N, blocks); return true;
      case 32: launch_nif_v3<HID, E, 8, 2, 32>(h, N, blocks); return true;
      case 34: launch_nif_v3<HID, E, 8, 2, 34>(h, N, blocks); return true;   // no LDS reads of A + clock stamps
      case 64: launch_nif_v3<HID, E, 8, 2, 64>(h, N, blocks); return true;   // half the LDS reads of A (the NB = 2 bound)
      case 96: launch_nif_v3<HID, E, 8, 2, 96>(h, N, blocks); return true;   // ... + clock stamps
      case 128: launch_nif_v3<HID, E, 8, 2, 128>(h, N, blocks); return true;  // every wave at raised priority (valid results)
      default: break;
    }
  }
  return false;
}
#endif

// Register-resident kernels, one instantiation per (padded hidden width, padded embedding): v3 keeps the bias tiles of
// at most 8 layers resident in LDS; deeper networks take the v2 ring (layer 0 resident).  A ring stage of v3 carries
// two output tiles where the tile count is even, else one.
template <int HID, int E>
int launch_nif_t(pt_handle h, const ptd::NifParams& N, int blocks) {
  constexpr int TPS = ((HID / 32) % 2 == 0) ? 2 : 1;
#ifdef PTMI_DIAG_BUILD
  if constexpr (HID == 320 && E == 12) {
    if (launch_nif_diag<HID, E>(h, N, blocks)) return PT_OK;
    // A/B switch of the profiling build over the v2 ring kernel (product code, other template arguments): 2 = FOUR waves x
    // 64 samples (NB = 2: one wave per SIMD, every weight fragment feeds two 32-sample tiles), 3 = eight waves x 32 samples
    const int variant = getenv("PTMI_NIF_VARIANT") ? atoi(getenv("PTMI_NIF_VARIANT")) : 0;
    if (variant == 2) return launch_nif_v2<HID, E, 2, 4>(h, N, blocks);
    if (variant == 3) return launch_nif_v2<HID, E, 1, 8>(h, N, blocks);
  }
#endif
  if (N.n_layers > (uint32_t)ptd::NifV3Geometry<HID, E, 8, TPS>::MAX_LAYERS) return launch_nif_v2<HID, E, 1, 8>(h, N, blocks);
  return launch_nif_v3<HID, E, 8, TPS>(h, N, blocks);
}

#ifdef PTMI_HEADLINE_ONLY
// Development build (seconds to compile): only the headline shape is instantiated.
template <int E>
int launch_nif_e(pt_handle h, const ptd::NifParams& N, int blocks) {
  if constexpr (E == 12) { if (h->nif_hidden == 320) return launch_nif_t<320, 12>(h, N, blocks); }
  return fail(h, PT_ERR_UNSUPPORTED_MODEL, "PTMI_HEADLINE_ONLY build: only hidden 320 / embedding 12 is instantiated");
}
#else
template <int E>
int launch_nif_e(pt_handle h, const ptd::NifParams& N, int blocks) {
  switch (h->nif_hidden) {
    case 32: return launch_nif_t<32, E>(h, N, blocks);
    case 64: return launch_nif_t<64, E>(h, N, blocks);
    case 96: return launch_nif_t<96, E>(h, N, blocks);
    case 128: return launch_nif_t<128, E>(h, N, blocks);
    case 160: return launch_nif_t<160, E>(h, N, blocks);
    case 192: return launch_nif_t<192, E>(h, N, blocks);
    case 224: return launch_nif_t<224, E>(h, N, blocks);
    case 256: return launch_nif_t<256, E>(h, N, blocks);
    case 288: return launch_nif_t<288, E>(h, N, blocks);
    case 320: return launch_nif_t<320, E>(h, N, blocks);
    default: break;
  }
  return fail(h, PT_ERR_UNSUPPORTED_MODEL, "no register-resident NIF kernel for hidden width " + std::to_string(h->nif_hidden));
}
#endif

// Fork the chunk streams off the NIF stream / join them back (pt_context::chunk_stream).
static int chunk_streams_fork(pt_handle h, int sets) {
  if (sets < 2) return PT_OK;
  PT_HIP(hipEventRecord(h->chunk_fork, h->stream));
  for (int i = 0; i + 1 < sets; ++i) PT_HIP(hipStreamWaitEvent(h->chunk_stream[i], h->chunk_fork, 0));
  return PT_OK;
}
static int chunk_streams_join(pt_handle h, int sets, int rc) {   // also after a failed launch: whatever was queued ends before the NIF stream goes on
  for (int i = 0; i + 1 < sets; ++i) {
    const hipError_t e1 = hipEventRecord(h->chunk_join[i], h->chunk_stream[i]), e2 = hipStreamWaitEvent(h->stream, h->chunk_join[i], 0);
    if (rc == PT_OK && (e1 != hipSuccess || e2 != hipSuccess)) rc = fail(h, PT_ERR_HIP, "joining a chunk stream failed");
  }
  return rc;
}
static int chunk_sets_for(pt_handle h, uint64_t max_tiles, uint32_t chunk) {
  int sets = (int)std::min<uint64_t>((max_tiles + chunk - 1) / chunk, (uint64_t)h->chunk_sets);
#ifdef PTMI_DIAG_BUILD
  if (const char* e = getenv("PTMI_CHUNK_STREAMS")) sets = std::max(1, std::min(sets, atoi(e)));   // A/B of the profiling build
#endif
  return std::max(sets, 1);
}

// Wide networks, one launch per layer over chunks of the queue (pt_nif_gemm.h).  The number of queue tiles is only
// known on the device, so chunks are launched up to the queue's capacity and those past its end return at once.
template <int E>
void launch_nifg16_encode(pt_handle h, hipStream_t st, const ptd::NifParams& N, uint32_t tile0, uint32_t chunk, uint4* feat) {
  hipLaunchKernelGGL((ptd::nifg16_encode_kernel<E>), dim3((chunk + 3u) / 4u), dim3(256), 0, st, N, h->d_tile_start, tile0, chunk, feat);
}

template <int FUSE>
int launch_nifg16_layer(pt_handle h, hipStream_t st, const ptd::NifGemmParams& G, uint32_t grid) {
  static std::atomic<unsigned long long> attr_set{0};
  if (int rc = set_dynamic_lds(h, reinterpret_cast<const void*>(&ptd::nifg16_layer_kernel<FUSE, 0>), ptd::kGemmLdsBytes, attr_set)) return rc;
#ifdef PTMI_DIAG_BUILD
  // timing-only ablations / clock stamps of the profiling build, read per launch (PTMI_GEMM_DIAG: see the kernel's DIAG bits)
  const int gdiag = getenv("PTMI_GEMM_DIAG") ? atoi(getenv("PTMI_GEMM_DIAG")) : 0;
#define PT_LAYER16(D)                                                                                                          \
  case D: {                                                                                                                    \
    PT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ptd::nifg16_layer_kernel<FUSE, D>),                               \
                               hipFuncAttributeMaxDynamicSharedMemorySize, ptd::kGemmLdsBytes));                               \
    ptd::NifGemmParams GS = G;                                                                                                 \
    GS.stamps = h->d_stamps;                                                                                                   \
    hipLaunchKernelGGL((ptd::nifg16_layer_kernel<FUSE, D>), dim3(grid), dim3(512), ptd::kGemmLdsBytes, st, GS);                 \
    PT_HIP(hipGetLastError());                                                                                                 \
    return PT_OK;                                                                                                              \
  }
  switch (gdiag) { PT_LAYER16(1) PT_LAYER16(2) PT_LAYER16(3) PT_LAYER16(8) PT_LAYER16(32) PT_LAYER16(64) PT_LAYER16(128) PT_LAYER16(256)
                   PT_LAYER16(16) PT_LAYER16(33) PT_LAYER16(40) PT_LAYER16(48) PT_LAYER16(160) PT_LAYER16(512) PT_LAYER16(544) PT_LAYER16(1024) PT_LAYER16(2048) PT_LAYER16(1056) PT_LAYER16(4096) PT_LAYER16(5120) PT_LAYER16(8192) PT_LAYER16(9216) PT_LAYER16(3072) default: break; }
#undef PT_LAYER16
#endif
  hipLaunchKernelGGL((ptd::nifg16_layer_kernel<FUSE, 0>), dim3(grid), dim3(512), ptd::kGemmLdsBytes, st, G);
  PT_HIP(hipGetLastError());
  return PT_OK;
}

int launch_nif_gemm(pt_handle h, const ptd::NifParams& N) {
  const uint32_t H = (uint32_t)h->nif_hidden, KS = H / 32, IS = ((uint32_t)h->nif_emb / 4 + 1) / 2, NT = H / 32, FB = NT / 8;
  const uint32_t n_layers = N.n_layers, chunk = h->gemm_chunk;
  if (!chunk) return fail(h, PT_ERR_NOT_READY, "wide-NIF buffers are not allocated");
  if (FB == 0 || NT % 8u) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "layer-by-layer NIF path needs a hidden width that is a multiple of 256");
  hipLaunchKernelGGL(ptd::nifg_scan_kernel, dim3(1), dim3(256), 0, h->stream, N.region_count, N.n_regions, h->d_tile_start);
  PT_HIP(hipGetLastError());
  const uint64_t max_tiles = (uint64_t)N.n_regions * ((N.region_cap + 31u) / 32u);
  uint32_t grid = ((uint32_t)h->n_cus / (8u * FB)) * 8u * FB;
  if (grid == 0) grid = 8u * FB;
  ptd::NifGemmParams G{};
  G.wpack = N.wpack;
  G.bpack = N.bpack;
  G.act_stride = KS;
  G.feat_stride = IS;
  G.total_tiles = h->d_tile_start + N.n_regions;
  G.chunk_tiles = chunk;
  G.n_ftiles = NT;
  G.head_piece_base = h->head_piece_base;
  G.partial_stride = chunk * 32u;
  h->nif_kernel = "nifg16_layer_kernel<0, 0> x " + std::to_string(n_layers - 2) + " + nifg16_layer_kernel<1, 0> (last hidden layer, head fused) + nifg16_encode_kernel<" +
                  std::to_string(h->nif_emb) + "> + nifg16_finish_kernel, per chunk of " + std::to_string(chunk) + " queue tiles, hidden " + std::to_string(H);
  const uint32_t lh = n_layers - 1;
  ptd::NifHeadParams Hd{};
  Hd.slices = 2u * FB;
  Hd.partial_stride = chunk * 32u;
  Hd.in_weights = ((N.concat_mask >> lh) & 1u) ? h->d_head_in : nullptr;
  Hd.n_in = (uint32_t)h->nif_emb;
  Hd.bias0 = h->head_bias[0]; Hd.bias1 = h->head_bias[1]; Hd.bias2 = h->head_bias[2];
  Hd.relu = (N.relu_mask >> lh) & 1u;
  Hd.chunk_tiles = chunk;
  // chunks alternate between the NIF stream and chunk_stream (each with its own buffer set): see pt_context::chunk_stream
  const int sets = chunk_sets_for(h, max_tiles, chunk);
  if (int frc = chunk_streams_fork(h, sets)) return frc;
  int rc = PT_OK;
  uint32_t set = 0;
  for (uint64_t tile0 = 0; tile0 < max_tiles && rc == PT_OK; tile0 += chunk, set = (set + 1u) % (uint32_t)sets) {
    hipStream_t st = set ? h->chunk_stream[set - 1u] : h->stream;
    uint4* const act[2] = {h->d_gemm_act[0] + set * h->gemm_act_set, h->d_gemm_act[1] + set * h->gemm_act_set};
    uint4* const feat = h->d_gemm_feat + set * h->gemm_feat_set;
    float4* const partial = h->d_head_partial + set * h->head_partial_set;
    G.tile0 = (uint32_t)tile0;
    G.feat = feat;
    switch (h->nif_emb) {
      case 4: launch_nifg16_encode<4>(h, st, N, G.tile0, chunk, feat); break;
      case 8: launch_nifg16_encode<8>(h, st, N, G.tile0, chunk, feat); break;
      case 12: launch_nifg16_encode<12>(h, st, N, G.tile0, chunk, feat); break;
      case 16: launch_nifg16_encode<16>(h, st, N, G.tile0, chunk, feat); break;
      default: rc = fail(h, PT_ERR_UNSUPPORTED_MODEL, "unsupported embedding dimension"); continue;
    }
    if (hipGetLastError() != hipSuccess) { rc = fail(h, PT_ERR_HIP, "wide-NIF encode launch failed"); continue; }
    for (uint32_t l = 0; l + 1 < n_layers && rc == PT_OK; ++l) {
      const bool concat = (N.concat_mask >> l) & 1u;
      const bool last = l + 2 == n_layers;   // the head rides in this layer's epilogue
      G.piece_base = N.piece_base[l];
      G.bias_base = N.bias_base[l];
      G.ks_act = l ? KS : 0u;
      G.ks_in = (l == 0 || concat) ? IS : 0u;
      G.relu = (N.relu_mask >> l) & 1u;
      G.act_in = act[(l + 1u) & 1u];
      G.act_out = last ? nullptr : act[l & 1u];
      G.head_partial = last ? partial : nullptr;
      rc = last ? launch_nifg16_layer<1>(h, st, G, grid) : launch_nifg16_layer<0>(h, st, G, grid);
    }
    if (rc) continue;
    Hd.tile0 = G.tile0;
    Hd.partial = partial;
    hipLaunchKernelGGL(ptd::nifg16_finish_kernel, dim3((chunk + 7u) / 8u), dim3(256), 0, st, N, Hd, h->d_tile_start);
    if (hipGetLastError() != hipSuccess) rc = fail(h, PT_ERR_HIP, "wide-NIF finish launch failed");
  }
  return chunk_streams_join(h, sets, rc);
}

// ---- float32 models (pt_nif_f32.h) -----------------------------------------------------------------------------
// Shapes as normalize_nif: hidden widths padded to a common multiple of 32 (zero weights, zero bias), the Fourier features
// to 4 x Ep with Ep = E rounded up to a multiple of 4 (zero rows for the padding slots, zero features).
struct HostLayerF32 { uint32_t rows, cols; std::vector<float> kernel, bias; bool has_bias, relu, f16 = false; };   // f16: binary16 variables, widened (a mixed model)

int pack_nif_f32(pt_handle h, const std::vector<HostLayerF32>& L, uint32_t E, std::vector<float>& blob,
                 std::vector<pt_context::F32Layer>& out, uint32_t& Hp_out, uint32_t& Ep_out) {
  const uint32_t n = (uint32_t)L.size();
  if (n < 2 || n > ptd::kMaxLayers) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF must have 2..16 dense layers");
  if (E == 0 || E > 16) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "embedding dimension must be in 1..16");
  const uint32_t in_dim = 4 * E, Ep = (E + 3u) / 4u * 4u, in_p = 4 * Ep;
  if (L[0].rows != in_dim) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "first layer must take the 4*embedding Fourier features");
  if (L[n - 1].cols != 3) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF head must have 3 outputs (BGR)");
  uint32_t widest = 0;
  for (uint32_t l = 0; l + 1 < n; ++l) widest = std::max(widest, L[l].cols);
  const uint32_t Hp = (widest + 31u) / 32u * 32u;
  if (Hp > kMaxGemmHidden) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "hidden layers wider than 2048 are not supported");
  out.clear();
  blob.clear();
  uint32_t prev = 0;
  for (uint32_t l = 0; l < n; ++l) {
    const HostLayerF32& Y = L[l];
    const bool head = l + 1 == n;
    bool concat = false;
    if (l == 0) {
    } else if (Y.rows == prev) {
    } else if (Y.rows == prev + in_dim) {   // NifModel.cpp:305-308
      concat = true;
    } else {
      return fail(h, PT_ERR_UNSUPPORTED_MODEL, "layer " + std::to_string(l) + ": input width " + std::to_string(Y.rows) +
                                                   " is neither the previous layer's width nor that plus the 4*embedding features");
    }
    pt_context::F32Layer F{};
    F.k_act = l ? Hp : 0u;
    F.k_in = (l == 0 || concat) ? in_p : 0u;
    F.ldw = head ? 4u : Hp;
    F.relu = Y.relu;
    F.half_out = Y.f16 ? 1u : 0u;
    F.cast_half = (!Y.f16 && !head && L[l + 1].f16) ? 1u : 0u;   // a binary16 layer reads its input as half (its matmul's type)
    F.w_off = blob.size();
    blob.resize(blob.size() + (size_t)(F.k_act + F.k_in) * F.ldw, 0.f);
    float* W = &blob[F.w_off];
    const uint32_t act_t = l ? prev : 0u;
    for (uint32_t r = 0; r < act_t; ++r)
      for (uint32_t c = 0; c < Y.cols; ++c) W[(size_t)r * F.ldw + c] = Y.kernel[(size_t)r * Y.cols + c];
    if (F.k_in)
      for (uint32_t f = 0; f < in_dim; ++f)   // feature order [sin u, sin v, cos u, cos v] x E (NifModel.cpp:216)
        for (uint32_t c = 0; c < Y.cols; ++c)
          W[(size_t)(F.k_act + (f / E) * Ep + (f % E)) * F.ldw + c] = Y.kernel[(size_t)(act_t + f) * Y.cols + c];
    F.b_off = blob.size();
    blob.resize(blob.size() + F.ldw, 0.f);
    if (Y.has_bias) for (uint32_t c = 0; c < Y.cols; ++c) blob[F.b_off + c] = Y.bias[c];
    blob.resize((blob.size() + 3) / 4 * 4, 0.f);   // keep every kernel 16-byte aligned (the head reads float4 rows)
    out.push_back(F);
    prev = Y.cols;
  }
  Hp_out = Hp;
  Ep_out = Ep;
  return PT_OK;
}

template <int E>
void launch_nif32_encode(pt_handle h, hipStream_t st, const ptd::NifParams& N, uint32_t tile0, uint32_t chunk, float* feat) {
  hipLaunchKernelGGL((ptd::nif32_encode_kernel<E>), dim3((chunk + 3u) / 4u), dim3(256), 0, st, N, h->d_tile_start, tile0, chunk, feat);
}

int launch_nif_f32(pt_handle h, const ptd::NifParams& N) {
  const uint32_t chunk = h->f32_chunk, n_layers = (uint32_t)h->f32_layers.size();
  if (!chunk) return fail(h, PT_ERR_NOT_READY, "float32 NIF buffers are not allocated");
  h->nif_kernel = "nif32_layer_kernel x " + std::to_string(n_layers - 1) + " + nif32_encode_kernel<" + std::to_string(h->nif_emb) + "> + nif32_head_kernel, per chunk of " +
                  std::to_string(chunk) + " queue tiles, hidden " + std::to_string(h->nif_hidden);
  hipLaunchKernelGGL(ptd::nifg_scan_kernel, dim3(1), dim3(256), 0, h->stream, N.region_count, N.n_regions, h->d_tile_start);
  PT_HIP(hipGetLastError());
  const uint64_t max_tiles = (uint64_t)N.n_regions * ((N.region_cap + 31u) / 32u);
  // chunks alternate between the NIF stream and chunk_stream (each with its own buffer set): see pt_context::chunk_stream
  const int sets = chunk_sets_for(h, max_tiles, chunk);
  if (int frc = chunk_streams_fork(h, sets)) return frc;
  int rc = PT_OK;
  uint32_t set = 0;
  for (uint64_t tile0 = 0; tile0 < max_tiles && rc == PT_OK; tile0 += chunk, set = (set + 1u) % (uint32_t)sets) {
    hipStream_t st = set ? h->chunk_stream[set - 1u] : h->stream;
    float* const act[2] = {h->d_f32_act[0] + set * h->f32_act_set, h->d_f32_act[1] + set * h->f32_act_set};
    float* const feat = h->d_f32_feat + set * h->f32_feat_set;
    switch (h->nif_emb) {
      case 4: launch_nif32_encode<4>(h, st, N, (uint32_t)tile0, chunk, feat); break;
      case 8: launch_nif32_encode<8>(h, st, N, (uint32_t)tile0, chunk, feat); break;
      case 12: launch_nif32_encode<12>(h, st, N, (uint32_t)tile0, chunk, feat); break;
      case 16: launch_nif32_encode<16>(h, st, N, (uint32_t)tile0, chunk, feat); break;
      default: rc = fail(h, PT_ERR_UNSUPPORTED_MODEL, "unsupported embedding dimension"); continue;
    }
    if (hipGetLastError() != hipSuccess) { rc = fail(h, PT_ERR_HIP, "float32 NIF encode launch failed"); continue; }
    for (uint32_t l = 0; l < n_layers && rc == PT_OK; ++l) {
      const pt_context::F32Layer& F = h->f32_layers[l];
      const float* in = act[(l + 1u) & 1u];
      if (l + 1 < n_layers) {
        ptd::NifF32Params G{};
        G.w = h->d_f32_weights + F.w_off; G.bias = h->d_f32_weights + F.b_off;
        G.ldw = F.ldw; G.k_act = F.k_act; G.k_in = F.k_in; G.relu = F.relu; G.half_out = F.half_out; G.cast_half = F.cast_half;
        G.act_in = in; G.feat = feat; G.act_out = act[l & 1u];
        G.lda = h->f32_lda; G.ldf = h->f32_ldf;
        G.total_tiles = h->d_tile_start + N.n_regions; G.tile0 = (uint32_t)tile0; G.chunk_tiles = chunk;
        const uint32_t blocks = chunk / 8u * ((F.ldw + 63u) / 64u);  // one 256-sample x 64-feature block per workgroup
        const dim3 grid((blocks + 7u) / 8u * 8u);
        if (F.half_out) hipLaunchKernelGGL(ptd::nif32_layer_kernel<1>, grid, dim3(256), 0, st, G);
        else if (F.cast_half) hipLaunchKernelGGL(ptd::nif32_layer_kernel<2>, grid, dim3(256), 0, st, G);
        else hipLaunchKernelGGL(ptd::nif32_layer_kernel<0>, grid, dim3(256), 0, st, G);
      } else {
        ptd::NifF32Head Hd{};
        Hd.w = h->d_f32_weights + F.w_off;
        Hd.k_act = F.k_act; Hd.k_in = F.k_in; Hd.relu = F.relu; Hd.half_out = F.half_out;
        Hd.bias0 = h->head_bias[0]; Hd.bias1 = h->head_bias[1]; Hd.bias2 = h->head_bias[2];
        Hd.act_in = in; Hd.feat = feat; Hd.lda = h->f32_lda; Hd.ldf = h->f32_ldf;
        Hd.tile0 = (uint32_t)tile0; Hd.chunk_tiles = chunk;
        hipLaunchKernelGGL(ptd::nif32_head_kernel, dim3((chunk + 7u) / 8u), dim3(256), 0, st, N, Hd, h->d_tile_start);
      }
      if (hipGetLastError() != hipSuccess) rc = fail(h, PT_ERR_HIP, "float32 NIF layer launch failed");
    }
  }
  return chunk_streams_join(h, sets, rc);
}

int launch_nif(pt_handle h, const ptd::NifParams& N, int blocks) {
  h->nif_kernel = "(profiling-build variant)";   // every product launcher below overwrites it with the kernel it dispatches
  if (h->nif_f32) return launch_nif_f32(h, N);
  if (h->nif_gemm) {
    return launch_nif_gemm(h, N);
  }
  switch (h->nif_emb) {
    case 4: return launch_nif_e<4>(h, N, blocks);
    case 8: return launch_nif_e<8>(h, N, blocks);
    case 12: return launch_nif_e<12>(h, N, blocks);
    case 16: return launch_nif_e<16>(h, N, blocks);
    default: break;
  }
  return fail(h, PT_ERR_UNSUPPORTED_MODEL, "unsupported embedding dimension");
}

void free_batch_buffers(pt_handle h) {
  for (auto& B : h->bb) {
    (void)hipFree(B.q_u); (void)hipFree(B.q_v); (void)hipFree(B.q_tr); (void)hipFree(B.q_tg); (void)hipFree(B.q_tb);
    (void)hipFree(B.q_path); (void)hipFree(B.survivors); (void)hipFree(B.states); (void)hipFree(B.region_count); (void)hipFree(B.plen);
    (void)hipFree(B.rad_r); (void)hipFree(B.rad_g); (void)hipFree(B.rad_b);
    if (B.traced) (void)hipEventDestroy(B.traced);
    if (B.accumulated) (void)hipEventDestroy(B.accumulated);
    B = pt_context::BatchBuffers();
  }
}

hipEvent_t get_event(pt_handle h, size_t i) {
  while (h->events.size() <= i) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    h->events.push_back(e);
  }
  return h->events[i];
}

}  // namespace
