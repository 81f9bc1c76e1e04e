// ptmi_nif_pack.h -- NIF shape normalisation and weight packing for the MFMA kernels (host)
// Part of the one translation unit ptmi.hip (host side of include/ptmi.h); included there, in this order:
// ptmi_context.h, ptmi_nif_pack.h, ptmi_nif_launch.h, [the entry points in ptmi.hip], ptmi_film_comm.h.
#pragma once

namespace {

// ---- NIF shape normalisation ---------------------------------------------------------------
// The reference builds whatever Dense stack the H5 describes (NifModel.cpp:295-326).  The MFMA kernels want a uniform
// hidden width (a multiple of 32 for the register-resident kernels, of 256 for the layer-by-layer path) and 4 | E, so
// the stack is zero-padded to that: a padded output feature has zero weights and zero bias (its activation is 0 with
// or without ReLU), a padded input row multiplies it by zero, and a padded frequency slot (E not a multiple of 4) has
// zero weights and a zero feature (NifParams::n_freq).  Arithmetic on the true entries is unchanged.
struct NifPlan {
  uint32_t E = 0, Ep = 0;   // frequencies per coordinate: true / padded to a multiple of 4
  uint32_t Hp = 0;          // padded uniform hidden width
  bool gemm = false;        // layer-by-layer path (pt_nif_gemm.h) instead of the register-resident kernels
};

constexpr uint32_t kMaxFusedHidden = 320;    // nif_kernel_v3/v2: two activation vectors of H halves per sample in VGPRs
constexpr uint32_t kMaxGemmHidden = 2048;    // nifg_layer_kernel: bias tiles of one layer in 4 KiB of LDS

int normalize_nif(pt_handle h, const std::vector<HostLayer>& L, uint32_t E, std::vector<HostLayer>& out, NifPlan& plan) {
  const uint32_t n = (uint32_t)L.size();
  if (n < 2 || n > ptd::kMaxLayers) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF must have 2..16 dense layers");
  if (E == 0 || E > 16) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "embedding dimension must be in 1..16");
  const uint32_t in_dim = 4 * E, Ep = (E + 3u) / 4u * 4u, in_p = 4 * Ep;
  if (L[0].rows != in_dim) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "first layer must take the 4*embedding Fourier features");
  if (L[n - 1].cols != 3) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF head must have 3 outputs (BGR)");
  uint32_t widest = 0;
  for (uint32_t l = 0; l + 1 < n; ++l) widest = std::max(widest, L[l].cols);
  plan.E = E;
  plan.Ep = Ep;
  plan.gemm = widest > kMaxFusedHidden;
  plan.Hp = plan.gemm ? (widest + 255u) / 256u * 256u : (widest + 31u) / 32u * 32u;
  if (plan.Hp > kMaxGemmHidden) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "hidden layers wider than 2048 are not supported");
  const uint32_t Hp = plan.Hp;
  out.assign(n, HostLayer());
  uint32_t prev = 0;   // true width of the previous layer's output
  for (uint32_t l = 0; l < n; ++l) {
    const HostLayer& Y = L[l];
    bool concat = false;
    if (l == 0) {
      // (rows == in_dim checked above)
    } else if (Y.rows == prev) {
    } else if (Y.rows == prev + in_dim) {   // NifModel.cpp:305-308: x = concat(x, input) when the widths differ
      concat = true;
    } else {
      return fail(h, PT_ERR_UNSUPPORTED_MODEL, "layer " + std::to_string(l) + ": input width " + std::to_string(Y.rows) +
                                                   " is neither the previous layer's width nor that plus the 4*embedding features");
    }
    HostLayer& Z = out[l];
    const bool feats = (l == 0) || concat;
    const uint32_t act_p = l ? Hp : 0u, act_t = l ? prev : 0u;
    Z.rows = act_p + (feats ? in_p : 0u);
    Z.cols = (l + 1 == n) ? 3u : Hp;
    Z.relu = Y.relu;
    Z.kernel.assign((size_t)Z.rows * Z.cols, 0);
    for (uint32_t r = 0; r < act_t; ++r)
      memcpy(&Z.kernel[(size_t)r * Z.cols], &Y.kernel[(size_t)r * Y.cols], (size_t)Y.cols * 2);
    if (feats)
      for (uint32_t f = 0; f < in_dim; ++f)   // feature order [sin u, sin v, cos u, cos v] x E (NifModel.cpp:216)
        memcpy(&Z.kernel[(size_t)(act_p + (f / E) * Ep + (f % E)) * Z.cols], &Y.kernel[(size_t)(act_t + f) * Y.cols], (size_t)Y.cols * 2);
    if (!Y.bias.empty()) {
      Z.bias.assign(Z.cols, 0);
      memcpy(Z.bias.data(), Y.bias.data(), (size_t)Y.cols * 2);
    }
    prev = Y.cols;
  }
  return PT_OK;
}

// ---- NIF weight packing -------------------------------------------------------------------
// Piece (l, j, s): the A operand of one v_mfma_f32_32x32x16_f16: lane (r = lane & 31, hh = lane >> 5)
// holds W^T[32 j + r][k(hh, 0..7)], where k() is the k-step's map onto rows of the Keras kernel:
//  * activation k-step s (from a previous accumulator tile t = s / 2, half s % 2):
//      k = 32 t + 16 (s % 2) + 8 (e >> 2) + 4 hh + (e & 3)     (accumulator-as-operand order)
//  * input k-step s' (Fourier features, NifModel.cpp:216 order [sin u, sin v, cos u, cos v]):
//      k = base + (e < 4 ? 0 : 2E) + hh E + 4 s' + (e & 3), base = H for a concat layer, else 0
int pack_nif(pt_handle h, const std::vector<HostLayer>& L, uint32_t E, std::vector<uint16_t>& wpack,
             std::vector<uint16_t>& bpack, ptd::NifParams& N) {
  const uint32_t n = (uint32_t)L.size();
  if (n < 2 || n > ptd::kMaxLayers) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF must have 2..16 dense layers");
  if (E == 0 || E % 4) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "embedding dimension must be a multiple of 4");
  const uint32_t in_dim = 4 * E, H = L[0].cols;
  if (L[0].rows != in_dim) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "first layer must take the 4*embedding Fourier features");
  if (H % 32) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "hidden size must be a multiple of 32");
  if (L[n - 1].cols != 3) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF head must have 3 outputs (BGR)");
  memset(&N, 0, sizeof(N));
  N.n_layers = n;
  uint32_t piece = 0, btile = 0;
  for (uint32_t l = 0; l < n; ++l) {
    const HostLayer& Y = L[l];
    const bool head = (l == n - 1);
    if (!head && Y.cols != H) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "all hidden layers must have the same width");
    bool concat = false;
    uint32_t act_steps = 0;
    if (l == 0) {
      if (Y.rows != in_dim) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "bad first layer shape");
    } else if (Y.rows == H) {
      act_steps = H / 16;
    } else if (Y.rows == H + in_dim) {  // NifModel.cpp:305-308: x = concat(x, input)
      act_steps = H / 16;
      concat = true;
    } else {
      return fail(h, PT_ERR_UNSUPPORTED_MODEL, "layer input width is neither hidden nor hidden+features");
    }
    const uint32_t in_steps = (l == 0 || concat) ? E / 4 : 0;
    const uint32_t ksteps = act_steps + in_steps;
    const uint32_t ntiles = (Y.cols + 31) / 32;
    N.piece_base[l] = piece;
    N.bias_base[l] = btile;
    if (concat) N.concat_mask |= 1u << l;
    if (Y.relu) N.relu_mask |= 1u << l;
    if (!Y.bias.empty()) N.bias_mask |= 1u << l;
    wpack.resize((size_t)(piece + ntiles * ksteps) * 512, 0);
    bpack.resize((size_t)(btile + ntiles) * 32, 0);
    for (uint32_t j = 0; j < ntiles; ++j) {
      for (uint32_t s = 0; s < ksteps; ++s) {
        uint16_t* dst = &wpack[(size_t)(piece + j * ksteps + s) * 512];
        for (uint32_t lane = 0; lane < 64; ++lane) {
          const uint32_t r = lane & 31, hh = lane >> 5, col = 32 * j + r;
          for (uint32_t e = 0; e < 8; ++e) {
            uint32_t k;
            if (s < act_steps) {
              k = 32 * (s / 2) + 16 * (s % 2) + 8 * (e >> 2) + 4 * hh + (e & 3);
            } else {
              const uint32_t sp = s - act_steps;
              k = (concat ? H : 0) + (e < 4 ? 0 : 2 * E) + hh * E + 4 * sp + (e & 3);
            }
            dst[lane * 8 + e] = (col < Y.cols) ? Y.kernel[(size_t)k * Y.cols + col] : (uint16_t)0;
          }
        }
      }
      // bias of n-tile j in accumulator order: lane half hh, register i -> row (i&3) + 8 (i>>2) + 4 hh
      for (uint32_t hh = 0; hh < 2; ++hh)
        for (uint32_t i = 0; i < 16; ++i) {
          const uint32_t col = 32 * j + (i & 3) + 8 * (i >> 2) + 4 * hh;
          bpack[(size_t)(btile + j) * 32 + hh * 16 + i] = (!Y.bias.empty() && col < Y.cols) ? Y.bias[col] : (uint16_t)0;
        }
    }
    piece += ntiles * ksteps;
    btile += ntiles;
  }
  wpack.resize(wpack.size() + 512, 0);   // the paired loaders may copy (never use) one piece past the last
  return PT_OK;
}

// Wide networks (pt_nif_gemm.h), v_mfma_f32_16x16x32_f16.  Piece (l, s, f): the A operand of k-step s (32 inputs) and
// feature tile f (16 outputs) of layer l; lane (r = lane & 15, q = lane >> 4) holds W^T[16 f + r][k(q, 0..7)] with
//  * activation k-step s:  k = 32 s + (e < 4 ? 4 q + e : 16 + 4 q + (e - 4))          (accumulator-as-operand order)
//  * input k-step s':      coordinate cd = q & 1, frequency f' = 4 (2 s' + (q >> 1)) + (e & 3);
//                          k = base + (e < 4 ? 0 : 2E) + cd E + f', or a zero weight where f' >= E (padding slots)
// Pieces of a layer are ordered [s][f] (the two feature tiles a wave loads per stage are adjacent).  The head is one
// 16-row tile (rows 0..2), activation k-steps only: its feature inputs, if any, go to `head_in` as plain floats
// [sin u, sin v, cos u, cos v][E] x (B, G, R, -) for nifg16_finish_kernel.  Bias of a 32-feature group: [q][8]:
// e < 4 -> feature 32 j + 4 q + e, e >= 4 -> 32 j + 16 + 4 q + (e - 4).
int pack_nif_g16(pt_handle h, const std::vector<HostLayer>& L, uint32_t E, std::vector<uint16_t>& wpack,
                 std::vector<uint16_t>& bpack, ptd::NifParams& N, std::vector<float>& head_in, float head_bias[3],
                 uint32_t& head_piece_base) {
  const uint32_t n = (uint32_t)L.size();
  if (n < 2 || n > ptd::kMaxLayers) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF must have 2..16 dense layers");
  if (E == 0 || E % 4) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "embedding dimension must be a multiple of 4");
  const uint32_t in_dim = 4 * E, H = L[0].cols;
  if (L[0].rows != in_dim) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "first layer must take the 4*embedding Fourier features");
  if (H % 256) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "layer-by-layer NIF path needs a hidden width that is a multiple of 256");
  if (L[n - 1].cols != 3) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "NIF head must have 3 outputs (BGR)");
  memset(&N, 0, sizeof(N));
  N.n_layers = n;
  const uint32_t in_steps_all = (E / 4 + 1) / 2;
  uint32_t piece = 0, btile = 0;
  head_in.clear();
  for (uint32_t l = 0; l < n; ++l) {
    const HostLayer& Y = L[l];
    const bool head = (l == n - 1);
    if (!head && Y.cols != H) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "all hidden layers must have the same width");
    bool concat = false;
    uint32_t act_steps = 0;
    if (l == 0) {
      if (Y.rows != in_dim) return fail(h, PT_ERR_UNSUPPORTED_MODEL, "bad first layer shape");
    } else if (Y.rows == H) {
      act_steps = H / 32;
    } else if (Y.rows == H + in_dim) {  // NifModel.cpp:305-308: x = concat(x, input)
      act_steps = H / 32;
      concat = true;
    } else {
      return fail(h, PT_ERR_UNSUPPORTED_MODEL, "layer input width is neither hidden nor hidden+features");
    }
    const uint32_t in_steps = (!head && (l == 0 || concat)) ? in_steps_all : 0;
    const uint32_t ksteps = act_steps + in_steps;
    const uint32_t nf16 = head ? 1u : H / 16;
    N.piece_base[l] = piece;
    N.bias_base[l] = btile;
    if (concat) N.concat_mask |= 1u << l;
    if (Y.relu) N.relu_mask |= 1u << l;
    if (!Y.bias.empty()) N.bias_mask |= 1u << l;
    wpack.resize((size_t)(piece + ksteps * nf16) * 512, 0);
    for (uint32_t s = 0; s < ksteps; ++s)
      for (uint32_t f = 0; f < nf16; ++f) {
        uint16_t* dst = &wpack[(size_t)(piece + s * nf16 + f) * 512];
        for (uint32_t lane = 0; lane < 64; ++lane) {
          const uint32_t r = lane & 15, q = lane >> 4, col = 16 * f + r;
          for (uint32_t e = 0; e < 8; ++e) {
            uint32_t k;
            bool zero = col >= Y.cols;
            if (s < act_steps) {
              k = 32 * s + (e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4));
            } else {
              const uint32_t sp = s - act_steps, cd = q & 1, fr = 4 * (2 * sp + (q >> 1)) + (e & 3);
              if (fr >= E) zero = true;
              k = (concat ? H : 0) + (e < 4 ? 0 : 2 * E) + cd * E + fr;
            }
            dst[lane * 8 + e] = zero ? (uint16_t)0 : Y.kernel[(size_t)k * Y.cols + col];
          }
        }
      }
    if (head) {
      head_piece_base = piece;
      for (int o = 0; o < 3; ++o) head_bias[o] = Y.bias.empty() ? 0.f : host_h2f(Y.bias[o]);
      if (concat) {
        head_in.assign((size_t)in_dim * 4, 0.f);
        for (uint32_t f = 0; f < in_dim; ++f)
          for (int o = 0; o < 3; ++o) head_in[(size_t)f * 4 + o] = host_h2f(Y.kernel[(size_t)(H + f) * 3 + o]);
      }
    } else {
      const uint32_t nj = H / 32;
      bpack.resize((size_t)(btile + nj) * 32, 0);
      for (uint32_t j = 0; j < nj; ++j)
        for (uint32_t q = 0; q < 4; ++q)
          for (uint32_t e = 0; e < 8; ++e) {
            const uint32_t col = 32 * j + (e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4));
            bpack[(size_t)(btile + j) * 32 + q * 8 + e] = Y.bias.empty() ? (uint16_t)0 : Y.bias[col];
          }
      btile += nj;
    }
    piece += ksteps * nf16;
  }
  bpack.resize(bpack.size() + 32, 0);
  return PT_OK;
}


}  // namespace
