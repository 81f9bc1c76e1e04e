"""Seeded NIF models whose arithmetic leaves the comfortable range -- stand-ins for what trained weights (which nobody
here can fetch: converted.hdf5 is absent from the reference checkout) might do to the kernels.  Every model is a plain
Dense stack the reference would build (NifModel.cpp:295-326); what is special is the VALUES:

  gated_inf          a hidden pre-activation overflows its type (binary16: > 65504; float32: > FLT_MAX) to +inf for the
                     samples where a gate feature is on, its twin to -inf (killed by the ReLU); downstream inf x w sums of
                     mixed sign give inf - inf = NaN ahead of a ReLU, which returns 0 (NifModel.cpp:323-325)
  inf_to_output      the same in the LAST hidden layer: +-inf reach the linear head and the decode (NifModel.cpp:221-245)
  nan_to_output      two infinite features meet with opposite signs in the linear head: NaN through the decode
  nan_linear_hidden  a NaN ahead of a LINEAR hidden layer (must pass through it, not be clamped)
  subnormal          a layer scaled by 2^-12 / 2^-20 (2^-130 / 2^-134 for float32 layers): subnormal weights and activations, scaled
                     back up by the next layer so that a flush to zero anywhere would change the output by O(1)
  bias_extremes      biases at +-65504 (+-FLT_MAX), one of them pushed over the top by its pre-activation
  decode_range       head outputs that put the decode's exp() argument at +100 (inf), -95 (a subnormal float), -200 (0), +85
  ..._linear_decode  log_tonemap = 0 variants (no exp)

`kinds[l]` is the numpy dtype of layer l's variables (float16 / float32): one list builds the fp16, float32 and mixed models.
"""
import numpy as np

from ipu_path_trace_amd import nif_assets

F16_MAX = 65504.0
F32_MAX = float(np.finfo(np.float32).max)


def _base(widths, emb, seed, skips=None):
    L = nif_assets.synthetic_nif(widths=widths, embedding_dim=emb, seed=seed, dtype=np.float32, skips=skips)
    return [[k.copy(), b.copy(), relu] for k, b, relu in L]


def _big(kind):       # a weight that overflows the layer's type once multiplied by an activation above ~1.1
    return 60000.0 if kind == np.float16 else 3.0e38


def _top(kind):
    return F16_MAX if kind == np.float16 else F32_MAX


def _tiny(kind, strong):   # scale that makes O(0.1) weights subnormal in the layer's type
    if kind == np.float16:
        return 2.0 ** (-20 if strong else -12)
    return 2.0 ** (-134 if strong else -130)


def _finish(L, kinds):
    out = []
    for (k, b, relu), kind in zip(L, kinds):
        kk, bb = k.astype(kind), None if b is None else b.astype(kind)
        assert np.isfinite(kk).all() and (bb is None or np.isfinite(bb).all())   # the WEIGHTS are always finite
        out.append((kk, bb, bool(relu)))
    return out


def _gate_columns(k, col_pos, col_neg, gate_rows, big):
    """Column col_pos = +big x (sum of the gate rows' activations), col_neg = -big x the same; every other row 0."""
    k[:, col_pos] = 0.0
    k[:, col_neg] = 0.0
    k[gate_rows, col_pos] = big
    k[gate_rows, col_neg] = -big


def build(case, widths, emb, kinds, seed=31):
    """-> (layers, log_tonemap).  widths: hidden widths (>= 3 hidden layers); kinds: dtype per dense layer incl. the head."""
    n_hidden = len(widths)
    assert n_hidden >= 3 and len(kinds) == n_hidden + 1
    L = _base(widths, emb, seed, skips=set())
    log_tonemap = not case.endswith("_linear_decode")
    base_case = case[:-len("_linear_decode")] if not log_tonemap else case
    last = n_hidden - 1                # index of the last hidden layer; the head is L[n_hidden]
    gates = [3, 5, 6, 9]               # features of the layer before: on (> 0) for some samples, exactly 0 for others

    if base_case == "gated_inf":
        _gate_columns(L[1][0], 0, 1, gates, _big(kinds[1]))
    elif base_case == "inf_to_output":
        _gate_columns(L[last][0], 0, 1, gates, _big(kinds[last]))
        L[last][0][:, 2] = 0.0
        L[last][0][gates[:2], 2] = _big(kinds[last])          # a second, differently gated +inf feature
        head = L[n_hidden][0]
        head[0, :] = [0.5, -0.5, 0.25]                        # +inf -> (+inf, -inf, +inf) per channel
        head[2, :] = [0.25, -0.5, 0.5]                        # same signs as row 0: never inf - inf
    elif base_case == "nan_to_output":
        _gate_columns(L[last][0], 0, 1, gates, _big(kinds[last]))
        L[last][0][:, 2] = 0.0
        L[last][0][gates[:2], 2] = _big(kinds[last])
        head = L[n_hidden][0]
        head[0, :] = [0.5, -0.5, 0.25]
        head[2, :] = [-0.25, 0.5, 0.5]                        # channels 0, 1: inf - inf = NaN where both features are inf
    elif base_case == "nan_linear_hidden":
        # layer last-1 (ReLU): features 0 and 1 -> +inf together (same gates); layer `last` is LINEAR:
        # feature 0 = inf - inf = NaN, every other feature = +inf; the head's weights are chosen so that a NaN that was
        # clamped to -inf on its way through the linear layer would give +inf instead of NaN
        big = _big(kinds[last - 1])
        k = L[last - 1][0]
        k[:, 0] = 0.0
        k[:, 1] = 0.0
        k[gates, 0] = big
        k[gates, 1] = big
        kl = L[last][0]
        kl[0, :] = 0.01
        kl[1, :] = 0.01
        kl[0, 0], kl[1, 0] = 1.0, -1.0
        L[last][2] = False
        head = L[n_hidden][0]
        head[:, :] = np.abs(head) + 1e-3
        head[0, :] = -0.5
    elif base_case in ("subnormal", "subnormal_strong"):
        strong = base_case == "subnormal_strong"
        # the scale that makes layer 1's weights and outputs subnormal -- in binary16 if layer 1 or the layer that reads
        # its output is binary16 (a float32 layer's output is then cast to a subnormal half), else in float32
        s = _tiny(np.float16 if np.float16 in (kinds[1], kinds[2]) else np.float32, strong)
        L[1][0] *= s
        L[1][1] *= s
        # ... and back up by layer 2, as far as ITS weight type allows (binary16 weights stay below 65504), so that the
        # signal is again comparable with the biases: a flush to zero anywhere in between changes the output by tens of percent
        up = min(1.0 / s, 2.0 ** 14) if kinds[2] == np.float16 else min(1.0 / s, 2.0 ** 126)
        L[2][0] *= up
        # what is still missing goes into the layer after (ReLU is positively homogeneous: with layer 2's bias scaled
        # alike the network computes the base network's function, so a flush anywhere shows as an O(1) change)
        L[2][1] *= s * up
        L[3][0] *= 1.0 / (s * up)
    elif base_case == "bias_extremes":
        top = _top(kinds[last])
        b = L[last][1]
        b[0], b[1], b[2] = top, -top, top
        k = L[last][0]
        # feature 2: a pre-activation far above half an ulp of `top` (16 for binary16, 1e31 for float32): top + x overflows
        k[:, 2] = np.abs(k[:, 2]) * (400.0 if kinds[last] == np.float16 else 1e34) + (1.0 if kinds[last] == np.float16 else 1e33)
        head = L[n_hidden][0]
        head[0, :] = np.array([1.0, -1.0, 0.5]) * (2e-4 if kinds[last] == np.float16 else 1e-38)   # top x w stays finite
        head[2, :] = [0.5, 0.5, -0.5]
    elif base_case == "decode_range":
        L[n_hidden][0] *= 0.02
        # exp() arguments o x max + mean of about +100 (-> inf), -95 (a subnormal float) and -200 (-> 0)
        L[n_hidden][1][:] = [29.84, -27.04, -57.7]
    elif base_case == "decode_range_finite":
        L[n_hidden][0] *= 0.02
        L[n_hidden][1][:] = [25.47, -24.0, 10.0]              # +85 (8e36), -84.6, +32: large but finite everywhere
    else:
        raise ValueError(case)
    return _finish(L, kinds), log_tonemap


CASES = ["gated_inf", "inf_to_output", "nan_to_output", "nan_linear_hidden", "subnormal", "subnormal_strong", "bias_extremes",
         "decode_range", "decode_range_finite", "inf_to_output_linear_decode", "nan_to_output_linear_decode",
         "bias_extremes_linear_decode"]


def classes(x):
    """Per element: 0 finite non-zero, 1 zero, 2 +inf, 3 -inf, 4 NaN.  The two smallest subnormal floats count as zero: exp()
    of an argument at the very end of its range (-103.3) lands on 0 or on 1.4e-45 depending on the last bit of the argument."""
    x = np.asarray(x)
    c = np.zeros(x.shape, dtype=np.int8)
    c[np.abs(x) < 4e-45] = 1
    c[np.isposinf(x)] = 2
    c[np.isneginf(x)] = 3
    c[np.isnan(x)] = 4
    return c


def sample_points(n, seed=17):
    rng = np.random.default_rng(seed)
    return rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32)
