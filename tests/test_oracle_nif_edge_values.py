"""The edge-value NIF models (tests/nif_edge_models.py) on the CPU: the oracle must really show the classes they were built
for -- NaN, +inf, -inf, exact zeros, subnormal floats -- at SOME positions and ordinary finite values at others, or the GPU
comparison (tests/test_gpu_nif_edge_values.py) would compare nothing.  Also pins the oracle's own rules for these values
against an independent numpy restatement (ReLU of a NaN is 0; a linear layer passes a NaN; subnormal binary16 values
survive every rounding point)."""
import numpy as np
import pytest

from ipu_path_trace_amd import nif_assets
from tests import nif_edge_models as M

META = nif_assets.URBAN_ALLEY_META
H, F = np.float16, np.float32


def _numpy_nif(L, emb, u, v, log_tonemap):
    """Independent restatement for all-binary16 models: float64 sums (exact for these sizes up to the final rounding),
    rounded to half at the reference's points (NifModel.cpp:314-325), decode in float32 (:221-245)."""
    un, vn = (u.astype(F) - F(1)) * F(2), (v.astype(F) - F(1)) * F(2)
    p = (2.0 ** np.arange(emb)).astype(F)
    au, av = (un[:, None] * p).astype(H).astype(F), (vn[:, None] * p).astype(H).astype(F)
    feat = np.concatenate([np.sin(au), np.sin(av), np.cos(au), np.cos(av)], axis=1).astype(H).astype(np.float64)
    x = feat
    with np.errstate(all="ignore"):
        for k, b, relu in L:
            kk = k.astype(np.float64)
            if x.shape[1] != kk.shape[0]:
                x = np.concatenate([x, feat], axis=1)
            # inf x 0 and inf - inf must come out as NaN exactly as in a sequential sum: numpy's matmul does that
            y = (x @ kk).astype(F).astype(H)
            if b is not None:
                y = (y.astype(F) + b.astype(F)).astype(H)
            y = y.astype(np.float64)
            if relu:
                y = np.where(y > 0, y, 0.0)
            x = y
        o = x[:, :3].astype(F) * F(META["max"]) + np.asarray(nif_assets.folded_mean(), dtype=F)
        return np.exp(o.astype(np.float64)).astype(F) if log_tonemap else o.astype(F)


@pytest.mark.parametrize("kinds", [[H] * 4, [F] * 4, [H, F, F, H]], ids=["binary16", "float32", "mixed"])
def test_edge_models_show_their_classes_in_the_oracle(oracle, kinds):
    u, v = M.sample_points(1500)
    mean = nif_assets.folded_mean()
    seen = np.zeros(5, dtype=np.int64)
    subnormal_out = 0
    for case in M.CASES:
        L, lt = M.build(case, [64] * 3, 4, kinds)
        ref = oracle.Nif(L, 4, META["max"], mean, log_tonemap=lt).infer(u, v)
        c = M.classes(ref)
        seen += np.bincount(c.ravel(), minlength=5)
        subnormal_out += int(((np.abs(ref) < np.finfo(F).tiny) & (ref != 0)).sum())
        if case.startswith("nan_to_output"):
            assert (c == 4).any() and (c != 4).any()                # NaN for some samples, not for others
        if case.startswith("inf_to_output"):
            assert (c == 2).any() and (c == 0).any()
        if case == "nan_linear_hidden":
            assert (c == 4).any() and (c == 0).any()
        if case.startswith("subnormal"):
            assert (c == 0).all()
            # zeroing the scaled-down layer changes the output by tens of percent: a flush to zero would be seen
            flushed = [(k * 0 if i == 1 else k, b * 0 if i == 1 else b, relu) for i, (k, b, relu) in enumerate(L)]
            other = oracle.Nif(flushed, 4, META["max"], mean, log_tonemap=lt).infer(u, v)
            assert np.median(np.abs(other - ref) / np.abs(ref)) > 0.2
            k1 = L[1][0]
            assert np.mean(np.abs(k1.astype(np.float64)) < np.finfo(k1.dtype).tiny) > 0.8   # the weights ARE subnormal (or zero)
    assert (seen > 0).all(), seen
    assert subnormal_out > 0                                          # decode_range: exp(-95) is a subnormal float


def test_oracle_edge_rules_against_numpy(oracle):
    """All-binary16 models: same classes as the numpy restatement everywhere; finite values within one half-ulp-of-a-hidden-
    activation's worth (the restatement sums in float64, the oracle in a float FMA chain)."""
    u, v = M.sample_points(800)
    mean = nif_assets.folded_mean()
    for case in M.CASES:
        L, lt = M.build(case, [64] * 3, 4, [H] * 4)
        ref = oracle.Nif(L, 4, META["max"], mean, log_tonemap=lt).infer(u, v)
        want = _numpy_nif(L, 4, u, v, lt)
        cr, cw = M.classes(ref), M.classes(want)
        # a float64 sum can stay finite where the float chain overflows only in cases built to sit far from the edge: none here
        assert np.mean(cr != cw) < 2e-3, (case, np.mean(cr != cw))
        fin = (cr == 0) & (cw == 0)
        if fin.any():
            rel = np.abs(ref[fin] - want[fin]) / np.abs(want[fin])
            assert np.median(rel) < 1e-3 and rel.max() < 5e-2, (case, np.median(rel), rel.max())
