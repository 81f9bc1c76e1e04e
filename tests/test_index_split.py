"""The round-up reciprocal the trace kernel splits a path index with (csrc/ptmi_context.h::item_divider, pt_trace.h::split_index):
(x * magic) >> shift must equal x // n for EVERY x < 2^31 -- the batch size keeps path indices below 2^31 (pt_create).  The
formula is restated here (the library's copy is host C++ inside libptmi.so) and checked where it could break: around every
multiple of n near the top of the range, at powers of two and their neighbours, and for the work-item counts the configs use."""
import numpy as np
import pytest


def item_divider(n):
    s = 0
    while (1 << s) < n:
        s += 1
    magic = ((1 << (31 + s)) // n) + 1
    assert magic < (1 << 32)
    return magic, 31 + s


CASES = [1, 2, 3, 5, 7, 63, 64, 65, 255, 256, 257, 1000, 4097, 65535, 65536, 65537, 138000, 1104000, 1104001, 8294400, (1 << 20) - 1, 1 << 20,
         (1 << 20) + 1, (1 << 30) - 1, 1 << 30, (1 << 30) + 1, (1 << 31) - 1]


@pytest.mark.parametrize("n", CASES)
def test_reciprocal_multiply_is_exact_below_two_to_the_31(n):
    magic, shift = item_divider(n)
    err = magic * n - (1 << shift)
    assert 0 < err <= (1 << (shift - 31))                   # Granlund & Montgomery's condition for 31-bit dividends
    top = (1 << 31) - 1
    rng = np.random.default_rng(n)
    xs = set(int(v) for v in rng.integers(0, 1 << 31, size=20000))
    for q in list(range(0, min(top // n, 2000) + 1)) + [top // n - d for d in range(0, 2000) if top // n - d >= 0]:
        for d in (-1, 0, 1):
            x = q * n + d
            if 0 <= x <= top:
                xs.add(x)
    xs.update([0, 1, top, top - 1])
    for x in xs:
        assert (x * magic) >> shift == x // n, (n, x)


def test_every_small_divisor_exhaustively_on_a_dense_range():
    for n in range(1, 300):
        magic, shift = item_divider(n)
        x = np.arange(0, 200000, dtype=np.uint64)
        assert np.array_equal((x * np.uint64(magic)) >> np.uint64(shift), x // np.uint64(n)), n
        x = np.arange((1 << 31) - 200000, 1 << 31, dtype=np.uint64)      # magic < 2^32, x < 2^31: the product fits 64 bits
        assert np.array_equal((x * np.uint64(magic)) >> np.uint64(shift), x // np.uint64(n)), n
