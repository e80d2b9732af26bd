"""The library's own radix sort and running maximum (gaml_amd/csrc/radix_sort.hip.h) -- what orders the aligner's hits of
a large batch (reference: alignments filed per window sorted by (position, read), first found survives,
graph.cc:841, 891, 895-897) and the PacBio coverage sweep's intervals (graph.cc:3198-3250) -- against numpy's stable sort.
Integer work: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _stable_sorted(keys, vals, begin_bit, end_bit):
    width = end_bit - begin_bit
    digits = (keys >> np.uint64(begin_bit)) & np.uint64((1 << width) - 1 if width < 64 else 0xFFFFFFFFFFFFFFFF) if width > 0 else np.zeros_like(keys)
    order = np.argsort(digits, kind="stable")
    return keys[order], None if vals is None else vals[order]


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 4095, 4096, 4097, 8191, 70_001, 1_300_003])
def test_sort_pairs_is_numpys_stable_sort(n):
    from gaml_amd import api
    ctx = api.Context(device=0)
    rng = np.random.default_rng(n)
    for begin_bit, end_bit, spread in ((0, 64, 64), (0, 56, 56), (0, 41, 41), (8, 20, 64), (0, 8, 3), (5, 5, 64)):
        keys = rng.integers(0, 1 << min(spread, 63), n, dtype=np.uint64)
        if spread == 64:
            keys |= rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(63)
        vals = np.arange(n, dtype=np.uint64) * np.uint64(3) + np.uint64(7)
        want_k, want_v = _stable_sorted(keys, vals, begin_bit, end_bit)
        got_k, got_v, got_m = ctx.debug_radix_sort(keys, vals, begin_bit, end_bit, running_max=True)
        assert (got_k == want_k).all() and (got_v == want_v).all(), (n, begin_bit, end_bit)
        assert (got_m == np.maximum.accumulate(want_v)).all(), (n, begin_bit, end_bit)
        got_k, got_v, got_m = ctx.debug_radix_sort(keys, None, begin_bit, end_bit, running_max=True)
        assert got_v is None and (got_k == want_k).all()
        assert (got_m == np.maximum.accumulate(want_k)).all()
    ctx.close()


def test_few_distinct_keys_keep_their_input_order():
    """Runs of equal keys longer than a tile, and a single key: stability is what the second (major) sort of the hits relies on."""
    from gaml_amd import api
    ctx = api.Context(device=0)
    n = 300_000
    for distinct in (1, 2, 5):
        keys = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) >> np.uint64(7)) % np.uint64(distinct) << np.uint64(33)
        vals = np.arange(n, dtype=np.uint64)
        got_k, got_v, _ = ctx.debug_radix_sort(keys, vals, 0, 40)
        want_k, want_v = _stable_sorted(keys, vals, 0, 40)
        assert (got_k == want_k).all() and (got_v == want_v).all()
    ctx.close()
