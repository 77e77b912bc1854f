"""The buffer-by-buffer feed used to model a live receiver with dropped samples (tests/helpers.py
oracle_live_feed) is the oracle's own file replay when nothing is dropped."""
import numpy as np
import pytest

from helpers import FIELDS, assert_same, oracle_live_feed


@pytest.mark.parametrize("n", [3 * 131072 + 777, 2 * 131072, 1000])
def test_buffer_feed_equals_file_replay(pkg, oracle, n):
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=5 + n % 11, msgs_per_sec=6000, n_aircraft=12), n)
    want, wstats = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0).replay(iq, cap=1 << 14)
    cut = (n // 131072 // 2) * 131072 * 2
    got, gstats = oracle_live_feed(oracle.Oracle(oracle.FMT_UC8, 58, 1, 0), [iq[:cut], iq[cut:]] if cut else [iq],
                                   [0, 0] if cut else [0])
    assert len(want) > 0 or n < 131072
    assert_same(got, gstats, want, wstats)


def test_dropped_samples_move_the_clock_and_blank_the_overlap(pkg, oracle):
    n = 4 * 131072
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=77, msgs_per_sec=6000, n_aircraft=12), n)
    a, b = iq[: n], iq[n:]
    same, _ = oracle_live_feed(oracle.Oracle(oracle.FMT_UC8, 58, 1, 0), [iq[: 2 * 131072 * 2], iq[2 * 131072 * 2:]], [0, 0])
    gap, _ = oracle_live_feed(oracle.Oracle(oracle.FMT_UC8, 58, 1, 0), [iq[: 2 * 131072 * 2], iq[2 * 131072 * 2:]], [0, 12345])
    first = same["timestampMsg"] < 2 * 131072 * 5
    assert np.array_equal(gap[: first.sum()], same[: first.sum()])
    later_same, later_gap = same[first.sum():], gap[first.sum():]
    # messages well inside the second half only move on the clock
    inner = later_same["timestampMsg"] > 2 * 131072 * 5 + 4000
    ts = set((later_gap["timestampMsg"] - 12345 * 5).tolist())
    assert inner.sum() > 10 and all(int(t) in ts for t in later_same["timestampMsg"][inner])
