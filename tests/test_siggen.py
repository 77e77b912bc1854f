"""The seeded capture generator: deterministic, position-addressable, three sample formats."""
import numpy as np
import pytest


def test_deterministic_and_block_addressable(pkg):
    cfg = pkg.siggen.make_cfg(seed=7)
    whole = pkg.siggen.generate(cfg, 40000, nthreads=3)
    again = pkg.siggen.generate(cfg, 40000, nthreads=1)
    assert np.array_equal(whole, again)
    part = pkg.siggen.generate(cfg, 40000 - 8192, first_sample=8192, nthreads=2)
    assert np.array_equal(whole[2 * 8192:], part)
    other = pkg.siggen.generate(pkg.siggen.make_cfg(seed=8), 40000)
    assert not np.array_equal(whole, other)


@pytest.mark.parametrize("fmt,lim", [("SC16", 32767), ("SC16Q11", 2047)])
def test_s16_formats_are_in_range(pkg, fmt, lim):
    cfg = pkg.siggen.make_cfg(seed=9, fmt=getattr(pkg.siggen, fmt))
    iq = pkg.siggen.generate(cfg, 50000).view("<i2")
    assert iq.size == 100000 and abs(int(iq.min())) <= lim and int(iq.max()) <= lim
    assert int(np.abs(iq).max()) > lim // 4  # frames are there, not just noise


def test_unaligned_start_is_rejected(pkg):
    with pytest.raises(ValueError):
        pkg.siggen.generate(pkg.siggen.make_cfg(seed=1), 100, first_sample=5)


def test_content_decodes(pkg, oracle):
    """~2000 frames/s are generated; most of them must come out of the oracle again."""
    n = 6 * 131072
    msgs, st = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0).replay(pkg.siggen.generate(pkg.siggen.make_cfg(seed=11), n))
    expected = n / 1200
    assert 0.5 * expected < len(msgs) < 1.05 * expected
    assert set(np.unique(msgs["msgtype"])) >= {0, 4, 5, 11, 17, 20}
    assert st["demod_accepted"][1] > 0  # single-bit repairs happen
