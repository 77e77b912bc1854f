"""Two readings of the reference against each other: oracle/modes_oracle.c (the checker of every parity test) and
tests/indep_demod.py (numpy / plain Python, written from demod_2400.c, mode_s.c, crc.c, icao_filter.c, convert.c and
mode_ac.c without looking at the oracle).  Same capture in, same ordered message list and the same demodulator counters
out -- bit for bit, floats included.  The reference cannot be built in this image, so this does not pin the oracle
to the reference; it does say that the oracle is not one person's single misreading.  CPU only; the GPU leg
(HIP path against this second reading directly, not through the oracle) is tests/test_gpu_indep_demod.py."""
import numpy as np
import pytest

import indep_demod as D
import indep_signal as S

OFMT = {"uc8": "FMT_UC8", "sc16": "FMT_SC16", "sc16q11": "FMT_SC16Q11"}
INT_STATS = ("demod_preambles", "demod_rejected_bad", "demod_rejected_unknown_icao", "demod_accepted",
             "demod_preamblePhase", "demod_bestPhase", "demod_modeac", "strong_signal_count", "noise_power_count",
             "signal_power_count")
FLOAT_STATS = ("noise_power_sum", "signal_power_sum", "peak_signal_power")


def assert_second_reading_agrees(got, gstats, want, wstats):
    """got: indep_demod's list of dicts; want: a MESSAGE_DTYPE array (the oracle's, or the HIP path's)"""
    assert len(got) == len(want), (len(got), len(want))
    for i, (g, w) in enumerate(zip(got, want)):
        nb = int(w["msgbits"]) // 8
        for k in ("timestampMsg", "sysTimestampMsg", "msgtype", "msgbits", "addr", "correctedbits"):
            assert g[k] == int(w[k]), (i, k, g, w)
        assert g["msg"][:nb] == bytes(w["msg"][:nb]), (i, g, w)
        if g["msgtype"] != 32:  # Mode S: what demodulate2400 itself decides and measures
            for k in ("score", "bestphase", "crc", "iid"):
                assert g[k] == int(w[k]), (i, k, g, w)
            assert g["signalLevel"] == float(w["signalLevel"]), (i, g, w)  # the same double, not a tolerance
    for k in INT_STATS:
        assert gstats[k] == wstats[k], (k, gstats[k], wstats[k])
    for k in FLOAT_STATS:
        assert np.array_equal(np.float64(gstats[k]), np.float64(wstats[k]), equal_nan=True), (k, gstats[k], wstats[k])


def both(oracle, fmt, raw, threshold=58, nfix=1, mode_ac=False, dc_filter=False):
    raw = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
    got, gstats = D.Receiver(fmt, threshold, nfix, mode_ac, dc_filter=dc_filter).replay(raw.tobytes())
    want, wstats = oracle.Oracle(getattr(oracle, OFMT[fmt]), threshold, nfix, 1 if mode_ac else 0,
                                 dc_filter=dc_filter).replay(raw, cap=1 << 18)
    assert_second_reading_agrees(got, gstats, want, wstats)
    return got


CORPUS = {
    "plain": dict(),
    "carrier_plus_minus_50kHz": dict(freq_offset_hz=50e3),
    "dc_offset": dict(dc=(0.06, -0.04)),
    "clipped": dict(clip_gain=1.6),
    "echo_750ns": dict(echo=(9, 0.35, 1.0)),
    "dense_noisy": dict(frames_per_sec=12000.0, noise=0.06, n_aircraft=400),
    "mode_ac": dict(ac_per_sec=3000.0),
    "random_bytes": dict(random_bytes=True),
}


@pytest.mark.parametrize("case", sorted(CORPUS))
@pytest.mark.parametrize("fmt", ["uc8", "sc16", "sc16q11"])
def test_numpy_corpus(oracle, fmt, case):
    """the independent corpus (tests/indep_signal.py), Mode A/C on, --fix"""
    iq, _ = S.capture(101 + sorted(CORPUS).index(case), 4 * 131072 + 1234, fmt=fmt, **CORPUS[case])
    got = both(oracle, fmt, iq, nfix=1, mode_ac=True)
    if case not in ("random_bytes",):
        assert len(got) > 50


@pytest.mark.parametrize("threshold", [40, 58, 75, 400])
@pytest.mark.parametrize("nfix", [0, 1, 2])
def test_thresholds_no_fix_and_aggressive(oracle, threshold, nfix):
    iq, _ = S.capture(33, 3 * 131072 + 77, fmt="uc8", frames_per_sec=4000.0, noise=0.04, flip_fraction=0.2)
    got = both(oracle, "uc8", iq, threshold=threshold, nfix=nfix)
    if threshold != 400:
        assert max(m["correctedbits"] for m in got) == nfix  # the repairs the option allows really happen


@pytest.mark.parametrize("bits", [56, 112])
def test_two_bit_tables_of_both_readings(oracle, bits):
    """prepareErrorTable(bits, 2, 4) twice: every entry of this reading's table is the oracle's answer, and the oracle
    has no answer where this table has none (every one- and two-bit pattern, a sample of three-bit ones, random ones)"""
    import itertools
    table = D.error_table(bits, 2)
    o = oracle.Oracle(oracle.FMT_UC8, 58, 2, 0)
    single = D.SINGLE[112 - bits:]
    syns = set(single) | {a ^ b for a, b in itertools.combinations(single, 2)}
    rng = np.random.default_rng(bits)
    tri = list(itertools.combinations(single[5:], 3))
    syns |= {a ^ b ^ c for a, b, c in (tri[i] for i in rng.integers(0, len(tri), 5000))}
    syns |= {int(x) for x in rng.integers(1, 1 << 24, 5000)}
    syns.discard(0)
    for s in syns:
        n, wrong = o.diagnose(s, bits)
        if s in table:
            assert n == len(table[s]) and tuple(wrong[:n]) == table[s], (hex(s), n, wrong, table[s])
        else:
            assert n == -1, (hex(s), n, wrong)


@pytest.mark.parametrize("fmt", ["uc8", "sc16", "sc16q11"])
def test_capture_that_ends_on_a_buffer_boundary(oracle, fmt):
    """the reader then hands over one more, empty buffer (sdr_ifile.c:200-216): its means are 0 / 0 and the noise power
    statistic is a NaN from there on -- in both readings"""
    iq, _ = S.capture(5, 2 * 131072, fmt=fmt, ac_per_sec=2000.0)
    both(oracle, fmt, iq, mode_ac=True)


@pytest.mark.parametrize("fmt", ["uc8", "sc16", "sc16q11"])
def test_dc_blocking_converters(oracle, fmt):
    """--dcfilter: convert_*_generic, the filter state carried through the stream, a receiver with a DC offset"""
    iq, _ = S.capture(77, 2 * 131072 + 3000, fmt=fmt, dc=(0.05, -0.03), ac_per_sec=1500.0)
    got = both(oracle, fmt, iq, mode_ac=True, dc_filter=True)
    assert len(got) > 50


@pytest.mark.parametrize("fmt", ["sc16", "sc16q11", "uc8"])
@pytest.mark.parametrize("extra", [1, 2, 7])
def test_last_buffer_of_a_few_samples_with_mode_ac(oracle, fmt, extra):
    """A float mean of squares can round below the square of the float mean when a buffer holds one or two samples:
    demodulate2400AC takes the square root of a negative number and casts the NaN to unsigned (demod_2400.c:530-531,
    undefined in C; 0 on the reference's x86-64 build).  Found by the fuzzer's third witness (case 200103)."""
    iq, _ = S.capture(9, 131072 + extra, fmt=fmt, ac_per_sec=3000.0)
    both(oracle, fmt, iq, mode_ac=True)


@pytest.mark.parametrize("seed", [10901, 10920])
@pytest.mark.parametrize("fmt", ["UC8", "SC16", "SC16Q11"])
def test_generator_of_the_benchmark(pkg, oracle, fmt, seed):
    """the benchmark's own content model (csrc/msd_siggen.c), five buffers of it"""
    cfg = pkg.siggen.make_cfg(seed=seed, fmt=getattr(pkg.siggen, fmt), ac_per_sec=1500)
    iq = pkg.siggen.generate(cfg, 5 * 131072 + 4096)
    got = both(oracle, fmt.lower(), iq, mode_ac=True)
    assert len(got) > 300 and {m["msgtype"] for m in got} >= {0, 4, 5, 11, 17, 20, 32}


def test_crowded_filter_and_the_short_cut_of_its_membership_test(pkg, oracle):
    """30 000 aircraft: the 8192-slot table (two entries per address) is full after the first seconds and
    icaoFilterAdd gives up (icao_filter.c:73-76).  The second reading walking the table literally, its set short cut and
    the oracle agree."""
    cfg = pkg.siggen.make_cfg(seed=31, msgs_per_sec=12000, n_aircraft=30000, noise_fs=0.005)
    iq = pkg.siggen.generate(cfg, 6 * 131072 + 4096)
    got = both(oracle, "uc8", iq)
    lit = D.Receiver("uc8", 58, 1, False, literal_filter=True)
    msgs, stats = lit.replay(iq.tobytes())
    assert [m["timestampMsg"] for m in msgs] == [m["timestampMsg"] for m in got] and len(got) > 500
    rng = np.random.default_rng(3)
    fast = D.IcaoFilter()
    slow = D.IcaoFilter(literal=True)
    for k, a in enumerate(rng.integers(1, 1 << 24, 9000).tolist()):
        fast.add(a)
        slow.add(a)
        if k == 4000:
            fast.expire(0)
            slow.expire(0)
    assert fast.a == slow.a and fast.b == slow.b
    assert D.IcaoFilter.EMPTY not in slow.b  # 5000 addresses after the flip: the active table is full, later adds gave up
    for a in rng.integers(1, 1 << 24, 300).tolist() + [x for x in slow.a[:200] if x != D.IcaoFilter.EMPTY]:
        assert fast.test(a) == slow.test(a)


def test_a_long_quiet_stretch_flips_the_filter(oracle):
    """more than 60 s between two messages of an aircraft that only ever sent DF4: the first table is wiped at the second
    flip (icao_filter.c:150-164), so the second one is rejected -- unless the squitter in between re-announced it.
    Synthetic magnitudes would need 150 M samples; the filter is driven directly instead."""
    f = D.IcaoFilter()
    f.expire(0)            # first backgroundTasks(): active = b
    f.add(0x4840D6)
    f.expire(59999)
    assert f.test(0x4840D6)
    f.expire(60000)        # active = a (wiped); b still holds it
    assert f.test(0x4840D6)
    f.expire(120000)       # active = b, wiped
    assert not f.test(0x4840D6)
    o = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0)
    assert o.filter_test(0x4840D6) == 0
    o.filter_add(0x4840D6)
    assert o.filter_test(0x4840D6) == 1 and o.filter_test(0x4840D7) == 0
    assert D.IcaoFilter.hash(0x4840D6) == D.IcaoFilter.hash(0x4840D6 | 0xFF000000)  # three bytes only


def test_pulse_train_of_preambles(oracle):
    """the adversarial capture of tests/test_gpu_adversarial.py (36 % of the positions are preamble hits, five trial
    phases each): one buffer and a bit, both readings"""
    from test_gpu_adversarial import pulse_train
    n = 131072 + 5000
    got, gstats = D.Receiver("uc8", 58, 1, True).replay(pulse_train(n, 7).tobytes())
    want, wstats = oracle.Oracle(oracle.FMT_UC8, 58, 1, 1).replay(pulse_train(n, 7), cap=1 << 16)
    assert gstats["demod_preambles"] > 0.3 * n
    assert_second_reading_agrees(got, gstats, want, wstats)


@pytest.mark.parametrize("fmt,mode_ac", [("uc8", False), ("sc16", True)])
@pytest.mark.parametrize("drop", [1, 131072 + 17, (1 << 32) + 12345])
def test_live_feed_with_dropped_samples(pkg, oracle, fmt, mode_ac, drop):
    """gaps in the stream: the sample clock runs on, the buffer behind a gap has no look-behind -- both readings"""
    from helpers import oracle_live_feed
    C, bps = 131072, 2 if fmt == "uc8" else 4
    n = 9 * C + 999
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=400, fmt=getattr(pkg.siggen, fmt.upper()), msgs_per_sec=7000,
                                                 n_aircraft=25, ac_per_sec=1000), n)
    cuts, drops = [0, 2 * C, 6 * C, n], [0, drop, 3 * drop + 1]
    segs = [iq[bps * a: bps * b] for a, b in zip(cuts[:-1], cuts[1:])]
    want, wstats = oracle_live_feed(oracle.Oracle(getattr(oracle, OFMT[fmt]), 58, 1, int(mode_ac)), segs, drops,
                                    bytes_per_sample=bps)
    got, gstats = D.Receiver(fmt, 58, 1, mode_ac).live_feed([s.tobytes() for s in segs], drops)
    assert len(got) > 100
    assert_second_reading_agrees(got, gstats, want, wstats)


def test_threshold_while_samples_were_dropped_recently(pkg, oracle):
    """demod_2400.c:285-290 in both restatements: with Modes.stats_15min.samples_dropped set the preamble tests use
    max(75, threshold) -- equal to a receiver configured with that maximum, and no change for a threshold above 75."""
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=77, msgs_per_sec=9000, n_aircraft=40), 3 * 131072)
    for configured in (58, 90):
        orc = oracle.Oracle(oracle.FMT_UC8, configured, 1, 0)
        orc.set_recently_dropped(True)
        want, wstats = orc.replay(iq, cap=1 << 16)
        plain, pstats = oracle.Oracle(oracle.FMT_UC8, max(75, configured), 1, 0).replay(iq, cap=1 << 16)
        assert np.array_equal(want, plain) and wstats["demod_preambles"] == pstats["demod_preambles"]
        second = D.Receiver("uc8", configured, 1, False)
        second.recently_dropped = True
        msgs, sstats = second.replay(iq.tobytes())
        assert_second_reading_agrees(msgs, sstats, want, wstats)
