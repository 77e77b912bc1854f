"""Parity of the HIP path (through the C-ABI) against the oracle on seeded synthetic captures."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COUNTERS = ("demod_preambles", "demod_rejected_bad", "demod_rejected_unknown_icao", "demod_accepted",
            "demod_preamblePhase", "demod_bestPhase", "demod_modeac", "strong_signal_count",
            "samples_processed", "noise_power_count", "signal_power_count", "buffers")
FIELDS = ("timestampMsg", "sysTimestampMsg", "signalLevel", "addr", "crc", "score", "msgtype", "msgbits",
          "correctedbits", "bestphase", "iid")


@pytest.fixture(params=["gpu-resolve", "host-resolve"], autouse=True)
def resolve_stage(request, monkeypatch):
    """Every parity case runs with the ordered resolve stage on the GPU (default) and on host threads."""
    monkeypatch.setenv("MSD_GPU_RESOLVE", "1" if request.param == "gpu-resolve" else "0")
    return request.param


def assert_same(got, gstats, want, wstats):
    assert len(got) == len(want), (len(got), len(want))
    for f in FIELDS:
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(got["msg"], want["msg"])
    for k in COUNTERS:
        assert gstats[k] == wstats[k], (k, gstats[k], wstats[k])
    for k in ("noise_power_sum", "signal_power_sum", "peak_signal_power"):
        assert np.array_equal(np.float64(gstats[k]), np.float64(wstats[k]), equal_nan=True), (k, gstats[k], wstats[k])


def run_case(pkg, oracle, torch, fmt, n, seed, nfix, batch=None, mode_ac=0, **cfgkw):
    cfg = pkg.siggen.make_cfg(seed=seed, fmt=fmt, **cfgkw)
    iq = pkg.siggen.generate(cfg, n)
    d_iq = torch.from_numpy(iq).to("cuda:0")
    max_batch = batch or max(pkg.CHUNK, ((n + pkg.CHUNK - 1) // pkg.CHUNK) * pkg.CHUNK)
    dem = pkg.Demodulator(fmt=fmt, nfix_crc=nfix, mode_ac=mode_ac, max_batch_samples=max_batch,
                          message_capacity=1 << 18)
    if batch:
        got = pkg.replay_device(dem, d_iq.data_ptr(), n, batch)
    else:
        got = dem.submit_device(d_iq.data_ptr(), n, last=True)
    want, wstats = oracle.Oracle(fmt, 58, nfix, mode_ac).replay(iq, cap=1 << 18)
    assert len(want) > 0
    assert_same(got, dem.stats(), want, wstats)
    return got, dem


@pytest.mark.parametrize("nfix", [0, 1, 2])
@pytest.mark.parametrize("n", [3 * 131072 + 4567, 4 * 131072, 131072 + 1])
def test_uc8_single_batch(pkg, oracle, torch_cuda, n, nfix):
    run_case(pkg, oracle, torch_cuda, pkg.FMT_UC8, n, seed=1090 + n % 7, nfix=nfix)


@pytest.mark.parametrize("fmt", ["uc8", "sc16"])
@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 325, 326, 327, 1000, 4095, 131071, 131073])
def test_short_captures(pkg, oracle, torch_cuda, n, fmt):
    """Captures shorter than a buffer, shorter than the 326-sample overlap, shorter than one 8-sample load
    group, and empty: same messages (possibly none), counters and per-buffer means as the oracle."""
    f, of = (pkg.FMT_UC8, oracle.FMT_UC8) if fmt == "uc8" else (pkg.FMT_SC16, oracle.FMT_SC16)
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=90 + n % 13, fmt=f, msgs_per_sec=8000, n_aircraft=4,
                                                 overlap_permille=0), max(n, 4096))[: n * (2 if fmt == "uc8" else 4)]
    d = torch_cuda.from_numpy(np.concatenate([iq, np.zeros(64, dtype=np.uint8)])).to("cuda:0")  # keep a valid pointer for n = 0
    dem = pkg.Demodulator(fmt=f, nfix_crc=1, max_batch_samples=2 * 131072, message_capacity=1 << 12)
    got = dem.submit_device(d.data_ptr(), n, last=True)
    want, wstats, wmeans = oracle.Oracle(of, 58, 1, 0).replay(iq, cap=1 << 12, want_means=True)
    assert_same(got, dem.stats(), want, wstats)
    gmeans = dem.buffer_means()
    assert np.array_equal(gmeans, wmeans[: len(gmeans)], equal_nan=True) and len(gmeans) == wstats["buffers"]
    if n >= 131071:
        assert len(want) > 0


def test_uc8_pipelined_batches(pkg, oracle, torch_cuda):
    run_case(pkg, oracle, torch_cuda, pkg.FMT_UC8, 20 * 131072 + 999, seed=5, nfix=1, batch=4 * 131072)


@pytest.mark.parametrize("pinned", [True, False])
def test_streaming_host_ingest(pkg, oracle, torch_cuda, pinned):
    """msd_launch_host / msd_collect: uploads run ahead of the kernels, three batches in flight, from
    page-locked buffers (msd_host_alloc) or ordinary host memory; same messages as the oracle."""
    n, batch = 22 * 131072 + 4321, 4 * 131072
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=31, msgs_per_sec=3000), n)
    dem = pkg.Demodulator(fmt=pkg.FMT_UC8, nfix_crc=1, max_batch_samples=batch, message_capacity=1 << 18)
    bufs = [dem.host_buffer(batch * 2) if pinned else np.empty(batch * 2, dtype=np.uint8) for _ in range(pkg.capi.PIPELINE_DEPTH)]
    parts, inflight, off, k = [], 0, 0, 0
    while off < n:
        m = min(batch, n - off)
        if inflight == pkg.capi.PIPELINE_DEPTH:
            parts.append(dem.collect())
            inflight -= 1
        b = bufs[k % len(bufs)]  # its previous batch has been collected by now
        b[: m * 2] = iq[off * 2:(off + m) * 2]
        dem.launch_host(b, m, last=off + m >= n)
        inflight += 1
        off += m
        k += 1
    while inflight:
        parts.append(dem.collect())
        inflight -= 1
    got = np.concatenate(parts)
    want, wstats = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0).replay(iq, cap=1 << 18)
    assert_same(got, dem.stats(), want, wstats)


@pytest.mark.parametrize("fmt", ["sc16", "sc16q11"])
def test_s16_formats(pkg, oracle, torch_cuda, fmt):
    f = pkg.FMT_SC16 if fmt == "sc16" else pkg.FMT_SC16Q11
    run_case(pkg, oracle, torch_cuda, f, 3 * 131072 + 77, seed=10920, nfix=1)


@pytest.mark.parametrize("fmt", ["uc8", "sc16q11"])
def test_mode_s_plus_mode_ac_fix(pkg, oracle, torch_cuda, fmt):
    """BASELINE.json configs[4]: Mode S + Mode A/C combined (demodulate2400AC) with --fix; six buffers, so the
    GPU resolve handles the Mode A/C skip-ahead too when it is on."""
    f = pkg.FMT_UC8 if fmt == "uc8" else pkg.FMT_SC16Q11
    got, dem = run_case(pkg, oracle, torch_cuda, f, 6 * 131072 + 31, seed=44, nfix=1, mode_ac=1, msgs_per_sec=200,
                        ac_per_sec=2000)
    assert dem.stats()["demod_modeac"] > 300
    assert (got["msgtype"] == 32).sum() == dem.stats()["demod_modeac"]


def test_mode_ac_pipelined(pkg, oracle, torch_cuda):
    run_case(pkg, oracle, torch_cuda, pkg.FMT_UC8, 12 * 131072, seed=45, nfix=1, mode_ac=1, msgs_per_sec=2000,
             ac_per_sec=1000, batch=4 * 131072)


@pytest.mark.parametrize("path", ["fused", "magbuf"])
def test_replay_cli_matches_oracle(pkg, oracle, torch_cuda, tmp_path, path):
    """BASELINE.json configs[0]: --ifile replay through the sdr.h-handler / mag_buf boundary (host C)."""
    import os
    import subprocess
    n = 24_000_000 if path == "fused" else 12 * 131072 + 5000   # 10 s of signal for the fast path
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=1090), n)
    f = tmp_path / "capture.uc8"
    iq.tofile(f)
    exe = os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "msd_replay")
    out = subprocess.run([exe, "--device-type", "ifile", "--ifile", str(f), "--iformat", "uc8", "--fix", "--mlat",
                          "--raw", "--path", path], capture_output=True, text=True, check=True)
    want, _ = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0).replay(iq, cap=1 << 17)
    lines = out.stdout.split()
    assert len(lines) == len(want) > 0
    for line, m in zip(lines, want):
        assert line == "@%012X%s;" % (int(m["timestampMsg"]), bytes(m["msg"][: m["msgbits"] // 8]).hex())


@pytest.mark.parametrize("fmt,mode_ac", [("uc8", 1), ("uc8", 0), ("sc16q11", 1)])
def test_header_fields_from_the_emit_kernel(pkg, oracle, torch_cuda, fmt, mode_ac, resolve_stage):
    """SURVEY.md 8(f) rank 1, first stage: MSD_CFG_DECODE_FIELDS / msd_collect_fields deliver the header
    fields of every accepted message (from the emit kernel, or from the same code on the host when the
    batch was resolved there), equal to the oracle's restatement -- including the altitude a Mode A/C
    reply inherits from an earlier reply of its buffer."""
    from test_fields import FIELD_NAMES
    f, of = (pkg.FMT_UC8, oracle.FMT_UC8) if fmt == "uc8" else (pkg.FMT_SC16Q11, oracle.FMT_SC16Q11)
    n, batch = 21 * 131072 + 999, 8 * 131072
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=61, fmt=f, msgs_per_sec=4000, ac_per_sec=3000 * mode_ac,
                                                 n_aircraft=120), n)
    d = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(fmt=f, nfix_crc=1, mode_ac=mode_ac, max_batch_samples=batch, message_capacity=1 << 18,
                          decode_fields=True)
    msgs, fields = [], []
    bps = dem.bytes_per_sample
    for off in range(0, n, batch):
        m = min(batch, n - off)
        dem.launch_device(d.data_ptr() + off * bps, m, off + m >= n)
        mm, ff = dem.collect_fields()
        msgs.append(mm)
        fields.append(ff)
    msgs, fields = np.concatenate(msgs), np.concatenate(fields)
    want, wfields, wstats = oracle.Oracle(of, 58, 1, mode_ac).replay_fields(iq, cap=1 << 18)
    assert_same(msgs, dem.stats(), want, wstats)
    for name in FIELD_NAMES:
        assert np.array_equal(fields[name], wfields[name]), name
    assert (fields["altitude_baro_valid"] == 1).sum() > 50
    if mode_ac:
        ac = msgs["msgtype"] == 32
        assert ac.sum() > 200 and (fields["altitude_baro_valid"][ac] == 1).sum() > 20


def test_replay_cli_wire_formats(pkg, oracle, torch_cuda, tmp_path):
    """the replay tool's --net-raw (AVR) and --beast outputs equal the oracle's
    restatement of net_io.c:769-835,870-896 applied to the oracle's messages."""
    import os
    import subprocess
    n = 9 * 131072 + 777
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=1091, msgs_per_sec=3000, ac_per_sec=800), n)
    f = tmp_path / "capture.uc8"
    iq.tofile(f)
    exe = os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "msd_replay")
    base = [exe, "--ifile", str(f), "--iformat", "uc8", "--fix", "--modeac", "--mlat", "--batch-buffers", "4"]
    want, _ = oracle.Oracle(oracle.FMT_UC8, 58, 1, 1).replay(iq, cap=1 << 17)
    assert (want["msgtype"] == 32).sum() > 20
    raw = subprocess.run(base + ["--net-raw"], capture_output=True, check=True).stdout
    assert raw == b"".join(oracle.avr_line(m, True) for m in want)
    beast = subprocess.run(base + ["--beast"], capture_output=True, check=True).stdout
    assert beast == b"".join(oracle.beast_frame(m) for m in want)


@pytest.mark.parametrize("threads", ["1", "3", "16"])
def test_resolve_threads_and_membership_churn(pkg, oracle, torch_cuda, monkeypatch, threads, resolve_stage):
    """The speculative buffer-parallel resolve must equal the sequential one for any thread count,
    also when new aircraft appear in nearly every buffer (forces re-resolution and the serial tail;
    the GPU resolve gives such a batch to the host resolver after its pass limit)."""
    monkeypatch.setenv("MSD_RESOLVE_THREADS", threads)
    got, dem = run_case(pkg, oracle, torch_cuda, pkg.FMT_UC8, 40 * 131072 + 17, seed=77, nfix=1, n_aircraft=30000,
                        msgs_per_sec=4000)
    t = dem.timing()
    if resolve_stage == "gpu-resolve":
        assert t["resolve_passes"] > 1 or t["resolve_fallback"] >= 1, t
    else:
        assert t["resolve_passes"] == 0 and t["resolve_fallback"] == 0, t


def test_gpu_resolve_runs_and_converges(pkg, oracle, torch_cuda, resolve_stage):
    """A stable aircraft population: the first batch needs a second pass (everything is new), later
    batches one; no batch falls back to the host."""
    if resolve_stage != "gpu-resolve":
        pytest.skip("GPU resolve only")
    got, dem = run_case(pkg, oracle, torch_cuda, pkg.FMT_UC8, 64 * 131072, seed=11, nfix=1, batch=16 * 131072,
                        n_aircraft=300, msgs_per_sec=3000)
    t = dem.timing()
    assert t["resolve_passes"] >= 1 and t["resolve_fallback"] == 0, t


@pytest.mark.parametrize("seed,overlap,aircraft", [(21, 300, 2000), (22, 600, 150), (23, 50, 20000)])
def test_gpu_resolve_overlaps_and_new_aircraft(pkg, oracle, torch_cuda, resolve_stage, seed, overlap, aircraft):
    """Garbled/overlapping frames (hidden messages, wrong add predictions) together with aircraft that
    keep appearing: the acceptance chain, the prediction table and its corrections, several batches.
    The 20000-aircraft case also fills the ICAO filter's active table (icaoFilterAdd then gives up,
    icao_filter.c:82-86), which only the exact sequential replay of the host resolver reproduces."""
    got, dem = run_case(pkg, oracle, torch_cuda, pkg.FMT_UC8, 48 * 131072 + 1234, seed=seed, nfix=1, batch=16 * 131072,
                        n_aircraft=aircraft, msgs_per_sec=6000, overlap_permille=overlap, flip_permille=50)


@pytest.mark.parametrize("inline_adds", ["40", "-1"], ids=["forty-inline", "none-inline"])
def test_gpu_resolve_long_add_lists(pkg, oracle, torch_cuda, resolve_stage, monkeypatch, inline_adds):
    """More unique addresses in one buffer than a per-buffer report holds inline (232; 40 here, the
    synthetic traffic peaks near 150 -- or none at all: msd_config.test_inline_adds < 0): the complete add
    lists are fetched instead."""
    if resolve_stage != "gpu-resolve":
        pytest.skip("GPU resolve only")
    monkeypatch.setenv("MSD_RESOLVE_INLINE_ADDS", inline_adds)
    got, dem = run_case(pkg, oracle, torch_cuda, pkg.FMT_UC8, 8 * 131072, seed=12, nfix=0, n_aircraft=1000,
                        msgs_per_sec=8000, overlap_permille=0)
    assert dem.timing()["resolve_long_lists"] >= 1, dem.timing()


def test_interference_storm_overflows_and_is_rerun_in_pieces(pkg, oracle, torch_cuda, monkeypatch):
    """strong wideband interference makes a large share of all positions candidates.
    The candidate arenas then overflow; the batch must be rescanned in pieces, never truncated."""
    rng = np.random.default_rng(5)
    n = 128 * 131072
    on = rng.random(n) < 0.3
    i = np.where(on, 128 + 100 * rng.choice([-1, 1], n), 128 + rng.integers(-2, 3, n)).clip(0, 255).astype(np.uint8)
    q = (128 + rng.integers(-3, 4, n)).clip(0, 255).astype(np.uint8)
    iq = np.stack([i, q], 1).reshape(-1).copy()
    d = torch_cuda.from_numpy(iq).to("cuda:0")
    monkeypatch.setenv("MSD_ARENA_SCALE_PERMILLE", "200")  # arenas sized for a fifth of the default density
    dem = pkg.Demodulator(fmt=pkg.FMT_UC8, nfix_crc=1, max_batch_samples=n, message_capacity=1 << 16)
    got = dem.submit_device(d.data_ptr(), n, last=True)
    assert dem.timing()["reruns"] >= 1, dem.timing()
    want, wstats = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0).replay(iq, cap=1 << 16)
    assert_same(got, dem.stats(), want, wstats)
    assert wstats["demod_preambles"] > 0.05 * n


@pytest.mark.parametrize("drop", [1, 12345, 131072, 5 * 131072 + 17])
def test_dropped_samples_between_batches(pkg, oracle, torch_cuda, drop):
    """msd_note_dropped: the batch behind a gap starts with a MAGBUF_DISCONTINUOUS buffer (zero look-behind),
    the sample clock runs on over the gap, msd_stats.samples_dropped counts it (sdr_rtlsdr.c:281-300,
    fifo.c:176-184, readsb.c:836).  Two gaps, batches in flight together."""
    from helpers import oracle_live_feed
    C = pkg.CHUNK
    n = 13 * C + 999
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=400 + drop % 9, msgs_per_sec=7000, n_aircraft=25), n)
    cuts = [0, 4 * C, 9 * C, n]
    drops = [0, drop, 3 * drop + 1]
    segs = [iq[2 * a: 2 * b] for a, b in zip(cuts[:-1], cuts[1:])]
    want, wstats = oracle_live_feed(oracle.Oracle(oracle.FMT_UC8, 58, 1, 0), segs, drops)
    d_iq = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(nfix_crc=1, max_batch_samples=5 * C, message_capacity=1 << 16)
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        if drops[i]:
            dem.note_dropped(drops[i])
        dem.launch_device(d_iq.data_ptr() + 2 * a, b - a, last=b == n)
    got = np.concatenate([dem.collect() for _ in range(3)])
    assert len(want) > 100
    assert_same(got, dem.stats(), want, wstats)
    assert dem.stats()["samples_dropped"] == sum(drops)
    with pytest.raises(pkg.MsdError):
        dem.note_dropped(1)  # the capture is over


@pytest.mark.parametrize("drop", [(1 << 32) - 6 * 131072 - 100, (1 << 32) + 12345, (1 << 36) + 3, (1 << 43) + 131071])
def test_sample_clock_beyond_32_bits(pkg, oracle, torch_cuda, drop):
    """A live receiver passes 2^32 samples after half an hour.  The sample clock is a 64-bit counter turned into 12 MHz
    ticks by a double expression (sdr_ifile.c:187, sdr_rtlsdr.c: `sampleCounter * 12e6 / sample_rate`, inexact once the
    product leaves 53 bits): a gap of that size in front of the second batch (the first drop makes the clock cross 2^32
    inside the batch behind it), compared with the oracle fed the same way."""
    from helpers import oracle_live_feed
    C = pkg.CHUNK
    n = 14 * C + 777
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=77, msgs_per_sec=7000, n_aircraft=25), n)
    cuts = [0, 3 * C, 9 * C, n]
    drops = [0, drop, 5]
    segs = [iq[2 * a: 2 * b] for a, b in zip(cuts[:-1], cuts[1:])]
    want, wstats = oracle_live_feed(oracle.Oracle(oracle.FMT_UC8, 58, 1, 0), segs, drops)
    d_iq = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(nfix_crc=1, max_batch_samples=6 * C, message_capacity=1 << 16)
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        if drops[i]:
            dem.note_dropped(drops[i])
        dem.launch_device(d_iq.data_ptr() + 2 * a, b - a, last=b == n)
    got = np.concatenate([dem.collect() for _ in range(3)])
    assert len(want) > 100 and int(want["timestampMsg"].max()) > 5 * drop
    assert_same(got, dem.stats(), want, wstats)


def test_more_batches_than_prediction_generations(pkg, oracle, torch_cuda):
    """The prediction table of a pipeline slot is never wiped between batches: its entries carry one of 255 generations
    (msd_pred_impl.h).  1100 batches of two buffers -- 275 per slot -- take every slot's generation round once, with new
    aircraft in every batch (30 000 of them), so stale entries of the generation that comes back would be believed."""
    import os
    if os.environ.get("MSD_GPU_RESOLVE") == "0":
        pytest.skip("the host resolver keeps no prediction table")
    C = pkg.CHUNK
    n = 2200 * C + 4096
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=255, msgs_per_sec=3000, n_aircraft=30000, noise_fs=0.01), n)
    want, wstats = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0).replay(iq, cap=1 << 20)
    dem = pkg.Demodulator(nfix_crc=1, max_batch_samples=2 * C, message_capacity=1 << 20)
    got = pkg.replay_device(dem, torch_cuda.from_numpy(iq).to("cuda:0").data_ptr(), n, 2 * C)
    assert len(want) > 100000
    assert_same(got, dem.stats(), want, wstats)


def test_preamble_threshold_change_applies_to_later_batches(pkg, oracle, torch_cuda):
    """msd_set_preamble_threshold (demod_2400.c:285-290 raises the threshold to 75 while samples were
    dropped recently): batches launched before the call keep the old value."""
    from helpers import oracle_live_feed
    C = pkg.CHUNK
    n = 8 * C
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=31, msgs_per_sec=7000, n_aircraft=25), n)
    d_iq = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(nfix_crc=1, preamble_threshold=58, max_batch_samples=4 * C, message_capacity=1 << 16)
    dem.launch_device(d_iq.data_ptr(), 4 * C, last=False)
    dem.set_preamble_threshold(75)
    dem.launch_device(d_iq.data_ptr() + 8 * C, 4 * C, last=True)
    got = np.concatenate([dem.collect(), dem.collect()])
    lo = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0).replay(iq, cap=1 << 16)[0]
    hi = oracle.Oracle(oracle.FMT_UC8, 75, 1, 0).replay(iq, cap=1 << 16)[0]
    border = 4 * C * 5
    # away from the border (the skip-ahead / filter state of the first half can differ) the halves are
    # those of the two thresholds
    first = got[got["timestampMsg"] < border - 10000]
    assert np.array_equal(first, lo[: len(first)])
    assert len(lo) != len(hi)
    st = dem.stats()
    assert st["demod_preambles"] < oracle.Oracle(oracle.FMT_UC8, 58, 1, 0).replay(iq, cap=1 << 16)[1]["demod_preambles"]
    with pytest.raises(pkg.MsdError):
        dem.set_preamble_threshold(0)


def test_threshold_while_samples_were_dropped_recently(pkg, oracle, torch_cuda):
    """demod_2400.c:285-290: while Modes.stats_15min.samples_dropped is not zero, preambles are tested against
    max(PREAMBLE_THRESHOLD_PIZERO = 75, Modes.preambleThreshold).  The 15-minute window is the host program's; the boundary
    contract is that the host calls msd_set_preamble_threshold(max(75, threshold)) when the window fills and the configured
    value again when it empties.  The oracle and the second reading model the flag itself: a whole capture under each
    state, and the product's two settings must equal them -- for a configured threshold below 75 and for one above it
    (which the flag must not lower)."""
    import indep_demod
    C = pkg.CHUNK
    n = 6 * C + 999
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=77, msgs_per_sec=9000, n_aircraft=40), n)
    d_iq = torch_cuda.from_numpy(iq).to("cuda:0")
    for configured in (58, 90):
        per_state = []
        for dropped in (False, True):
            orc = oracle.Oracle(oracle.FMT_UC8, configured, 1, 0)
            orc.set_recently_dropped(dropped)
            want, wstats = orc.replay(iq, cap=1 << 16)
            second = indep_demod.Receiver("uc8", configured, 1, False)
            second.recently_dropped = dropped
            smsgs, sstats = second.replay(iq.tobytes())
            assert len(smsgs) == len(want) and sstats["demod_preambles"] == wstats["demod_preambles"]
            dem = pkg.Demodulator(nfix_crc=1, preamble_threshold=configured, max_batch_samples=4 * C, message_capacity=1 << 16)
            dem.set_preamble_threshold(max(75, configured) if dropped else configured)   # the host's side of the contract
            got = pkg.replay_device(dem, d_iq.data_ptr(), n, 4 * C)
            assert_same(got, dem.stats(), want, wstats)
            per_state.append(len(want))
        assert (per_state[0] != per_state[1]) == (configured < 75)


@pytest.mark.parametrize("fmt,mode_ac", [("uc8", 0), ("uc8", 1), ("sc16", 0)])
def test_aggressive_two_bit_correction(pkg, oracle, torch_cuda, fmt, mode_ac):
    """--aggressive (Modes.nfix_crc = 2, readsb.c:542): DF17/18 with two wrong bits are corrected against the
    (2, 4) tables of crc.c:374-379, DF11 never with more than one (mode_s.c:352-356); several batches, header
    fields from the corrected bytes."""
    from helpers import fmt_ids
    f, of = fmt_ids(pkg, oracle, fmt)
    n = 24 * 131072 + 4321
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=2024, fmt=f, msgs_per_sec=9000, n_aircraft=60), n)
    d_iq = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(fmt=f, nfix_crc=2, mode_ac=mode_ac, max_batch_samples=8 * 131072, message_capacity=1 << 17,
                          decode_fields=True)
    bps = 2 if fmt == "uc8" else 4
    parts, off, inflight = [], 0, 0
    while off < n:
        if inflight == 3:
            parts.append(dem.collect_fields())
            inflight -= 1
        m = min(8 * 131072, n - off)
        dem.launch_device(d_iq.data_ptr() + off * bps, m, last=off + m >= n)
        off += m
        inflight += 1
    for _ in range(inflight):
        parts.append(dem.collect_fields())
    got = np.concatenate([p[0] for p in parts])
    gfields = np.concatenate([p[1] for p in parts])
    want, wfields, wstats = oracle.Oracle(of, 58, 2, mode_ac).replay_fields(iq, cap=1 << 17)
    assert_same(got, dem.stats(), want, wstats)
    assert gfields.tobytes() == wfields.tobytes()
    assert wstats["demod_accepted"][2] > 20 and (want["correctedbits"] == 2).sum() == wstats["demod_accepted"][2]
    assert not ((want["correctedbits"] == 2) & (want["msgtype"] == 11)).any()


@pytest.mark.parametrize("fmt,mode_ac", [("uc8", 0), ("sc16", 0), ("sc16q11", 1)])
def test_dc_filter(pkg, oracle, torch_cuda, fmt, mode_ac):
    """--dcfilter (MSD_CFG_DC_FILTER): the converters with the 1 Hz DC block (convert.c:113-213,374-423).  Their
    filter state runs through the whole stream (two batches and a ragged end here), the level / power sums
    restart with every buffer; a receiver with a DC offset on both channels."""
    from helpers import fmt_ids
    f, of = fmt_ids(pkg, oracle, fmt)
    n = 6 * 131072 + 555
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=808, fmt=f, msgs_per_sec=6000, n_aircraft=15,
                                                 ac_per_sec=400 if mode_ac else 0), n)
    if fmt == "uc8":
        v = iq.reshape(-1, 2).astype(np.int32) + np.array([9, -6])
        iq = np.clip(v, 0, 255).astype(np.uint8).reshape(-1)
    else:
        full = 32768 if fmt == "sc16" else 2048
        v = iq.view("<i2").reshape(-1, 2).astype(np.int32) + np.array([full // 25, -full // 40])
        iq = np.clip(v, -full, full - 1).astype("<i2").reshape(-1).view(np.uint8)
    bps = 2 if fmt == "uc8" else 4
    d_iq = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(fmt=f, nfix_crc=1, mode_ac=mode_ac, max_batch_samples=4 * 131072, message_capacity=1 << 16,
                          dc_filter=True)
    dem.launch_device(d_iq.data_ptr(), 4 * 131072, last=False)
    dem.launch_device(d_iq.data_ptr() + 4 * 131072 * bps, n - 4 * 131072, last=True)
    got = dem.collect()
    means = [dem.buffer_means()]
    got = np.concatenate([got, dem.collect()])
    means.append(dem.buffer_means())
    want, wstats, wmeans = oracle.Oracle(of, 58, 1, mode_ac, dc_filter=True).replay(iq, cap=1 << 16, want_means=True)
    plain = oracle.Oracle(of, 58, 1, mode_ac).replay(iq, cap=1 << 16)[0]
    assert len(want) > 300 and want.tobytes() != plain.tobytes()  # the filter matters for this capture
    assert_same(got, dem.stats(), want, wstats)
    gm = np.concatenate(means)
    assert np.array_equal(gm, wmeans[: len(gm)], equal_nan=True) and len(gm) == wstats["buffers"]
    # a new capture starts from a zero filter state again
    dem.reset()
    again = dem.submit_device(d_iq.data_ptr(), 4 * 131072, last=True)
    w2 = oracle.Oracle(of, 58, 1, mode_ac, dc_filter=True).replay(iq[: 4 * 131072 * bps], cap=1 << 16)[0]
    assert again.tobytes() == w2.tobytes()


def test_replay_cli_aggressive_dcfilter(pkg, oracle, torch_cuda, tmp_path):
    """The option pair that completes the converter / CRC flags of the path (readsb.c:486,542) through the
    --ifile handler: msd_replay --aggressive --dcfilter equals the oracle with nfix 2 and the DC block."""
    import os
    import subprocess
    n = 9 * 131072 + 321
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=1092, msgs_per_sec=5000, flip_permille=200), n)
    iq = np.clip(iq.reshape(-1, 2).astype(np.int32) + np.array([7, -4]), 0, 255).astype(np.uint8).reshape(-1)
    f = tmp_path / "capture.uc8"
    iq.tofile(f)
    exe = os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "msd_replay")
    out = subprocess.run([exe, "--ifile", str(f), "--iformat", "uc8", "--aggressive", "--dcfilter", "--mlat", "--raw",
                          "--batch-buffers", "4"], capture_output=True, text=True, check=True)
    want, wstats = oracle.Oracle(oracle.FMT_UC8, 58, 2, 0, dc_filter=True).replay(iq, cap=1 << 16)
    assert wstats["demod_accepted"][2] > 0
    lines = out.stdout.split()
    assert len(lines) == len(want) > 100
    for line, m in zip(lines, want):
        assert line == "@%012X%s;" % (int(m["timestampMsg"]), bytes(m["msg"][: m["msgbits"] // 8]).hex())
    # the literal drop-in (init_converter(..., filter_dc = 1) + demodulate2400 on struct mag_buf): the same messages
    mb = subprocess.run([exe, "--ifile", str(f), "--iformat", "uc8", "--aggressive", "--dcfilter", "--mlat", "--raw", "--path", "magbuf"],
                        capture_output=True, text=True, check=True)
    assert mb.stdout.split() == lines


@pytest.mark.parametrize("pattern", ["full-scale", "alternating", "ramp"])
def test_level_and_power_sums_at_full_scale(pkg, oracle, torch_cuda, pattern):
    """The scan kernel sums the squared magnitudes of a buffer modulo 2^32 plus a bracketing sum of the
    truncated squares (msd_kernels.hip power_sum): saturated input makes that wrap hundreds of times per lane."""
    n = 3 * 131072 + 4097
    if pattern == "full-scale":
        iq = np.full(2 * n, 255, dtype=np.uint8)
    elif pattern == "alternating":
        iq = np.tile(np.array([255, 0, 128, 127, 0, 0, 255, 255], dtype=np.uint8), (2 * n + 7) // 8)[: 2 * n].copy()
    else:
        iq = (np.arange(2 * n, dtype=np.uint32) * 37 // 3).astype(np.uint8)
    d = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(nfix_crc=1, max_batch_samples=4 * 131072, message_capacity=1 << 12)
    got = dem.submit_device(d.data_ptr(), n, last=True)
    want, wstats, wmeans = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0).replay(iq, cap=1 << 12, want_means=True)
    assert_same(got, dem.stats(), want, wstats)
    gm = dem.buffer_means()
    assert np.array_equal(gm, wmeans[: len(gm)], equal_nan=True) and len(gm) == wstats["buffers"]
    if pattern == "full-scale":
        assert gm[0, 1] > 0.99  # mean power of a saturated buffer


@pytest.mark.parametrize("dc", [False, True])
def test_restart_behind_a_draining_capture(pkg, oracle, torch_cuda, dc):
    """msd_restart: the first batches of a new capture are launched while the last ones of the previous capture
    are still in flight; each capture must come out as if it had run alone (empty filter, zero clock, zero
    counters), and the old capture's counters stay readable until the new one's first batch is collected."""
    C = pkg.CHUNK
    caps = []
    for seed, n in ((501, 9 * C + 77), (502, 6 * C), (503, 5 * C + 1)):
        iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=seed, msgs_per_sec=7000, n_aircraft=40), n)
        caps.append((iq, n, torch_cuda.from_numpy(iq).to("cuda:0")))
    dem = pkg.Demodulator(nfix_crc=1, max_batch_samples=4 * C, message_capacity=1 << 16, dc_filter=dc)
    inflight, got, stats = [], {}, {}

    def collect_one():
        cid, is_last = inflight.pop(0)
        got.setdefault(cid, []).append(dem.collect())
        if is_last:
            stats[cid] = dem.stats()

    for cid, (iq, n, d) in enumerate(caps):
        dem.restart()
        off = 0
        while off < n:
            if len(inflight) == pkg.capi.PIPELINE_DEPTH:
                collect_one()
            m = min(4 * C, n - off)
            dem.launch_device(d.data_ptr() + 2 * off, m, last=off + m >= n)
            inflight.append((cid, off + m >= n))
            off += m
    while inflight:
        collect_one()
    for cid, (iq, n, d) in enumerate(caps):
        want, wstats = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0, dc_filter=dc).replay(iq, cap=1 << 16)
        assert len(want) > 100
        assert_same(np.concatenate(got[cid]), stats[cid], want, wstats)
    with pytest.raises(pkg.MsdError):
        d2 = caps[0][2]
        dem.restart()
        dem.launch_device(d2.data_ptr(), 4 * C, last=False)
        dem.restart()  # the running capture has not been closed


def test_field_decoder_on_the_device_matches_the_oracle(pkg, oracle, torch_cuda):
    """msd_decode_fields_device runs the emit kernel's decoder (msd_fields_impl.h compiled for gfx950) on crafted
    messages the synthetic captures never contain: every Comm-B register with plausible and borderline content
    (comm_b.c -- the coordinated-turn check goes through the device's double-precision tan), target state and
    operational status squitters of every version, and random payloads of every downlink format."""
    from test_fields import FIELD_NAMES, es_record, make_commb
    rng = np.random.default_rng(6061)
    recs = []
    for k in range(20000):
        raw = bytearray(rng.integers(0, 256, 14, dtype=np.uint8).tobytes())
        sel = k % 4
        if sel == 0:  # Comm-B
            kind = (10, 17, 20, 30, 40, 50, 60, 50)[(k // 4) % 8]
            raw[0] = ((20 + (k // 32) % 2) << 3) | (raw[0] & 7)
            raw[1] &= 0x07 if k % 64 else 0xFF  # DR = 0 for almost all
            raw[4:11] = make_commb(rng, kind)
        elif sel == 1:  # ES 29 / 31
            raw[0] = ((17 + (k // 4) % 2) << 3) | (raw[0] & 7)
            raw[4] = ((29 if (k // 8) % 2 else 31) << 3) | int(rng.integers(0, 2))
            raw[9] = (raw[9] & 0x1F) | (int(rng.integers(0, 4)) << 5)
        elif sel == 2:  # any ES type
            raw[0] = (17 << 3) | 5
        else:  # any downlink format the demodulator accepts
            raw[0] = (int(rng.choice([0, 4, 5, 11, 16, 17, 18, 20, 21, 24])) << 3) | (raw[0] & 7)
        n = 14 if raw[0] & 0x80 else 7
        recs.append(es_record(pkg, bytes(raw[:n]).hex()))
    msgs = np.array(recs, dtype=pkg.capi.MESSAGE_DTYPE)
    dem = pkg.Demodulator(nfix_crc=1, max_batch_samples=131072)
    got = dem.decode_fields_device(msgs)
    seen = np.zeros(10, dtype=int)
    for m, g in zip(msgs, got):
        want = oracle.fields_of(m)
        for f in FIELD_NAMES:
            assert g[f] == want[f], (f, m["msg"].tobytes().hex(), int(m["msgtype"]), g[f], want[f])
        seen[want["commb_format"]] += 1
    assert (seen[3:] > 30).all(), seen
    # and byte for byte against the host build of the same header
    host = np.array([pkg.capi.decode_fields(m) for m in msgs[:3000]])
    assert host.tobytes() == got[:3000].tobytes()


def test_more_clean_squitter_addresses_than_the_prediction_table_holds(pkg, oracle, torch_cuda, resolve_stage):
    """One batch with far more distinct CRC-clean DF17 / DF11 addresses than the scan kernel's prediction table has
    slots (65536; its list 32768): the probes must end (msd_pred_impl.h bounds them and stops inserting once the list
    is full), the batch goes to the host resolver, and the messages are the oracle's."""
    n = 800 * 131072
    got, dem = run_case(pkg, oracle, torch_cuda, pkg.FMT_UC8, n, seed=4242, nfix=1, n_aircraft=1000000, msgs_per_sec=8000,
                        overlap_permille=0)
    # the capture really holds that many: every accepted DF11 / DF17 carries its address in the clear
    squitters = got[(got["msgtype"] == 17) | (got["msgtype"] == 11)]
    assert len(np.unique(squitters["addr"])) > 66000, len(np.unique(squitters["addr"]))
    if resolve_stage == "gpu-resolve":
        assert dem.timing()["resolve_fallback"] >= 1


@pytest.mark.parametrize("fmt,noise_fs", [("sc16", 0.002), ("sc16", 0.3), ("sc16q11", 0.02)])
def test_float_sums_of_a_full_batch_that_ends_on_a_buffer_boundary(pkg, oracle, torch_cuda, fmt, noise_fs):
    """convert.c:241-252 at the benchmark's batch size: 512 full buffers of 16-bit IQ in one batch, which the reader counts
    as 513 (sdr_ifile.c:192-216: the end of the file is only noticed by a short read), the last one empty.  The float sums
    of every buffer -- sequential in the reference, evaluated block-parallel by msd_fm_functions_kernel /
    msd_fm_apply_kernel -- must equal the oracle's bit for bit, and the empty buffer must not be read (round 4: the apply
    kernel fetched its first sample, one word past the capture; only a capture that ends on a page boundary faults).
    A quiet band puts the power sum below 16, where every block holds ties; a loud one drives both sums through
    seventeen binades."""
    f, of = (pkg.FMT_SC16, oracle.FMT_SC16) if fmt == "sc16" else (pkg.FMT_SC16Q11, oracle.FMT_SC16Q11)
    n = 512 * 131072
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=20260, fmt=f, msgs_per_sec=3000, noise_fs=noise_fs), n)
    d = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(fmt=f, nfix_crc=0, max_batch_samples=n, message_capacity=1 << 18)
    got = dem.submit_device(d.data_ptr(), n, last=True)
    want, wstats, wmeans = oracle.Oracle(of, 58, 0, 0).replay(iq, cap=1 << 18, want_means=True)
    assert wstats["buffers"] == 513
    assert_same(got, dem.stats(), want, wstats)
    gm = dem.buffer_means()
    assert len(gm) == 513 and np.array_equal(gm, wmeans[:513], equal_nan=True)
    assert np.isfinite(gm[:512]).all() and gm[:512, 0].min() > 0


@pytest.mark.gpu
def test_a_probation_confirmed_by_a_message_the_round_then_dropped(pkg, oracle, torch_cuda):
    """The fuzzer's case 702780 (round 6), the first four buffers of it: a new aircraft's clean squitter confirms its entry
    of the resolve kernel's probation table (later tries of its address are staged as known while the message that will add
    it is still to be accepted), the same round's probation check then moves the cut in front of that message, and in the
    re-evaluation another message hides it: it is never accepted, the address never added (mode_s.c:717-726) -- and a
    corrected DF17 of that address 0.1 s later must score 700, not 900 (mode_s.c:376-381).  The confirmation of a message
    behind the final cut is void (msd_resolve_kernels.hip)."""
    n = 4 * 131072
    kw = dict(msgs_per_sec=12000, n_aircraft=800, overlap_permille=0, flip_permille=20, noise_fs=0.005, ac_per_sec=500)
    got, dem = run_case(pkg, oracle, torch_cuda, pkg.FMT_UC8, n, 702780, 1, batch=4 * 131072, **kw)   # compares with the oracle
    i = int(np.flatnonzero(got["addr"] == 0xF0F741)[0])
    assert len(got) > 600 and got["score"][i] == 700 and got["correctedbits"][i] == 1 and got["msgtype"][i] == 17
