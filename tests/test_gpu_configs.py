"""BASELINE.json configurations that the other parity tests do not already replay at their own seeds,
and a bounded slice of the randomised sweep (tests/fuzz_parity.py) so that every driver run repeats it."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", range(10901, 10909))
def test_configs3_the_eight_captures_one_after_the_other(pkg, oracle, torch_cuda, seed):
    """configs[3]: eight independent UC8 captures, seeds 10901..10908, here one after the
    other on cuda:0 -- 40 buffers of each through the pipelined path with its own context, as every rank does
    with its capture -- against the oracle, message for message and counter for counter."""
    n = 40 * pkg.CHUNK + 777
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=seed), n)
    d = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(fmt=pkg.FMT_UC8, nfix_crc=0, max_batch_samples=16 * pkg.CHUNK, message_capacity=1 << 18)
    got = pkg.replay_device(dem, d.data_ptr(), n, 16 * pkg.CHUNK)
    want, wstats = oracle.Oracle(oracle.FMT_UC8, 58, 0, 0).replay(iq, cap=1 << 18)
    assert len(want) > 1000
    assert_same(got, dem.stats(), want, wstats)


def test_bounded_slice_of_the_randomised_sweep():
    """24 cases of tests/fuzz_parity.py from a fixed first seed (format, length, batch size, traffic model,
    --fix level, Mode A/C, fields, resolve path drawn per case); the tool exits non-zero on any difference."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "24", "5000"], capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "failures: 0" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("env", [{"MSD_CHAIN_INLINE": "0"}, {"MSD_CHAIN_INLINE": "1"}, {"MSD_EMIT_FUSED": "0"},
                                 {"MSD_NO_HELPER": "1"}, {"MSD_LEAN": "0"}, {"MSD_RESOLVE_AHEAD": "0"},
                                 {"MSD_POWER_FUSED": "0", "MSD_CHAIN_INLINE": "1"}, {"MSD_POWER_FUSED": "1", "MSD_CHAIN_INLINE": "0"},
                                 {"MSD_WAIT_INPUTS_ON_STREAM": "1"}, {"MSD_LEAN": "0", "MSD_RESOLVE_AHEAD": "0", "MSD_POWER_FUSED": "0"}],
                         ids=["side-streams", "in-order", "record-kernel", "no-helper", "gather-kernel", "no-resolve-ahead",
                              "power-kernel-in-order", "power-in-resolve-side-streams", "stream-waits-for-inputs", "round-2-layout"])
@pytest.mark.parametrize("mode", ["uc8", "uc8-ac-fix", "sc16"])
def test_every_stream_layout_gives_the_same_messages(pkg, oracle, torch_cuda, monkeypatch, env, mode):
    """The stream layouts of DESIGN.md 4.6 (chain in order with the records written by the next scan, chain on
    side streams, a record kernel of its own, no helper thread; with and without the gather kernel, the signal power
    kernel, the resolve passes one batch ahead of the delivery) are scheduling only: each of them, forced through
    its environment switch, must deliver the oracle's messages and counters -- two captures back to back on one
    context (msd_restart), so that the early start across a capture boundary is on the path too."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    fmt, ofmt, fix, ac = {"uc8": (pkg.FMT_UC8, oracle.FMT_UC8, 0, 0), "uc8-ac-fix": (pkg.FMT_UC8, oracle.FMT_UC8, 1, 1),
                          "sc16": (pkg.FMT_SC16, oracle.FMT_SC16, 1, 0)}[mode]
    n, batch = 36 * pkg.CHUNK + 4321, 8 * pkg.CHUNK
    dem = pkg.Demodulator(fmt=fmt, nfix_crc=fix, mode_ac=ac, max_batch_samples=batch, message_capacity=1 << 18)
    bps = dem.bytes_per_sample
    caps, dev = [], []
    for seed in (4711, 4712):
        iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=seed, fmt=fmt, ac_per_sec=600 if ac else 0), n)
        caps.append(iq)
        dev.append(torch_cuda.from_numpy(iq).to("cuda:0"))
    got, stats, inflight = {0: [], 1: []}, {}, []

    def collect_one():
        cap, last = inflight.pop(0)
        got[cap].append(dem.collect())
        if last:
            stats[cap] = dem.stats() # the capture's own counters: the next capture's first batch is still out

    for cap in (0, 1):
        if cap == 1:
            dem.restart() # the first capture's last batches are still in flight
        off = 0
        while off < n:
            if len(inflight) == pkg.capi.PIPELINE_DEPTH:
                collect_one()
            m = min(batch, n - off)
            dem.launch_device(dev[cap].data_ptr() + off * bps, m, off + m >= n)
            inflight.append((cap, off + m >= n))
            off += m
    while inflight:
        collect_one()
    for cap in (0, 1):
        want, wstats = oracle.Oracle(ofmt, 58, fix, ac).replay(caps[cap], cap=1 << 18)
        assert len(want) > 500
        assert_same(np.concatenate(got[cap]), stats[cap], want, wstats)


@pytest.mark.parametrize("growth", ["bigger-slices-and-one-more-scan", "in-pieces-through-the-host"])
def test_overflowing_batch_in_the_middle_of_a_pipelined_stream(pkg, oracle, torch_cuda, monkeypatch, growth):
    """Four batches in flight, the third one an interference storm that overflows its candidate arenas (lean layout:
    only its first resolve pass tells, possibly a msd_collect early -- the resolve passes run one batch ahead).  Round 5:
    its slot gets region slices the densest region fits, the batch is scanned once more and stays on the GPU resolve
    (grow_and_rescan; the other slots follow before they meet the same traffic).  With MSD_CFG_NO_ARENA_GROWTH -- or on a
    device without room for the slices -- it is rescanned in pieces and resolved on the host as before; the batches
    around it stay on the GPU, and what the pieces' scans note must not leak into the slot's next prediction table."""
    if growth == "in-pieces-through-the-host":
        monkeypatch.setenv("MSD_ARENA_GROWTH", "0")
    C, nb = pkg.CHUNK, 96
    n = 5 * nb * C + 4321
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=777, msgs_per_sec=5000, n_aircraft=400), n).copy()
    rng = np.random.default_rng(6)
    lo, hi = 2 * nb * C, 3 * nb * C
    on = rng.random(hi - lo) < 0.3
    iq[2 * lo:2 * hi:2] = np.where(on, 128 + 100 * rng.choice([-1, 1], hi - lo), 128 + rng.integers(-2, 3, hi - lo)).clip(0, 255).astype(np.uint8)
    iq[2 * lo + 1:2 * hi:2] = (128 + rng.integers(-3, 4, hi - lo)).clip(0, 255).astype(np.uint8)
    monkeypatch.setenv("MSD_ARENA_SCALE_PERMILLE", "200")
    d = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(fmt=pkg.FMT_UC8, nfix_crc=1, max_batch_samples=nb * C, message_capacity=1 << 18)
    got = pkg.replay_device(dem, d.data_ptr(), n, nb * C)
    t = dem.timing()
    assert t["reruns"] >= 1 and (t["resolve_fallback"] >= 1) == (growth == "in-pieces-through-the-host"), t
    want, wstats = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0).replay(iq, cap=1 << 18)
    assert len(want) > 1000
    assert_same(got, dem.stats(), want, wstats)


def test_mode_ac_replies_of_a_batch_rescanned_in_pieces(pkg, oracle, torch_cuda, monkeypatch):
    """Found by the fuzzer once it drew arena sizes (case 500110): a GPU-resolve context whose batch overflows its slices and
    -- growth off, or no memory for it -- is rescanned in pieces.  The pieces' Mode A/C lists are stitched on the host; the
    fallback then copied the device's list (the last piece's, with piece-relative positions) over them: a reply of the second
    piece came out six buffers early.  Ordinary traffic with Mode A/C replies in every piece."""
    monkeypatch.setenv("MSD_ARENA_SCALE_PERMILLE", "50")
    monkeypatch.setenv("MSD_ARENA_GROWTH", "0")
    n = 11 * pkg.CHUNK
    cfg = pkg.siggen.make_cfg(seed=500110, msgs_per_sec=12000, n_aircraft=800, overlap_permille=10, flip_permille=200, noise_fs=0.02, ac_per_sec=4000)
    iq = pkg.siggen.generate(cfg, n)
    d = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(fmt=pkg.FMT_UC8, preamble_threshold=40, nfix_crc=1, mode_ac=1, max_batch_samples=64 * pkg.CHUNK, message_capacity=1 << 18)
    got = pkg.replay_device(dem, d.data_ptr(), n, 64 * pkg.CHUNK)
    t = dem.timing()
    assert t["reruns"] >= 1 and t["resolve_fallback"] >= 1, t
    want, wstats = oracle.Oracle(oracle.FMT_UC8, 40, 1, 1).replay(iq, cap=1 << 18)
    ac = want[want["msgtype"] == 32]
    assert len(ac) >= 5 and (ac["timestampMsg"] >= 6 * pkg.CHUNK * 5).any()   # replies in the second piece too
    assert_same(got, dem.stats(), want, wstats)
