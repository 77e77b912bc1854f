"""The HIP path against the second reading of the reference (tests/indep_demod.py: numpy / plain Python, written from
the reference's sources without looking at oracle/modes_oracle.c) -- directly, not through the oracle: ordered message
list (timestamps, bytes, address, CRC, score, phase, corrected bits, signal level as the same double) and the demodulator
counters, Mode S and Mode A/C, three sample formats, both resolve stages."""
import numpy as np
import pytest

import indep_demod as D
import indep_signal as S
from helpers import fmt_ids
from test_indep_demod import CORPUS, assert_second_reading_agrees

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["gpu-resolve", "host-resolve"])
def resolve_stage(request, monkeypatch):
    monkeypatch.setenv("MSD_GPU_RESOLVE", "1" if request.param == "gpu-resolve" else "0")
    return request.param


def hip_and_second_reading(pkg, oracle, torch, fmt, iq, threshold=58, nfix=1, mode_ac=True, batch=4 * 131072, dc_filter=False):
    raw = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
    n = raw.size // (2 if fmt == "uc8" else 4)
    f, _ = fmt_ids(pkg, oracle, fmt)
    dem = pkg.Demodulator(fmt=f, preamble_threshold=threshold, nfix_crc=nfix, mode_ac=1 if mode_ac else 0,
                          max_batch_samples=batch, message_capacity=1 << 17, dc_filter=dc_filter)
    got = pkg.replay_device(dem, torch.from_numpy(raw).to("cuda:0").data_ptr(), n, batch)
    want, wstats = D.Receiver(fmt, threshold, nfix, mode_ac, dc_filter=dc_filter).replay(raw.tobytes())
    assert_second_reading_agrees(want, wstats, got, dem.stats())
    return got


@pytest.mark.parametrize("case", sorted(CORPUS))
@pytest.mark.parametrize("fmt", ["uc8", "sc16", "sc16q11"])
def test_numpy_corpus(pkg, oracle, torch_cuda, resolve_stage, fmt, case):
    iq, _ = S.capture(201 + sorted(CORPUS).index(case), 9 * 131072 + 555, fmt=fmt, **CORPUS[case])
    hip_and_second_reading(pkg, oracle, torch_cuda, fmt, iq)


@pytest.mark.parametrize("threshold,nfix", [(40, 1), (58, 0), (75, 1), (400, 0), (58, 2), (40, 2)])
def test_thresholds_no_fix_and_aggressive(pkg, oracle, torch_cuda, resolve_stage, threshold, nfix):
    iq, _ = S.capture(34, 6 * 131072 + 77, fmt="uc8", frames_per_sec=4000.0, noise=0.04, flip_fraction=0.2)
    hip_and_second_reading(pkg, oracle, torch_cuda, "uc8", iq, threshold=threshold, nfix=nfix, mode_ac=False)


@pytest.mark.parametrize("fmt", ["UC8", "SC16", "SC16Q11"])
def test_generator_of_the_benchmark(pkg, oracle, torch_cuda, resolve_stage, fmt):
    """the benchmark's content model and seed: 24 buffers in three batches"""
    cfg = pkg.siggen.make_cfg(seed=10901, fmt=getattr(pkg.siggen, fmt), ac_per_sec=1500)
    iq = pkg.siggen.generate(cfg, 24 * 131072 + 4096)
    got = hip_and_second_reading(pkg, oracle, torch_cuda, fmt.lower(), iq, batch=8 * 131072)
    assert len(got) > 1500


def test_capture_that_ends_on_a_buffer_boundary(pkg, oracle, torch_cuda, resolve_stage):
    iq, _ = S.capture(6, 4 * 131072, fmt="uc8", ac_per_sec=2000.0)
    hip_and_second_reading(pkg, oracle, torch_cuda, "uc8", iq)


@pytest.mark.parametrize("fmt", ["uc8", "sc16", "sc16q11"])
def test_dc_blocking_converters(pkg, oracle, torch_cuda, resolve_stage, fmt):
    """--dcfilter: the filter state runs through the stream, across buffers and batches"""
    iq, _ = S.capture(78, 3 * 131072 + 3000, fmt=fmt, dc=(0.05, -0.03), ac_per_sec=1500.0)
    hip_and_second_reading(pkg, oracle, torch_cuda, fmt, iq, batch=2 * 131072, dc_filter=True)


@pytest.mark.parametrize("fmt,mode_ac", [("uc8", False), ("sc16", True)])
@pytest.mark.parametrize("drop", [1, 131072 + 17, (1 << 32) + 12345])
def test_live_feed_with_dropped_samples(pkg, oracle, torch_cuda, resolve_stage, fmt, mode_ac, drop):
    """msd_note_dropped between batches in flight: the HIP path against the second reading's live feed"""
    C, bps = 131072, 2 if fmt == "uc8" else 4
    n = 9 * C + 999
    f, _ = fmt_ids(pkg, oracle, fmt)
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=401, fmt=f, msgs_per_sec=7000, n_aircraft=25, ac_per_sec=1000), n)
    cuts, drops = [0, 2 * C, 6 * C, n], [0, drop, 3 * drop + 1]
    want, wstats = D.Receiver(fmt, 58, 1, mode_ac).live_feed([iq[bps * a: bps * b].tobytes() for a, b in zip(cuts[:-1], cuts[1:])], drops)
    d_iq = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(fmt=f, nfix_crc=1, mode_ac=int(mode_ac), max_batch_samples=4 * C, message_capacity=1 << 16)
    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        if drops[i]:
            dem.note_dropped(drops[i])
        dem.launch_device(d_iq.data_ptr() + bps * a, b - a, last=b == n)
    got = np.concatenate([dem.collect() for _ in range(3)])
    assert len(want) > 100
    assert_second_reading_agrees(want, wstats, got, dem.stats())
