"""HIP path vs oracle on captures that do not come from csrc/msd_siggen.c (tests/indep_signal.py: numpy PPM frames
with their own parity, Mode A/C replies on the 1.45 us grid, carrier offset, DC offset, clipping, an echo, dense noisy
traffic, uniformly random bytes), 64 buffers each.  Two independent questions per capture: (1) the GPU's ordered message
list and counters equal the oracle's; (2) the frames that were transmitted strongly, alone and unflipped with their
address in the clear (DF17 / DF11) come out of the GPU with the transmitted bytes and the transmitted time -- a
known-answer check at system level that does not go through the oracle at all."""
import numpy as np
import pytest

import indep_signal as S
from helpers import fmt_ids
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu

N = 64 * 131072 + 7777
_captures = {}


def cached_capture(key, *args, **kw):
    if key not in _captures:
        _captures[key] = S.capture(*args, **kw)
    return _captures[key]


@pytest.fixture(params=["gpu-resolve", "host-resolve"])
def resolve_stage(request, monkeypatch):
    """the ordered resolve stage on the GPU (default) and on host threads"""
    monkeypatch.setenv("MSD_GPU_RESOLVE", "1" if request.param == "gpu-resolve" else "0")
    return request.param

CASES = {
    "plain": dict(),
    "carrier_plus_minus_50kHz": dict(freq_offset_hz=50e3),
    "dc_offset": dict(dc=(0.06, -0.04)),
    "clipped": dict(clip_gain=1.6),
    "echo_750ns": dict(echo=(9, 0.35, 1.0)),
    "dense_noisy": dict(frames_per_sec=12000.0, noise=0.06, n_aircraft=400),
    "mode_ac": dict(ac_per_sec=3000.0),
    "random_bytes": dict(random_bytes=True),
}


def run(pkg, oracle, torch, fmt, iq, n, nfix, mode_ac, threshold=58, batch=16 * 131072):
    f, of = fmt_ids(pkg, oracle, fmt)
    d = torch.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(fmt=f, preamble_threshold=threshold, nfix_crc=nfix, mode_ac=mode_ac, max_batch_samples=batch,
                          message_capacity=1 << 19)
    got = pkg.replay_device(dem, d.data_ptr(), n, batch)
    want, wstats = oracle.Oracle(of, threshold, nfix, mode_ac).replay(iq, cap=1 << 19)
    assert_same(got, dem.stats(), want, wstats)
    return got


def check_transmitted(got, frames, min_fraction):
    strong = [f for f in S.isolated_strong_frames(frames) if f["df"] in (17, 11)]
    assert len(strong) > 50
    by_bytes = {}
    for m in got:
        by_bytes.setdefault(bytes(m["msg"][: m["msgbits"] // 8]), []).append(int(m["timestampMsg"]))
    found = 0
    for f in strong:
        ts = by_bytes.get(S.frame_bytes(f["bits"]))
        if not ts:
            continue
        # timestampMsg (demod_2400.c:358): 12 MHz ticks, (8 + 56) us behind the start of the preamble whatever the
        # frame's length, late by the 326-sample overlap, to within the phase resolution
        expect = f["tick"] + 326 * 5 + (8 + 56) * 12
        if min(abs(t - expect) for t in ts) <= 4:
            found += 1
    assert found >= min_fraction * len(strong), (found, len(strong))


@pytest.mark.parametrize("case", list(CASES))
def test_uc8_corpus(pkg, oracle, torch_cuda, case, resolve_stage):
    kw = CASES[case]
    iq, frames = cached_capture(case, 1000 + len(case), N, fmt="uc8", **kw)
    mode_ac = 1 if case == "mode_ac" else 0
    got = run(pkg, oracle, torch_cuda, "uc8", iq, N, nfix=1, mode_ac=mode_ac)
    if case == "random_bytes":
        return
    check_transmitted(got, frames, 0.85 if case == "dense_noisy" else 0.97)
    if mode_ac:
        sent = sum(1 for f in frames if f["kind"] == "AC")
        assert (got["msgtype"] == 32).sum() > 0.2 * sent   # (replies alone: 68 %; here they share the air with 1500 Mode S frames per second)


@pytest.mark.parametrize("fmt,mode_ac", [("sc16", 0), ("sc16q11", 1)])
def test_16bit_corpus(pkg, oracle, torch_cuda, fmt, mode_ac):
    iq, frames = S.capture(2000 + mode_ac, N, fmt=fmt, ac_per_sec=2000.0 * mode_ac, freq_offset_hz=20e3)
    got = run(pkg, oracle, torch_cuda, fmt, iq, N, nfix=1, mode_ac=mode_ac)
    check_transmitted(got, frames, 0.97)


@pytest.mark.parametrize("threshold", [40, 75, 400])
def test_thresholds_on_dense_traffic(pkg, oracle, torch_cuda, threshold):
    """--preamble-threshold over its whole range (readsb.c:503-505: 40..400) on dense, noisy traffic: at 40 one position
    in thirteen tries a phase, at 400 almost none does."""
    n = 24 * 131072
    iq, frames = S.capture(3000 + threshold, n, fmt="uc8", frames_per_sec=12000.0, noise=0.06, n_aircraft=400)
    run(pkg, oracle, torch_cuda, "uc8", iq, n, nfix=1, mode_ac=0, threshold=threshold, batch=8 * 131072)
