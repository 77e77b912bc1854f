"""A capture built to hurt: a periodic pulse train (found by random search over short periodic UC8 patterns, the
preamble tests of the second reading as the objective) in which 36 % of all positions pass the preamble tests and every
one of them asks for all five trial phases -- 24 x the hit density of the benchmark capture, far beyond what the candidate
arenas are sized for.  The batch must be scanned again -- into bigger region slices on the GPU resolve path, in pieces
through the host resolver where that is not possible -- never truncated, and the result must still be the oracle's, message
for message and counter for counter."""
import numpy as np
import pytest

from helpers import fmt_ids
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu

PATTERN = np.array([[127, 128], [127, 128], [127, 128], [127, 128], [239, 111], [132, 12], [110, 164], [168, 108],
                    [127, 128], [127, 128], [127, 128]], dtype=np.uint8)


def pulse_train(n, noise_seed=None, frames=None):
    iq = np.tile(PATTERN, (n // len(PATTERN) + 1, 1))[:n].copy()
    if noise_seed is not None:  # a little receiver noise on top: the hits stay, the sliced bits vary
        rng = np.random.default_rng(noise_seed)
        iq = np.clip(iq.astype(np.int16) + rng.integers(-3, 4, size=iq.shape), 0, 255).astype(np.uint8)
    return iq.reshape(-1)


@pytest.fixture(params=["gpu-resolve", "host-resolve"])
def resolve_stage(request, monkeypatch):
    monkeypatch.setenv("MSD_GPU_RESOLVE", "1" if request.param == "gpu-resolve" else "0")
    return request.param


@pytest.mark.parametrize("arenas", ["base-size", "base-size-no-growth", "default"])
@pytest.mark.parametrize("noise_seed", [None, 7])
@pytest.mark.parametrize("nfix,mode_ac", [(0, 0), (1, 1)])
def test_pulse_train_of_preambles(pkg, oracle, torch_cuda, resolve_stage, monkeypatch, noise_seed, nfix, mode_ac, arenas):
    """base-size: arenas of one hit per 8 and one live try per 16 samples (msd_config.test_arena_permille = 1000) -- they
    overflow: the slot's slices grow and the batch is scanned again (no-growth, MSD_CFG_NO_ARENA_GROWTH: rescanned in
    pieces and resolved on the host, the path a device without spare memory takes); default (four times that): these
    short batches fit and stay on the GPU"""
    if arenas.startswith("base-size"):
        monkeypatch.setenv("MSD_ARENA_SCALE_PERMILLE", "1000")
    if arenas == "base-size-no-growth":
        monkeypatch.setenv("MSD_ARENA_GROWTH", "0")
    n = 12 * 131072 + 4321
    iq = pulse_train(n, noise_seed)
    f, of = fmt_ids(pkg, oracle, "uc8")
    dem = pkg.Demodulator(fmt=f, preamble_threshold=58, nfix_crc=nfix, mode_ac=mode_ac, max_batch_samples=8 * 131072,
                          message_capacity=1 << 19)
    got = pkg.replay_device(dem, torch_cuda.from_numpy(iq).to("cuda:0").data_ptr(), n, 8 * 131072)
    want, wstats = oracle.Oracle(of, 58, nfix, mode_ac).replay(iq, cap=1 << 19)
    assert wstats["demod_preambles"] > 0.3 * n  # it is as dense as advertised
    assert_same(got, dem.stats(), want, wstats)
    t = dem.timing()
    if arenas.startswith("base-size"):
        assert t["reruns"] > 0 or t["resolve_fallback"] > 0  # the arenas did overflow; nothing was cut short
        if arenas == "base-size-no-growth" and resolve_stage == "gpu-resolve":
            assert t["resolve_fallback"] > 0, t
        if arenas == "base-size" and resolve_stage == "gpu-resolve" and not mode_ac:
            # ... and the batch stayed on the GPU: its slot got region slices the densest region fits, one more scan, the
            # GPU resolve again (grow_and_rescan) -- no batch went through the host resolver
            assert t["reruns"] > 0 and t["resolve_fallback"] == 0, t


@pytest.mark.parametrize("arenas", ["base-size", "default"])
def test_pulse_train_inside_ordinary_traffic(pkg, oracle, torch_cuda, resolve_stage, monkeypatch, arenas):
    """three buffers of the pulse train in the middle of an ordinary capture: the batches around them are untouched"""
    if arenas == "base-size":
        monkeypatch.setenv("MSD_ARENA_SCALE_PERMILLE", "1000")
    n = 24 * 131072
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=515), n).copy()
    a, b = 9 * 131072 + 1000, 12 * 131072 + 500
    iq[2 * a: 2 * b] = pulse_train(b - a, 3)
    f, of = fmt_ids(pkg, oracle, "uc8")
    dem = pkg.Demodulator(fmt=f, preamble_threshold=58, nfix_crc=1, mode_ac=0, max_batch_samples=8 * 131072, message_capacity=1 << 19)
    got = pkg.replay_device(dem, torch_cuda.from_numpy(iq).to("cuda:0").data_ptr(), n, 8 * 131072)
    want, wstats = oracle.Oracle(of, 58, 1, 0).replay(iq, cap=1 << 19)
    assert_same(got, dem.stats(), want, wstats)
