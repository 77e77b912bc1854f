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
