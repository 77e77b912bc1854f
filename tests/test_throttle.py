"""--throttle (sdr_ifile.c:168-169,218-226): the ifile handler releases a buffer when the one before it has
"played" at 2.4 MSPS.  The pacer is host C and is timed here without a GPU; the throttled replay itself needs one."""
import ctypes as C
import os
import subprocess
import time

import numpy as np
import pytest


class Timespec(C.Structure):
    _fields_ = [("tv_sec", C.c_long), ("tv_nsec", C.c_long)]


class Pacer(C.Structure):
    _fields_ = [("next", Timespec), ("sample_rate", C.c_double)]


def host_lib(pkg):
    L = C.CDLL(os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "libmsd_host.so"))
    L.msd_pacer_start.argtypes = [C.POINTER(Pacer), C.c_double]
    L.msd_pacer_wait.argtypes = [C.POINTER(Pacer), C.c_uint64]
    L.msd_pacer_start.restype = L.msd_pacer_wait.restype = None
    return L


def test_pacer_releases_buffers_at_the_sample_rate(pkg):
    """First buffer at once, every later one samples / rate after its predecessor, deadlines absolute: a consumer
    that dawdles between two buffers does not push the later ones back (clock_nanosleep TIMER_ABSTIME)."""
    L = host_lib(pkg)
    rate, block = 2.4e6, 4 * 131072  # four buffers a step = 218 ms: the tolerances below survive a busy test host
    period, slack = block / rate, 0.1

    def attempt():
        p = Pacer()
        L.msd_pacer_start(C.byref(p), rate)
        t0 = time.monotonic()
        release = []
        for i in range(5):
            L.msd_pacer_wait(C.byref(p), block)
            release.append(time.monotonic() - t0)
            if i == 2:
                time.sleep(0.15)  # a slow consumer, shorter than a step (and longer than the slack)
        # a ragged last block moves the deadline by its own length only
        L.msd_pacer_wait(C.byref(p), 1000)
        t1 = time.monotonic()
        L.msd_pacer_wait(C.byref(p), 0)
        return release, time.monotonic() - t1

    # never early -- on any attempt; on time within the slack -- on one attempt of three (the host may be busy)
    for _ in range(3):
        release, ragged = attempt()
        for i in range(1, 5):
            assert release[i] >= i * period - 1e-3, (i, release)
        if release[0] < slack and ragged < slack and all(release[i] < i * period + slack for i in range(1, 5)):
            break
    else:
        raise AssertionError(("late", release, ragged))


def test_pacer_carries_nanoseconds_into_seconds(pkg):
    L = host_lib(pkg)
    p = Pacer()
    L.msd_pacer_start(C.byref(p), 1e3)  # 1000 samples per second
    p.next.tv_sec, p.next.tv_nsec = 0, 999_999_000  # long past: no sleeping in this test
    L.msd_pacer_wait(C.byref(p), 2500)  # + 2.5 s
    assert (p.next.tv_sec, p.next.tv_nsec) == (3, 499_999_000)


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["fused", "magbuf"])
def test_throttled_replay_runs_in_signal_time_and_delivers_the_same_messages(pkg, oracle, torch_cuda, tmp_path, path):
    """parity is defined on the throttled (lossless) feed.  Eight buffers are 0.44 s of signal:
    the throttled replay takes at least seven buffer periods, the unthrottled one does not, both print the oracle's list."""
    n = 8 * 131072 + 3000
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=1093, msgs_per_sec=3000), n)
    f = tmp_path / "capture.uc8"
    iq.tofile(f)
    exe = os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "msd_replay")
    base = [exe, "--device-type", "ifile", "--ifile", str(f), "--iformat", "uc8", "--fix", "--mlat", "--raw", "--path", path]
    want, _ = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0).replay(iq, cap=1 << 16)
    expect = ["@%012X%s;" % (int(m["timestampMsg"]), bytes(m["msg"][: m["msgbits"] // 8]).hex()) for m in want]
    def run_seconds(res):
        return float([l for l in res.stderr.splitlines() if l.startswith("run_seconds")][0].split()[1])

    fast = subprocess.run(base + ["--stats"], capture_output=True, text=True, check=True)
    slow = subprocess.run(base + ["--stats", "--throttle"], capture_output=True, text=True, check=True)
    assert fast.stdout.split() == expect and len(expect) > 100
    assert slow.stdout.split() == expect
    # the reader's own clock around msd_ifileRun (process start-up and GPU initialisation are not in it): eight full
    # buffers are released one buffer period apart, the first at once
    assert run_seconds(slow) >= 8 * 131072 / 2.4e6 - 0.01, (run_seconds(fast), run_seconds(slow))
