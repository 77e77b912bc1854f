"""bench.py's host-side pieces that need no GPU: the two-thread CPU baseline stream (SURVEY.md 8(d) form (b): reader
thread + demodulator thread per stream, every thread on a CPU of its own) and the list / counter diff."""
import argparse
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args():
    return argparse.Namespace(threshold=58, fix=1, mode_ac=False, sc16q11_table_bits=0)


def test_two_thread_stream_delivers_the_oracles_messages(pkg, oracle):
    import bench
    nb = 6
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=10901), nb * pkg.CHUNK, nthreads=2)
    want, _ = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0).replay(iq)
    two = bench.two_thread_stream(oracle, pkg, iq, 2, oracle.FMT_UC8, _args(), nb)
    # the stream ends with its last whole buffer (no end-of-file buffer behind it): the replay's list up to there
    limit = nb * pkg.CHUNK * 5
    assert two["messages"] >= int((want["timestampMsg"] < limit - 2000 * 5).sum()) > 50
    assert two["messages"] <= len(want)
    assert two["cores"] == 2 and two["reader_thread_cpu_s"] >= 0 and two["demod_thread_cpu_s"] > 0 and two["value"] > 0


def test_n_streams_pin_every_thread_to_its_own_cpu(pkg, oracle):
    import bench
    avail = sorted(os.sched_getaffinity(0))
    if len(avail) < 4:
        import pytest
        pytest.skip("needs four CPUs")
    before = os.sched_getaffinity(0)
    nb, world = 3, 2
    iqs = [pkg.siggen.generate(pkg.siggen.make_cfg(seed=10901 + r), nb * pkg.CHUNK, nthreads=2) for r in range(world)]
    results, threads = [[] for _ in range(world)], []
    for r in range(world):
        th = threading.Thread(target=bench.two_thread_stream,
                              args=(oracle, pkg, iqs[r], 2, oracle.FMT_UC8, _args(), nb, (avail[2 * r], avail[2 * r + 1]), results[r]))
        th.start()
        threads.append(th)
    for th in threads:
        th.join()
    got = [x[0] for x in results]
    assert [x["cpus"] for x in got] == [[avail[0], avail[1]], [avail[2], avail[3]]]
    assert got[0]["messages"] != got[1]["messages"] and min(x["messages"] for x in got) > 20
    assert os.sched_getaffinity(0) == before          # the calling thread's mask is its own


def test_diff_against_oracle_counts_fields_and_counters(pkg, oracle):
    import bench
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=10901), 2 * pkg.CHUNK, nthreads=2)
    want, wstats = oracle.Oracle(oracle.FMT_UC8, 58, 0, 0).replay(iq)
    assert bench.diff_against_oracle(want, wstats, want, wstats) == 0
    other = want.copy()
    other["addr"][3] ^= 1
    other["msg"][5][2] ^= 0x10
    assert bench.diff_against_oracle(other, wstats, want, wstats) == 2
    assert bench.diff_against_oracle(want[:-1], wstats, want, wstats) == 1
    st2 = dict(wstats)
    st2["demod_preambles"] = wstats["demod_preambles"] + 1
    assert bench.diff_against_oracle(want, st2, want, wstats) == 1
    assert bench.cpu_model()
