"""The mag_buf FIFO of the host boundary (fifo.h:34-120), against its contract and against the
reference's own fifo.c (oracle/_ref/libref_fifo.so, compiled unmodified from /root/reference)."""
import ctypes as C
import os
import shutil
import threading
import time

import numpy as np
import pytest


class MagBuf(C.Structure):
    pass


MagBuf._fields_ = [("data", C.POINTER(C.c_uint16)), ("totalLength", C.c_uint), ("validLength", C.c_uint),
                   ("overlap", C.c_uint), ("sampleTimestamp", C.c_uint64), ("sysTimestamp", C.c_uint64),
                   ("flags", C.c_int), ("mean_level", C.c_double), ("mean_power", C.c_double),
                   ("dropped", C.c_uint), ("next", C.POINTER(MagBuf))]

DISCONTINUOUS = 1


class Fifo:
    """Either implementation behind the same eight calls."""

    def __init__(self, path, prefix):
        self.h = C.CDLL(path)
        f = lambda name: getattr(self.h, prefix + name)
        self.create, self.destroy, self.drain, self.halt = f("create"), f("destroy"), f("drain"), f("halt")
        self.acquire, self.enqueue, self.dequeue, self.release = f("acquire"), f("enqueue"), f("dequeue"), f("release")
        self.create.restype = C.c_bool
        self.create.argtypes = [C.c_uint, C.c_uint, C.c_uint]
        for fn in (self.acquire, self.dequeue):
            fn.restype = C.POINTER(MagBuf)
            fn.argtypes = [C.c_uint32]
        self.enqueue.argtypes = [C.POINTER(MagBuf)]
        self.release.argtypes = [C.POINTER(MagBuf)]


def ours(pkg):
    return Fifo(os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "libmsd_host.so"), "msd_fifo_")


def reference(oracle, tmp_path):
    path = oracle.build_ref()
    if path is None:
        pytest.skip("oracle/_ref/libref_fifo.so is not there and /root/reference is not available to build it")
    # a private copy per test: the reference keeps its state in file statics and fifo_create does not
    # clear fifo_halted, so one loaded instance serves one create .. halt cycle
    mine = tmp_path / "libref_fifo_instance.so"
    shutil.copy(path, mine)
    return Fifo(str(mine), "fifo_")


def feed(q, blocks, overlap, size, depth_one=True, flags=None):
    """Produce `blocks` (arrays of new samples), consuming as the reference's main loop does; returns what
    the consumer saw: (sampleTimestamp, copy of data[0:validLength])."""
    got = []

    def consume_all():
        while True:
            b = q.dequeue(0)
            if not b:
                return
            n = b.contents.validLength
            got.append((b.contents.sampleTimestamp, np.ctypeslib.as_array(b.contents.data, shape=(size,))[:n].copy()))
            q.release(b)

    for k, new in enumerate(blocks):
        b = q.acquire(10)
        assert b and b.contents.validLength == overlap and b.contents.overlap == overlap
        assert b.contents.totalLength == size and b.contents.flags == 0
        data = np.ctypeslib.as_array(b.contents.data, shape=(size,))
        data[overlap:overlap + len(new)] = new
        b.contents.validLength = overlap + len(new)
        b.contents.sampleTimestamp = k
        if flags and flags[k]:
            b.contents.flags = flags[k]
        q.enqueue(b)
        if depth_one:
            consume_all()
    consume_all()
    return got


def make_blocks(rng, n, new):
    return [rng.integers(1, 65535, size=new).astype(np.uint16) for _ in range(n)]


def test_queue_deeper_than_one_keeps_every_buffer_and_overlap_rule(pkg):
    q = ours(pkg)
    overlap, new = 326, 1000
    assert q.create(4, overlap + new, overlap)
    try:
        blocks = make_blocks(np.random.default_rng(1), 3, new)
        got = feed(q, blocks, overlap, overlap + new, depth_one=False)  # three queued before anything is consumed
        assert [g[0] for g in got] == [0, 1, 2]           # the reference loses the middle ones (fifo.c:192-197)
        assert (got[0][1][:overlap] == 0).all()           # first buffer: zero overlap
        for k in (1, 2):                                  # fifo.h:34-55: previous buffer's last 326 samples
            assert np.array_equal(got[k][1][:overlap], blocks[k - 1][-overlap:])
            assert np.array_equal(got[k][1][overlap:], blocks[k])
    finally:
        q.halt()
        q.destroy()


def test_same_stream_as_the_reference_fifo_on_a_depth_one_feed(pkg, oracle, tmp_path):
    """The feed under which the reference is lossless: every buffer is consumed before the
    next one is enqueued.  Ragged last block, a discontinuity in the middle, a block shorter than the overlap."""
    overlap, new = 326, 4096
    rng = np.random.default_rng(7)
    blocks = make_blocks(rng, 7, new)
    blocks[3] = blocks[3][:100]      # shorter than the overlap: the carried tail then spans two blocks
    blocks[6] = blocks[6][:1234]     # ragged end of the capture
    flags = [0, 0, DISCONTINUOUS, 0, 0, 0, 0]
    seen = {}
    for name, q in (("ref", reference(oracle, tmp_path)), ("ours", ours(pkg))):
        assert q.create(12, overlap + new, overlap)
        try:
            seen[name] = feed(q, blocks, overlap, overlap + new, depth_one=True, flags=flags)
        finally:
            q.halt()
            q.destroy()
    assert len(seen["ref"]) == len(seen["ours"]) == len(blocks)
    for (ts_r, d_r), (ts_o, d_o) in zip(seen["ref"], seen["ours"]):
        assert ts_r == ts_o and np.array_equal(d_r, d_o)
    assert (seen["ours"][2][1][:overlap] == 0).all()  # behind the gap: silence (fifo.c:179-182)


def test_the_reference_fifo_loses_buffers_once_two_are_queued(oracle, tmp_path):
    """The defect this build does not reproduce, shown on the reference's own code: with three buffers
    queued only the first is ever dequeued (fifo.c:192-197 never advances the tail pointer)."""
    q = reference(oracle, tmp_path)
    overlap, new = 326, 1000
    assert q.create(4, overlap + new, overlap)
    try:
        got = feed(q, make_blocks(np.random.default_rng(2), 3, new), overlap, overlap + new, depth_one=False)
        assert len(got) < 3
    finally:
        q.halt()
        q.destroy()


def test_timeouts_and_nonblocking_calls(pkg):
    q = ours(pkg)
    assert q.create(2, 400, 100)
    try:
        assert not q.dequeue(0)                      # empty, non-blocking
        t0 = time.monotonic()
        assert not q.dequeue(50)                     # empty: really times out (fifo.c:219 never does)
        assert 0.03 < time.monotonic() - t0 < 1.0
        a, b = q.acquire(0), q.acquire(0)
        assert a and b
        t0 = time.monotonic()
        assert not q.acquire(50)                     # both buffers are out
        assert 0.03 < time.monotonic() - t0 < 1.0
        q.release(a)
        assert q.acquire(0)                          # release makes it available again
    finally:
        q.halt()
        q.destroy()


def test_halt_wakes_a_blocked_consumer_and_refuses_new_work(pkg):
    q = ours(pkg)
    assert q.create(3, 400, 100)
    try:
        b = q.acquire(0)
        b.contents.validLength = 400
        q.enqueue(b)
        first = q.dequeue(0)
        assert first
        result = {}
        th = threading.Thread(target=lambda: result.setdefault("buf", bool(q.dequeue(5000))))
        t0 = time.monotonic()
        th.start()
        time.sleep(0.05)
        q.halt()                                     # fifo.h:89-95: waiters return NULL at once
        th.join(2.0)
        assert not th.is_alive() and result["buf"] is False and time.monotonic() - t0 < 2.0
        assert not q.acquire(10) and not q.dequeue(10)
        q.release(first)
    finally:
        q.destroy()


def test_drain_returns_when_the_consumer_has_taken_everything(pkg):
    q = ours(pkg)
    assert q.create(4, 400, 100)
    try:
        for _ in range(3):
            b = q.acquire(0)
            b.contents.validLength = 400
            q.enqueue(b)

        def consumer():
            for _ in range(3):
                time.sleep(0.02)
                q.release(q.dequeue(1000))

        th = threading.Thread(target=consumer)
        th.start()
        q.drain()                                    # fifo.h:86-87
        assert not q.dequeue(0)
        th.join(2.0)
        assert not th.is_alive()
    finally:
        q.halt()
        q.destroy()
