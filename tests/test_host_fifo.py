"""The mag_buf FIFO of the host boundary (fifo.h semantics with the tail bug fixed)."""
import ctypes as C
import os

import numpy as np


class MagBuf(C.Structure):
    pass


MagBuf._fields_ = [("data", C.POINTER(C.c_uint16)), ("totalLength", C.c_uint), ("validLength", C.c_uint),
                   ("overlap", C.c_uint), ("sampleTimestamp", C.c_uint64), ("sysTimestamp", C.c_uint64),
                   ("flags", C.c_int), ("mean_level", C.c_double), ("mean_power", C.c_double),
                   ("dropped", C.c_uint), ("next", C.POINTER(MagBuf))]


def _host(pkg):
    h = C.CDLL(os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "libmsd_host.so"))
    h.msd_fifo_create.restype = C.c_bool
    h.msd_fifo_create.argtypes = [C.c_uint, C.c_uint, C.c_uint]
    h.msd_fifo_acquire.restype = C.POINTER(MagBuf)
    h.msd_fifo_acquire.argtypes = [C.c_uint32]
    h.msd_fifo_dequeue.restype = C.POINTER(MagBuf)
    h.msd_fifo_dequeue.argtypes = [C.c_uint32]
    h.msd_fifo_enqueue.argtypes = [C.POINTER(MagBuf)]
    h.msd_fifo_release.argtypes = [C.POINTER(MagBuf)]
    return h


def test_queue_deeper_than_one_keeps_every_buffer_and_overlap_rule(pkg):
    h = _host(pkg)
    overlap, new = 326, 1000
    assert h.msd_fifo_create(4, overlap + new, overlap)
    try:
        sent = []
        for k in range(3):                      # three buffers queued before anything is consumed
            b = h.msd_fifo_acquire(10)
            assert b and b.contents.validLength == overlap and b.contents.overlap == overlap
            data = np.ctypeslib.as_array(b.contents.data, shape=(overlap + new,))
            data[overlap:] = np.arange(new, dtype=np.uint16) + 1000 * (k + 1)
            b.contents.validLength = overlap + new
            b.contents.sampleTimestamp = k
            sent.append(data[overlap:].copy())
            h.msd_fifo_enqueue(b)
        got = []
        while True:
            b = h.msd_fifo_dequeue(0)
            if not b:
                break
            data = np.ctypeslib.as_array(b.contents.data, shape=(overlap + new,)).copy()
            got.append((b.contents.sampleTimestamp, data))
            h.msd_fifo_release(b)
        assert [g[0] for g in got] == [0, 1, 2]           # the reference loses the middle ones (fifo.c:192-197)
        assert (got[0][1][:overlap] == 0).all()           # first buffer: zero overlap
        for k in (1, 2):                                  # fifo.c:179-188: previous buffer's last 326 samples
            assert np.array_equal(got[k][1][:overlap], sent[k - 1][-overlap:])
        assert not h.msd_fifo_acquire(0) is None
    finally:
        h.msd_fifo_halt()
        h.msd_fifo_destroy()
