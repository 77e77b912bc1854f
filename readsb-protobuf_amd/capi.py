"""ctypes binding of include/modes_hip.h (libmodes_hip.so) -- the Python face of the C-ABI.

The binding mirrors the reference's interfaces for this path:
  Demodulator.convert(...)            <-> iq_convert_fn            convert.h:33-38
  Demodulator.demodulate_magbuf(...)  <-> demodulate2400[AC] + icaoFilterExpire   demod_2400.h:37-38
  Demodulator.submit_*/launch/collect <-> ifileRun + the consumer loop   sdr_ifile.c:164-237
There is no CPU fallback: loading or creating a context without the HIP library / a GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSD_LIBMODES_HIP: another build of the same library (kernel experiments: scripts/r4_variant_build.sh); never a fallback
LIB_PATH = os.environ.get("MSD_LIBMODES_HIP") or os.path.join(_HERE, "csrc", "libmodes_hip.so")

FMT_UC8, FMT_SC16, FMT_SC16Q11, FMT_MAG16 = 0, 1, 2, 3
CHUNK = 131072
OVERLAP = 326
PIPELINE_DEPTH = 4

MESSAGE_DTYPE = np.dtype(
    [
        ("timestampMsg", "<u8"),
        ("sysTimestampMsg", "<u8"),
        ("signalLevel", "<f8"),
        ("addr", "<u4"),
        ("crc", "<u4"),
        ("score", "<i4"),
        ("msgtype", "u1"),
        ("msgbits", "u1"),
        ("correctedbits", "u1"),
        ("bestphase", "u1"),
        ("msg", "u1", (14,)),
        ("iid", "u1"),
        ("pad", "u1"),
    ],
    align=True,
)
assert MESSAGE_DTYPE.itemsize == 56

FIELDS_DTYPE = np.dtype(
    [("altitude_baro", "<i4"), ("AC", "<u2"), ("ID", "<u2"), ("squawk", "<u2"), ("altitude_baro_valid", "u1"),
     ("altitude_baro_unit", "u1"), ("squawk_valid", "u1"), ("airground", "u1"), ("alert", "u1"), ("alert_valid", "u1"),
     ("spi", "u1"), ("spi_valid", "u1"), ("CA", "u1"), ("CC", "u1"), ("CF", "u1"), ("DR", "u1"), ("FS", "u1"),
     ("KE", "u1"), ("ND", "u1"), ("RI", "u1"), ("SL", "u1"), ("UM", "u1"), ("VS", "u1"), ("source", "u1"),
     ("addrtype", "u1"), ("imf", "u1"), ("addr", "<u4"), ("metype", "u1"), ("mesub", "u1"), ("cpr_valid", "u1"),
     ("cpr_type", "u1"), ("cpr_odd", "u1"), ("nic_b_valid", "u1"), ("nic_b", "u1"), ("callsign_valid", "u1"),
     ("callsign", "S8"), ("cpr_lat", "<u4"), ("cpr_lon", "<u4"), ("altitude_geom", "<i4"),
     ("altitude_geom_valid", "u1"), ("altitude_geom_unit", "u1"), ("category", "u1"), ("category_valid", "u1"),
     ("nac_v_valid", "u1"), ("nac_v", "u1"), ("velocity_valid", "u1"), ("heading_valid", "u1"), ("ew_vel", "<i2"),
     ("ns_vel", "<i2"), ("heading_raw", "<u2"), ("heading_type", "u1"), ("movement", "u1"), ("ias", "<u2"),
     ("tas", "<u2"), ("ias_valid", "u1"), ("tas_valid", "u1"), ("baro_rate_valid", "u1"), ("geom_rate_valid", "u1"),
     ("baro_rate", "<i2"), ("geom_rate", "<i2"), ("geom_delta", "<i2"), ("geom_delta_valid", "u1"),
     ("emergency_valid", "u1"), ("emergency", "u1"),
     ("nav_valid", "u1"), ("nav_altitude_source", "u1"), ("nav_modes", "u1"), ("nav_heading_type", "u1"),
     ("acc_valid", "u1"), ("nac_p", "u1"), ("nic_baro", "u1"), ("nic_a", "u1"), ("nic_c", "u1"), ("gva", "u1"),
     ("sda", "u1"), ("sil", "u1"), ("sil_type", "u1"), ("cc_antenna_offset", "u1"), ("commb_format", "u1"),
     ("nav_heading_raw", "<u2"), ("nav_qnh_raw", "<u2"), ("nav_mcp_altitude", "<i4"), ("nav_fms_altitude", "<i4"),
     ("opstatus", "<u4"), ("roll_q", "<i2"), ("track_rate_q", "<i2"), ("gs", "<u2"), ("mach_raw", "<u2"),
     ("commb_valid", "u1"), ("pad2", "u1", (3,))],
    align=True,
)
assert FIELDS_DTYPE.itemsize == 140
CFG_DECODE_FIELDS = 1
CFG_DC_FILTER = 2
CFG_HOST_RESOLVE, CFG_CHAIN_IN_ORDER, CFG_CHAIN_SIDE_STREAMS, CFG_NO_LEAN, CFG_NO_RESOLVE_AHEAD = 1 << 4, 1 << 5, 1 << 6, 1 << 7, 1 << 8
CFG_POWER_KERNEL, CFG_POWER_IN_RESOLVE, CFG_EMIT_KERNEL, CFG_WAIT_INPUTS_ON_STREAM = 1 << 9, 1 << 10, 1 << 11, 1 << 12
CFG_NO_HELPER, CFG_REPASS_AUX, CFG_RECORDS_DMA, CFG_TRACE, CFG_NO_ARENA_GROWTH = 1 << 13, 1 << 14, 1 << 15, 1 << 16, 1 << 17
CFG_DC_SEQUENTIAL, CFG_DC_ONE_PASS, CFG_DC_FUSED_LAUNCH = 1 << 18, 1 << 19, 1 << 20


def layout_from_environment():
    """The library itself never reads the environment (modes_hip.h: every switch is a field of msd_config).  The test
    suite and the benchmark scripts select stream layouts and test settings for a whole process through MSD_*
    variables; this turns them into (flags, fields) for msd_create -- a convenience of the Python binding only."""
    e = os.environ.get
    flags = 0
    if e("MSD_GPU_RESOLVE", "1") == "0":
        flags |= CFG_HOST_RESOLVE
    if e("MSD_CHAIN_INLINE") not in (None, ""):
        flags |= CFG_CHAIN_IN_ORDER if e("MSD_CHAIN_INLINE") != "0" else CFG_CHAIN_SIDE_STREAMS
    if e("MSD_LEAN", "1") == "0":
        flags |= CFG_NO_LEAN
    if e("MSD_RESOLVE_AHEAD", "1") == "0":
        flags |= CFG_NO_RESOLVE_AHEAD
    if e("MSD_POWER_FUSED") not in (None, ""):
        flags |= CFG_POWER_IN_RESOLVE if e("MSD_POWER_FUSED") != "0" else CFG_POWER_KERNEL
    if e("MSD_EMIT_FUSED", "1") == "0":
        flags |= CFG_EMIT_KERNEL
    for name, bit in (("MSD_WAIT_INPUTS_ON_STREAM", CFG_WAIT_INPUTS_ON_STREAM), ("MSD_NO_HELPER", CFG_NO_HELPER),
                      ("MSD_REPASS_AUX", CFG_REPASS_AUX), ("MSD_RESOLVE_TRACE", CFG_TRACE)):
        if e(name) is not None:
            flags |= bit
    if e("MSD_RECORDS_DMA", "0") not in ("0", ""):
        flags |= CFG_RECORDS_DMA
    if e("MSD_ARENA_GROWTH", "1") == "0":
        flags |= CFG_NO_ARENA_GROWTH
    if e("MSD_DC", "") == "sequential":
        flags |= CFG_DC_SEQUENTIAL
    elif e("MSD_DC", "") == "one_pass":
        flags |= CFG_DC_ONE_PASS
    elif e("MSD_DC", "") == "fused":
        flags |= CFG_DC_FUSED_LAUNCH
    fields = dict(resolve_threads=int(e("MSD_RESOLVE_THREADS", "0") or 0),
                  test_arena_permille=int(e("MSD_ARENA_SCALE_PERMILLE", "0") or 0),
                  test_inline_adds=int(e("MSD_RESOLVE_INLINE_ADDS", "0") or 0),
                  debug_flags=int(e("MSD_DEBUG_FLAGS", "0") or 0))
    return flags, fields
INVALID_ALTITUDE = -9999


class Config(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("format", C.c_int32),
        ("preamble_threshold", C.c_int32),
        ("nfix_crc", C.c_int32),
        ("mode_ac", C.c_int32),
        ("flags", C.c_int32),
        ("max_batch_samples", C.c_uint64),
        ("stream", C.c_void_p),
        ("resolve_threads", C.c_int32),
        ("test_arena_permille", C.c_int32),
        ("test_inline_adds", C.c_int32),
        ("debug_flags", C.c_int32),
        ("sc16q11_table_bits", C.c_int32),
        ("reserved0", C.c_int32),
        ("sample_rate", C.c_double),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("demod_preambles", C.c_uint64),
        ("demod_rejected_bad", C.c_uint64),
        ("demod_rejected_unknown_icao", C.c_uint64),
        ("demod_accepted", C.c_uint64 * 3),
        ("demod_preamblePhase", C.c_uint64 * 5),
        ("demod_bestPhase", C.c_uint64 * 5),
        ("demod_modeac", C.c_uint64),
        ("strong_signal_count", C.c_uint64),
        ("samples_processed", C.c_uint64),
        ("noise_power_count", C.c_uint64),
        ("signal_power_count", C.c_uint64),
        ("noise_power_sum", C.c_double),
        ("signal_power_sum", C.c_double),
        ("peak_signal_power", C.c_double),
        ("buffers", C.c_uint64),
        ("samples_dropped", C.c_uint64),
    ]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out


class Timing(C.Structure):
    _fields_ = [
        ("scan_kernel_ms", C.c_float),
        ("other_kernels_ms", C.c_float),
        ("d2h_ms", C.c_float),
        ("resolve_ms", C.c_float),
        ("hits", C.c_uint64),
        ("tries", C.c_uint64),
        ("reruns", C.c_uint64),
        ("resolve_passes", C.c_uint64),
        ("resolve_fallback", C.c_uint64),
        ("resolve_long_lists", C.c_uint64),
        ("timed_batches", C.c_uint64),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class _SinkState(C.Structure):
    _fields_ = [("out", C.c_void_p), ("cap", C.c_size_t), ("count", C.c_size_t)]


class _FieldsSinkState(C.Structure):
    _fields_ = [("out", C.c_void_p), ("fields", C.c_void_p), ("cap", C.c_size_t), ("count", C.c_size_t)]


EXPORTS = [
    "msd_create", "msd_destroy", "msd_last_error", "msd_submit_device", "msd_submit_host", "msd_reset",
    "msd_launch_device", "msd_launch_host", "msd_host_alloc", "msd_host_free", "msd_collect", "msd_get_stats",
    "msd_get_timing", "msd_get_buffer_means", "msd_convert", "msd_demodulate_magbuf", "msd_array_sink",
    "msd_collect_fields", "msd_decode_fields", "msd_fields_to_float", "msd_array_fields_sink",
    "msd_note_dropped", "msd_set_preamble_threshold", "msd_set_timing_interval", "msd_restart", "msd_decode_fields_device",
    "msd_arena_permille", "msd_host_register", "msd_host_unregister", "msd_demodulate_magbufs",
    "msd_convert_begin", "msd_convert_end", "msd_thread_attach", "msd_dc_filter_status",
]

_lib = None


def lib():
    """Load libmodes_hip.so (built in-tree by __graft_entry__.build()); fail loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`; "
                               "there is no CPU fallback for the demodulator")
        L = C.CDLL(LIB_PATH)
        L.msd_create.restype = C.c_int
        L.msd_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        L.msd_destroy.argtypes = [C.c_void_p]
        L.msd_last_error.restype = C.c_char_p
        L.msd_last_error.argtypes = [C.c_void_p]
        for name in ("msd_submit_device", "msd_submit_host"):
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
        L.msd_reset.restype = C.c_int
        L.msd_reset.argtypes = [C.c_void_p]
        L.msd_decode_fields_device.restype = C.c_int
        L.msd_decode_fields_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.msd_restart.restype = C.c_int
        L.msd_restart.argtypes = [C.c_void_p]
        L.msd_note_dropped.restype = C.c_int
        L.msd_note_dropped.argtypes = [C.c_void_p, C.c_uint64]
        L.msd_set_timing_interval.restype = C.c_int
        L.msd_set_timing_interval.argtypes = [C.c_void_p, C.c_uint32]
        L.msd_set_preamble_threshold.restype = C.c_int
        L.msd_set_preamble_threshold.argtypes = [C.c_void_p, C.c_int]
        for name in ("msd_launch_device", "msd_launch_host"):
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        L.msd_host_alloc.restype = C.c_int
        L.msd_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.msd_collect_fields.restype = C.c_int
        L.msd_collect_fields.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.msd_decode_fields.restype = None
        L.msd_decode_fields.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.msd_host_free.restype = None
        L.msd_host_free.argtypes = [C.c_void_p, C.c_void_p]
        L.msd_collect.restype = C.c_int
        L.msd_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.msd_get_stats.restype = C.c_int
        L.msd_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.msd_get_timing.restype = C.c_int
        L.msd_get_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
        L.msd_get_buffer_means.restype = C.c_int
        L.msd_get_buffer_means.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.msd_convert.restype = C.c_int
        L.msd_convert.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.POINTER(C.c_double),
                                  C.POINTER(C.c_double)]
        L.msd_demodulate_magbuf.restype = C.c_int
        L.msd_demodulate_magbuf.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint64, C.c_uint64,
                                            C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.msd_demodulate_magbufs.restype = C.c_int
        L.msd_demodulate_magbufs.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p]
        L.msd_convert_begin.restype = C.c_int
        L.msd_convert_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
        L.msd_convert_end.restype = C.c_int
        L.msd_convert_end.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        for name in ("msd_thread_attach", "msd_arena_permille"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [C.c_void_p]
        L.msd_dc_filter_status.restype = C.c_int
        L.msd_dc_filter_status.argtypes = [C.c_void_p, C.c_void_p]
        L.msd_host_register.restype = C.c_int
        L.msd_host_register.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.msd_host_unregister.restype = None
        L.msd_host_unregister.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class MsdError(RuntimeError):
    pass


def _raw_copy(a):
    """Copy of a contiguous record array as one memcpy (numpy copies structured arrays field by field)."""
    out = np.empty(a.shape, dtype=a.dtype)
    out.view(np.uint8)[:] = a.view(np.uint8)
    return out


class Demodulator:
    """One receiver context on one GPU (its own ICAO filter, clock, counters and HIP streams)."""

    def __init__(self, fmt=FMT_UC8, preamble_threshold=58, nfix_crc=1, mode_ac=0, device=0,
                 max_batch_samples=CHUNK, stream=None, message_capacity=1 << 16, decode_fields=False, dc_filter=False,
                 flags=None, **fields):
        """flags / fields: msd_config.flags (CFG_*) and its tuning / test fields; None = from the MSD_* environment
        variables of the test suite (layout_from_environment)."""
        self._h = C.c_void_p()
        self.fmt = fmt
        if flags is None:
            flags, env_fields = layout_from_environment()
            fields = {**env_fields, **fields}
        self.flags = flags | (CFG_DECODE_FIELDS if decode_fields else 0) | (CFG_DC_FILTER if dc_filter else 0)
        cfg = Config(device=device, format=fmt, preamble_threshold=preamble_threshold, nfix_crc=nfix_crc,
                     mode_ac=mode_ac, flags=self.flags, max_batch_samples=max_batch_samples,
                     stream=C.c_void_p(stream) if stream else None, **fields)
        rc = lib().msd_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise MsdError(f"msd_create failed: {os.strerror(-rc)} ({rc})")
        self._sink_fn = C.cast(lib().msd_array_sink, C.c_void_p)
        self._buf = np.zeros(message_capacity, dtype=MESSAGE_DTYPE)
        self._chunks = []

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().msd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dc_filter_status(self):
        """(exact, passes, guessed, blocks) of the most recent batch's DC block: exact = 1 when the parallel-in-time kernels
        did it, 0 when the in-order kernel behind them had to."""
        out = (C.c_uint32 * 4)()
        self._check(lib().msd_dc_filter_status(self._h, out))
        return tuple(int(v) for v in out)

    def arena_permille(self):
        """Candidate arenas in thousandths of the base size: 4000, or 1000 after msd_create's out-of-memory retry."""
        return int(lib().msd_arena_permille(self._h))

    @property
    def bytes_per_sample(self):
        return 2 if self.fmt in (FMT_UC8, FMT_MAG16) else 4

    def _check(self, rc):
        if rc != 0:
            raise MsdError(f"{lib().msd_last_error(self._h).decode()} ({os.strerror(-rc)}, {rc})")

    def _run(self, call, copy=True):
        """Run one C call that delivers messages; returns them as a structured array (a view into the
        context's message buffer, valid until the next call, when copy=False)."""
        while True:
            st = _SinkState(self._buf.ctypes.data, self._buf.size, 0)
            rc = call(self._sink_fn, C.byref(st))
            self._check(rc)
            if st.count <= self._buf.size:
                return _raw_copy(self._buf[: st.count]) if copy else self._buf[: st.count]
            # a context is stateful, so a too-small array cannot simply be retried: grow ahead of time
            raise MsdError(f"message array too small ({st.count} > {self._buf.size}); "
                           "construct the Demodulator with a larger message_capacity")

    def collect_fields(self, copy=True):
        """msd_collect_fields: (messages, header fields) of the oldest outstanding batch.  copy=False
        returns views of the context's arrays, valid until the next collect."""
        if not hasattr(self, "_fbuf") or self._fbuf.size != self._buf.size:
            self._fbuf = np.zeros(self._buf.size, dtype=FIELDS_DTYPE)
        st = _FieldsSinkState(self._buf.ctypes.data, self._fbuf.ctypes.data, self._buf.size, 0)
        self._check(lib().msd_collect_fields(self._h, C.cast(lib().msd_array_fields_sink, C.c_void_p), C.byref(st)))
        if st.count > self._buf.size:
            raise MsdError(f"message array too small ({st.count} > {self._buf.size})")
        if not copy:
            return self._buf[: st.count], self._fbuf[: st.count]
        return _raw_copy(self._buf[: st.count]), _raw_copy(self._fbuf[: st.count])

    def reserve_messages(self, n):
        if n > self._buf.size:
            self._buf = np.zeros(n, dtype=MESSAGE_DTYPE)

    # --- streaming interface -------------------------------------------------------------------
    def submit_device(self, dptr, nsamples, last=True):
        return self._run(lambda fn, st: lib().msd_submit_device(self._h, C.c_void_p(dptr), nsamples, int(last), fn, st))

    def submit_host(self, iq, nsamples=None, last=True):
        iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
        if nsamples is None:
            nsamples = iq.size // self.bytes_per_sample
        return self._run(lambda fn, st: lib().msd_submit_host(self._h, iq.ctypes.data, nsamples, int(last), fn, st))

    def launch_device(self, dptr, nsamples, last=False):
        self._check(lib().msd_launch_device(self._h, C.c_void_p(dptr), nsamples, int(last)))

    def launch_host(self, iq, nsamples=None, last=False):
        """Asynchronous ingest of host samples; `iq` (uint8 view) must stay alive and unchanged until collected.
        Use host_buffer() for page-locked memory (the upload is then a DMA at PCIe rate)."""
        iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
        if nsamples is None:
            nsamples = iq.size // self.bytes_per_sample
        self._check(lib().msd_launch_host(self._h, iq.ctypes.data, nsamples, int(last)))

    def host_buffer(self, nbytes):
        """A page-locked uint8 numpy array (msd_host_alloc); freed when the array is garbage collected."""
        p = C.c_void_p()
        self._check(lib().msd_host_alloc(self._h, nbytes, C.byref(p)))
        raw = (C.c_uint8 * nbytes).from_address(p.value)
        arr = np.frombuffer(raw, dtype=np.uint8)
        import weakref
        h, L = self._h, lib()
        weakref.finalize(raw, lambda: L.msd_host_free(h, p))
        return arr

    def collect(self, copy=True):
        return self._run(lambda fn, st: lib().msd_collect(self._h, fn, st), copy=copy)

    def reset(self):
        self._check(lib().msd_reset(self._h))

    def decode_fields_device(self, messages):
        """msd_decode_fields_device: the emit kernel's field decoder on an array of message records."""
        messages = np.ascontiguousarray(messages, dtype=MESSAGE_DTYPE)
        out = np.zeros(len(messages), dtype=FIELDS_DTYPE)
        self._check(lib().msd_decode_fields_device(self._h, messages.ctypes.data, len(messages), out.ctypes.data))
        return out

    def restart(self):
        """msd_restart: a new capture behind one whose last batches are still in flight."""
        self._check(lib().msd_restart(self._h))

    def note_dropped(self, nsamples):
        """msd_note_dropped: the receiver lost nsamples in front of the next batch (MAGBUF_DISCONTINUOUS)."""
        self._check(lib().msd_note_dropped(self._h, nsamples))

    def set_timing_interval(self, every):
        """msd_set_timing_interval: record the kernel timing events for one batch in `every` (0: never)."""
        self._check(lib().msd_set_timing_interval(self._h, every))

    def set_preamble_threshold(self, threshold):
        self._check(lib().msd_set_preamble_threshold(self._h, threshold))

    def stats(self):
        st = Stats()
        self._check(lib().msd_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def timing(self):
        t = Timing()
        self._check(lib().msd_get_timing(self._h, C.byref(t)))
        return t.as_dict()

    def buffer_means(self, cap=1 << 16):
        out = np.zeros((cap, 2), dtype=np.float64)
        n = lib().msd_get_buffer_means(self._h, out.ctypes.data, cap)
        if n < 0:
            self._check(n)
        return out[: min(n, cap)].copy()

    # --- iq_convert_fn ---------------------------------------------------------------------------
    def convert(self, iq, nsamples):
        iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
        mag = np.zeros(nsamples, dtype=np.uint16)
        ml, mp = C.c_double(), C.c_double()
        self._check(lib().msd_convert(self._h, iq.ctypes.data, mag.ctypes.data, nsamples, C.byref(ml), C.byref(mp)))
        return mag, ml.value, mp.value

    # --- demodulate2400(struct mag_buf *) ----------------------------------------------------------
    def demodulate_magbuf(self, data, valid_length=None, overlap=OVERLAP, sample_timestamp=0, sys_timestamp=0,
                          mean_level=0.0, mean_power=0.0):
        data = np.ascontiguousarray(data, dtype=np.uint16)
        if valid_length is None:
            valid_length = data.size
        return self._run(lambda fn, st: lib().msd_demodulate_magbuf(
            self._h, data.ctypes.data, valid_length, overlap, sample_timestamp, sys_timestamp, mean_level,
            mean_power, fn, st))


class MagbufView(C.Structure):
    _fields_ = [("data", C.c_void_p), ("validLength", C.c_uint), ("overlap", C.c_uint), ("sampleTimestamp", C.c_uint64),
                ("sysTimestamp", C.c_uint64), ("mean_level", C.c_double), ("mean_power", C.c_double)]


def demodulate_magbufs(demod, bufs):
    """msd_demodulate_magbufs: bufs = [(data, valid_length, overlap, sample_timestamp, sys_timestamp, mean_level, mean_power), ...]
    of consecutive buffers, one GPU batch."""
    keep = [np.ascontiguousarray(b[0], dtype=np.uint16) for b in bufs]
    views = (MagbufView * len(bufs))()
    for i, b in enumerate(bufs):
        views[i] = MagbufView(keep[i].ctypes.data, b[1], b[2], b[3], b[4], b[5], b[6])
    return demod._run(lambda fn, st: lib().msd_demodulate_magbufs(demod._h, views, len(bufs), fn, st))


def replay_device(demod, dptr, nsamples, batch_samples):
    """Replay a device-resident capture through the two-deep pipeline; returns all messages."""
    bps = demod.bytes_per_sample
    out = []
    off = 0
    inflight = 0
    while True:
        n = min(batch_samples, nsamples - off)
        last = off + n >= nsamples
        if inflight == PIPELINE_DEPTH:
            out.append(demod.collect())
            inflight -= 1
        demod.launch_device(dptr + off * bps, n, last)
        inflight += 1
        off += n
        if last:
            break
    while inflight:
        out.append(demod.collect())
        inflight -= 1
    return np.concatenate(out) if out else np.zeros(0, dtype=MESSAGE_DTYPE)


FIELDS_FLOAT_DTYPE = np.dtype(
    [("gs_v0", "<f4"), ("gs_v2", "<f4"), ("gs_selected", "<f4"), ("heading", "<f4"), ("track_rate", "<f4"),
     ("roll", "<f4"), ("nav_qnh", "<f4"), ("nav_heading", "<f4"), ("mach", "<f8"), ("gs_valid", "u1"),
     ("heading_valid", "u1"), ("heading_type", "u1"), ("track_rate_valid", "u1"), ("roll_valid", "u1"),
     ("mach_valid", "u1"), ("nav_qnh_valid", "u1"), ("nav_heading_valid", "u1")], align=True)
assert FIELDS_FLOAT_DTYPE.itemsize == 48


def fields_to_float(fields):
    """msd_fields_to_float: the float-valued members of struct modesMessage for one msd_fields record."""
    rec = np.zeros(1, dtype=FIELDS_DTYPE)
    rec[0] = fields
    out = np.zeros(1, dtype=FIELDS_FLOAT_DTYPE)
    lib().msd_fields_to_float.restype = None
    lib().msd_fields_to_float.argtypes = [C.c_void_p, C.c_void_p]
    lib().msd_fields_to_float(rec.ctypes.data, out.ctypes.data)
    return out[0]


def decode_fields(message, carry=None):
    """msd_decode_fields on one message record; carry = fields of the previous Mode A/C reply of the buffer."""
    rec = np.ascontiguousarray(message).reshape(1)
    out = np.zeros(1, dtype=FIELDS_DTYPE)
    cp = np.ascontiguousarray(carry).reshape(1).ctypes.data if carry is not None else None
    lib().msd_decode_fields(rec.ctypes.data, cp, out.ctypes.data)
    return out[0]
