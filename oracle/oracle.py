"""ctypes binding of oracle/libmodes_oracle.so -- the CPU restatement used as the parity checker.

TEST INFRASTRUCTURE ONLY ("parity unpinned" for the demodulator, see modes_oracle.h).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmodes_oracle.so")

FMT_UC8, FMT_SC16, FMT_SC16Q11 = 0, 1, 2
CHUNK = 131072
OVERLAP = 326

# numpy mirror of orc_message (modes_oracle.h)
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

# numpy mirror of orc_fields
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
    ]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out


def build(force=False):
    """Compile the restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "modes_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "modes_oracle.h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libmodes_oracle.so"])
    return _LIB


REF_DIR = "/root/reference"
REF_FIFO = os.path.join(_HERE, "_ref", "libref_fifo.so")


def build_ref():
    """oracle/_ref: the reference's own fifo.c, compiled where it lies (the only hot-path file of the
    reference that builds in this image, see oracle/Makefile).  Returns the library path, or None when
    neither the reference tree nor a prebuilt library is there (the GPU box has only the latter)."""
    if os.path.isdir(REF_DIR):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_ref"])
    return REF_FIFO if os.path.exists(REF_FIFO) else None


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int] * 4
        L.orc_set_dc_filter.restype = None
        L.orc_set_dc_filter.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_recently_dropped.restype = None
        L.orc_set_recently_dropped.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_sc16q11_table_bits.restype = None
        L.orc_set_sc16q11_table_bits.argtypes = [C.c_void_p, C.c_int]
        L.orc_sc16q11_table.restype = C.POINTER(C.c_uint16)
        L.orc_sc16q11_table.argtypes = [C.c_void_p]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_replay.restype = C.c_uint64
        L.orc_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t,
                                 C.POINTER(C.c_size_t), C.c_void_p, C.c_size_t]
        L.orc_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.orc_convert.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.POINTER(C.c_double),
                                  C.POINTER(C.c_double)]
        L.orc_uc8_table.restype = C.POINTER(C.c_uint16)
        L.orc_checksum.restype = C.c_uint32
        L.orc_checksum.argtypes = [C.c_char_p, C.c_int]
        L.orc_diagnose.restype = C.c_int
        L.orc_diagnose.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_int * 2)]
        L.orc_score.restype = C.c_int
        L.orc_score.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.orc_filter_add.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_filter_test.restype = C.c_int
        L.orc_filter_test.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_slice.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
        L.orc_demod_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint64, C.c_uint64,
                                       C.c_double, C.c_double, C.c_void_p, C.c_size_t,
                                       C.POINTER(C.c_size_t)]
        L.orc_set_fields_out.restype = None
        L.orc_set_fields_out.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_decode_ac13.restype = C.c_int
        L.orc_decode_ac13.argtypes = [C.c_uint, C.POINTER(C.c_int)]
        L.orc_decode_id13.restype = C.c_uint
        L.orc_decode_id13.argtypes = [C.c_uint]
        L.orc_mode_a_to_mode_c.restype = C.c_int
        L.orc_mode_a_to_mode_c.argtypes = [C.c_uint]
        L.orc_avr_line.restype = C.c_size_t
        L.orc_avr_line.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.orc_beast_frame.restype = C.c_size_t
        L.orc_beast_frame.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def avr_line(message, mlat=False):
    """AVR raw line of one message record (numpy void of MESSAGE_DTYPE), net_io.c:870-896."""
    rec = np.ascontiguousarray(message).reshape(1)
    buf = C.create_string_buffer(64)
    n = lib().orc_avr_line(rec.ctypes.data, int(mlat), buf)
    return buf.raw[:n]


def beast_frame(message):
    """Beast binary frame of one message record, net_io.c:769-835."""
    rec = np.ascontiguousarray(message).reshape(1)
    buf = (C.c_uint8 * 64)()
    n = lib().orc_beast_frame(rec.ctypes.data, buf)
    return bytes(buf[:n])


class Oracle:
    """One receiver context (its own ICAO filter, clock and counters)."""

    def __init__(self, fmt=FMT_UC8, preamble_threshold=58, nfix_crc=1, mode_ac=0, dc_filter=False, sc16q11_table_bits=0):
        self._h = lib().orc_create(fmt, preamble_threshold, nfix_crc, mode_ac)
        if not self._h:
            raise ValueError("orc_create rejected the configuration")
        if dc_filter:
            lib().orc_set_dc_filter(self._h, 1)
        self.sc16q11_table_bits = sc16q11_table_bits
        if sc16q11_table_bits: # a reference built with -DSC16Q11_TABLE_BITS=n (convert.c:264-328)
            lib().orc_set_sc16q11_table_bits(self._h, sc16q11_table_bits)
        self.fmt = fmt

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def bytes_per_sample(self):
        return 2 if self.fmt == FMT_UC8 else 4

    def set_recently_dropped(self, on):
        """Modes.stats_15min.samples_dropped != 0 (demod_2400.c:285-290) for the buffers demodulated from now on."""
        lib().orc_set_recently_dropped(self._h, 1 if on else 0)

    def replay_fields(self, iq, cap):
        """replay() that also decodes the header fields: returns (messages, fields, stats)."""
        self._fields = np.zeros(cap, dtype=FIELDS_DTYPE)
        lib().orc_set_fields_out(self._h, self._fields.ctypes.data, cap)
        msgs, st = self.replay(iq, cap=cap)
        lib().orc_set_fields_out(self._h, None, 0)
        return msgs, self._fields[: len(msgs)].copy(), st

    def replay(self, iq, cap=None, want_means=False):
        """Run a whole capture (bytes-like / uint8 array).  Returns (messages, stats[, means])."""
        iq = np.ascontiguousarray(np.frombuffer(iq, dtype=np.uint8) if not isinstance(iq, np.ndarray)
                                  else iq.view(np.uint8).reshape(-1))
        nsamples = iq.size // self.bytes_per_sample
        if cap is None:
            cap = max(1024, nsamples // 100)
        nbuf_max = nsamples // CHUNK + 1
        means = np.zeros((nbuf_max, 2), dtype=np.float64) if want_means else None
        while True:
            out = np.zeros(cap, dtype=MESSAGE_DTYPE)
            n = C.c_size_t(0)
            lib().orc_replay(self._h, iq.ctypes.data, nsamples, out.ctypes.data, cap, C.byref(n),
                             means.ctypes.data if want_means else None, nbuf_max)
            if n.value <= cap:
                break
            raise RuntimeError(f"oracle produced {n.value} messages, cap {cap}: use a fresh Oracle "
                               "with a larger cap (contexts are stateful)")
        msgs = out[: n.value]
        return (msgs, self.stats(), means) if want_means else (msgs, self.stats())

    def stats(self):
        st = Stats()
        lib().orc_get_stats(self._h, C.byref(st))
        return st.as_dict()

    def convert(self, iq, nsamples):
        iq = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
        mag = np.zeros(nsamples, dtype=np.uint16)
        ml, mp = C.c_double(), C.c_double()
        lib().orc_convert(self._h, iq.ctypes.data, mag.ctypes.data, nsamples, C.byref(ml), C.byref(mp))
        return mag, ml.value, mp.value

    def score(self, msg: bytes, validbits=None):
        return lib().orc_score(self._h, bytes(msg), validbits if validbits is not None else len(msg) * 8)

    def diagnose(self, syndrome, bitlen):
        bits = (C.c_int * 2)()
        n = lib().orc_diagnose(self._h, syndrome, bitlen, C.byref(bits))
        return n, list(bits)

    def filter_add(self, addr):
        lib().orc_filter_add(self._h, addr)

    def filter_test(self, addr):
        return bool(lib().orc_filter_test(self._h, addr))

    def demod_buffer(self, data, sample_ts=0, sys_ts=0, mean_level=0.0, mean_power=0.0, cap=4096):
        data = np.ascontiguousarray(data, dtype=np.uint16)
        out = np.zeros(cap, dtype=MESSAGE_DTYPE)
        n = C.c_size_t(0)
        lib().orc_demod_buffer(self._h, data.ctypes.data, data.size, sample_ts, sys_ts, mean_level,
                               mean_power, out.ctypes.data, cap, C.byref(n))
        return out[: n.value]


def fields_of(message):
    """orc_fields_of: the decoded fields of one Mode S message record (MESSAGE_DTYPE scalar)."""
    rec = np.zeros(1, dtype=MESSAGE_DTYPE)
    rec[0] = message
    out = np.zeros(1, dtype=FIELDS_DTYPE)
    lib().orc_fields_of.restype = None
    lib().orc_fields_of.argtypes = [C.c_void_p, C.c_void_p]
    lib().orc_fields_of(rec.ctypes.data, out.ctypes.data)
    return out[0]


FIELDS_FLOAT_DTYPE = np.dtype(
    [("gs_v0", "<f4"), ("gs_v2", "<f4"), ("gs_selected", "<f4"), ("heading", "<f4"), ("track_rate", "<f4"),
     ("roll", "<f4"), ("nav_qnh", "<f4"), ("nav_heading", "<f4"), ("mach", "<f8"), ("gs_valid", "u1"),
     ("heading_valid", "u1"), ("heading_type", "u1"), ("track_rate_valid", "u1"), ("roll_valid", "u1"),
     ("mach_valid", "u1"), ("nav_qnh_valid", "u1"), ("nav_heading_valid", "u1")], align=True)


def fields_float_of(message):
    """orc_fields_float_of: the float-valued members of struct modesMessage, computed from the message bytes where
    and how the reference computes them."""
    rec = np.zeros(1, dtype=MESSAGE_DTYPE)
    rec[0] = message
    out = np.zeros(1, dtype=FIELDS_FLOAT_DTYPE)
    lib().orc_fields_float_of.restype = None
    lib().orc_fields_float_of.argtypes = [C.c_void_p, C.c_void_p]
    lib().orc_fields_float_of(rec.ctypes.data, out.ctypes.data)
    return out[0]


def checksum(msg: bytes) -> int:
    lib().orc_create(0, 58, 0, 0)  # makes sure the static tables exist (leaks one tiny ctx once)
    return lib().orc_checksum(bytes(msg), len(msg) * 8)


def uc8_table():
    return np.ctypeslib.as_array(lib().orc_uc8_table(), shape=(65536,)).copy()


def slice_bytes(mag, j, try_phase, nbytes):
    mag = np.ascontiguousarray(mag, dtype=np.uint16)
    out = np.zeros(nbytes, dtype=np.uint8)
    lib().orc_slice(mag.ctypes.data, j, try_phase, nbytes, out.ctypes.data)
    return out
