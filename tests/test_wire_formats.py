"""AVR raw lines and Beast binary frames of accepted messages: the host C
encoders of libmsd_host.so against hand-built known answers from the format definitions
(net_io.c:769-835, 870-896) and against the oracle's restatement on golden message lists."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import golden_names, load_golden


@pytest.fixture(scope="module")
def wire(pkg):
    host = C.CDLL(os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "libmsd_host.so"))
    host.msd_avr_line.restype = C.c_size_t
    host.msd_avr_line.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    host.msd_beast_frame.restype = C.c_size_t
    host.msd_beast_frame.argtypes = [C.c_void_p, C.c_void_p]

    class W:
        @staticmethod
        def avr(rec, mlat):
            rec = np.ascontiguousarray(rec).reshape(1)
            buf = C.create_string_buffer(64)
            n = host.msd_avr_line(rec.ctypes.data, int(mlat), buf)
            return buf.raw[:n]

        @staticmethod
        def beast(rec):
            rec = np.ascontiguousarray(rec).reshape(1)
            buf = (C.c_uint8 * 64)()
            n = host.msd_beast_frame(rec.ctypes.data, buf)
            return bytes(buf[:n])

    return W


def make(pkg, hexmsg, ts=0, level=0.0, msgtype=17):
    rec = np.zeros(1, dtype=pkg.capi.MESSAGE_DTYPE)
    raw = bytes.fromhex(hexmsg)
    rec["msg"][0, : len(raw)] = np.frombuffer(raw, dtype=np.uint8)
    rec["msgbits"] = 8 * len(raw)
    rec["timestampMsg"] = ts
    rec["signalLevel"] = level
    rec["msgtype"] = msgtype
    return rec[0]


def test_avr_known_answers(pkg, wire):
    m = make(pkg, "8D4840D6202CC371C32CE0576098", ts=0x0123456789AB)
    assert wire.avr(m, False) == b"*8D4840D6202CC371C32CE0576098;\n"
    assert wire.avr(m, True) == b"@0123456789AB8D4840D6202CC371C32CE0576098;\n"
    short = make(pkg, "5D4840D6A1B2C3", ts=0)
    assert wire.avr(short, True) == b"*5D4840D6A1B2C3;\n"  # no timestamp: '*' even with --mlat (net_io.c:877)
    ac = make(pkg, "7700", ts=5, msgtype=32)
    assert wire.avr(ac, True) == b"@0000000000057700;\n"


def test_beast_known_answers(pkg, wire):
    m = make(pkg, "8D4840D6202CC371C32CE0576098", ts=0x0123456789AB, level=0.25)
    assert wire.beast(m) == bytes.fromhex("1a33" "0123456789ab" "80" "8d4840d6202cc371c32ce0576098")  # sqrt(.25)*255 = 127.5 -> 128
    # every 0x1A after the type byte is doubled: timestamp, signal (26/255)^2, payload
    esc = make(pkg, "1A4840D61A2CC3", ts=0x001A00001A00, level=(26 / 255.0) ** 2)
    assert wire.beast(esc) == bytes.fromhex("1a32" "00" "1a1a" "0000" "1a1a" "00" "1a1a" "1a1a" "4840d6" "1a1a" "2cc3")
    weak = make(pkg, "5D4840D6A1B2C3", ts=1, level=1e-9)
    assert wire.beast(weak)[8] == 1  # a non-zero level never encodes as 0 (net_io.c:820-821)
    loud = make(pkg, "5D4840D6A1B2C3", ts=1, level=1.5)
    assert wire.beast(loud)[8] == 255
    ac = make(pkg, "7700", ts=5, msgtype=32)
    assert wire.beast(ac) == bytes.fromhex("1a31" "000000000005" "00" "7700")


@pytest.mark.parametrize("name", golden_names())
def test_wire_formats_match_the_oracle_on_golden_lists(pkg, oracle, wire, name):
    meta, z = load_golden(name)
    n = len(z["timestampMsg"])
    recs = np.zeros(n, dtype=pkg.capi.MESSAGE_DTYPE)
    for f in recs.dtype.names:
        if f in z.files:
            recs[f] = z[f]
    step = max(1, n // 400)
    for rec in recs[::step]:
        for mlat in (False, True):
            assert wire.avr(rec, mlat) == oracle.avr_line(rec, mlat)
        assert wire.beast(rec) == oracle.beast_frame(rec)
