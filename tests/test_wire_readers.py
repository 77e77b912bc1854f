"""Round 6, SURVEY.md 8(f) rank 2 finished: modesQueueOutput's forwarding rule (net_io.c:1263-1290), the bytes as
received for --net-verbatim (mode_s.c:427-429, net_io.c:775,874), and the readers of both wire formats
(net_io.c:1486-1627 + the scanner at :2504-2569, decodeHexMessage :1656-1764) -- host C of libmsd_host.so against
hand-built known answers, public frames and the oracle's messages.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from test_wire_formats import make


class BeastReader(C.Structure):
    _fields_ = [("buf", C.c_uint8 * 256), ("len", C.c_size_t), ("mode_ac", C.c_int), ("frames", C.c_uint64),
                ("modeac_ignored", C.c_uint64), ("other_frames", C.c_uint64), ("garbage_bytes", C.c_uint64)]


@pytest.fixture(scope="module")
def wire(pkg):
    host = C.CDLL(os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "libmsd_host.so"))
    SINK = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
    host.msd_wire_forwards.restype = C.c_int
    host.msd_wire_forwards.argtypes = [C.c_void_p, C.c_int]
    host.msd_wire_verbatim.restype = C.c_int
    host.msd_wire_verbatim.argtypes = [C.c_void_p, C.c_void_p]
    host.msd_avr_line_out.restype = C.c_size_t
    host.msd_avr_line_out.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
    host.msd_beast_frame_out.restype = C.c_size_t
    host.msd_beast_frame_out.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    host.msd_beast_reader_init.argtypes = [C.c_void_p, C.c_int]
    host.msd_beast_reader_feed.restype = C.c_size_t
    host.msd_beast_reader_feed.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, SINK, C.c_void_p]
    host.msd_avr_parse_line.restype = C.c_int
    host.msd_avr_parse_line.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
    dt = pkg.capi.MESSAGE_DTYPE

    class W:
        @staticmethod
        def forwards(rec, verbatim):
            rec = np.ascontiguousarray(rec).reshape(1)
            return bool(host.msd_wire_forwards(rec.ctypes.data, int(verbatim)))

        @staticmethod
        def verbatim(rec):
            rec = np.ascontiguousarray(rec).reshape(1)
            out = (C.c_uint8 * 14)()
            n = host.msd_wire_verbatim(rec.ctypes.data, out)
            return n, bytes(out[: int(rec["msgbits"][0]) // 8])

        @staticmethod
        def avr_out(rec, mlat, verbatim):
            rec = np.ascontiguousarray(rec).reshape(1)
            buf = C.create_string_buffer(64)
            n = host.msd_avr_line_out(rec.ctypes.data, int(mlat), int(verbatim), buf)
            return buf.raw[:n]

        @staticmethod
        def beast_out(rec, verbatim):
            rec = np.ascontiguousarray(rec).reshape(1)
            buf = (C.c_uint8 * 64)()
            n = host.msd_beast_frame_out(rec.ctypes.data, int(verbatim), buf)
            return bytes(buf[:n])

        @staticmethod
        def read_beast(stream, mode_ac=True, chunks=None):
            r = BeastReader()
            host.msd_beast_reader_init(C.byref(r), int(mode_ac))
            got = []

            def sink(p, user):
                got.append(np.frombuffer(C.string_at(p, dt.itemsize), dtype=dt)[0].copy())

            cb = SINK(sink)
            pos, delivered = 0, 0
            sizes = iter(chunks) if chunks is not None else None
            while pos < len(stream):
                k = next(sizes) if sizes is not None else len(stream)
                delivered += host.msd_beast_reader_feed(C.byref(r), stream[pos:pos + k], min(k, len(stream) - pos), cb, None)
                pos += k
            assert delivered == len(got) == r.frames
            return (np.array(got, dtype=dt) if got else np.zeros(0, dtype=dt)), r

        @staticmethod
        def parse_avr(line, mode_ac=True, keep_ts=True):
            rec = np.zeros(1, dtype=dt)
            ok = host.msd_avr_parse_line(line if isinstance(line, bytes) else line.encode(), int(mode_ac), int(keep_ts), rec.ctypes.data)
            return rec[0] if ok else None

    return W


def crc24(raw):
    """modesChecksum by long division (crc.c:31,67-82)."""
    rem = 0
    for byte in raw[:-3]:
        rem ^= byte << 16
        for _ in range(8):
            rem = ((rem << 1) ^ 0xFFF409) & 0xFFFFFF if rem & 0x800000 else (rem << 1) & 0xFFFFFF
    return rem ^ int.from_bytes(raw[-3:], "big")


def flipped(raw, bits):
    b = bytearray(raw)
    for i in bits:
        b[i >> 3] ^= 0x80 >> (i & 7)
    return bytes(b)


def corrected_record(pkg, clean_hex, bits):
    """What the demodulator hands over for a frame received with `bits` flipped: the repaired bytes, the syndrome of the received ones."""
    clean = bytes.fromhex(clean_hex)
    rec = make(pkg, clean_hex, ts=0x123456, level=0.1, msgtype=clean[0] >> 3)
    rec = rec.copy()
    rec["correctedbits"] = len(bits)
    rec["crc"] = crc24(flipped(clean, bits))
    return rec, flipped(clean, bits)


def test_forwarding_rule(pkg, wire):
    """net_io.c:1272-1285: two repaired bits only with --net-verbatim; everything else always."""
    for nbits in (0, 1, 2):
        rec, _ = corrected_record(pkg, "8D4840D6202CC371C32CE0576098", [40, 77][:nbits])
        assert wire.forwards(rec, False) == (nbits < 2)
        assert wire.forwards(rec, True)
        assert bool(wire.avr_out(rec, True, False)) == (nbits < 2) and bool(wire.beast_out(rec, False)) == (nbits < 2)


@pytest.mark.parametrize("clean_hex,bits", [
    ("8D4840D6202CC371C32CE0576098", [40]),          # DF17, one bit in ME
    ("8D4840D6202CC371C32CE0576098", [9]),           # ... in AA
    ("8D4840D6202CC371C32CE0576098", [111]),         # ... the last parity bit
    ("8D4840D6202CC371C32CE0576098", [5, 100]),      # two bits (--aggressive)
    ("8D4840D6202CC371C32CE0576098", [33, 34]),      # adjacent
    ("5D4840D6F1B2A3"[:8] + "%06X" % 0, [20]),       # placeholder, replaced below
])
def test_verbatim_bytes_known_answers(pkg, wire, clean_hex, bits):
    if clean_hex.startswith("5D") and len(clean_hex) == 14:
        # a DF11 all-call reply from interrogator 5: parity = remainder ^ IID (crc & 0x7f carries the IID)
        head = bytes.fromhex("5D4840D6")
        clean_hex = (head + (crc24(head + b"\0\0\0") ^ 5).to_bytes(3, "big")).hex()
        assert crc24(bytes.fromhex(clean_hex)) == 5
    else:
        assert crc24(bytes.fromhex(clean_hex)) == 0   # the public frame is clean
    rec, received = corrected_record(pkg, clean_hex, bits)
    n, out = wire.verbatim(rec)
    assert n == len(bits) and out == received
    # and both writers send exactly those bytes with --net-verbatim, the repaired ones without
    assert wire.avr_out(rec, False, True) == b"*" + received.hex().upper().encode() + b";\n"
    if len(bits) < 2:
        assert wire.avr_out(rec, False, False) == b"*" + bytes.fromhex(clean_hex).hex().upper().encode() + b";\n"
    assert wire.beast_out(rec, True)[9:] == received.replace(b"\x1a", b"\x1a\x1a")
    clean = make(pkg, clean_hex, ts=1, msgtype=17)
    assert wire.verbatim(clean) == (0, bytes.fromhex(clean_hex))


def test_verbatim_on_the_oracles_repaired_messages(pkg, oracle, wire):
    """Every message the oracle repaired (--aggressive: one and two bits, DF11 and DF17): the bytes put back have the
    message's own pre-repair syndrome and differ from the repaired bytes in exactly `correctedbits` places -- there is
    one such pattern (crc.c:184-354), so this is the received message."""
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=4242, msgs_per_sec=6000, flip_permille=300), 40 * 131072, nthreads=2)
    msgs, _ = oracle.Oracle(oracle.FMT_UC8, 58, 2, 0).replay(iq, cap=1 << 17)
    seen = {1: 0, 2: 0}
    for m in msgs:
        if not m["correctedbits"]:
            continue
        n, raw = wire.verbatim(m)
        nbytes = int(m["msgbits"]) // 8
        assert n == int(m["correctedbits"])
        assert crc24(raw) == int(m["crc"])
        diff = sum(bin(a ^ b).count("1") for a, b in zip(raw, bytes(m["msg"][:nbytes])))
        assert diff == n
        seen[n] += 1
    assert seen[1] > 50 and seen[2] > 5, seen


def test_beast_reader_round_trip_with_escapes_and_any_chunking(pkg, oracle, wire):
    """writer -> reader on the oracle's messages plus frames full of 0x1A, fed in pieces of every size (a frame, a pair of
    0x1A bytes, the type byte split across calls)."""
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=1091, msgs_per_sec=3000, ac_per_sec=800), 12 * 131072, nthreads=2)
    msgs, _ = oracle.Oracle(oracle.FMT_UC8, 58, 1, 1).replay(iq, cap=1 << 17)
    recs = [m for m in msgs]
    recs.append(make(pkg, "1A1A1A1A1A1A1A", ts=0x1A1A1A1A1A1A, level=(26 / 255.0) ** 2, msgtype=3))
    recs.append(make(pkg, "8D1A1A1A1A1A1A1A1A1A1A1A1A1A", ts=0x00001A00001A, level=0.5, msgtype=17))
    recs.append(make(pkg, "1A1A", ts=0x1A, msgtype=32))
    stream = b"".join(oracle.beast_frame(m) for m in recs)
    rng = np.random.default_rng(5)
    for chunks in (None, [1] * len(stream), list(rng.integers(1, 9, size=len(stream))), list(rng.integers(30, 400, size=len(stream)))):
        got, r = wire.read_beast(stream, True, chunks)
        assert len(got) == len(recs) and r.garbage_bytes == 0 and r.len == 0
        for g, m in zip(got, recs):
            nbytes = int(m["msgbits"]) // 8
            assert int(g["timestampMsg"]) == int(m["timestampMsg"]) & 0xFFFFFFFFFFFF
            assert int(g["msgbits"]) == int(m["msgbits"]) and bytes(g["msg"][:nbytes]) == bytes(m["msg"][:nbytes])
            sig = oracle.beast_frame(m).replace(b"\x1a\x1a", b"\x1a")[8]
            assert float(g["signalLevel"]) == (sig / 255.0) ** 2          # net_io.c:1563-1565
            if nbytes == 2:
                modeac = (int(m["msg"][0]) << 8) | int(m["msg"][1])
                assert int(g["msgtype"]) == 32 and int(g["addr"]) == (modeac & 0xFF7F) | (1 << 24)   # mode_ac.c:168-202
            else:
                assert int(g["msgtype"]) == m["msg"][0] >> 3
                assert int(g["crc"]) == crc24(bytes(m["msg"][:nbytes]))
    # what the oracle's own records say for the messages that needed no repair: same address, same checksum
    got, _ = wire.read_beast(b"".join(oracle.beast_frame(m) for m in msgs), True)
    clean = msgs["correctedbits"] == 0
    assert np.array_equal(got["addr"][clean], msgs["addr"][clean]) and np.array_equal(got["crc"][clean], msgs["crc"][clean])
    assert clean.sum() > 300 and (msgs["msgtype"] == 32).sum() > 20


def test_beast_reader_resynchronises_and_counts(pkg, wire):
    """net_io.c:2504-2569: garbage in front of a frame is skipped, a 0x1A followed by an unknown type is not a frame, type '1'
    frames only count with mode_ac off (net_io.c:1500-1508), the frames this reader has no use for ('4', '5') are consumed whole."""
    f3 = bytes.fromhex("1a33" "0123456789ab" "80" "8d4840d6202cc371c32ce0576098")
    f1 = bytes.fromhex("1a31" "000000000005" "00" "7700")
    f5 = bytes.fromhex("1a35" "000000000000" "00") + bytes(range(14))
    stream = b"xyz" + f3 + b"\x1a\x07" + f1 + f5 + b"\x1a"      # ends inside a frame start
    got, r = wire.read_beast(stream, mode_ac=False)
    assert len(got) == 1 and bytes(got[0]["msg"]) == bytes.fromhex("8d4840d6202cc371c32ce0576098")
    assert int(got[0]["addr"]) == 0x4840D6 and int(got[0]["crc"]) == 0 and int(got[0]["timestampMsg"]) == 0x0123456789AB
    assert (r.modeac_ignored, r.other_frames, r.garbage_bytes, r.len) == (1, 1, 3 + 2, 1)
    got, r = wire.read_beast(stream, mode_ac=True)
    assert len(got) == 2 and int(got[1]["msgtype"]) == 32 and int(got[1]["addr"]) == (0x7700 & 0xFF7F) | (1 << 24)


def test_avr_reader_known_answers(pkg, wire):
    """decodeHexMessage's framing (net_io.c:1656-1740)."""
    m = wire.parse_avr("*8D4840D6202CC371C32CE0576098;")
    assert m is not None and bytes(m["msg"]) == bytes.fromhex("8D4840D6202CC371C32CE0576098") and int(m["addr"]) == 0x4840D6
    assert int(m["msgtype"]) == 17 and int(m["msgbits"]) == 112 and int(m["crc"]) == 0 and int(m["timestampMsg"]) == 0
    m = wire.parse_avr("  @0123456789AB8d4840d6202cc371c32ce0576098;\r\n")
    assert m is not None and int(m["timestampMsg"]) == 0x0123456789AB and float(m["signalLevel"]) == 0.0
    assert wire.parse_avr("@0123456789AB8d4840d6202cc371c32ce0576098;", keep_ts=False)["timestampMsg"] == 0
    m = wire.parse_avr("<0123456789AB808D4840D6202CC371C32CE0576098;")
    assert m is not None and float(m["signalLevel"]) == (0x80 / 255.0) ** 2
    m = wire.parse_avr(":5D4840D6A1B2C3;")
    assert m is not None and int(m["msgbits"]) == 56 and int(m["msgtype"]) == 11 and int(m["iid"]) == crc24(bytes.fromhex("5D4840D6A1B2C3")) & 0x7F
    m = wire.parse_avr("*20000F1F684A6C;")     # DF4: the checksum is the address
    assert int(m["addr"]) == crc24(bytes.fromhex("20000F1F684A6C")) == int(m["crc"])
    assert wire.parse_avr("*7700;") is not None and wire.parse_avr("*7700;", mode_ac=False) is None
    for bad in ("*8D4840D6202CC371C32CE0576098", "#8D4840D6;", "*8D4840D6202CC371C32CE05760;", "*8D4840D6202CC371C32CE05760ZZ;", ";", ""):
        assert wire.parse_avr(bad) is None, bad


def test_avr_writer_reader_round_trip(pkg, oracle, wire):
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=77, msgs_per_sec=3000, ac_per_sec=500), 6 * 131072, nthreads=2)
    msgs, _ = oracle.Oracle(oracle.FMT_UC8, 58, 1, 1).replay(iq, cap=1 << 17)
    assert len(msgs) > 100
    for m in msgs[::3]:
        g = wire.parse_avr(oracle.avr_line(m, True))
        nbytes = int(m["msgbits"]) // 8
        assert g is not None and bytes(g["msg"][:nbytes]) == bytes(m["msg"][:nbytes])
        assert int(g["timestampMsg"]) == int(m["timestampMsg"]) & 0xFFFFFFFFFFFF
