"""Header fields of accepted messages (first stage): the altitude / identity
codes against an independent Python statement of the encodings (exhaustive over all 8192 codes) and
published Gillham values, the per-format fields against hand-built messages, and the host entry
point (the code the emit kernel shares) against the oracle on replayed captures."""
import ctypes as C

import numpy as np
import pytest

from helpers import load_golden

FIELD_NAMES = ("altitude_baro", "AC", "ID", "squawk", "altitude_baro_valid", "altitude_baro_unit", "squawk_valid",
               "airground", "alert", "alert_valid", "spi", "spi_valid", "CA", "CC", "CF", "DR", "FS", "KE", "ND", "RI",
               "SL", "UM", "VS", "source", "addrtype", "imf", "addr", "metype", "mesub", "cpr_valid", "cpr_type",
               "cpr_odd", "nic_b_valid", "nic_b", "callsign_valid", "callsign", "cpr_lat", "cpr_lon", "altitude_geom",
               "altitude_geom_valid", "altitude_geom_unit", "category", "category_valid", "nac_v_valid", "nac_v",
               "velocity_valid", "heading_valid", "ew_vel", "ns_vel", "heading_raw", "heading_type", "movement", "ias",
               "tas", "ias_valid", "tas_valid", "baro_rate_valid", "geom_rate_valid", "baro_rate", "geom_rate",
               "geom_delta", "geom_delta_valid", "emergency_valid", "emergency", "nav_valid", "nav_altitude_source",
               "nav_modes", "nav_heading_type", "acc_valid", "nac_p", "nic_baro", "nic_a", "nic_c", "gva", "sda", "sil",
               "sil_type", "cc_antenna_offset", "commb_format", "nav_heading_raw", "nav_qnh_raw", "nav_mcp_altitude",
               "nav_fms_altitude", "opstatus", "roll_q", "track_rate_q", "gs", "mach_raw", "commb_valid")


def code_bits(code13):
    b = lambda n: (code13 >> n) & 1  # noqa: E731
    return dict(C1=b(12), A1=b(11), C2=b(10), A2=b(9), C4=b(8), A4=b(7), B1=b(5), D1=b(4), B2=b(3), D2=b(2), B4=b(1),
                D4=b(0))


def squawk_py(code13):
    """identity / altitude code -> the four octal digits A B C D, one hex digit each"""
    k = code_bits(code13)
    digit = lambda x: k[x + "4"] * 4 + k[x + "2"] * 2 + k[x + "1"]  # noqa: E731
    return (digit("A") << 12) | (digit("B") << 8) | (digit("C") << 4) | digit("D")


def gray_to_bin(bits):
    acc, n = 0, 0
    for g in bits:
        acc ^= g
        n = (n << 1) | acc
    return n


def gillham_feet_py(code13):
    """Altitude code without Q: D2 D4 A1 A2 A4 B1 B2 B4 is a Gray-coded count of 500 ft, C1 C2 C4 a Gray-coded
    count of 100 ft (1..4 and 7 standing for 5) that runs backwards in odd 500 ft blocks; offset -1300 ft."""
    k = code_bits(code13)
    if k["D1"] or not (k["C1"] or k["C2"] or k["C4"]):
        return None
    n500 = gray_to_bin([k[x] for x in ("D2", "D4", "A1", "A2", "A4", "B1", "B2", "B4")])
    n100 = gray_to_bin([k["C1"], k["C2"], k["C4"]])
    if n100 == 7:
        n100 = 5
    elif n100 in (5, 6):
        return None
    if n500 & 1:
        n100 = 6 - n100
    return n500 * 500 + n100 * 100 - 1300


def ac13_feet_py(ac13):
    if ac13 & 0x40:  # M bit: metric, never decoded
        return None
    if ac13 & 0x10:  # Q bit: 25 ft steps
        return ((((ac13 & 0x1F80) >> 2) | ((ac13 & 0x20) >> 1) | (ac13 & 0xF)) * 25) - 1000
    return gillham_feet_py(ac13)


def msg_record(pkg, df, payload27=0):
    """A message record with DF and the 27 bits behind it (bits 6..32) set; the rest zero."""
    rec = np.zeros(1, dtype=pkg.capi.MESSAGE_DTYPE)
    word = (df << 27) | (payload27 & 0x7FFFFFF)
    rec["msg"][0, :4] = np.frombuffer(int(word).to_bytes(4, "big"), dtype=np.uint8)
    rec["msgtype"] = df
    rec["msgbits"] = 112 if df >= 16 else 56
    return rec[0]


def test_identity_and_altitude_codes_exhaustive(pkg, oracle):
    lib = oracle.lib()
    for code in range(8192):
        sq = squawk_py(code)
        assert lib.orc_decode_id13(code) == sq
        f = pkg.capi.decode_fields(msg_record(pkg, 5, code))
        assert (f["ID"], f["squawk"], f["squawk_valid"]) == (code, sq if code else 0, 1 if code else 0)
        want = ac13_feet_py(code)
        unit = C.c_int(0)
        got = lib.orc_decode_ac13(code, C.byref(unit))
        assert got == (-9999 if want is None else want), code
        assert unit.value == (1 if code & 0x40 else 0)
        f = pkg.capi.decode_fields(msg_record(pkg, 20, code))
        assert f["AC"] == code
        if code == 0:
            assert (f["altitude_baro"], f["altitude_baro_valid"], f["altitude_baro_unit"]) == (0, 0, 0)
        else:
            assert f["altitude_baro"] == (-9999 if want is None else want), code
            assert f["altitude_baro_valid"] == (0 if want is None else 1)
            assert f["altitude_baro_unit"] == (1 if code & 0x40 else 0)


def test_gillham_table_properties_and_published_values(oracle):
    lib = oracle.lib()
    seen = {}
    for i in range(4096):
        a = (i & 0o7) | ((i & 0o70) << 1) | ((i & 0o700) << 2) | ((i & 0o7000) << 3)
        c = lib.orc_mode_a_to_mode_c(a)
        if c != -9999:
            assert c not in seen
            seen[c] = a
    assert sorted(seen) == list(range(-12, 1268))  # 1280 codes: -1200 ft .. 126700 ft in 100 ft steps
    for c in range(-12, 1267):  # neighbouring altitudes differ in exactly one pulse
        assert bin(seen[c] ^ seen[c + 1]).count("1") == 1
    for feet, octal in {-1200: 0o0040, -1000: 0o0020, 0: 0o0620, 100: 0o0630, 500: 0o0220}.items():  # ABCD
        hexcoded = ((octal >> 9) & 7) << 12 | ((octal >> 6) & 7) << 8 | ((octal >> 3) & 7) << 4 | (octal & 7)
        assert seen[feet // 100] == hexcoded, (feet, oct(octal))
    assert lib.orc_mode_a_to_mode_c(0x8620 | 0x0808) == lib.orc_mode_a_to_mode_c(0x0620)  # stray bits are ignored


def test_per_format_header_fields(pkg):
    # FS 0..7 for DF4/5/20/21 (mode_s.c:613-650): (airground, alert, spi, valid)
    fs_table = {0: (3, 0, 0, 1), 1: (1, 0, 0, 1), 2: (3, 1, 0, 1), 3: (1, 1, 0, 1), 4: (3, 1, 1, 1), 5: (3, 0, 1, 1),
                6: (0, 0, 0, 0), 7: (0, 0, 0, 0)}
    for df in (4, 5, 20, 21):
        for fs, (ag, alert, spi, valid) in fs_table.items():
            f = pkg.capi.decode_fields(msg_record(pkg, df, (fs << 24) | (0b10101 << 19) | (0b110011 << 13)))
            assert (f["FS"], f["airground"], f["alert"], f["spi"], f["alert_valid"], f["spi_valid"]) == \
                   (fs, ag, alert, spi, valid, valid)
            assert (f["DR"], f["UM"]) == (0b10101, 0b110011)
    for df in (11, 17):  # CA (mode_s.c:577-597): 1..3 leave airground alone
        for ca, ag in {0: 3, 1: 0, 2: 0, 3: 0, 4: 1, 5: 2, 6: 3, 7: 3}.items():
            f = pkg.capi.decode_fields(msg_record(pkg, df, ca << 24))
            assert (f["CA"], f["airground"]) == (ca, ag)
    for df in (0, 16):  # VS bit 6, CC bit 7 (DF0), SL bits 9-11, RI bits 14-17
        for vs in (0, 1):
            f = pkg.capi.decode_fields(msg_record(pkg, df, (vs << 26) | (1 << 25) | (0b101 << 21) | (0b1001 << 15)))
            assert (f["VS"], f["airground"], f["SL"], f["RI"]) == (vs, 1 if vs else 3, 0b101, 0b1001)
            assert f["CC"] == (1 if df == 0 else 0)
    assert pkg.capi.decode_fields(msg_record(pkg, 18, 0b110 << 24))["CF"] == 0b110
    for df in (24, 27, 31):  # KE is bit 4 (inside the 5-bit DF number), ND bits 5-8
        f = pkg.capi.decode_fields(msg_record(pkg, df, 0b011 << 24))
        assert (f["KE"], f["ND"]) == ((df >> 1) & 1, ((df & 1) << 3) | 0b011)
    f = pkg.capi.decode_fields(msg_record(pkg, 11, 5 << 24))
    assert all(f[x] == 0 for x in ("AC", "ID", "squawk", "FS", "DR", "UM", "RI", "SL", "VS", "altitude_baro_valid"))


def test_mode_ac_reply_fields_and_altitude_carry(pkg):
    def ac(code):
        rec = np.zeros(1, dtype=pkg.capi.MESSAGE_DTYPE)
        rec["msgtype"], rec["msgbits"] = 32, 16
        rec["msg"][0, 0], rec["msg"][0, 1] = code >> 8, code & 0xFF
        return rec[0]
    a = pkg.capi.decode_fields(ac(0x0620))  # the Mode C code of 0 ft
    assert (a["squawk"], a["squawk_valid"], a["spi"], a["spi_valid"]) == (0x0620, 1, 0, 1)
    assert (a["altitude_baro"], a["altitude_baro_valid"]) == (0, 1)
    hi = pkg.capi.decode_fields(ac(0x0630))
    assert (hi["altitude_baro"], hi["altitude_baro_valid"]) == (100, 1)
    b = pkg.capi.decode_fields(ac(0x7777), carry=hi)  # not a Mode C code: the buffer's record keeps the altitude
    assert (b["squawk"], b["altitude_baro"], b["altitude_baro_valid"]) == (0x7777, 100, 1)
    assert pkg.capi.decode_fields(ac(0x7777))["altitude_baro_valid"] == 0
    d = pkg.capi.decode_fields(ac(0x06A0), carry=b)  # ident pulse: SPI, no altitude of its own (mode_ac.c:186-197)
    assert (d["spi"], d["squawk"], d["altitude_baro"], d["altitude_baro_valid"]) == (1, 0x0620, 100, 1)


def es_record(pkg, hexmsg):
    rec = np.zeros(1, dtype=pkg.capi.MESSAGE_DTYPE)
    raw = bytes.fromhex(hexmsg)
    rec["msg"][0, : len(raw)] = np.frombuffer(raw, dtype=np.uint8)
    rec["msgtype"] = raw[0] >> 3
    rec["msgbits"] = 8 * len(raw)
    rec["addr"] = int.from_bytes(raw[1:4], "big")
    return rec[0]


def test_extended_squitter_published_examples(pkg):
    """Messages and decoded values from the public ADS-B decoding literature (J. Sun, 'The 1090 MHz riddle')."""
    f = pkg.capi.decode_fields(es_record(pkg, "8D4840D6202CC371C32CE0576098"))  # identification
    assert (f["metype"], f["callsign_valid"], f["callsign"], f["category"], f["category_valid"]) == \
           (4, 1, b"KLM1023 ", 0xA0, 1)
    assert (f["source"], f["addrtype"], f["addr"], f["CA"], f["airground"]) == (7, 0, 0x4840D6, 5, 2)
    even = pkg.capi.decode_fields(es_record(pkg, "8D40621D58C382D690C8AC2863A7"))  # airborne position, even frame
    odd = pkg.capi.decode_fields(es_record(pkg, "8D40621D58C386435CC412692AD6"))   # odd frame of the pair
    for f, is_odd, lat, lon in ((even, 0, 93000, 51372), (odd, 1, 74158, 50194)):
        assert (f["metype"], f["cpr_valid"], f["cpr_type"], f["cpr_odd"], f["cpr_lat"], f["cpr_lon"]) == \
               (11, 1, 1, is_odd, lat, lon)
        assert (f["altitude_baro"], f["altitude_baro_valid"], f["altitude_geom_valid"]) == (38000, 1, 0)
        assert (f["alert_valid"], f["alert"], f["spi_valid"], f["spi"], f["nic_b_valid"]) == (1, 0, 1, 0, 1)
    v = pkg.capi.decode_fields(es_record(pkg, "8D485020994409940838175B284F"))  # velocity, subtype 1 (ground speed)
    assert (v["metype"], v["mesub"], v["velocity_valid"], v["ew_vel"], v["ns_vel"]) == (19, 1, 1, -8, -159)
    assert abs(np.hypot(-8.0, -159.0) - 159.20) < 0.01                       # 159.20 kt ...
    assert abs(np.degrees(np.arctan2(-8.0, -159.0)) % 360 - 182.88) < 0.01   # ... on track 182.88 deg
    assert (v["geom_rate_valid"], v["geom_rate"], v["baro_rate_valid"]) == (1, -832, 0)
    assert (v["geom_delta_valid"], v["geom_delta"]) == (1, 550)
    a = pkg.capi.decode_fields(es_record(pkg, "8DA05F219B06B6AF189400CBC33F"))  # velocity, subtype 3 (airspeed)
    assert (a["metype"], a["mesub"], a["heading_valid"], a["heading_type"], a["velocity_valid"]) == (19, 3, 1, 4, 0)
    assert abs(a["heading_raw"] * 360.0 / 1024.0 - 243.98) < 0.01
    assert (a["tas_valid"], a["tas"], a["ias_valid"]) == (1, 375, 0)
    assert (a["baro_rate_valid"], a["baro_rate"]) == (1, -2304)


def test_extended_squitter_df18_address_qualifiers(pkg):
    """DF18 control field and IMF bit (mode_s.c:1379-1426,770-792): where the address stops being an ICAO one."""
    def df18(cf, me_hex):
        return es_record(pkg, "%02X" % ((18 << 3) | cf) + "ABCDEF" + me_hex + "000000")
    airborne_imf = "%014X" % ((11 << 51) | (1 << 48))  # type 11, bit 8 set
    airborne = "%014X" % (11 << 51)
    table = {  # cf: (me, source, addrtype, imf, non_icao)
        0: (airborne_imf, 7, 1, 0, 0),  # ADS-B from a non-transponder device: bit 8 is NIC-B there
        1: (airborne, 7, 4, 0, 1),
        2: (airborne, 5, 3, 0, 0),
        5: (airborne, 5, 7, 0, 1),
        6: (airborne, 6, 2, 0, 0),
        4: (airborne, 7, 9, 0, 1),
    }
    for cf, (me, source, addrtype, imf, non_icao) in table.items():
        f = pkg.capi.decode_fields(df18(cf, me))
        assert (f["CF"], f["source"], f["addrtype"], f["imf"], f["addr"]) == \
               (cf, source, addrtype, imf, 0xABCDEF | (non_icao << 24)), cf
    f = pkg.capi.decode_fields(df18(2, airborne_imf))  # fine TIS-B with IMF: track file number instead of an address
    assert (f["source"], f["addrtype"], f["imf"], f["addr"], f["nic_b_valid"]) == (5, 6, 1, 0x1ABCDEF, 0)
    f = pkg.capi.decode_fields(df18(6, airborne_imf))  # ADS-R with IMF
    assert (f["source"], f["addrtype"], f["imf"], f["addr"]) == (6, 5, 1, 0x1ABCDEF)
    f = pkg.capi.decode_fields(df18(3, "%014X" % (1 << 55)))  # coarse TIS-B: only the IMF bit (bit 1 of ME) is looked at
    assert (f["source"], f["addrtype"], f["imf"], f["cpr_valid"], f["callsign_valid"]) == (5, 6, 1, 0, 0)
    f = pkg.capi.decode_fields(df18(0, airborne_imf))
    assert (f["nic_b_valid"], f["nic_b"]) == (1, 1)


def test_extended_squitter_surface_status_and_fault_pattern(pkg):
    me = (7 << 51) | (17 << 44) | (1 << 43) | (33 << 36) | (1 << 34) | (39195 << 17) | 110320  # type 7 surface position
    f = pkg.capi.decode_fields(es_record(pkg, "8C484175" + "%014X" % me + "000000"))
    assert (f["metype"], f["airground"], f["cpr_valid"], f["cpr_type"], f["cpr_odd"], f["cpr_lat"], f["cpr_lon"]) == \
           (7, 1, 1, 0, 1, 39195, 110320)
    assert (f["movement"], f["heading_valid"], f["heading_raw"], f["heading_type"]) == (17, 1, 33, 5)
    me = (28 << 51) | (1 << 48) | (2 << 45) | (0b1010101010101 << 32)  # aircraft status: emergency 2 and a squawk
    f = pkg.capi.decode_fields(es_record(pkg, "8D484175" + "%014X" % me + "000000"))
    assert (f["metype"], f["mesub"], f["emergency_valid"], f["emergency"], f["squawk_valid"]) == (28, 1, 1, 2, 1)
    assert f["squawk"] == squawk_py(0b1010101010101)
    me = (15 << 51) | (0x1F000 << 17)  # type 15, altitude 0, longitude 0, zeros in the latitude LSBs: the known fault
    f = pkg.capi.decode_fields(es_record(pkg, "8D484175" + "%014X" % me + "000000"))
    assert (f["metype"], f["cpr_valid"], f["cpr_lat"], f["altitude_baro_valid"]) == (15, 0, 0x1F000, 0)
    me = (21 << 51) | (0xC38 << 36) | (5 << 17) | 9  # type 21: geometric altitude
    f = pkg.capi.decode_fields(es_record(pkg, "8D484175" + "%014X" % me + "000000"))
    assert (f["altitude_geom_valid"], f["altitude_geom"], f["altitude_baro_valid"], f["cpr_valid"]) == (1, 38000, 0, 1)
    on_ground = pkg.capi.decode_fields(es_record(pkg, "8C484175" + "%014X" % ((11 << 51) | (0xC38 << 36)) + "000000"))
    assert (on_ground["CA"], on_ground["airground"], on_ground["altitude_baro_valid"]) == (4, 1, 0)  # CA 4: no altitude


@pytest.mark.parametrize("name", ["uc8_fix_modeac", "sc16q11_fix_modeac", "uc8_nofix"])
def test_host_decode_matches_oracle_on_replayed_captures(pkg, oracle, name):
    """msd_decode_fields (the code the emit kernel shares) against the oracle's independent restatement:
    the golden capture is regenerated from its seed, the oracle decodes fields during its replay, the
    host entry point decodes the same messages one by one (Mode A/C replies with the buffer's carry)."""
    meta, z = load_golden(name)
    fmt = {"uc8": pkg.FMT_UC8, "sc16": pkg.FMT_SC16, "sc16q11": pkg.FMT_SC16Q11}[meta["format"]]
    ofmt = {"uc8": oracle.FMT_UC8, "sc16": oracle.FMT_SC16, "sc16q11": oracle.FMT_SC16Q11}[meta["format"]]
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=meta["seed"], fmt=fmt, **meta["gen"]), meta["nsamples"])
    msgs, fields, _ = oracle.Oracle(ofmt, 58, meta["nfix_crc"], meta["mode_ac"]).replay_fields(iq, cap=1 << 17)
    assert len(msgs) == len(z["timestampMsg"]) and np.array_equal(msgs["timestampMsg"], z["timestampMsg"])
    # buffer of a Mode A/C reply: its F1 position is at most 131071, so f2_clock / 5 < 131072 * 5 + 244 ticks
    carry, carry_buf, nac = None, None, 0
    for m, want in zip(msgs, fields):
        if m["msgtype"] == 32:
            buf = (int(m["timestampMsg"]) - 244) // (131072 * 5)
            got = pkg.capi.decode_fields(m, carry if carry_buf == buf else None)
            carry, carry_buf = got, buf
            nac += 1
        else:
            got = pkg.capi.decode_fields(m)
        for f in FIELD_NAMES:
            assert got[f] == want[f], (f, int(got[f]), int(want[f]), int(m["msgtype"]))
    assert nac > 0 or not meta["mode_ac"]


def test_target_state_published_example(pkg):
    """ME type 29 subtype 1 (DO-260B target state and status), the example of 'The 1090 MHz riddle': selected
    altitude 16992 ft from the MCP/FCU, QNH 1012.8 hPa, selected heading 66.8 deg, NACp 9, NICbaro 1, SIL 3,
    autopilot / VNAV / approach-free / TCAS / LNAV mode bits."""
    f = pkg.capi.decode_fields(es_record(pkg, "8DA05629EA21485CBF3F8CADAEEB"))
    assert (f["metype"], f["mesub"]) == (29, 1)
    assert f["nav_valid"] & 4 and f["nav_mcp_altitude"] == 16992 and not f["nav_valid"] & 8
    assert f["nav_valid"] & 16 and abs(800.0 + (f["nav_qnh_raw"] - 1) * 0.8 - 1012.8) < 1e-9
    assert f["nav_valid"] & 2 and f["nav_valid"] & 32 and abs(f["nav_heading_raw"] * 180.0 / 256.0 - 66.8) < 0.05
    assert (f["nac_p"], f["nic_baro"], f["sil"], f["sil_type"], f["acc_valid"] & 3) == (9, 1, 3, 1, 3)
    assert f["nav_valid"] & 1 and f["nav_modes"] == (1 | 2 | 16 | 32)  # autopilot, VNAV, LNAV, TCAS


def test_target_state_and_operational_status_against_the_oracle(pkg, oracle):
    """ME types 29 and 31 (mode_s.c:1058-1370), every version and subtype, DF17 and DF18 with every control
    field: the shared host/device decoder against the oracle's restatement on random payloads."""
    rng = np.random.default_rng(2931)
    n = 0
    for k in range(6000):
        raw = bytearray(rng.integers(0, 256, 14, dtype=np.uint8).tobytes())
        df = 17 if k % 3 else 18
        raw[0] = (df << 3) | int(rng.integers(0, 8))
        metype = 29 if k % 2 else 31
        raw[4] = (metype << 3) | (raw[4] & 7)
        if metype == 31 and k % 5:  # mostly the subtypes that are decoded, and versions 0..2
            raw[4] = (metype << 3) | int(rng.integers(0, 2))
            raw[9] = (raw[9] & 0x1F) | (int(rng.integers(0, 3)) << 5)  # ME bits 41-43
        if metype == 29 and k % 5:
            raw[4] = (metype << 3) | (int(rng.integers(0, 2)) << 1) | (raw[4] & 1)  # ME bits 6-7: subtype 0 or 1
        m = es_record(pkg, bytes(raw).hex())
        got, want = pkg.capi.decode_fields(m), oracle.fields_of(m)
        for f in FIELD_NAMES:
            assert got[f] == want[f], (f, bytes(raw).hex(), int(got[f]), int(want[f]))
        n += bool(want["opstatus"] & 1) + bool(want["nav_valid"] or want["acc_valid"])
    assert n > 3000


def test_comm_b_published_examples(pkg):
    """Comm-B register inference and decode (comm_b.c) on the examples of 'The 1090 MHz riddle' / pyModeS:
    BDS 5,0 roll 2.1 deg, track 114.258 deg, 438 kt over ground, 0.125 deg/s, TAS 424 kt; BDS 6,0 heading
    42.715 deg, IAS 252 kt, Mach 0.42, -1920 ft/min both rates; BDS 4,0 MCP and FMS 3008 ft, QNH 1020.0 hPa;
    BDS 2,0 callsigns."""
    f = pkg.capi.decode_fields(es_record(pkg, "A000139381951536E024D4CCF6B5"))
    assert (f["commb_format"], f["commb_valid"], f["heading_valid"], f["heading_type"], f["tas_valid"]) == (8, 7, 1, 1, 1)
    assert (round(f["roll_q"] * 45 / 256, 1), f["heading_raw"] * 90 / 512, f["gs"], f["track_rate_q"] / 32, f["tas"]) == \
           (2.1, 114.2578125, 438, 0.125, 424)
    f = pkg.capi.decode_fields(es_record(pkg, "A00004128F39F91A7E27C46ADC21"))
    assert (f["commb_format"], f["commb_valid"], f["heading_type"], f["ias_valid"], f["ias"]) == (9, 8, 3, 1, 252)
    assert (f["heading_raw"] * 90 / 512, round(f["mach_raw"] * 2.048 / 512, 3)) == (42.71484375, 0.42)
    assert (f["baro_rate_valid"], f["baro_rate"], f["geom_rate_valid"], f["geom_rate"]) == (1, -1920, 1, -1920)
    f = pkg.capi.decode_fields(es_record(pkg, "A000029C85E42F313000007047D3"))
    assert (f["commb_format"], f["nav_mcp_altitude"], f["nav_fms_altitude"], 800 + f["nav_qnh_raw"] * 0.1) == (7, 3008, 3008, 1020.0)
    assert f["nav_valid"] & (4 | 8 | 16 | 64) == (4 | 8 | 16 | 64)
    for hx, cs in (("A000083E202CC371C31DE0AA1CCF", b"KLM1017 "), ("A0001838201584F23468207CDFA5", b"EXS2MF  ")):
        f = pkg.capi.decode_fields(es_record(pkg, hx))
        assert (f["commb_format"], f["callsign_valid"], f["callsign"]) == (5, 1, cs)
    # a GICB capability report (BDS 1,7) must have its last 32 bits clear (comm_b.c:129-132)
    assert pkg.capi.decode_fields(es_record(pkg, "A0000000" + "FA010100000000" + "000000"))["commb_format"] == 4
    assert pkg.capi.decode_fields(es_record(pkg, "A0000000" + "FA010180000001" + "000000"))["commb_format"] != 4
    # an all-zero MB field is "empty response"; DR set: not looked at at all
    assert pkg.capi.decode_fields(es_record(pkg, "A0000000" + "00" * 7 + "000000"))["commb_format"] == 2
    assert pkg.capi.decode_fields(es_record(pkg, "A0080000" + "00" * 7 + "000000"))["commb_format"] == 0


def test_comm_b_integer_thresholds_equal_the_float_comparisons():
    """The device decoder tests raw integers where comm_b.c compares floats: same verdict for every raw value
    (float32 results of double expressions, as the reference's C computes them)."""
    f32 = np.float32
    for raw in range(4096):  # BDS 4,0 pressure setting, comm_b.c:324-333
        setting = f32(800 + raw * 0.1)
        assert (setting >= 900 and setting <= 1100) == (1000 <= raw <= 3000) or raw == 0
    for raw in range(1024):  # BDS 6,0 Mach, :651-660
        mach = f32(raw * 2.048 / 512)
        assert (float(mach) >= 0.1 and float(mach) <= 0.9) == (25 <= raw <= 225), raw
    for raw in range(512):
        for sign in (0, 1):
            roll = f32(raw * 45.0 / 256.0)
            if sign:
                roll = f32(float(roll) - 90.0)
            assert (roll >= -40 and roll < 40) == ((raw >= 285) if sign else (raw <= 227)), (raw, sign)  # :464-476
            rate = f32(raw * 8.0 / 256.0)
            if sign:
                rate = f32(rate - f32(16))
            assert (rate >= -10.0 and rate <= 10.0) == (-320 <= raw - 512 * sign <= 320), (raw, sign)  # :512-524


def make_commb(rng, kind):
    """an MB field that looks like BDS `kind` (plausible values, sometimes off the edge)"""
    def put(bits, first, last, value):
        for i in range(last - first + 1):
            bits[first - 1 + i] = (value >> (last - first - i)) & 1
    bits = [0] * 56
    edge = rng.random() < 0.25
    if kind == 10:
        put(bits, 1, 8, 0x10)
        put(bits, 9, 56, int(rng.integers(0, 1 << 48)))
        if not edge:
            put(bits, 10, 14, 0)
    elif kind == 17:
        put(bits, 1, 24, int(rng.integers(0, 1 << 24)) if edge else int(rng.choice([0xFA8180, 0x020000, 0xFE8101, 0xFA0000])))
    elif kind == 20:
        put(bits, 1, 8, 0x20)
        chars = [int(rng.choice([1, 2, 11, 12, 13, 26, 32, 48, 49, 55, 57])) for _ in range(8)]
        if edge:
            chars[int(rng.integers(0, 8))] = int(rng.choice([0, 27, 33, 63]))
        for i, c in enumerate(chars):
            put(bits, 9 + 6 * i, 14 + 6 * i, c)
    elif kind == 30:
        put(bits, 1, 8, 0x30)
        put(bits, 9, 56, int(rng.integers(0, 1 << 48)))
    elif kind == 40:
        for v, lo in ((1, 2), (14, 15)):
            if rng.random() < 0.8:
                bits[v - 1] = 1
                alt = int(rng.choice([3008, 36000, 35984, 12000, 500, 51200, 10240])) if not edge else int(rng.integers(0, 65536))
                put(bits, lo, lo + 11, (alt // 16) & 0xFFF)
        if rng.random() < 0.8:
            bits[26] = 1
            put(bits, 28, 39, int(rng.integers(990, 3010)) if not edge else int(rng.integers(0, 4096)))
        if rng.random() < 0.5:
            bits[47] = 1
            put(bits, 49, 51, int(rng.integers(0, 8)))
        if rng.random() < 0.5:
            bits[53] = 1
            put(bits, 55, 56, int(rng.integers(0, 4)))
        if edge and rng.random() < 0.3:
            put(bits, 40, 47, 1)
    elif kind == 50:
        bits[0] = bits[11] = bits[23] = bits[45] = 1
        roll = int(rng.integers(-230, 230)) if not edge else int(rng.integers(-512, 512))
        put(bits, 2, 11, roll & 0x3FF)
        put(bits, 13, 23, int(rng.integers(0, 2048)))
        put(bits, 25, 34, int(rng.integers(20, 360)) if not edge else int(rng.integers(0, 1024)))
        if rng.random() < 0.8:
            bits[34] = 1
            put(bits, 36, 45, int(rng.integers(-330, 330)) & 0x3FF)
        put(bits, 47, 56, int(rng.integers(20, 360)) if not edge else int(rng.integers(0, 1024)))
    else:
        bits[0] = bits[12] = bits[23] = 1
        put(bits, 2, 12, int(rng.integers(0, 2048)))
        put(bits, 14, 23, int(rng.integers(40, 710)) if not edge else int(rng.integers(0, 1024)))
        put(bits, 25, 34, int(rng.integers(20, 230)) if not edge else int(rng.integers(0, 1024)))
        for v, lo in ((35, 36), (46, 47)):
            if rng.random() < 0.8:
                bits[v - 1] = 1
                put(bits, lo, lo + 9, int(rng.integers(-200, 200)) & 0x3FF)
    return int("".join(map(str, bits)), 2).to_bytes(7, "big")


def test_comm_b_against_the_oracle(pkg, oracle):
    """decodeCommB (comm_b.c:50-744): the integer decoder shared by the emit kernel and the host against the
    oracle's float restatement -- plausible and borderline contents of every register, and random MB fields."""
    rng = np.random.default_rng(4050)
    seen = np.zeros(10, dtype=int)
    for k in range(12000):
        kind = (10, 17, 20, 30, 40, 50, 60, 0, 40, 50, 60, -1)[k % 12]
        mb = (make_commb(rng, kind) if kind > 0 else
              rng.integers(0, 256, 7, dtype=np.uint8).tobytes() if kind == 0 else bytes(7) if k % 24 else b"\x00" * 6 + b"\x01")
        df = 20 if k % 2 else 21
        head = bytes([(df << 3) | int(rng.integers(0, 8)), 0 if k % 16 else int(rng.integers(0, 256)) & 0xF8,
                      int(rng.integers(0, 256)), int(rng.integers(0, 256))])
        m = es_record(pkg, (head + mb + b"\x00\x00\x00").hex())
        got, want = pkg.capi.decode_fields(m), oracle.fields_of(m)
        for f in FIELD_NAMES:
            assert got[f] == want[f], (f, m["msg"].tobytes().hex(), int(got[f]), int(want[f]))
        seen[want["commb_format"]] += 1
    assert (seen[2:] > 50).all() and seen[1] > 10, seen  # every format, and ambiguity, occurred


FLOAT_NAMES = ("gs_v0", "gs_v2", "gs_selected", "heading", "track_rate", "roll", "nav_qnh", "nav_heading", "mach",
               "gs_valid", "heading_valid", "heading_type", "track_rate_valid", "roll_valid", "mach_valid",
               "nav_qnh_valid", "nav_heading_valid")


def assert_floats_equal(pkg, oracle, m):
    """msd_fields -> msd_fields_to_float (product, from the delivered integers) against the oracle's values, which
    are computed from the message bytes at the places the reference computes them: bit for bit."""
    got = pkg.capi.fields_to_float(pkg.capi.decode_fields(m))
    want = oracle.fields_float_of(m)
    for f in FLOAT_NAMES:
        a, b = got[f], want[f]
        assert a.tobytes() == b.tobytes(), (f, m["msg"].tobytes().hex(), float(a), float(b))
    return want


def test_float_fields_published_examples(pkg, oracle):
    """The float-valued members of struct modesMessage (readsb.h:423-438,533-534) on the published examples:
    159.20 kt on track 182.88 deg, heading 243.98 deg, BDS 5,0 / 6,0 / 4,0 values, QNH 1012.8 hPa."""
    v = assert_floats_equal(pkg, oracle, es_record(pkg, "8D485020994409940838175B284F"))
    assert (v["gs_valid"], v["heading_valid"], v["heading_type"]) == (1, 1, 1)
    assert abs(v["gs_selected"] - 159.20) < 0.01 and abs(v["heading"] - 182.88) < 0.01 and v["gs_v0"] == v["gs_v2"] == v["gs_selected"]
    a = assert_floats_equal(pkg, oracle, es_record(pkg, "8DA05F219B06B6AF189400CBC33F"))
    assert (a["gs_valid"], a["heading_valid"], a["heading_type"]) == (0, 1, 4) and abs(a["heading"] - 243.98) < 0.01
    t = assert_floats_equal(pkg, oracle, es_record(pkg, "A000139381951536E024D4CCF6B5"))  # BDS 5,0
    assert (t["roll_valid"], t["track_rate_valid"], t["gs_valid"], t["heading_type"]) == (1, 1, 1, 1)
    assert (round(float(t["roll"]), 1), float(t["heading"]), float(t["gs_selected"]), float(t["track_rate"])) == (2.1, 114.2578125, 438.0, 0.125)
    h = assert_floats_equal(pkg, oracle, es_record(pkg, "A00004128F39F91A7E27C46ADC21"))  # BDS 6,0
    assert (h["mach_valid"], h["heading_type"], float(h["heading"])) == (1, 3, 42.71484375)
    assert round(float(h["mach"]), 3) == 0.42 and h["mach"] == np.float32(h["mach"])  # a float, widened
    q = assert_floats_equal(pkg, oracle, es_record(pkg, "A000029C85E42F313000007047D3"))  # BDS 4,0
    assert q["nav_qnh_valid"] == 1 and q["nav_qnh"] == np.float32(800 + 2200 * 0.1)
    s = assert_floats_equal(pkg, oracle, es_record(pkg, "8DA05629EA21485CBF3F8CADAEEB"))  # target state (DO-260B example)
    assert s["nav_qnh_valid"] == 1 and abs(s["nav_qnh"] - 1012.8) < 0.01
    assert s["nav_heading_valid"] == 1 and abs(s["nav_heading"] - 66.8) < 0.1


def test_float_fields_every_velocity_and_movement_code(pkg, oracle):
    """sqrtf / atan2 ground speed and track for all four quadrants and both subtypes on a lattice of the 1023 x 1023
    speed components (incl. the largest, where ns^2 + ew^2 + 0.5 no longer fits a float), the heading scale of
    subtypes 3 and 4, and every surface movement code with both tables (mode_s.c:216-259,826-855,911-924)."""
    def me_record(df, me):
        raw = bytes([(df << 3) | 5, 0x48, 0x40, 0xD6]) + me.to_bytes(7, "big") + bytes(3)
        return es_record(pkg, raw.hex())
    rng = np.random.default_rng(19)
    comps = sorted(set([1, 2, 3, 512, 1022, 1023] + [int(x) for x in rng.integers(1, 1024, 40)]))
    n = 0
    for sub in (1, 2):
        for ew in comps:
            for ns in comps[:: 3 if sub == 2 else 1]:
                for signs in range(4):
                    me = (19 << 51) | (sub << 48) | ((signs & 1) << 42) | (ew << 32) | ((signs >> 1) << 31) | (ns << 21) | (1 << 10) | 5
                    w = assert_floats_equal(pkg, oracle, me_record(17, me))
                    assert w["gs_valid"] == 1 and (w["heading_valid"] == 1) == (w["gs_selected"] > 0)
                    n += 1
    for sub in (3, 4):
        for hdg in list(range(0, 1024, 7)) + [1023]:
            me = (19 << 51) | (sub << 48) | (1 << 42) | (hdg << 32) | (1 << 31) | (300 << 21)
            w = assert_floats_equal(pkg, oracle, me_record(17, me))
            assert w["heading"] == np.float32(hdg * 360.0 / 1024.0) and w["gs_valid"] == 0
    for movement in range(128):
        for trk in (0, 1, 77, 127):
            me = (6 << 51) | (movement << 44) | (1 << 43) | (trk << 36) | 12345
            w = assert_floats_equal(pkg, oracle, me_record(17, me))
            assert w["gs_valid"] == (1 if 0 < movement < 125 else 0) and w["heading"] == np.float32(trk * 360.0 / 128.0)
    assert n > 5000


def test_float_fields_against_the_oracle_on_random_payloads(pkg, oracle):
    """Random ME type 29 payloads (both layouts: QNH and selected heading) and crafted Comm-B registers 4,0 / 5,0 /
    6,0 (roll, track, track rate, Mach, QNH): product floats from the integers equal the oracle's from the bytes."""
    rng = np.random.default_rng(2950)
    seen = dict(qnh=0, nav_heading=0, roll=0, mach=0, track_rate=0)
    for k in range(4000):
        raw = bytearray(rng.integers(0, 256, 14, dtype=np.uint8).tobytes())
        raw[0] = (17 << 3) | 5
        raw[4] = (29 << 3) | (int(rng.integers(0, 2)) << 1) | (raw[4] & 1)
        w = assert_floats_equal(pkg, oracle, es_record(pkg, bytes(raw).hex()))
        seen["qnh"] += int(w["nav_qnh_valid"])
        seen["nav_heading"] += int(w["nav_heading_valid"])
    for k in range(6000):
        kind = (40, 50, 60)[k % 3]
        head = bytes([((20 if k % 2 else 21) << 3) | int(rng.integers(0, 8)), 0, int(rng.integers(0, 256)), int(rng.integers(0, 256))])
        w = assert_floats_equal(pkg, oracle, es_record(pkg, (head + make_commb(rng, kind) + bytes(3)).hex()))
        seen["roll"] += int(w["roll_valid"])
        seen["mach"] += int(w["mach_valid"])
        seen["track_rate"] += int(w["track_rate_valid"])
        seen["qnh"] += int(w["nav_qnh_valid"])
    assert all(v > 300 for v in seen.values()), seen


@pytest.mark.parametrize("name", ["uc8_fix_modeac", "sc16q11_fix_modeac"])
def test_float_fields_on_replayed_captures(pkg, oracle, name):
    """... and on every Mode S message of two golden captures (regenerated from their seeds, replayed by the oracle)."""
    meta, z = load_golden(name)
    fmt = {"uc8": pkg.FMT_UC8, "sc16": pkg.FMT_SC16, "sc16q11": pkg.FMT_SC16Q11}[meta["format"]]
    ofmt = {"uc8": oracle.FMT_UC8, "sc16": oracle.FMT_SC16, "sc16q11": oracle.FMT_SC16Q11}[meta["format"]]
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=meta["seed"], fmt=fmt, **meta["gen"]), meta["nsamples"])
    msgs, _, _ = oracle.Oracle(ofmt, 58, meta["nfix_crc"], meta["mode_ac"]).replay_fields(iq, cap=1 << 17)
    assert len(msgs) == len(z["timestampMsg"])
    n = 0
    for m in msgs:
        if m["msgtype"] == 32:
            continue
        w = assert_floats_equal(pkg, oracle, m)
        n += int(w["gs_valid"]) + int(w["heading_valid"])
    assert n > 0
