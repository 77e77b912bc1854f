"""Round 6: the replay tool's raw / Beast outputs under --aggressive follow
modesQueueOutput (net_io.c:1263-1290), and a Beast feed read back by the host reader decodes on the GPU to the same
fields the record kernel produced."""
import os
import subprocess

import numpy as np
import pytest

from test_wire_readers import wire  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def test_replay_cli_forwards_like_modesQueueOutput(pkg, oracle, wire, torch_cuda, tmp_path):
    """--aggressive --net-raw / --beast: messages that needed two repairs stay home unless --net-verbatim is given, and then
    every message goes out with the bytes as received (mode_s.c:427-429, net_io.c:775,874)."""
    n = 24 * 131072 + 999
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=2024, msgs_per_sec=9000, n_aircraft=60, flip_permille=200), n)
    f = tmp_path / "capture.uc8"
    iq.tofile(f)
    want, _ = oracle.Oracle(oracle.FMT_UC8, 58, 2, 0).replay(iq, cap=1 << 18)
    two = int((want["correctedbits"] == 2).sum())
    assert two > 10 and (want["correctedbits"] == 1).sum() > 100
    exe = os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "msd_replay")
    base = [exe, "--ifile", str(f), "--iformat", "uc8", "--aggressive", "--mlat", "--batch-buffers", "8"]
    fwd = [m for m in want if m["correctedbits"] < 2]                     # net_io.c:1272, :1278
    raw = subprocess.run(base + ["--net-raw"], capture_output=True, check=True).stdout
    assert raw == b"".join(oracle.avr_line(m, True) for m in fwd)
    beast = subprocess.run(base + ["--beast"], capture_output=True, check=True).stdout
    assert beast == b"".join(oracle.beast_frame(m) for m in fwd)
    # --net-verbatim: everything, unrepaired
    received = []
    for m in want:
        k, bytes_ = wire.verbatim(m)
        assert k == int(m["correctedbits"])
        v = m.copy()
        v["msg"][: len(bytes_)] = np.frombuffer(bytes_, dtype=np.uint8)
        received.append(v)
    raw = subprocess.run(base + ["--net-raw", "--net-verbatim"], capture_output=True, check=True).stdout
    assert raw == b"".join(oracle.avr_line(m, True) for m in received)
    beast = subprocess.run(base + ["--beast", "--net-verbatim"], capture_output=True, check=True).stdout
    assert beast == b"".join(oracle.beast_frame(m) for m in received)
    assert len(received) == len(fwd) + two
    # the display dump (--raw, mode_s.c:1786-1798) is not a network output: it shows every message, repaired
    shown = subprocess.run(base, capture_output=True, check=True).stdout
    assert shown.count(b"\n") == len(want)


def test_beast_feed_through_the_reader_decodes_on_the_gpu(pkg, oracle, wire, torch_cuda):
    """GPU messages -> Beast frames (the writer) -> the reader, fed in ragged pieces -> msd_decode_fields_device: the same
    fields the record kernel decoded for those messages, Mode A/C replies included."""
    from test_fields import FIELD_NAMES
    n = 16 * 131072 + 333
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=606, msgs_per_sec=5000, ac_per_sec=600, n_aircraft=80), n)
    d_iq = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(nfix_crc=1, mode_ac=1, max_batch_samples=16 * 131072 + 131072, message_capacity=1 << 17, decode_fields=True)
    dem.launch_device(d_iq.data_ptr(), n, last=True)
    msgs, fields = dem.collect_fields()
    assert len(msgs) > 500 and (msgs["msgtype"] == 32).sum() > 20
    stream = b"".join(oracle.beast_frame(m) for m in msgs)
    rng = np.random.default_rng(9)
    got, r = wire.read_beast(stream, True, list(rng.integers(1, 700, size=len(stream))))
    assert len(got) == len(msgs) and r.garbage_bytes == 0
    nb = msgs["msgbits"] // 8
    for g, m, k in zip(got, msgs, nb):
        assert bytes(g["msg"][:k]) == bytes(m["msg"][:k]) and int(g["msgtype"]) == int(m["msgtype"])
    assert np.array_equal(got["addr"], msgs["addr"])   # AA after the repair, or the checksum of an address/parity format
    again = dem.decode_fields_device(got)
    mode_s = msgs["msgtype"] != 32
    for name in FIELD_NAMES:
        assert np.array_equal(again[name][mode_s], fields[name][mode_s]), name
    # a Mode A/C reply decodes on its own from the wire (the carried altitude of demod_2400.c:523-528 is a property of the
    # demodulator's buffer, not of the frame): compare with the host decoder, which is given no carry either
    for g, a in zip(got[~mode_s][:200], again[~mode_s][:200]):
        h = pkg.capi.decode_fields(g)
        for name in FIELD_NAMES:
            assert h[name] == a[name], name
