"""Random runs of the replay tool (not collected by pytest): python tests/fuzz_replay.py [cases] [first_seed]
msd_replay -- readsb's part of the boundary: option keys, the ifile handler, its reader threads and page-locked ring, the mag_buf
FIFO and demodulate2400(struct mag_buf *) -- on capture files of random length (turn-size multiples and their neighbours among
them), format, path (fused / magbuf), --batch-buffers, --fix / --aggressive, --dcfilter, --modeac, from /dev/shm, against the
oracle's message list line by line ("@<12-hex timestamp><message>;", mode_s.c:1786-1798 with --mlat)."""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
orc = g.load_oracle()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
first = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
exe = os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "msd_replay")
CH = 131072
bad = 0
for case in range(first, first + ncases):
    rng = np.random.default_rng(case)
    fmt_name = str(rng.choice(["uc8", "uc8", "sc16", "sc16q11"]))
    fmt, ofmt = {"uc8": (pkg.FMT_UC8, orc.FMT_UC8), "sc16": (pkg.FMT_SC16, orc.FMT_SC16), "sc16q11": (pkg.FMT_SC16Q11, orc.FMT_SC16Q11)}[fmt_name]
    path = str(rng.choice(["fused", "fused", "magbuf"]))
    bb = int(rng.choice([1, 2, 16, 17, 64]))
    turns = int(rng.integers(0, 7))
    n = turns * bb * CH + int(rng.choice([0, 0, 1, CH - 1, CH, CH + 1, 5 * CH + 4321, 63 * CH]))
    n = max(1, min(n, 300 * CH))
    nfix = int(rng.integers(0, 3))
    dc = bool(rng.integers(0, 3) == 0)
    mode_ac = int(rng.integers(0, 3) == 0)
    kw = dict(msgs_per_sec=int(rng.choice([2000, 6000, 12000])), n_aircraft=int(rng.choice([20, 400, 5000])),
              overlap_permille=int(rng.choice([0, 200])), flip_permille=int(rng.choice([0, 100])), noise_fs=float(rng.choice([0.005, 0.03])),
              ac_per_sec=2000 if mode_ac else 0)
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=case, fmt=fmt, **kw), n)
    f = "/dev/shm/fuzz_replay_%d.bin" % os.getpid()
    iq.tofile(f)
    args = [exe, "--ifile", f, "--iformat", fmt_name, "--mlat", "--raw", "--path", path, "--batch-buffers", str(bb),
            ["--no-fix", "--fix", "--aggressive"][nfix]] + (["--dcfilter"] if dc else []) + (["--modeac"] if mode_ac else [])
    desc = f"case {case}: {fmt_name} n={n} path={path} batch-buffers={bb} nfix={nfix} dc={int(dc)} ac={mode_ac} {kw}"
    try:
        out = subprocess.run(args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, "exit %d: %s" % (out.returncode, out.stderr[-300:])
        want, _ = orc.Oracle(ofmt, 58, nfix, mode_ac, dc_filter=dc).replay(iq, cap=1 << 20)
        lines = out.stdout.split()
        assert len(lines) == len(want), "lines %d, oracle %d" % (len(lines), len(want))
        for k, (line, m) in enumerate(zip(lines, want)):
            exp = "@%012X%s;" % (int(m["timestampMsg"]), bytes(m["msg"][: m["msgbits"] // 8]).hex().upper() if False else bytes(m["msg"][: m["msgbits"] // 8]).hex())
            assert line == exp, "line %d: %s != %s" % (k, line, exp)
        print("ok  ", desc, "msgs", len(want), flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL", desc, repr(e)[:300], flush=True)
    finally:
        try:
            os.unlink(f)
        except OSError:
            pass
print("failures:", bad)
sys.exit(1 if bad else 0)
