"""Randomised parity sweep (not collected by pytest): python tests/fuzz_parity.py [cases] [first_seed]
Each case draws a traffic model, a format, a batch size and the resolve path at random and compares the
HIP path with the oracle message for message and counter for counter; captures of up to eight buffers also with the
second reading of the reference (tests/indep_demod.py), directly."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402
from tests.test_gpu_parity import assert_same  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import indep_demod  # noqa: E402  (the second reading of the reference: a third witness for the short captures)
from test_indep_demod import assert_second_reading_agrees  # noqa: E402

pkg = g.load_package()
orc = g.load_oracle()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
bad = 0
for case in range(first, first + ncases):
    rng = np.random.default_rng(case)
    fmt_name = rng.choice(["uc8", "uc8", "uc8", "sc16", "sc16q11"])
    fmt, ofmt = {"uc8": (pkg.FMT_UC8, orc.FMT_UC8), "sc16": (pkg.FMT_SC16, orc.FMT_SC16),
                 "sc16q11": (pkg.FMT_SC16Q11, orc.FMT_SC16Q11)}[fmt_name]
    nbuf = int(rng.integers(1, 40))
    n = nbuf * 131072 + int(rng.choice([0, 1, 7, 8, 1234, 65536, 131071]))
    batch = int(rng.choice([1, 2, 4, 8, 16, 64])) * 131072
    kw = dict(msgs_per_sec=int(rng.choice([200, 2000, 6000, 12000])), n_aircraft=int(rng.choice([3, 50, 800, 5000, 30000])),
              overlap_permille=int(rng.choice([0, 10, 200, 700])), flip_permille=int(rng.choice([0, 20, 200])),
              noise_fs=float(rng.choice([0.005, 0.02, 0.06])), ac_per_sec=int(rng.choice([0, 0, 500, 4000])))
    nfix = int(rng.integers(0, 3))
    mode_ac = int(kw["ac_per_sec"] > 0 and rng.integers(0, 2))
    gpu_resolve = int(rng.integers(0, 2))
    thr = int(rng.choice([58, 58, 58, 40, 75, 400]))  # --preamble-threshold, readsb.c:503-505
    os.environ["MSD_GPU_RESOLVE"] = str(gpu_resolve)
    os.environ["MSD_RESOLVE_THREADS"] = str(int(rng.choice([1, 4, 16])))
    cfg = pkg.siggen.make_cfg(seed=case, fmt=fmt, **kw)
    iq = pkg.siggen.generate(cfg, n)
    d = torch.from_numpy(iq).to("cuda:0")
    with_fields = int(rng.integers(0, 2))
    dc = bool(rng.integers(0, 8) < 2)  # --dcfilter: a quarter of the cases since round 6 (the DC block is parallel in time now; it ran at 0.13 GS/s)
    q11 = int(rng.choice([0, 0, 7, 8, 11])) if fmt_name == "sc16q11" and not dc else 0  # a -DSC16Q11_TABLE_BITS build (convert.c:264-328)
    # round 5: candidate arenas far too small for the traffic now and then (their region slices overflow: grown and
    # rescanned on the GPU, or -- growth switched off -- in pieces through the host resolver)
    arena = int(rng.choice([0, 0, 0, 50, 200, 1000]))
    growth = int(rng.integers(0, 4) != 0)
    os.environ["MSD_ARENA_SCALE_PERMILLE"] = str(arena)
    os.environ["MSD_ARENA_GROWTH"] = str(growth)
    dem = pkg.Demodulator(fmt=fmt, preamble_threshold=thr, nfix_crc=nfix, mode_ac=mode_ac, max_batch_samples=batch, message_capacity=1 << 19,
                          decode_fields=bool(with_fields), dc_filter=dc, **({"sc16q11_table_bits": q11} if q11 else {}))
    desc = (f"case {case}: {fmt_name} n={n} batch={batch // 131072} nfix={nfix} ac={mode_ac} gpu_resolve={gpu_resolve} "
            f"fields={with_fields} dc={int(dc)} thr={thr} q11_table_bits={q11} arena_permille={arena} growth={growth} {kw}")
    if with_fields:
        parts, fparts, bps = [], [], dem.bytes_per_sample
        for off in list(range(0, n, batch)) or [0]:
            m = min(batch, n - off)
            dem.launch_device(d.data_ptr() + off * bps, m, off + m >= n)
            mm, ff = dem.collect_fields()
            parts.append(mm)
            fparts.append(ff)
        got, gfields = np.concatenate(parts), np.concatenate(fparts)
        want, wfields, wstats = orc.Oracle(ofmt, thr, nfix, mode_ac, dc_filter=dc, sc16q11_table_bits=q11).replay_fields(iq, cap=1 << 19)
    else:
        got = pkg.replay_device(dem, d.data_ptr(), n, batch)
        want, wstats = orc.Oracle(ofmt, thr, nfix, mode_ac, dc_filter=dc, sc16q11_table_bits=q11).replay(iq, cap=1 << 19)
    try:
        assert_same(got, dem.stats(), want, wstats)
        if with_fields:
            for name in gfields.dtype.names:
                assert np.array_equal(gfields[name], wfields[name]), "field " + name
        second = ""
        if nbuf <= 8 and not q11:
            smsgs, sstats = indep_demod.Receiver(fmt_name, thr, nfix, bool(mode_ac), dc_filter=dc).replay(iq.tobytes())
            assert_second_reading_agrees(smsgs, sstats, got, dem.stats())
            second = " second-reading ok"
        print("ok  ", desc, "msgs", len(want), "passes", dem.timing()["resolve_passes"], "fallback", dem.timing()["resolve_fallback"], "reruns", dem.timing()["reruns"], second)
    except AssertionError as e:
        bad += 1
        print("FAIL", desc, str(e)[:200])
    del dem
print("failures:", bad)
sys.exit(1 if bad else 0)
