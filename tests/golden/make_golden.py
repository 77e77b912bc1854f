#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

Each fixture is (generator config -> expected ordered message list + demodulator counters).  The IQ
bytes are NOT stored: they are regenerated from the seed by the integer-only generator
(readsb-protobuf_amd/csrc/msd_siggen.c), whose output is itself pinned by a SHA-256 in the fixture.

Provenance: the expected outputs come from oracle/modes_oracle.c (our CPU restatement), because the
reference cannot be built in this image ("Oracle").  They pin the oracle and the HIP
path against regressions and against each other; they are not outputs of the reference binary.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as g  # noqa: E402

CASES = {
    # name: (format, nsamples, seed, nfix, mode_ac, generator kwargs)
    "uc8_nofix": ("uc8", 3 * 131072 + 4567, 10901, 0, 0, {}),
    "uc8_fix_exact_multiple": ("uc8", 2 * 131072, 1090, 1, 0, {}),
    "uc8_fix_modeac": ("uc8", 4 * 131072 + 99, 44, 1, 1, {"msgs_per_sec": 200, "ac_per_sec": 2000}),
    "sc16_fix": ("sc16", 2 * 131072 + 1000, 10920, 1, 0, {}),
    "sc16q11_fix_modeac": ("sc16q11", 2 * 131072 + 77, 10921, 1, 1, {"msgs_per_sec": 300, "ac_per_sec": 1500}),
    "uc8_short": ("uc8", 1000, 3, 1, 0, {"msgs_per_sec": 20000}),
}
FIELDS = ("timestampMsg", "sysTimestampMsg", "signalLevel", "addr", "crc", "score", "msgtype", "msgbits",
          "correctedbits", "bestphase", "iid", "msg")


def main():
    pkg = g.load_package()
    O = g.load_oracle()
    fmts = {"uc8": (pkg.FMT_UC8, O.FMT_UC8), "sc16": (pkg.FMT_SC16, O.FMT_SC16), "sc16q11": (pkg.FMT_SC16Q11, O.FMT_SC16Q11)}
    for name, (fmt, n, seed, nfix, mode_ac, kw) in CASES.items():
        pf, of = fmts[fmt]
        cfg = pkg.siggen.make_cfg(seed=seed, fmt=pf, **kw)
        iq = pkg.siggen.generate(cfg, n, nthreads=4)
        msgs, stats, means = O.Oracle(of, 58, nfix, mode_ac).replay(iq, want_means=True)
        meta = {"format": fmt, "nsamples": n, "seed": seed, "nfix_crc": nfix, "mode_ac": mode_ac, "gen": kw,
                "iq_sha256": hashlib.sha256(iq.tobytes()).hexdigest(), "stats": stats}
        arrays = {f: msgs[f] for f in FIELDS}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), meta=json.dumps(meta), means=means, **arrays)
        print(name, len(msgs), "messages", stats["demod_modeac"], "mode a/c")


if __name__ == "__main__":
    main()
