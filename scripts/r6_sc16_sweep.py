#!/usr/bin/env python3
"""SURVEY.md 7.1 step 5 / VERDICT r05 #8: the exhaustive sweep of convert_sc16_nodc (convert.c:215-253) -- all 2^32 (I, Q)
int16 pairs -- through the converter entry of the C-ABI (msd_convert, the iq_convert_fn replacement) on the GPU, against the
oracle's converter on the host cores: magnitudes AND the two sequential float means of every 131072-sample block, bit for
bit.  32768 blocks, block b holds I = 2b - 32768 and I + 1 against every Q.  Test infrastructure (uses oracle/); run on the
GPU box: python scripts/r6_sc16_sweep.py [sc16|sc16q11] [first_block] [blocks]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def main():
    fmt = sys.argv[1] if len(sys.argv) > 1 else "sc16"
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 32768 - first
    pkg, O = graft.load_package(), graft.load_oracle()
    f = {"sc16": pkg.FMT_SC16, "sc16q11": pkg.FMT_SC16Q11}[fmt]
    of = {"sc16": O.FMT_SC16, "sc16q11": O.FMT_SC16Q11}[fmt]
    dem = pkg.Demodulator(fmt=f, nfix_crc=0, max_batch_samples=pkg.CHUNK)
    orc = O.Oracle(of, 58, 0, 0)
    n = pkg.CHUNK
    blk = np.empty(2 * n, dtype=np.int16)
    q_all = np.arange(-32768, 32768, dtype=np.int32).astype(np.int16)
    blk[1:2 * 65536:2] = q_all
    blk[2 * 65536 + 1::2] = q_all
    bad_mag = bad_mean = 0
    hist_max = 0
    t0 = time.time()
    for b in range(first, first + blocks):
        i0 = 2 * b - 32768
        blk[0:2 * 65536:2] = i0
        blk[2 * 65536::2] = i0 + 1
        raw = blk.view(np.uint8)
        mg, lg, pg = dem.convert(raw, n)
        mo, lo, po = orc.convert(raw, n)
        bad_mag += int((mg != mo).sum())
        bad_mean += int(not (np.float64(lg) == np.float64(lo) and np.float64(pg) == np.float64(po)))
        hist_max = max(hist_max, int(mg.max()))
    dt = time.time() - t0
    print("%s: %d blocks = %d (I, Q) pairs through msd_convert on the GPU against the oracle's convert.c restatement: "
          "%d magnitudes differ, %d of %d blocks differ in a mean (sequential float sums, convert.c:241-252); largest magnitude %d; %.0f s"
          % (fmt, blocks, blocks * n, bad_mag, bad_mean, blocks, hist_max, dt))
    return 1 if bad_mag or bad_mean else 0


if __name__ == "__main__":
    sys.exit(main())
