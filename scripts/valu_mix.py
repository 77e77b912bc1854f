"""Static mix of 2-cycle and 4-cycle vector-ALU instructions of the scan kernel's tile loop, by phase.

Usage: python scripts/valu_mix.py [asm-file]   (default: compiles msd_kernels.hip for gfx950 into /tmp)
The phases are delimited by the s_setprio instructions the tile loop issues (conversion 0 -> tests 1 -> candidate rounds
2 / step B 3 -> back to 0).  Conversion and preamble tests are straight-line code executed once per tile, so their
static counts ARE their dynamic counts per tile; the candidate rounds are loops and their static mix is only a proxy.
Issue cycles per wave64 instruction from the microbenchmark (profiles/r03_valu_issue.txt): the instructions listed in
FULL_RATE issue in 2 cycles unless they carry an SGPR / SDWA / DPP operand; everything else in 4 (v_mad_u64_u32 ~5.6)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL_RATE = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32",
             "v_mov_b32", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_f16", "v_add_u16", "v_sub_u16",
             "v_not_b32", "v_subrev_f32", "v_max_f32", "v_min_f32"}


def cycles(line):
    m = re.match(r"\s+(v_[a-z0-9_]+?)(_e32|_e64|_sdwa|_dpp)?\s", line)
    if not m:
        return None
    op, form = m.group(1), m.group(2) or ""
    if op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        return 4
    operands = line.split(None, 1)[1] if len(line.split(None, 1)) > 1 else ""
    sgpr = re.search(r"(?<![a-z])s\[?\d", operands) is not None or "vcc" in operands or "exec" in operands
    if op in FULL_RATE and form not in ("_sdwa", "_dpp") and not sgpr:
        return 2
    return 4


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/msd_kernels_mix.s"
    if len(sys.argv) <= 1:
        csrc = os.path.join(ROOT, "readsb-protobuf_amd", "csrc")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm",
                               "-disable-machine-licm", "-I.", "-I../../include", "--cuda-device-only", "-S", "msd_kernels.hip",
                               "-o", path], cwd=csrc, stderr=subprocess.DEVNULL)
    text = open(path).read()
    start = text.index("_ZN12_GLOBAL__N_115msd_scan_kernelILi0ELb0ELb1EEEv13MsdScanParams:")
    body = text[start:text.index("s_endpgm", start)].splitlines()
    # the tile loop: from the loop header in front of the first `s_setprio 1` to the last `s_setprio 0`
    prio = [(i, int(l.split()[1])) for i, l in enumerate(body) if l.strip().startswith("s_setprio")]
    first1 = next(i for i, p in prio if p == 1)
    last0 = [i for i, p in prio if p == 0][-1]
    head = max(i for i, l in enumerate(body[:first1]) if "Loop Header: Depth=1" in l)
    phases = {"conversion (prio 0)": (head, first1), "preamble tests (prio 1)": (first1, next(i for i, p in prio if p == 2 and i > first1)),
              "candidate rounds (prio 2 / 3, loops: static proxy)": (next(i for i, p in prio if p == 2 and i > first1), last0)}
    total = {}
    for name, (a, b) in phases.items():
        n2 = n4 = other = 0
        for l in body[a:b]:
            c = cycles(l)
            if c == 2:
                n2 += 1
            elif c == 4:
                n4 += 1
            elif l.startswith("\t") and not l.strip().startswith((";", ".")):
                other += 1
        total[name] = (n2, n4, other)
        print("%-52s 2-cycle VALU %5d  4-cycle VALU %5d  other (SALU, LDS, VMEM, waits) %5d  mean VALU issue %.2f cycles"
              % (name, n2, n4, other, (2 * n2 + 4 * n4) / max(1, n2 + n4)))
    return total


if __name__ == "__main__":
    main()
