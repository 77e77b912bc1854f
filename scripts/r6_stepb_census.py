#!/usr/bin/env python3
"""Static census of the scan kernel's step-B loop body (msd_kernels.hip candidate_round, the loop between s_setprio 3 and
s_setprio 2) by what the instructions are for -- VERDICT r05 #2 asks for the split of the ~245 wave-instructions per
64-item iteration.  Usage: r6_stepb_census.py <hipcc -S output> [kernel mangled-name substring]
(hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -I. -I../../include -S --cuda-device-only msd_kernels.hip)"""
import collections
import re
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "msd_scan_kernelILi0ELb0ELb1EE"
lines = open(path).read().split("\n")
k0 = [i for i, l in enumerate(lines) if l.startswith("_Z") and want in l and "MsdScanParams:" in l][0]
k1 = [i for i in range(k0, len(lines)) if lines[i].strip().startswith("s_endpgm")][0]
p3 = [i for i in range(k0, k1) if lines[i].strip() == "s_setprio 3"][0]
# the branch into the loop follows s_setprio 3; its target is the loop header; the loop is the header's block plus every
# block whose label line says "in Loop: Header=<that block>"
m = [re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", lines[i]) for i in range(p3, p3 + 12)]
head = [x.group(1) for x in m if x][0]
tag = "Header=" + head[2:] + " "
body, cur_in = [], False
for i in range(k0, k1):
    l = lines[i]
    if l.startswith(".LBB") or l.startswith("; %bb."):
        cur_in = (tag in l + " ") or l.startswith(head + ":")
    elif cur_in:
        s = l.strip()
        if s and s[0] not in ";." and not s.endswith(":"):
            body.append(s.split()[0])
ops = collections.Counter(body)
cats = [
    ("LDS reads: samples, slot records, syndrome tables", lambda o: o.startswith("ds_read")),
    ("LDS atomics: ds_xor (CRC), ds_or (message words)", lambda o: o.startswith(("ds_xor", "ds_or", "ds_and", "ds_add", "ds_max", "ds_min"))),
    ("correlator sums: v_mad_i32_i24 / 24-bit multiplies", lambda o: o.startswith(("v_mad_i32_i24", "v_mul_i32_i24", "v_mul_u32_u24", "v_mad_u32_u24", "v_mul_lo", "v_mul_hi"))),
    ("sign-bit pushes: v_alignbit", lambda o: o.startswith("v_alignbit")),
    ("unpack / rotation / masks: bfe, shifts, and, or, perm", lambda o: o.startswith(("v_bfe", "v_lshr", "v_lshl_or", "v_lshlrev", "v_and", "v_or", "v_xor", "v_bfi", "v_perm", "v_not", "v_ashr", "v_and_or", "v_or3", "v_sdwa"))),
    ("address / index arithmetic: add, sub, lshl_add, mad_u", lambda o: o.startswith(("v_add", "v_sub", "v_lshl_add", "v_mad_u", "v_add3", "v_add_lshl"))),
    ("compares and selects", lambda o: o.startswith(("v_cmp", "v_cndmask"))),
    ("moves, lane reads", lambda o: o.startswith(("v_mov", "v_readlane", "v_readfirstlane", "v_writelane", "v_accvgpr"))),
    ("waits and other scalar: s_waitcnt, branches, exec masks", lambda o: o.startswith("s_")),
]
left = dict(ops)
total = sum(ops.values())
print("step-B loop of %s: header %s, %d instructions per 64-item iteration (static)" % (want, head, total))
for name, f in cats:
    ks = [k for k in left if f(k)]
    n = sum(left.pop(k) for k in ks)
    print("  %4d  %4.1f %%  %s" % (n, 100.0 * n / total, name))
print("  %4d  other: %s" % (sum(left.values()), left))
valu = sum(v for k, v in ops.items() if k.startswith("v_"))
print("  vector ALU %d, LDS %d, scalar %d; 15 message bits per item: %.1f vector instructions per bit"
      % (valu, sum(v for k, v in ops.items() if k.startswith("ds_")), sum(v for k, v in ops.items() if k.startswith("s_")), valu / 15.0))
print("  most frequent:", ", ".join("%s %d" % kv for kv in ops.most_common(14)))
