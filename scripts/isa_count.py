#!/usr/bin/env python3
"""Static instruction census of the kernels in a hipcc -save-temps .s file: per kernel, counts by class
(2-cycle / 4-cycle VALU as measured by scripts/micro/valu_issue.hip, MFMA, LDS, VMEM, SALU)."""
import collections
import sys

TWO = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32",
       "v_mov_b32", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_fmac_f32", "v_fma_f32", "v_not_b32", "v_subrev_f32",
       "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_add_nc_u32", "v_sub_nc_u32", "v_add_i32", "v_sub_i32"}


def census(path, only=None):
    name, counts = None, None
    out = []
    for line in open(path):
        s = line.strip()
        if not s or s[0] in ";." or s.startswith("//"):
            continue
        head = s.split(";")[0].strip()
        if head.endswith(":") and not head.startswith(".L") and " " not in head:
            if name is not None:
                out.append((name, counts))
            name, counts = head[:-1], collections.Counter()
            continue
        if name is None or head.endswith(":"):
            continue
        op = s.split()[0]
        if op.startswith(("v_", "s_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
            counts[op] += 1
    if name is not None:
        out.append((name, counts))
    for name, c in out:
        if only and only not in name:
            continue
        if not c:
            continue
        mfma = sum(v for k, v in c.items() if k.startswith("v_mfma"))
        valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        sdwa_dpp = 0
        two = sum(v for k, v in c.items() if k.split("_e32")[0].split("_e64")[0] in TWO)
        print(f"{name}: {sum(c.values())} instructions: VALU {valu} (2-cycle class {two}, other {valu - two}), MFMA {mfma}, "
              f"LDS {sum(v for k, v in c.items() if k.startswith('ds_'))}, VMEM {sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'flat_')))}, "
              f"scratch {sum(v for k, v in c.items() if k.startswith('scratch_'))}, SALU {sum(v for k, v in c.items() if k.startswith('s_'))}")
        print("   " + ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda x: -x[1])[:30]))


if __name__ == "__main__":
    census(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
