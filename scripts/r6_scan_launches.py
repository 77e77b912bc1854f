#!/usr/bin/env python3
"""Per-launch durations of msd_scan_kernel from a rocprofv3 --kernel-trace csv -> profiles/rNN_scan_launches.json, the
file bench.py computes roofline.frac_rocprof from (call-weighted mean AND median, from >= 30 launches,
the run's first launch of each instantiation excluded BY COUNT -- it pays for the code object and the cold tables).
Usage: r6_scan_launches.py <trace dir> <out.json> <samples per launch> <bytes per sample> "<command>" """
import csv
import glob
import json
import statistics
import sys

src, dst, samples, bps, cmd = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
rows = []
for f in glob.glob(src + "/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "msd_scan_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
kinds = {}
for r in rows:
    name = r["Kernel_Name"]
    targs = name[name.index("msd_scan_kernel<"):].split(">")[0] + ">"
    kinds.setdefault(targs, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"command": cmd, "samples_per_launch": samples, "algorithmic_bytes_per_launch": samples * bps, "excluded": "the first launch of every instantiation, by count",
       "instantiations": {}}
kept = []
for k, v in sorted(kinds.items()):
    use = v[1:] if len(v) > 1 else v
    kept += use
    out["instantiations"][k] = {"launches": len(v), "first_launch_us": round(v[0], 1), "kept": len(use), "mean_us": round(statistics.mean(use), 2),
                                "median_us": round(statistics.median(use), 2), "min_us": round(min(use), 1), "max_us": round(max(use), 1)}
out["launches_kept"] = len(kept)
out["mean_us"] = round(statistics.mean(kept), 2)          # call-weighted over both instantiations
out["median_us"] = round(statistics.median(kept), 2)
out["durations_us"] = [round(x, 1) for x in kept]
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "durations_us"}))
