#!/bin/bash
# HBM traffic of the scan kernel from PMC counters, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in separate --pmc passes (no trace domains mixed in).  Writes
# gpurun_out/traffic.json with per-launch KB figures for the default bench workload's launch size.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT/pmc_tmp
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-check --batch $((1<<26))"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_tmp/$ctr -o pmc -- $CMD > $OUT/pmc_tmp/$ctr.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/pmc_tmp/{ctr}/**/*counter_collection.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    vals = {"with_records": collections.defaultdict(float), "scan_only": collections.defaultdict(float)}
    for r in rows:
        if "msd_scan_kernel" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            # msd_scan_kernel<FMT, FIX2, EMIT>: EMIT = the launch also writes the previous batch's message records
            name = r["Kernel_Name"]
            targs = name[name.index("msd_scan_kernel<"):].split(">")[0]
            kind = "with_records" if targs.endswith("true") else "scan_only"
            vals[kind][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for kind, d in vals.items():
        v = sorted(d.values())
        if v:
            suffix = "" if kind == "with_records" else "_scan_only"
            res[ctr + "_KB_per_launch" + suffix] = v[len(v) // 2]
            res[ctr + "_launches" + suffix] = len(v)
res["samples_per_launch"] = 1 << 26
res["note"] = ("rocprofv3 --pmc, median over launches of msd_scan_kernel<UC8> with the record slice (the default layout; _scan_only: the launches without); gfx950 FETCH_SIZE counts 64 B per "
               "128 B request on wide coalesced reads (MI355X_MICROARCH.md), so fetch bytes = 2 * FETCH_SIZE * 1024")
json.dump(res, open(f"{out}/traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
rm -rf $OUT/pmc_tmp
