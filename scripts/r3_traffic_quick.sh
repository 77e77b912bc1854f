#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the UC8 scan kernel (separate --pmc passes), median over launches
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/tq; rm -rf $O; mkdir -p $O
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $ctr --output-format csv -d $O/$ctr -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --settle-seconds 0 --no-cpu-baseline --no-check --no-also "$@" > $O/$ctr.log 2>&1)
  python3 - $O/$ctr $ctr <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == sys.argv[2] and "msd_scan_kernel" in r["Kernel_Name"]:
        d[r["Dispatch_Id"]] += float(r["Counter_Value"])
v = sorted(d.values()); print(sys.argv[2], "KB per scan launch (median of %d):" % len(v), v[len(v)//2])
PY
done
rm -rf $O
