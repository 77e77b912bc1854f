#!/bin/bash
# round 6 (GPU box): does the resolve kernel's time depend on how thinly its inputs are spread?  The region slices are 3 % full at the
# benchmark's density (4096 slices of 128 KB / 256 KB, 4 KB / 11 KB used); MSD_ARENA_SCALE_PERMILLE makes them 4 x and 8 x smaller.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/r6h/arena.txt}; mkdir -p $(dirname $OUT)
for pm in 4000 1000 500 4000 1000; do
  T=/tmp/r6a_$pm; rm -rf $T
  (cd /tmp && MSD_ARENA_SCALE_PERMILLE=$pm timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-check --no-also > $T.json 2>/dev/null)
  python - $T $pm $T.json <<'PY' | tee -a $OUT
import csv, glob, sys, json
try:
    d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1]); extra = "value %.0f reruns %s" % (d["value"], (d.get("pipeline_ms") or {}).get("reruns"))
except Exception as e:
    extra = "bench line: %r" % e
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if "msd_resolve" in n or "msd_scan_kernel<0, false, true>" in n:
            print("[permille %s] %-40s %4s calls avg %8.1f us min %8.1f max %8.1f   %s" % (sys.argv[2], n[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, extra))
PY
  rm -rf $T $T.json
done
