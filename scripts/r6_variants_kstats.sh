#!/bin/bash
# round-6 helper (GPU box): for prebuilt library variants (scripts/r4_variant_build.sh) the bench line with --check and the
# rocprofv3 --kernel-trace --stats averages of the resolve and scan kernels.  Usage: r6_variants_kstats.sh <name> ...
# (env BENCHARGS, STEPS, OUT; the name "default" = the in-tree library)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/r6_variants.txt}; mkdir -p $(dirname $OUT)
for name in "$@"; do
  lib=$GRAFT_REPO_ROOT/readsb-protobuf_amd/csrc/variants/$name/libmodes_hip.so
  [ "$name" = default ] && lib=$GRAFT_REPO_ROOT/readsb-protobuf_amd/csrc/libmodes_hip.so
  [ -f $lib ] || { echo "[$name] no library" | tee -a $OUT; continue; }
  line=$(MSD_LIBMODES_HIP=$lib timeout 600 python bench.py --steps ${STEPS:-10} --warmup 2 --settle-seconds ${SETTLE:-2} --no-cpu-baseline --check --no-also $BENCHARGS 2>&1 | tail -1)
  echo "$line" | python -c "
import sys,json
name=sys.argv[1]
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']
    print('[%s] value %.0f ms/step %.3f scan_ms %.4f diff %s msgs %d' % (name, d['value'], d['ms_per_step'], r['avg_launch_ms'], d.get('message_set_diff_vs_oracle'), d['messages_per_step']))
except Exception as e:
    print('[%s] FAILED: %r' % (name, e))
" $name | tee -a $OUT
  T=/tmp/r6v_$name; rm -rf $T
  (cd /tmp && MSD_LIBMODES_HIP=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-check --no-also $BENCHARGS > /dev/null 2>&1)
  python - $T $name <<'PY' | tee -a $OUT
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if "msd_" in n:
            print("    [%s] %-40s %4s calls avg %8.1f us min %8.1f max %8.1f" % (sys.argv[2], n[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  rm -rf $T
done
