#!/bin/bash
# round-4 helper (GPU box): time prebuilt library variants (scripts/r4_variant_build.sh) with the default bench.
# Usage: r4_variants_run.sh <name> ...   (env BENCHARGS, STEPS, OUT)
cd $GRAFT_REPO_ROOT
OUT=${OUT:-gpurun_out/r4_variants.txt}; mkdir -p $(dirname $OUT)
for name in "$@"; do
  lib=$GRAFT_REPO_ROOT/readsb-protobuf_amd/csrc/variants/$name/libmodes_hip.so
  [ -f $lib ] || { echo "[$name] no library" | tee -a $OUT; continue; }
  line=$(MSD_LIBMODES_HIP=$lib timeout 600 python bench.py --steps ${STEPS:-10} --warmup 2 --settle-seconds ${SETTLE:-2} --no-cpu-baseline --check --no-also $BENCHARGS 2>&1 | tail -1)
  echo "$line" | python -c "
import sys,json
name=sys.argv[1]
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; p=d.get('pipeline_ms') or {}
    print('[%s] value %.0f ms/step %.3f scan_ms %.4f (n=%d; with records %s, alone %s) other_ms %s diff %s msgs %d' % (name, d['value'], d['ms_per_step'], r['avg_launch_ms'], r['launches_timed'], r.get('avg_launch_ms_with_records'), r.get('avg_launch_ms_scan_only'), p.get('other_kernels_ms'), d.get('message_set_diff_vs_oracle'), d['messages_per_step']))
except Exception as e:
    print('[%s] FAILED: %r' % (name, e))
" $name | tee -a $OUT
  [ -n "$VERBOSE" ] && echo "$line" >> $OUT.full
done
