#!/bin/bash
# round 5 (last session): what the record slice costs the scan kernel, and which half of it: MSD_DEBUG_FLAGS 64 builds the
# records and does not store them, 128 skips the slice (the run's messages are wrong by construction: --no-check).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b_emit_slice.txt
run() { echo -n "$1 : " >> $O; shift; env "$@" timeout 600 python bench.py --steps 10 --warmup 2 --settle-seconds 5 --no-cpu-baseline --no-also --no-dropin --no-check 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('scan with records', r.get('avg_launch_ms_with_records'), 'without', r.get('avg_launch_ms_scan_only'), 'ms/pass', d['ms_per_step'])" >> $O; }
echo "# $(date -u)" >> $O
for rep in 1 2; do
run "default" X=1
run "records built, not stored (flag 64)" MSD_DEBUG_FLAGS=64
run "slice skipped (flag 128)" MSD_DEBUG_FLAGS=128
done
cat $O
