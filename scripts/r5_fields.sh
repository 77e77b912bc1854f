#!/bin/bash
# --fields (196 B per message home): where did 197.6 -> 185.7 GS/s go between rounds 3 and 4?  The arenas (4 x), the record
# kernel, or the box?  r5_fields.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/fields.txt
run() { echo -n "$1 : " >> $O; shift; env "$@" timeout 600 python bench.py --fields --steps 10 --warmup 2 --settle-seconds 2 --no-cpu-baseline --no-also --no-check 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d.get('pipeline_ms') or {}
print(d['value'], d['ms_per_step'], 'scan', d['roofline']['avg_launch_ms'], 'other', p.get('other_kernels_ms'), 'msgs', d['messages_per_step'])" >> $O; }
echo "# $(date -u)" >> $O
run "default --fields" X=1
run "arenas at the base size" MSD_ARENA_SCALE_PERMILLE=1000
run "records by DMA" MSD_RECORDS_DMA=1
run "no helper thread" MSD_NO_HELPER=1
run "default again" X=1
echo -n "without --fields : " >> $O; timeout 600 python bench.py --steps 10 --warmup 2 --settle-seconds 2 --no-cpu-baseline --no-also --no-check 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O
cat $O
