#!/bin/bash
# round 5 (last session): --fields with the records AND the field records written to HBM and fetched by a copy on an idle stream
# (MSD_RECORDS_DMA=1; until now only the 56-byte records went that way and the 140-byte field records stayed kernel stores to host memory)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b_fields_dma.txt
run() { echo -n "$1 : " >> $O; shift; env "$@" timeout 600 python bench.py --fields --steps 10 --warmup 2 --settle-seconds 4 --no-cpu-baseline --no-also --no-dropin --check 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d.get('pipeline_ms') or {}
print(d['value'], d['ms_per_step'], 'scan', d['roofline']['avg_launch_ms'], 'diff', d.get('message_set_diff_vs_oracle'), 'msgs', d['messages_per_step'])" >> $O; }
echo "# $(date -u)" >> $O
for rep in 1 2; do
run "--fields, kernel stores to host memory (default)" X=1
run "--fields, records and fields by DMA" MSD_RECORDS_DMA=1
done
cat $O
