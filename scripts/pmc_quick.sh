#!/bin/bash
# two PMC passes (instruction mix, wait states) for several MSD_DEBUG_FLAGS settings
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq
mkdir -p $OUT
cd /tmp
for flags in ${FLAGS:-0 4 1}; do
 i=0
 for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH"; do
  i=$((i+1))
  MSD_DEBUG_FLAGS=$flags rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --settle-seconds 0 --no-cpu-baseline --samples $((1<<26)) --batch $((1<<26)) > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" $flags <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(float)
for r in rows:
    if 'msd_scan' in r['Kernel_Name']:
        agg[r['Counter_Name']]+=float(r['Counter_Value'])
print('flags',sys.argv[2], {k: round(v/1e6,2) for k,v in sorted(agg.items())})
PY
  rm -rf $OUT/p$i
 done
done
