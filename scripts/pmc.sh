#!/bin/bash
# PMC counter passes for the scan kernel (separate passes; no trace domains mixed in)
TAG=${1:-pmc}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --samples $((1<<27)) --batch $((1<<26)) ${BENCH_EXTRA}"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o pmc -- $CMD > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
disp=collections.Counter()
seen=set()
for r in rows:
    key=(r['Kernel_Name'][:60], r.get('Dispatch_Id'))
    if key not in seen:
        seen.add(key); disp[key[0]]+=1
for k,v in agg.items():
    if 'scan' in k or 'gather' in k or 'power' in k:
        print(k, 'dispatches', disp[k], {c: round(x/max(1,disp[k]),1) for c,x in v.items()})
PY
  else
    tail -3 $OUT/p$i.log
  fi
  rm -rf $OUT/p$i
done
