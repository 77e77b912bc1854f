#!/bin/bash
# round-4 PMC passes for the scan kernel: instruction fetch, LDS / TA queues, pipe cycles (separate passes, no trace domains).
# Usage: r4_pmc.sh <tag> [bench args...]   env: MSD_LIBMODES_HIP (variant library), SETS="A B C D"
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_pmc; mkdir -p $OUT
declare -A S
S[A]="SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"
S[B]="SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"
S[C]="SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
S[D]="TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
S[E]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
S[W]="WRITE_SIZE"
S[R]="FETCH_SIZE"
S[F]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH"
cd /tmp
for k in ${SETS:-A B C D}; do
  rm -rf $OUT/p
  timeout 180 rocprofv3 --pmc ${S[$k]} --output-format csv -d $OUT/p -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --settle-seconds 0 --no-cpu-baseline --no-check --no-also --samples $((1<<26)) --batch $((1<<26)) "$@" > $OUT/p.log 2>&1
  f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
  python3 - "$f" "$TAG" $k <<'PY' | tee -a $OUT/$TAG.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(float); disp=set()
for r in rows:
    if (__import__('os').environ.get('KFILTER') or 'msd_scan') in r['Kernel_Name']:
        agg[r['Counter_Name']]+=float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
n=max(1,len(disp))
print(sys.argv[2], 'set', sys.argv[3], 'scan dispatches', n, {k: round(v/n/1e6,3) for k,v in sorted(agg.items())})
PY
  else
    echo "$TAG set $k: no counters"; tail -5 $OUT/p.log
  fi
done
rm -rf $OUT/p
