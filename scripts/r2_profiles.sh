#!/bin/bash
# Round-2 evidence set -> gpurun_out/r02/ (copied to profiles/r02_* afterwards)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash scripts/profile.sh r02 > $O/profile.log 2>&1
cp gpurun_out/prof_r02/kernel_stats.csv $O/kernel_stats.csv
cp gpurun_out/prof_r02/kernel_trace_head.csv $O/scan_kernel_trace_head.csv
cp gpurun_out/prof_r02/bench_under_profiler.json $O/bench_under_profiler.json
bash scripts/pmc_traffic.sh > $O/traffic.log 2>&1; cp gpurun_out/traffic.json $O/traffic.json
FLAGS="0 4 1 2" bash scripts/pmc_quick.sh > $O/pmc_counters.txt 2>&1
bash scripts/r2_timeline.sh r02_tl > /dev/null 2>&1; cp gpurun_out/r02_tl/timeline.txt $O/timeline.txt; cp gpurun_out/r02_tl/host_trace.txt $O/host_trace.txt
MSD_CHAIN_INLINE=0 bash scripts/r2_timeline.sh r02_tl_side > /dev/null 2>&1; cp gpurun_out/r02_tl_side/timeline.txt $O/timeline_side_streams.txt
: > $O/configs.txt
for f in "" "--fix 1" "--fix 2" "--fields" "--mode-ac --fix 1" "--format sc16 --samples 268435456" "--format sc16q11 --samples 268435456" "--format sc16 --samples 268435456 --mode-ac --fix 1"; do
  echo -n "bench.py $f : " >> $O/configs.txt
  python bench.py --no-cpu-baseline --check $f 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'scan_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'msgs', d['messages_per_step'], 'diff', d.get('message_set_diff_vs_oracle'))" >> $O/configs.txt
done
echo -n "MSD_EMIT_FUSED=0 (record kernel of its own) : " >> $O/configs.txt; MSD_EMIT_FUSED=0 python bench.py --no-cpu-baseline --no-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'scan_ms', d['roofline']['avg_launch_ms'])" >> $O/configs.txt
for m in 0 1; do echo -n "MSD_CHAIN_INLINE=$m : " >> $O/configs.txt; MSD_CHAIN_INLINE=$m python bench.py --no-cpu-baseline --no-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'scan_ms', d['roofline']['avg_launch_ms'])" >> $O/configs.txt; done
python scripts/pcie_rate.py >> $O/configs.txt 2>&1
cat $O/configs.txt
tail -1 $O/bench_default.json | cut -c1-300
