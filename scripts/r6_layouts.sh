#!/bin/bash
# round 6 (GPU box): the stream-layout switches on the two side-stream workloads (Mode A/C + --fix, 16-bit IQ), which round 5 swept on the default workload only
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_layouts.txt; : > $O
for w in "--mode-ac --fix 1" "--format sc16 --samples 268435456"; do
  for e in "" "MSD_CHAIN_INLINE=1" "MSD_POWER_FUSED=1" "MSD_CHAIN_INLINE=1 MSD_POWER_FUSED=1" "MSD_CHAIN_INLINE=1 MSD_POWER_FUSED=0" "MSD_EMIT_FUSED=0" "MSD_REPASS_AUX=1" "MSD_WAIT_INPUTS_ON_STREAM=1"; do
    echo -n "bench.py $w  [$e] : " >> $O
    env $e timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --check $w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'scan_ms', d['roofline']['avg_launch_ms'], 'diff', d.get('message_set_diff_vs_oracle'))" >> $O 2>&1
  done
done
cat $O
