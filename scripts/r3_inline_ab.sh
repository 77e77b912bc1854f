#!/bin/bash
# in-order chain against side streams for the configurations with follow-up kernels on the scan stream
cd $GRAFT_REPO_ROOT
run() { echo -n "[$*] "; timeout 600 env "$1" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-also --check "${@:2}" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], 'diff', d.get('message_set_diff_vs_oracle'))"; }
for ci in MSD_CHAIN_INLINE=0 MSD_CHAIN_INLINE=1; do
run $ci --mode-ac --fix 1
run $ci --format sc16 --samples 268435456
run $ci --format sc16 --samples 268435456 --mode-ac --fix 1
done
