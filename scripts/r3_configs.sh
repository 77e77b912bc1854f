#!/bin/bash
# every configuration of DESIGN.md's table with --check (whole-capture diff against the oracle)
cd $GRAFT_REPO_ROOT
run() { echo -n "[$*] "; timeout 600 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --check "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'msgs', d['messages_per_step'], 'diff', d.get('message_set_diff_vs_oracle'))"; }
run
run --fix 1
run --fix 2
run --fields
run --mode-ac --fix 1
run --format sc16 --samples 268435456
run --format sc16q11 --samples 268435456
run --format sc16 --samples 268435456 --mode-ac --fix 1
