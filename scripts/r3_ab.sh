#!/bin/bash
# A/B of environment switches on the default bench, interleaved repeats: r3_ab.sh "ENV1=.. ENV2=.." "..." ...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for e in "$@"; do
    echo -n "[$e] "
    env $e python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline $BENCHARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done
