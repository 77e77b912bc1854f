#!/bin/bash
# GPU timeline of a bench configuration under rocprofv3 (kernel + memory-copy trace): r3_timeline.sh <tag> [bench args...]
TAG=${1:-tl}; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --settle-seconds 0 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/rocprof.log
python $GRAFT_REPO_ROOT/scripts/timeline.py $OUT/trace ${BACK:-6} ${ROWS:-60} > $OUT/timeline.txt
rm -rf $OUT/trace
cat $OUT/timeline.txt
