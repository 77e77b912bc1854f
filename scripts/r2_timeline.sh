#!/bin/bash
# GPU timeline of the default bench under rocprofv3 (kernel + memory-copy trace) -> gpurun_out/<tag>/timeline.txt
TAG=${1:-tl}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/rocprof.log
python $GRAFT_REPO_ROOT/scripts/timeline.py $OUT/trace 6 60 > $OUT/timeline.txt
rm -rf $OUT/trace
cat $OUT/timeline.txt
MSD_RESOLVE_TRACE=1 python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2> $OUT/host_trace.txt > /dev/null
tail -12 $OUT/host_trace.txt
