#!/bin/bash
# rocprofv3 per-kernel stats of a bench variant: r2_kprof.sh <tag> <bench args...>
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kprof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check "$@" > $OUT/bench.json 2> $OUT/rocprof.log
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
rm -rf $OUT/trace
python $GRAFT_REPO_ROOT/scripts/kstats.py $OUT
tail -1 $OUT/bench.json | cut -c1-200
