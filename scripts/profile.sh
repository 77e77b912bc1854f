#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command; summaries go to gpurun_out/prof_<tag>/
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_profiler.json 2> $OUT/rocprof.log
find $OUT/trace -name "*kernel_stats*" | head -3
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
for f in $(find $OUT/trace -name "*kernel_trace.csv"); do head -1 $f > $OUT/kernel_trace_head.csv; grep msd_scan $f | head -40 >> $OUT/kernel_trace_head.csv; done
rm -rf $OUT/trace
cat $OUT/kernel_stats.csv | head -12
tail -1 $OUT/bench_under_profiler.json | cut -c1-400
