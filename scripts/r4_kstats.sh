#!/bin/bash
# round 4: rocprofv3 --kernel-trace --stats summary of a bench run -> gpurun_out/r4_kstats/<tag>_kernel_stats.csv (+ the run's own line)
# Usage: r4_kstats.sh <tag> [bench args...]
TAG=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_kstats; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-also "$@" > $O/${TAG}_bench_under_profiler.json 2> $O/${TAG}_rocprof.log)
for f in $(find $O/trace_$TAG -name "*kernel_stats.csv"); do cp $f $O/${TAG}_kernel_stats.csv; done
for f in $(find $O/trace_$TAG -name "*kernel_trace.csv"); do head -1 $f > $O/${TAG}_scan_kernel_trace_head.csv; grep msd_scan $f | head -12 >> $O/${TAG}_scan_kernel_trace_head.csv; cp $f $O/${TAG}_kernel_trace_full.csv; done
rm -rf $O/trace_$TAG
python3 - $O/${TAG}_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    n=r['Name']; n=n[n.find('msd_'):][:60] if 'msd_' in n else n[:60]
    print('%-62s calls %5s avg %10.1f us  min %9.1f  max %9.1f  %5s%%' % (n, r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, r['Percentage']))
PY
