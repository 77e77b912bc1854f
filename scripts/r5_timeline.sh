#!/bin/bash
# GPU timeline of a prebuilt library variant under rocprofv3 (kernel trace only): r5_timeline.sh <tag> <variant|-> [bench args...]
# env passes through (MSD_CHAIN_INLINE=0 for the side-stream layout)
TAG=$1; VAR=$2; shift 2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5/$TAG
mkdir -p $OUT
[ "$VAR" != "-" ] && export MSD_LIBMODES_HIP=$GRAFT_REPO_ROOT/readsb-protobuf_amd/csrc/variants/$VAR/libmodes_hip.so
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-check --no-also "$@" > $OUT/bench.json 2> $OUT/rocprof.log
python $GRAFT_REPO_ROOT/scripts/timeline.py $OUT/trace ${BACK:-6} ${ROWS:-40} > $OUT/timeline.txt
rm -rf $OUT/trace
echo "== $TAG"; cat $OUT/timeline.txt
