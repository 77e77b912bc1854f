#!/bin/bash
# round-2 GPU session helper: tests, bench, optional section timers.  Usage: r2_run.sh <tag> [tests] [bench] [timing]
TAG=$1; shift
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
for what in "$@"; do
case $what in
tests) timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log;;
bench) timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -1 $OUT/bench.json | cut -c1-900;;
benchq) timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/benchq.json 2> $OUT/benchq.err; echo "benchq rc=$?"; tail -1 $OUT/benchq.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'], d.get('pipeline_ms'))";;
timing) (cd readsb-protobuf_amd/csrc && MSD_EXTRA_DEFS=-DMSD_KERNEL_TIMING bash build.sh > /dev/null 2>&1); MSD_KERNEL_TIMING=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/timing.json 2> $OUT/timing.err; grep "section cycles" $OUT/timing.err; (cd readsb-protobuf_amd/csrc && bash build.sh > /dev/null 2>&1);;
ablate) for f in ${ABL:-0 4 1 2}; do echo -n "flags=$f: "; MSD_DEBUG_FLAGS=$f timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"; done;;
variant) # build with $VDEFS, quick bench (scan stream alone: MSD_CHAIN_INLINE), rebuild default
  (cd readsb-protobuf_amd/csrc && MSD_EXTRA_DEFS="$VDEFS" bash build.sh > /dev/null 2>&1); for f in ${ABL:-0}; do echo -n "[$VDEFS] flags=$f: "; MSD_CHAIN_INLINE=${INL:-1} MSD_DEBUG_FLAGS=$f timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'])"; done; (cd readsb-protobuf_amd/csrc && bash build.sh > /dev/null 2>&1);;
esac
done
