#!/bin/bash
# round-3 helper: build the library with extra definitions and time the default bench under several stream layouts.
# Usage: r3_variants.sh "<defs A>" "<defs B>" ...   (env LAYOUTS="1 0" = MSD_CHAIN_INLINE values, BENCHARGS)
cd $GRAFT_REPO_ROOT
for defs in "$@"; do
  (cd readsb-protobuf_amd/csrc && MSD_EXTRA_DEFS="$defs" bash build.sh > /dev/null 2>&1) || { echo "[$defs] build failed"; continue; }
  for inl in ${LAYOUTS:-1 0}; do
    echo -n "[$defs] inline=$inl: "
    MSD_CHAIN_INLINE=$inl timeout 300 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline $BENCHARGS 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d.get('message_set_diff_vs_oracle'))"
  done
done
(cd readsb-protobuf_amd/csrc && bash build.sh > /dev/null 2>&1)
