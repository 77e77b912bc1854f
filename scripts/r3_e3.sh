#!/bin/bash
cd $GRAFT_REPO_ROOT
for defs in "-DMSD_SCAN_WAVES=8" ""; do
  (cd readsb-protobuf_amd/csrc && MSD_EXTRA_DEFS="$defs" bash build.sh > /dev/null 2>&1) || { echo "[$defs] build failed"; continue; }
  for e in "MSD_CHAIN_INLINE=0" "MSD_CHAIN_INLINE=0 MSD_POWER_FUSED=1" "MSD_CHAIN_INLINE=1"; do
    for rep in 1 2; do
    echo -n "[$defs] $e: "
    env $e timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
    done
  done
done
(cd readsb-protobuf_amd/csrc && bash build.sh > /dev/null 2>&1)
