#!/bin/bash
# round 5 (GPU box): the resolve beside the scan.  Prebuilt variants (scripts/r4_variant_build.sh):
#   w12          -DMSD_SCAN_WAVES=12                                   12-wave scan, 512-thread resolve
#   w12r256      + -DMSD_RESOLVE_WG=256 -DMSD_RESOLVE_SEG=512          slim resolve: 4 wavefronts, 44 KB of LDS, 90 registers
#   w12r256s640  the same with 640-hit segments
#   r256         the slim resolve behind the 16-wave scan
# each in order on one stream and (MSD_CHAIN_INLINE=0) with the resolve chain on side streams.
cd $GRAFT_REPO_ROOT
export OUT=${OUT:-gpurun_out/r5/coresident.txt} STEPS=${STEPS:-10}
mkdir -p $(dirname $OUT)
echo "# $(date -u) in order" >> $OUT
bash scripts/r4_variants_run.sh w12 w12r256 r256
echo "# side streams (MSD_CHAIN_INLINE=0)" >> $OUT
MSD_CHAIN_INLINE=0 bash scripts/r4_variants_run.sh w12 w12r256 w12r256s640 r256
echo "# default library, in order / side streams" >> $OUT
line=$(timeout 600 python bench.py --steps $STEPS --warmup 2 --settle-seconds 2 --no-cpu-baseline --check --no-also 2>&1 | tail -1)
echo "[base] $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('message_set_diff_vs_oracle'))")" | tee -a $OUT
line=$(MSD_CHAIN_INLINE=0 timeout 600 python bench.py --steps $STEPS --warmup 2 --settle-seconds 2 --no-cpu-baseline --check --no-also 2>&1 | tail -1)
echo "[base side] $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('message_set_diff_vs_oracle'))")" | tee -a $OUT
