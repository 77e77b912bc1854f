#!/bin/bash
# round 5: what would the tests phase cost if only the pre-check's survivors were tested?  Timing proxy (results are wrong by
# construction): the scan kernel stopped after the preamble tests (debug flag 1) / after the conversion (2), with the three
# tests computed for every position (default library), for one position in six (abl6) and one in sixteen (abl16), the
# pre-check for all.  Variants from scripts/r4_variant_build.sh.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/tests_ablation.txt
run() { local name=$1 lib=$2 flags=$3; echo -n "[$name, debug flags $flags] " >> $O
  MSD_DEBUG_FLAGS=$flags MSD_LIBMODES_HIP=$lib timeout 600 python bench.py --steps 10 --warmup 2 --settle-seconds 2 --no-cpu-baseline --no-also --no-check 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('scan %.4f ms per launch (n=%d), job %.3f ms per pass' % (r['avg_launch_ms'], r['launches_timed'], d['ms_per_step']))" >> $O; }
echo "# $(date -u)" >> $O
D=$GRAFT_REPO_ROOT/readsb-protobuf_amd/csrc
run "all positions tested" $D/libmodes_hip.so 1
run "conversion only" $D/libmodes_hip.so 2
run "one position in 6 tested" $D/variants/abl6/libmodes_hip.so 1
run "one position in 16 tested" $D/variants/abl16/libmodes_hip.so 1
run "all positions tested" $D/libmodes_hip.so 1
run "whole kernel" $D/libmodes_hip.so 0
cat $O
