#!/bin/bash
# round 5: how much of the launch is the table gather?  The scan kernel stopped after the conversion (debug flag 2) and after
# the preamble tests (1), with the 136-pitch table (variants/base) and with the scan kernel's own 256-pitch swizzled copy
# (variants/s43 = the default).  Timing proxies: the results are wrong by construction.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b_lut_ablation.txt
run() { local name=$1 lib=$2 flags=$3; echo -n "[$name, debug flags $flags] " >> $O
  MSD_DEBUG_FLAGS=$flags MSD_LIBMODES_HIP=$lib timeout 600 python bench.py --steps 10 --warmup 2 --settle-seconds 4 --no-cpu-baseline --no-also --no-check --no-dropin 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('scan %.4f ms per launch (n=%d), job %.3f ms per pass' % (r['avg_launch_ms'], r['launches_timed'], d['ms_per_step']))" >> $O; }
echo "# $(date -u)" >> $O
D=$GRAFT_REPO_ROOT/readsb-protobuf_amd/csrc/variants
for rep in 1 2; do
for v in base s43; do
  run "$v conversion only" $D/$v/libmodes_hip.so 2
  run "$v conversion + tests" $D/$v/libmodes_hip.so 1
  run "$v whole kernel" $D/$v/libmodes_hip.so 0
done; done
cat $O
