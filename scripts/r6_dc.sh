#!/bin/bash
# round 6 (GPU box): the --dcfilter chain walking along the lanes (MSD_DC_SYSTOLIC=1, the in-tree build) against round 4's one-lane
# chain (variants/dcold): the parity tests that carry --dcfilter, then the rate of both on 16 Mi samples with the whole-capture
# diff against the oracle; then the exhaustive SC16 converter sweep (scripts/r6_sc16_sweep.py).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary_entries.py tests/test_gpu_indep_demod.py -m gpu -q -k "dc or DC" > $O/dc_tests.txt 2>&1
tail -3 $O/dc_tests.txt
for v in new old; do
  lib=$GRAFT_REPO_ROOT/readsb-protobuf_amd/csrc/libmodes_hip.so
  [ $v = old ] && lib=$GRAFT_REPO_ROOT/readsb-protobuf_amd/csrc/variants/dcold/libmodes_hip.so
  for f in uc8 sc16; do
    line=$(MSD_LIBMODES_HIP=$lib timeout 900 python bench.py --dcfilter --format $f --samples 16777216 --steps 3 --warmup 1 --settle-seconds 0 --no-cpu-baseline --check --no-also 2>&1 | tail -1)
    echo "$line" | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('[dcfilter %s %s] value %.1f MS/s ms/step %.2f diff %s msgs %d' % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], d.get('message_set_diff_vs_oracle'), d['messages_per_step']))
except Exception as e:
    print('[dcfilter %s %s] FAILED %r' % (sys.argv[1], sys.argv[2], e))
" $v $f | tee -a $O/dc_rate.txt
  done
done
timeout 1500 python scripts/r6_sc16_sweep.py sc16 2>&1 | tail -1 | tee $O/sc16_sweep.txt
