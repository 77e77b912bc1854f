#!/bin/bash
# round 4: how the whole-job rate depends on the candidate density -> gpurun_out/r4_density.txt (-> profiles/r04_density.txt).
# 1 GiB UC8 through bench.py --check (zero diff against the oracle on every line, or the run aborts).
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4_density.txt
echo "# bench.py --steps 6 --warmup 2 --check, 1 GiB UC8, one MI355X; hits / tries per sample from msd_timing (per 128 Mi-sample batch)" > $OUT
run() {
  line=$(timeout 900 python bench.py --steps 6 --warmup 2 --settle-seconds 2 --no-cpu-baseline --check --no-also "$@" 2>&1 | tail -1)
  echo "$line" | python -c "
import sys,json
a=' '.join(sys.argv[1:]) or '(default)'
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; p=d['pipeline_ms']; n=r['samples_per_launch']
    print('%-62s %8.1f GS/s  scan %.4f ms  hits/sample %.4f  tries/sample %.4f  msgs %7d  reruns %d  host-resolved batches %d  diff %s' % (a, d['value']/1e3, r['avg_launch_ms'], p['hits']/n, p['tries']/n, d['messages_per_step'], p['reruns'], p['resolve_fallback'], d.get('message_set_diff_vs_oracle')))
except Exception as e:
    print('%-62s FAILED %r' % (a, e))
" "$@" | tee -a $OUT
}
run
run --threshold 40
run --threshold 75
run --threshold 400
run --msgs-per-sec 12000 --noise-fs 0.06
run --msgs-per-sec 12000 --noise-fs 0.06 --threshold 40
run --msgs-per-sec 12000 --noise-fs 0.06 --threshold 75
run --msgs-per-sec 12000 --noise-fs 0.06 --threshold 400
run --msgs-per-sec 12000 --noise-fs 0.06 --fix 1
run --input random
run --input random --threshold 40
run --input random --threshold 400
run --msgs-per-sec 0 --noise-fs 0.005
