#!/bin/bash
# round 6 (GPU box): the parallel-in-time --dcfilter: its tests, the rate of bench.py --dcfilter with the whole-capture diff, and the
# kernel timeline of one batch under rocprofv3 -> gpurun_out/dc/
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/dc; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dc_parallel.py -q 2>&1 | tail -8 > $O/tests_new.txt; cat $O/tests_new.txt
timeout 600 python -m pytest tests -m gpu -q -k "dc" 2>&1 | tail -3 > $O/tests_dc.txt; cat $O/tests_dc.txt
: > $O/dc_rate.txt
for f in uc8 sc16; do
  for n in 16777216 134217728; do
    line=$(timeout 600 python bench.py --dcfilter --format $f --samples $n --steps 5 --warmup 1 --settle-seconds 0 --no-cpu-baseline --check --no-also 2>/dev/null | tail -1)
    echo "$line" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[dcfilter %s %s samples] value %.1f MS/s ms/step %.3f diff %s msgs %d' % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], d.get('message_set_diff_vs_oracle'), d['messages_per_step']))
" $f $n | tee -a $O/dc_rate.txt
  done
done
export TMPDIR=/tmp; cd /tmp
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o dc -- python $GRAFT_REPO_ROOT/bench.py --dcfilter --samples ${PROF_SAMPLES:-16777216} --steps 5 --warmup 1 --settle-seconds 0 --no-check --no-cpu-baseline --no-also > $O/bench_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python3 - <<'PY' | tee $O/timeline.txt
import csv, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/dc/prof/"
rows = list(csv.DictReader(open(O + "dc_kernel_stats.csv")))
for r in rows[:8]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), "total ms %8.3f" % (int(r['TotalDurationNs']) / 1e6), "avg us %8.1f" % (float(r['AverageNs']) / 1e3))
rows = list(csv.DictReader(open(O + "dc_kernel_trace.csv")))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'dcp_init' in r['Kernel_Name']]
i0 = idx[-1]; t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0 + 70]:
    n = r['Kernel_Name']; n = n[n.find('msd_'):][:28]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if d > 8 or 'dcp' not in n:
        print("%9.1f %8.1f %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, d, n))
PY
rm -rf $O/prof
