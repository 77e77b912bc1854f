#!/bin/bash
# round 6 (GPU box): what msd_resolve_kernel costs when there is next to nothing to resolve -- rocprofv3 kernel statistics of bench.py at the
# default density, at --threshold 75 / 400 and on a quiet band: is the kernel's time the tries' or the kernel's own?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
: > $O/resolve_floor.txt
run() { # <label> <bench args...>
  local label=$1; shift
  rm -rf $O/trace_rf
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_rf -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-also --no-check "$@" > $O/rf_line.json 2>/dev/null)
  python3 - "$label" $O <<'PY' >> $O/resolve_floor.txt
import csv, glob, json, sys
label, O = sys.argv[1], sys.argv[2]
d = json.loads(open(O + "/rf_line.json").read().strip().splitlines()[-1])
f = glob.glob(O + "/trace_rf/**/*kernel_stats.csv", recursive=True)[0]
rows = {r["Name"]: r for r in csv.DictReader(open(f))}
def avg(sub):
    sel = [(int(r["Calls"]), float(r["AverageNs"]) / 1e3) for n, r in rows.items() if sub in n]
    c = sum(x for x, _ in sel)
    return (sum(x * y for x, y in sel) / c if c else float("nan")), c
rs, rc = avg("msd_resolve_kernel"); sc, scc = avg("msd_scan_kernel")
tm = d.get("timing_per_batch") or {}
print("%-34s value %7.1f GS/s  period %6.1f us per batch  scan %6.1f us (%d)  resolve %6.1f us (%d launches)  msgs %d" % (
    label, d["value"] / 1e3, d["ms_per_step"] * 1e3 / (d["config"]["samples_per_gpu"] / d["config"]["batch_samples"]), sc, scc, rs, rc, d["messages_per_step"]))
PY
  rm -rf $O/trace_rf $O/rf_line.json
}
run "(default)"
run "--threshold 75" --threshold 75
run "--threshold 400" --threshold 400
run "quiet band, no traffic" --msgs-per-sec 0 --noise-fs 0.005
run "12000 frames/s, sigma 0.06" --msgs-per-sec 12000 --noise-fs 0.06
run "--threshold 40" --threshold 40
cat $O/resolve_floor.txt
