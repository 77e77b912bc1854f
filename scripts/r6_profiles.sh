#!/bin/bash
# Round-6 evidence set -> gpurun_out/r06/ (copied to profiles/r06_* afterwards).  One gpurun call; every profiler run under `timeout`.
# New against r5_profiles.sh: the kernel statistics come from >= 30 launches with the first launch of each instantiation excluded by
# count (r06_scan_launches.json, what bench.py's roofline.frac_rocprof reads), the conversion-only floor, the density lines.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
PARTS=${PARTS:-bench kstats traffic timeline floor configs density}
has() { [[ " $PARTS " == *" $1 "* ]]; }

has bench && timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err

kstats() { # <name> <steps> <samples per launch> <bytes per sample> <bench args...>
  local name=$1 steps=$2 spl=$3 bps=$4; shift 4
  local cmd="rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps $steps --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-also --no-check $*"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps $steps --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-also --no-check "$@" > $O/${name}_bench_under_profiler.json 2> $O/${name}_rocprof.log)
  for f in $(find $O/trace_$name -name "*kernel_stats.csv"); do cp $f $O/${name}_kernel_stats.csv; done
  for f in $(find $O/trace_$name -name "*kernel_trace.csv"); do head -1 $f > $O/${name}_scan_kernel_trace_head.csv; grep msd_scan $f | head -12 >> $O/${name}_scan_kernel_trace_head.csv; done
  python scripts/r6_scan_launches.py $O/trace_$name $O/${name}_scan_launches.json $spl $bps "$cmd"
  rm -rf $O/trace_$name $O/${name}_rocprof.log
}
if has kstats; then
  kstats uc8 9 134217728 2                                   # 40 scan launches (4 per pass, 10 passes with the warm-up)
  kstats sc16 9 67108864 4 --format sc16 --samples 268435456
  kstats modeac 9 134217728 2 --mode-ac --fix 1
fi

traffic() { # <name> <bench args...>: FETCH_SIZE / WRITE_SIZE per kernel, separate --pmc passes, no trace domains
  local name=$1; shift
  mkdir -p $O/pmc_$name
  for ctr in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$name/$ctr -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --settle-seconds 0 --no-cpu-baseline --no-check --no-also "$@" > $O/pmc_$name/$ctr.log 2>&1)
  done
  python3 - $O/pmc_$name $O/${name}_traffic.json "$*" <<'PY'
import csv, glob, json, sys, collections
src, dst, args = sys.argv[1], sys.argv[2], sys.argv[3]
res = {"bench_args": args}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{src}/{ctr}/**/*counter_collection.csv", recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    vals = {"with_records": collections.defaultdict(float), "scan_only": collections.defaultdict(float)}
    other = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        if r["Counter_Name"] != ctr:
            continue
        name = r["Kernel_Name"]
        if "msd_scan_kernel" in name:
            targs = name[name.index("msd_scan_kernel<"):].split(">")[0]
            vals["with_records" if targs.endswith("true") else "scan_only"][r["Dispatch_Id"]] += float(r["Counter_Value"])
        elif "msd_" in name:
            short = name[name.index("msd_"):].split("(")[0].split("<")[0]
            other[short][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for kind, d in vals.items():
        v = sorted(d.values())
        if v:
            suffix = "" if kind == "with_records" else "_scan_only"
            res[ctr + "_KB_per_launch" + suffix] = v[len(v) // 2]
            res[ctr + "_launches" + suffix] = len(v)
    res[ctr + "_KB_per_launch_other_kernels"] = {k: sorted(d.values())[len(d) // 2] for k, d in other.items()}
line = [l for l in open(f"{src}/FETCH_SIZE.log") if l.startswith("{")][-1]
res["samples_per_launch"] = json.loads(line)["config"]["batch_samples"]  # the bench's own batch size in that run
res["note"] = ("rocprofv3 --pmc, median over launches of msd_scan_kernel with the record slice (_scan_only: the launches without); "
               "gfx950 FETCH_SIZE counts 64 B per 128 B request on wide coalesced reads (MI355X_MICROARCH.md), so fetch bytes = 2 * FETCH_SIZE * 1024; "
               "_other_kernels: median per launch of every other kernel of the run")
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res))
PY
  rm -rf $O/pmc_$name
}
if has traffic; then
  traffic uc8
  traffic sc16 --format sc16 --samples 268435456
  traffic modeac --mode-ac --fix 1
fi

if has timeline; then
for cfg in "uc8:" "sc16:--format sc16 --samples 268435456" "modeac:--mode-ac --fix 1"; do
  name=${cfg%%:*}; args=${cfg#*:}
  mkdir -p $O/tl_$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl_$name/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-also --no-check $args > /dev/null 2>&1)
  python scripts/timeline.py $O/tl_$name/trace 6 40 > $O/${name}_timeline.txt 2>&1
  rm -rf $O/tl_$name
done
fi

if has floor; then # the scan kernel stopped after the conversion (MSD_DEBUG_FLAGS=2) and after the tests (1): HIP-event launch times
  : > $O/floor.txt
  for fl in 2 1 0; do
    echo -n "MSD_DEBUG_FLAGS=$fl : " >> $O/floor.txt
    MSD_DEBUG_FLAGS=$fl timeout 600 python bench.py --steps 10 --warmup 2 --settle-seconds 2 --no-cpu-baseline --no-also --no-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('scan_ms', r['avg_launch_ms'], 'min_max', r['launch_ms_min_max'], 'launches', r['launches_timed'])" >> $O/floor.txt
  done
  cat $O/floor.txt
fi

if has configs; then
: > $O/configs.txt
for f in "" "--fix 1" "--fix 2" "--fields" "--mode-ac --fix 1" "--format sc16 --samples 268435456" "--format sc16q11 --samples 268435456" "--format sc16q11 --samples 268435456 --sc16q11-table-bits 8" "--format sc16 --samples 268435456 --mode-ac --fix 1"; do
  echo -n "bench.py $f : " >> $O/configs.txt
  timeout 600 python bench.py --no-cpu-baseline --no-also --check $f 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'scan_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'msgs', d['messages_per_step'], 'diff', d.get('message_set_diff_vs_oracle'))" >> $O/configs.txt
done
cat $O/configs.txt
fi

if has density; then
  bash scripts/r4_density.sh > /dev/null 2>&1
  cp gpurun_out/r4_density.txt $O/density.txt
  cat $O/density.txt
fi
tail -1 $O/bench_default.json 2>/dev/null | cut -c1-400
