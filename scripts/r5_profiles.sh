#!/bin/bash
# Round-5 evidence set -> gpurun_out/r05/ (copied to profiles/r05_* afterwards).  One gpurun call; every profiler run under `timeout`.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err

kstats() { # <name> <bench args...>: rocprofv3 --kernel-trace --stats summary + the scan kernel's first dispatches
  local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-also "$@" > $O/${name}_bench_under_profiler.json 2> $O/${name}_rocprof.log)
  for f in $(find $O/trace_$name -name "*kernel_stats.csv"); do cp $f $O/${name}_kernel_stats.csv; done
  for f in $(find $O/trace_$name -name "*kernel_trace.csv"); do head -1 $f > $O/${name}_scan_kernel_trace_head.csv; grep msd_scan $f | head -12 >> $O/${name}_scan_kernel_trace_head.csv; done
  rm -rf $O/trace_$name $O/${name}_rocprof.log
}
kstats uc8
kstats sc16 --format sc16 --samples 268435456
kstats modeac --mode-ac --fix 1

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
traffic uc8
traffic sc16 --format sc16 --samples 268435456
traffic modeac --mode-ac --fix 1

for cfg in "uc8:" "sc16:--format sc16 --samples 268435456" "modeac:--mode-ac --fix 1"; do
  name=${cfg%%:*}; args=${cfg#*:}
  mkdir -p $O/tl_$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tl_$name/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-also $args > /dev/null 2>&1)
  python scripts/timeline.py $O/tl_$name/trace 6 40 > $O/${name}_timeline.txt 2>&1
  rm -rf $O/tl_$name
done

: > $O/configs.txt
for f in "" "--fix 1" "--fix 2" "--fields" "--mode-ac --fix 1" "--format sc16 --samples 268435456" "--format sc16q11 --samples 268435456" "--format sc16q11 --samples 268435456 --sc16q11-table-bits 8" "--format sc16 --samples 268435456 --mode-ac --fix 1"; do
  echo -n "bench.py $f : " >> $O/configs.txt
  timeout 600 python bench.py --no-cpu-baseline --no-also --check $f 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'scan_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'msgs', d['messages_per_step'], 'diff', d.get('message_set_diff_vs_oracle'))" >> $O/configs.txt
done
for e in "MSD_EMIT_FUSED=0" "MSD_LEAN=0" "MSD_RESOLVE_AHEAD=0" "MSD_POWER_FUSED=0" "MSD_WAIT_INPUTS_ON_STREAM=1" "MSD_CHAIN_INLINE=0" "MSD_LEAN=0 MSD_RESOLVE_AHEAD=0 MSD_POWER_FUSED=0 MSD_WAIT_INPUTS_ON_STREAM=1"; do
  echo -n "$e : " >> $O/configs.txt; env $e timeout 600 python bench.py --no-cpu-baseline --no-also --no-check 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'scan_ms', d['roofline']['avg_launch_ms'])" >> $O/configs.txt
done
cat $O/configs.txt
tail -1 $O/bench_default.json | cut -c1-400
