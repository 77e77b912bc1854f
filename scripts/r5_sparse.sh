#!/bin/bash
# round 5: the sparse preamble tests (-DMSD_TESTS_SPARSE=1) against the default build: bench --check (zero diff on 1 GiB), then
# a slice of the GPU suite against the variant library.  r5_sparse.sh <variant> [more variants]
cd $GRAFT_REPO_ROOT
export OUT=gpurun_out/r5/sparse.txt STEPS=10
echo "# $(date -u)" >> $OUT
bash scripts/r4_variants_run.sh "$@"
line=$(timeout 600 python bench.py --steps $STEPS --warmup 2 --settle-seconds 2 --no-cpu-baseline --check --no-also 2>&1 | tail -1)
echo "[default] $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.0f ms/step %.3f scan_ms %.4f diff %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], d.get('message_set_diff_vs_oracle')))")" | tee -a $OUT
bash scripts/r4_variants_run.sh "$@"
