cd $GRAFT_REPO_ROOT
for f in 0 1 2; do echo "== debug_flags=$f"; MSD_DEBUG_FLAGS=$f python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['pipeline_ms'])"; done
echo "== noise only"; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --msgs-per-sec 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['pipeline_ms'])"
