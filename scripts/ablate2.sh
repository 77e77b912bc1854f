cd $GRAFT_REPO_ROOT
for f in 0 16 8 4 1 2; do echo -n "flags=$f: "; MSD_RESOLVE_THREADS=48 MSD_DEBUG_FLAGS=$f python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_launch_ms'], d['pipeline_ms']['hits'], d['pipeline_ms']['tries'])"; done
