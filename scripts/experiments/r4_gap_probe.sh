# round-4 experiment: the gaps between scan and resolve with / without the record-completion event (MSD_DEBUG_FLAGS=256 skips it; not a product mode)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for f in 0 256; do
  O=$GRAFT_REPO_ROOT/gpurun_out/gap_$f; rm -rf $O; mkdir -p $O
  (cd /tmp && MSD_DEBUG_FLAGS=$f timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --settle-seconds 0 --no-cpu-baseline --no-check --no-also --timing-interval 1000 > /dev/null 2>&1)
  echo "== MSD_DEBUG_FLAGS=$f"; python scripts/timeline.py $O/trace 6 40 2>&1 | tail -14
done
