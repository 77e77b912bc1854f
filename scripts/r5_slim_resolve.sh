export OUT=gpurun_out/r5/slim_inorder.txt STEPS=12
for i in 1 2; do
  bash scripts/r4_variants_run.sh r256s384 r256s416 r256s320 r256s384o6 r256s448
  line=$(timeout 600 python bench.py --steps 12 --warmup 2 --settle-seconds 2 --no-cpu-baseline --check --no-also 2>&1 | tail -1)
  echo "[default] $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.0f ms/step %.3f scan_ms %.4f diff %s' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], d.get('message_set_diff_vs_oracle')))")" | tee -a $OUT
done
