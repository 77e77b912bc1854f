for t in 3 8 1000 3 1000; do
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-check --no-also --timing-interval $t 2>/dev/null | tail -1 > /tmp/ti_out.json
  python - $t <<'PY'
import sys,json
d=json.loads(open('/tmp/ti_out.json').read()); print("interval", sys.argv[1], d["value"], d["ms_per_step"], d["roofline"].get("launches_timed"))
PY
done
