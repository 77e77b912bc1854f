#!/bin/bash
# What does this box look like?  bench value, host-side waits per batch, GPU timeline gaps.
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-also 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['pipeline_ms']['scan_kernel_ms'], d['settle_ms_per_pass'][-3:])"
MSD_RESOLVE_TRACE=1 python bench.py --no-also --no-cpu-baseline --settle-seconds 1 2> gpurun_out/diag.err > /dev/null
grep "ahead:" gpurun_out/diag.err | tail -60 | python3 -c "
import sys,re
rows=[list(map(float,re.findall(r'([0-9.]+) ms', l))) for l in sys.stdin]
n=len(rows); print('ahead: n', n, 'mean wait/replay', [round(sum(r[k] for r in rows)/n,4) for k in range(2)], 'waits<5us:', sum(1 for r in rows if r[0]<0.005))"
grep "gpu resolve:" gpurun_out/diag.err | tail -60 | python3 -c "
import sys,re
rows=[list(map(float,re.findall(r'([0-9.]+) ms', l))) for l in sys.stdin]
n=len(rows); print('collect: n', n, 'mean [waits, replay, commit+next, power stats, waited for helper]', [round(sum(r[k] for r in rows)/n,4) for k in range(len(rows[0]))])"
grep "launch: at" gpurun_out/diag.err | tail -60 | python3 -c "
import sys,re
rows=[list(map(float,re.findall(r'([0-9.]+) ms', l))) for l in sys.stdin]
n=len(rows); d=[rows[i+1][0]-rows[i][0] for i in range(n-1)]; print('launch: enqueue mean', round(sum(r[1] for r in rows)/n,4), 'interval mean', round(sum(d)/len(d),4), 'max', round(max(d),3))"
BACK=20 ROWS=60 bash scripts/r3_timeline.sh diag_tl --settle-seconds 1 > /dev/null 2>&1; tail -1 gpurun_out/diag_tl/timeline.txt
python - <<'PY'
import re
rows=[l.split() for l in open('gpurun_out/diag_tl/timeline.txt') if 'msd_' in l]
ev=[(float(r[0]),float(r[1]),r[2]) for r in rows]
gaps=[]
for i in range(len(ev)-1):
    if 'scan' in ev[i][2] and 'resolve' in ev[i+1][2]: gaps.append(('s->r', round(ev[i+1][0]-ev[i][0]-ev[i][1],1)))
    if 'resolve' in ev[i][2] and 'scan' in ev[i+1][2]: gaps.append(('r->s', round(ev[i+1][0]-ev[i][0]-ev[i][1],1)))
print('gaps', gaps[-12:])
PY
