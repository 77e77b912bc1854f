#!/bin/bash
# run-to-run spread of the whole-job rate on this box against the host-side waits of the same runs
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
MSD_RESOLVE_TRACE=1 python bench.py --no-also --no-cpu-baseline --steps 20 --settle-seconds 1 2> gpurun_out/j.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', d['value'], d['ms_per_step'], end=' | ')"
grep "ahead: commit" gpurun_out/j.err | tail -150 | python3 -c "
import sys,re
rows=[list(map(float,re.findall(r'([0-9.]+) ms', l))) for l in sys.stdin]
n=len(rows); print('commit mean %.3f max %.3f, begin mean %.3f max %.3f' % (sum(r[0] for r in rows)/n, max(r[0] for r in rows), sum(r[1] for r in rows)/n, max(r[1] for r in rows)), end=' | ')"
grep "ahead: next" gpurun_out/j.err | tail -150 | python3 -c "
import sys,re
rows=[list(map(float,re.findall(r'([0-9.]+) ms', l))) for l in sys.stdin]
n=len(rows); w=sorted(r[0] for r in rows); print('ahead wait mean %.3f p10 %.3f min %.3f  <10us: %d of %d' % (sum(w)/n, w[n//10], w[0], sum(1 for x in w if x<0.010), n), end=' | ')"
grep "gpu resolve:" gpurun_out/j.err | tail -150 | python3 -c "
import sys,re
rows=[list(map(float,re.findall(r'([0-9.]+) ms', l))) for l in sys.stdin]
n=len(rows); print('commit+next mean %.3f max %.3f, helper wait mean %.3f max %.3f' % (sum(r[2] for r in rows)/n, max(r[2] for r in rows), sum(r[4] for r in rows)/n, max(r[4] for r in rows)))"
done
