#!/bin/bash
# steady state of the replay tool's two paths on a long capture (1024 buffers), to separate the fixed start-up cost
# (module load, first launches) from the per-buffer cost: r5_dropin_long.sh [samples]
cd $GRAFT_REPO_ROOT
N=${1:-134217728}
python - <<PY
import __graft_entry__ as g
P = g.load_package()
iq = P.siggen.generate(P.siggen.make_cfg(seed=10901), $N)
iq.tofile("/dev/shm/r5_long.uc8")
PY
for path in magbuf fused; do
  for rep in 1 2; do
    echo -n "$path $N samples: "; ./readsb-protobuf_amd/csrc/msd_replay --ifile /dev/shm/r5_long.uc8 --iformat uc8 --no-fix --no-output --timing --path $path 2>&1 | tail -1
  done
done
rm -f /dev/shm/r5_long.uc8
