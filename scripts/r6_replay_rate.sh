#!/bin/bash
# round 6 (GPU box): the fused ifile handler on a capture file in /dev/shm by capture length and --batch-buffers (the turns of a regular
# file are read by four threads with pread since round 6): msd_replay --timing, wall time and rate.
cd $GRAFT_REPO_ROOT
python - <<PY
import __graft_entry__ as g
P = g.load_package()
iq = P.siggen.generate(P.siggen.make_cfg(seed=10901), 536870912)
iq.tofile("/dev/shm/r6_long.uc8")
iq[: 2 * 134217728].tofile("/dev/shm/r6_128.uc8")
iq[: 2 * 24000000].tofile("/dev/shm/r6_10s.uc8")
PY
for f in r6_10s r6_128 r6_long; do
  for bb in 64 256; do
    for rep in 1 2; do
      echo -n "$f --batch-buffers $bb: "; ./readsb-protobuf_amd/csrc/msd_replay --ifile /dev/shm/$f.uc8 --iformat uc8 --no-fix --no-output --timing --path fused --batch-buffers $bb 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('samples %d wall %.1f ms  %.0f Msamples/s  messages %d' % (d['samples'], d['wall_s']*1e3, d['msamples_per_s'], d['messages']))"
    done
  done
done
rm -f /dev/shm/r6_long.uc8 /dev/shm/r6_128.uc8 /dev/shm/r6_10s.uc8
