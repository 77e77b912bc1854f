#!/bin/bash
# round-4 helper (runs in the build container, no GPU needed): build libmodes_hip.so with extra definitions into
# readsb-protobuf_amd/csrc/variants/<name>/ so that one gpurun call can time many variants without compiling on the
# GPU box.  Usage: r4_variant_build.sh <name> "<defs>" [<name> "<defs>" ...]; select one with MSD_LIBMODES_HIP=<path>
# (readsb-protobuf_amd/capi.py; an experiment switch of the Python binding, not of the library).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/readsb-protobuf_amd/csrc
build_one() {
  local name=$1 defs=$2 tmp=/tmp/msd_variant_$1
  rm -rf $tmp && mkdir -p $tmp/pkg/csrc $tmp/include
  cp -r $SRC/*.c $SRC/*.h $SRC/*.hip $SRC/*.cpp $SRC/build.sh $SRC/host $tmp/pkg/csrc/ 2>/dev/null
  cp $ROOT/include/*.h $tmp/include/
  # build.sh uses -I../../include relative to csrc: mirror that layout
  (cd $tmp/pkg/csrc && MSD_EXTRA_DEFS="$defs" bash build.sh > $tmp/build.log 2>&1) || { echo "[$name] build FAILED"; tail -5 $tmp/build.log; return 1; }
  mkdir -p $SRC/variants/$name
  cp $tmp/pkg/csrc/libmodes_hip.so $SRC/variants/$name/
  echo "[$name] $defs -> variants/$name/libmodes_hip.so"
}
while [ $# -ge 2 ]; do build_one "$1" "$2" & shift 2; done
wait
