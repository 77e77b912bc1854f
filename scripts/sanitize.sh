#!/bin/bash
# Sanitizer builds of the host C (the reference builds without any, Makefile:13, and carries
# fifo.c:141,192-197,219).  scripts/sanitize.sh asan|tsan OUTDIR builds, with gcc -fsanitize=address,undefined or
# -fsanitize=thread:
#   OUTDIR/libmsd_host.so   every file of csrc/host/ (FIFO, ifile handler with its reader / consumer threads, wire formats,
#                           converter and demodulator adapters)
#   OUTDIR/libmodes_hip.so  the library's plain-C parts (msd_tables.c, msd_resolve.c with its resolver threads,
#                           msd_fields.c) instrumented, relinked with the HIP objects of the ordinary build (device code
#                           and its C++ launcher are not gcc's to instrument)
#   OUTDIR/msd_replay       the replay tool
#   OUTDIR/fifo_stress      tests/c/fifo_stress.c: producer twelve buffers ahead, consumer, a halt in mid-stream
# Run the python tests against them with MSD_LIBMODES_HIP=OUTDIR/libmodes_hip.so and LD_PRELOAD=$(gcc -print-file-name=libasan.so)
# (tests/test_sanitizers.py does).
set -e
MODE=$1
OUT=$(mkdir -p "$2" && cd "$2" && pwd)
cd "$(dirname "$0")/../readsb-protobuf_amd/csrc"
# asan: gcc (its shared runtime can be preloaded into python).  tsan: the ROCm clang for everything, the C++ launcher
# msd_capi.cpp included (hipcc instruments its host side; only the device code is left out) -- gcc 11's ThreadSanitizer
# runtime cannot start on the GPU boxes' kernel ("unexpected memory mapping") and, with msd_capi.cpp uninstrumented, takes
# the launcher's own std::mutex / std::atomic hand-overs to the C threads for races.
CC=gcc
CAPI_OBJ=msd_capi.o
case "$MODE" in
asan) SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer" ;;
tsan) SAN="-fsanitize=thread -fno-omit-frame-pointer"; CC=/opt/rocm/lib/llvm/bin/clang ;;
*) echo "usage: $0 asan|tsan OUTDIR" >&2; exit 2 ;;
esac
test -f msd_kernels.o -a -f msd_dc_kernels.o -a -f msd_resolve_kernels.o -a -f msd_capi.o || { echo "run build.sh first (the HIP objects are reused)" >&2; exit 1; }
INC="-I. -I../../include -Ihost"
CF="-std=c11 -O1 -g -Wall -Wextra -fPIC $SAN $INC"
$CC $CF -ffp-contract=off -c msd_tables.c -o "$OUT/msd_tables.o"
$CC $CF -ffp-contract=off -c msd_resolve.c -o "$OUT/msd_resolve.o"
$CC $CF -c msd_fields.c -o "$OUT/msd_fields.o"
if [ "$MODE" = tsan ]; then
    hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC $SAN -Wno-option-ignored $INC -c msd_capi.cpp -o "$OUT/msd_capi.o"
    CAPI_OBJ="$OUT/msd_capi.o"
fi
# (the sanitizer runtime comes from LD_PRELOAD or from the instrumented executable: the shared objects leave it undefined)
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libmodes_hip.so" msd_kernels.o msd_dc_kernels.o msd_resolve_kernels.o $CAPI_OBJ \
    "$OUT/msd_tables.o" "$OUT/msd_resolve.o" "$OUT/msd_fields.o" -lm -lpthread
for f in msd_fifo msd_sdr_ifile msd_wire msd_converter msd_demod; do
    $CC $CF -c host/$f.c -o "$OUT/$f.o"
done
$CC -shared -fPIC $SAN -o "$OUT/libmsd_host.so" "$OUT"/msd_fifo.o "$OUT"/msd_sdr_ifile.o "$OUT"/msd_wire.o "$OUT"/msd_converter.o "$OUT"/msd_demod.o \
    -L"$OUT" -lmodes_hip -Wl,-rpath,'$ORIGIN' -lpthread -lm
$CC $CF host/msd_replay_main.c "$OUT"/msd_sdr_ifile.o "$OUT"/msd_fifo.o "$OUT"/msd_wire.o "$OUT"/msd_converter.o -o "$OUT/msd_replay" \
    -L"$OUT" -lmodes_hip -Wl,-rpath,'$ORIGIN' -lpthread -lm
$CC $CF ../../tests/c/fifo_stress.c "$OUT"/msd_fifo.o -o "$OUT/fifo_stress" -lpthread
$CC $CF ../../tests/c/host_units.c "$OUT"/msd_wire.o "$OUT"/msd_tables.o "$OUT"/msd_fields.o "$OUT"/msd_sdr_ifile.o "$OUT"/msd_fifo.o "$OUT"/msd_converter.o \
    -o "$OUT/host_units" -L"$OUT" -lmodes_hip -Wl,-rpath,'$ORIGIN' -lpthread -lm
echo "sanitizer build ($MODE): $(ls "$OUT" | grep -v '\.o$' | tr '\n' ' ')"
