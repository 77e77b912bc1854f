#!/bin/bash
# Builds libmodes_hip.so (HIP kernels + C-ABI + host resolve) and libmsd_siggen.so for gfx950.
set -e
cd "$(dirname "$0")"
INC="-I. -I../../include $MSD_EXTRA_DEFS"
gcc -std=c11 -O2 -g -Wall -Wextra -fPIC -ffp-contract=off $INC -c msd_tables.c -o msd_tables.o
gcc -std=c11 -O2 -g -Wall -Wextra -fPIC -ffp-contract=off $INC -c msd_resolve.c -o msd_resolve.o
gcc -std=c11 -O2 -g -Wall -Wextra -fPIC $INC -c msd_fields.c -o msd_fields.o
# -disable-machine-licm: hoisting the LDS addresses and constants of every loop to the top of a kernel costs the big ones
# (scan, resolve) vector registers for their whole length -- the scan kernel spills at its 128, the resolve kernel at the
# 128 its two-workgroups-per-CU layout allows; without the hoisting both fit, and nothing got slower
LICM="-mllvm -disable-machine-licm"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $LICM $MSD_EXTRA_HIPFLAGS $INC -c msd_kernels.hip -o msd_kernels.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $LICM $INC -c msd_resolve_kernels.hip -o msd_resolve_kernels.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $INC -c msd_dc_kernels.hip -o msd_dc_kernels.o
hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC $INC -c msd_capi.cpp -o msd_capi.o
hipcc --offload-arch=gfx950 -shared -fPIC -o libmodes_hip.so msd_kernels.o msd_dc_kernels.o msd_resolve_kernels.o msd_capi.o msd_tables.o msd_resolve.o msd_fields.o -lm -lpthread
gcc -std=c11 -O2 -g -Wall -Wextra -fPIC $INC -Ihost -c host/msd_fifo.c -o host/msd_fifo.o
gcc -std=c11 -O2 -g -Wall -Wextra -fPIC $INC -Ihost -c host/msd_sdr_ifile.c -o host/msd_sdr_ifile.o
gcc -std=c11 -O2 -g -Wall -Wextra -fPIC $INC -Ihost -c host/msd_wire.c -o host/msd_wire.o
gcc -std=c11 -O2 -g -Wall -Wextra -fPIC $INC -Ihost -c host/msd_converter.c -o host/msd_converter.o
gcc -std=c11 -O2 -g -Wall -Wextra -fPIC $INC -Ihost -c host/msd_demod.c -o host/msd_demod.o
gcc -shared -fPIC -o libmsd_host.so host/msd_fifo.o host/msd_sdr_ifile.o host/msd_wire.o host/msd_converter.o host/msd_demod.o -L. -lmodes_hip -Wl,-rpath,'$ORIGIN' -lpthread -lm
gcc -std=c11 -O2 -g -Wall -Wextra $INC -Ihost host/msd_replay_main.c host/msd_sdr_ifile.o host/msd_fifo.o host/msd_wire.o host/msd_converter.o -o msd_replay \
    -L. -lmodes_hip -Wl,-rpath,'$ORIGIN' -lpthread -lm
gcc -std=c11 -O2 -Wall -Wextra -fPIC -shared -o libmsd_siggen.so msd_siggen.c -lpthread
echo built: $(ls -1 *.so)
