#!/bin/bash
# Builds libmodes_hip.so (HIP kernels + C-ABI + host resolve) and libmsd_siggen.so for gfx950.
set -e
cd "$(dirname "$0")"
INC="-I. -I../../include"
gcc -std=c11 -O2 -g -Wall -Wextra -fPIC -ffp-contract=off $INC -c msd_tables.c -o msd_tables.o
gcc -std=c11 -O2 -g -Wall -Wextra -fPIC -ffp-contract=off $INC -c msd_resolve.c -o msd_resolve.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $INC -c msd_kernels.hip -o msd_kernels.o
hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC $INC -c msd_capi.cpp -o msd_capi.o
hipcc --offload-arch=gfx950 -shared -fPIC -o libmodes_hip.so msd_kernels.o msd_capi.o msd_tables.o msd_resolve.o -lm
gcc -std=c11 -O2 -Wall -Wextra -fPIC -shared -o libmsd_siggen.so msd_siggen.c -lpthread
echo built: $(ls -1 *.so)
