#!/bin/bash
# round 6: lane occupancy of the scan kernel's phases: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) for the
# whole kernel and its ablations (MSD_DEBUG_FLAGS 0: whole, 1: stopped after the tests, 2: after the conversion, 4: no step B),
# per 64 Mi-sample launch, with the instruction counts (set E) of the same runs.  scripts/r4_pmc.sh does the passes.
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/r4_pmc/r06b_flags*.txt
for fl in 0 1 2 4; do MSD_DEBUG_FLAGS=$fl SETS="C E" bash scripts/r4_pmc.sh r06b_flags$fl "$@"; done
cat gpurun_out/r4_pmc/r06b_flags*.txt > gpurun_out/r06b_lane_counters.txt
KFILTER=msd_resolve SETS="C E F" bash scripts/r4_pmc.sh r06b_resolve "$@"
cat gpurun_out/r4_pmc/r06b_resolve.txt >> gpurun_out/r06b_lane_counters.txt
cat gpurun_out/r06b_lane_counters.txt
