#!/bin/bash
# round 5 (last session): SQ counters of the scan kernel and its ablations (MSD_DEBUG_FLAGS 0: whole, 1: stopped after the
# tests, 2: after the conversion, 4: no step B) per 64 Mi-sample launch, for profiles/r05_binding.json.  scripts/r4_pmc.sh does the passes.
cd $GRAFT_REPO_ROOT
for fl in 0 1 2 4; do MSD_DEBUG_FLAGS=$fl SETS="E F" bash scripts/r4_pmc.sh r05b_flags$fl; done
cat gpurun_out/r4_pmc/r05b_flags*.txt > gpurun_out/r05b_pmc_counters.txt
cat gpurun_out/r05b_pmc_counters.txt
