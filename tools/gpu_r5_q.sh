#!/bin/bash
# round 5, call Q: SQ / HBM counters of the affinity build at B = 1024, N = 200: the shipped tile kernel and the decoupled pool variant side by side
cd $GRAFT_REPO_ROOT
NS=200 RELPOSE_AFF_SEL=tile bash tools/gpu_r4_affinity_pmc.sh r5tile 1024 > /dev/null 2>&1
NS=200 RELPOSE_AFF_SEL=pool bash tools/gpu_r4_affinity_pmc.sh r5pool 1024 > /dev/null 2>&1
{ echo "######################## tile kernel (shipped; RELPOSE_AFF_SEL=tile)"; cat gpurun_out/r5tile_affinity_pmc.txt; echo; echo "######################## pool variant (RELPOSE_AFF_SEL=pool)"; cat gpurun_out/r5pool_affinity_pmc.txt; } > gpurun_out/r05_affinity_pmc.txt
cut -c1-260 gpurun_out/r05_affinity_pmc.txt
