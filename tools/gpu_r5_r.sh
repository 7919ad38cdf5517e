#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_r3_hbm16.sh f16x3 > /dev/null 2>&1; cp gpurun_out/hbm_pmc_f16x3.txt gpurun_out/r05_scnet_hbm_pmc_f16x3.txt
bash tools/gpu_r3_hbm16.sh f16 > /dev/null 2>&1; cp gpurun_out/hbm_pmc_f16.txt gpurun_out/r05_scnet_hbm_pmc_f16.txt
tail -2 gpurun_out/r05_scnet_hbm_pmc_f16x3.txt; tail -2 gpurun_out/r05_scnet_hbm_pmc_f16.txt
