#!/bin/bash
# usage (GPU box): bash tools/gpu_r3_hbm16.sh [precision] -> rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate counter-only passes) of one SCNet forward at 64 images
P=${1:-f16x3}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o p -- python tools/scnet_only.py 64 2 $P > gpurun_out/pmc_$c.log 2>&1
done
python tools/pmc_summary.py gpurun_out > gpurun_out/hbm_pmc_$P.txt 2>&1
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
cat gpurun_out/hbm_pmc_$P.txt | cut -c1-120
