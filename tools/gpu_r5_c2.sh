#!/bin/bash
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fit.py -m gpu -q --maxfail 8 -p no:cacheprovider -k "affinity" 2>&1 | tail -3
echo "== pool (suncg sigmas)"; RELPOSE_AFF_SEL=pool timeout 300 python tools/affinity_pmc.py 1024 10 200 400 2>&1 | grep -v amdgpu.ids
rm -rf gpurun_out/prof_aff
RELPOSE_AFF_SEL=pool timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_aff -o aff -- python tools/affinity_pmc.py 1024 5 200 > gpurun_out/prof_aff.log 2>&1
python tools/kernel_stats.py gpurun_out/prof_aff/aff_results.db 2>&1 | grep -E "aff3|affinity|fill|kernel  " | cut -c1-160
rm -rf gpurun_out/prof_aff
