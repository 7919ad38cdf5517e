#!/bin/bash
# usage (GPU box): bash tools/gpu_r3_matcher.sh TAG -> matcher alone (HIP events) at configs[1] / configs[2] sizes + its kernel breakdown
TAG=${1:-m}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
{ timeout 300 python tools/matcher_time.py 1; timeout 300 python tools/matcher_time.py 2; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_matcher_time.txt
for c in 1 2; do
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_mt$c -o p -- python tools/matcher_time.py $c > gpurun_out/${TAG}_mt$c.log 2>&1
python tools/kernel_stats.py gpurun_out/${TAG}_mt$c/p_results.db 2>&1 | grep -E "pair_|affinity|fit_pair|kernel  " >> gpurun_out/${TAG}_matcher_time.txt
rm -rf gpurun_out/${TAG}_mt$c
done
cat gpurun_out/${TAG}_matcher_time.txt
