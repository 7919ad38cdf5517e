#!/bin/bash
# usage (GPU box): bash tools/gpu_r4_matcher.sh TAG -> fit / matcher / pipeline parity tests, then the matcher alone (HIP events) at configs[1] / configs[2]
# sizes with every fit variant (tools/matcher_time.py) and its kernel breakdown
TAG=${1:-m4}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_matcher.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/${TAG}_tests.txt
{ timeout 300 python tools/matcher_time.py 1; timeout 300 python tools/matcher_time.py 2; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_matcher_time.txt
for c in 1 2; do
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/${TAG}_mt$c -o p -- python tools/matcher_time.py $c > gpurun_out/${TAG}_mt$c.log 2>&1
python tools/kernel_stats.py gpurun_out/${TAG}_mt$c/p_results.db 2>&1 | grep -E "pair_|affinity|fit_pair|kernel  " >> gpurun_out/${TAG}_matcher_time.txt
rm -rf gpurun_out/${TAG}_mt$c
done
cat gpurun_out/${TAG}_matcher_time.txt
