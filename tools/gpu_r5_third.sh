#!/bin/bash
# round 5, third GPU call: heads_kernel at <= 168 VGPRs (accumulator sets one after the other, launch bounds 3 waves / SIMD): SCNet + ops + keypoint tests, then the headline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_scnet.py tests/test_gpu_ops.py tests/test_gpu_keypoints.py tests/test_gpu_tune.py -m gpu -q --maxfail 10 -p no:cacheprovider > gpurun_out/r5_tests3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_tests3.log
tail -8 gpurun_out/r5_tests3.log
for i in 1 2; do
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r5_cfg1_heads_$i.json 2> gpurun_out/r5_cfg1_heads_$i.err; echo "cfg1 rc=$?"
python - <<PY
import json
r=json.loads(open("gpurun_out/r5_cfg1_heads_$i.json").read().strip().split("\n")[-1]); print("cfg1", round(r["value"],1), r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["in_loop"]["frac"], r["roofline"]["ms_per_forward_gemm"], r["pcie_inclusive"]["value"])
PY
done
timeout 400 python bench.py --config 2 --no-cpu-baseline --no-aux > gpurun_out/r5_cfg2_heads.json 2> gpurun_out/r5_cfg2_heads.err
python - <<PY
import json
r=json.loads(open("gpurun_out/r5_cfg2_heads.json").read().strip().split("\n")[-1]); print("cfg2", round(r["value"],1), r["ms_per_step"], r["pcie_inclusive"]["value"])
PY
