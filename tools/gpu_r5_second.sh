#!/bin/bash
# round 5, second GPU call: the two tests the first call failed (provider bug, f16 bound), then the PCIe-inclusive A/B (uploads on the slot
# stream vs one in-flight depth ahead on a copy stream) at configs[4] and configs[1]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_e2e.py tests/test_gpu_pipeline.py -m gpu -q --maxfail 10 -p no:cacheprovider > gpurun_out/r5_tests2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_tests2.log
tail -12 gpurun_out/r5_tests2.log
for mode in slot lookahead; do
  timeout 400 python bench.py --config 4 --h2d-mode $mode --no-cpu-baseline --no-aux > gpurun_out/r5_cfg4_h2d_$mode.json 2> gpurun_out/r5_cfg4_h2d_$mode.err; echo "cfg4 $mode rc=$?"
  python - <<PY
import json
r=json.loads(open("gpurun_out/r5_cfg4_h2d_$mode.json").read().strip().split("\n")[-1]); print("cfg4 $mode", round(r["value"],1), r["pcie_inclusive"]["value"], r["pcie_inclusive"]["mode"])
PY
done
for mode in slot lookahead; do
  timeout 400 python bench.py --h2d-mode $mode --no-cpu-baseline --no-aux > gpurun_out/r5_cfg1_h2d_$mode.json 2> gpurun_out/r5_cfg1_h2d_$mode.err; echo "cfg1 $mode rc=$?"
  python - <<PY
import json
r=json.loads(open("gpurun_out/r5_cfg1_h2d_$mode.json").read().strip().split("\n")[-1]); print("cfg1 $mode", round(r["value"],1), r["pcie_inclusive"]["value"], r["pcie_inclusive"]["mode"])
PY
done
