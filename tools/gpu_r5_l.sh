#!/bin/bash
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  local name=$1; shift
  timeout 500 python bench.py --no-cpu-baseline --no-aux "$@" 2>gpurun_out/r5l_$name.err | tail -1 > gpurun_out/r5l_$name.json
  python - $name <<'PY'
import json, sys
c = sys.argv[1]
try:
    r = json.loads(open(f"gpurun_out/r5l_{c}.json").read())
    print(c, round(r["value"], 1), "pairs/s", round(r["ms_per_step"], 2), "ms  pcie", round(r["pcie_inclusive"]["value"], 1), r["pcie_inclusive"]["mode"], flush=True)
except Exception as e:
    print(c, "FAILED", e, open(f"gpurun_out/r5l_{c}.err").read()[-800:], flush=True)
PY
}
run c4_default --config 4
run c1_default
run c2_default --config 2
run c3_default --config 3
timeout 900 python -m pytest tests/test_gpu_bench.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
