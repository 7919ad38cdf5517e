#!/bin/bash
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bench.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for c in 1 2 3 4; do
  timeout 500 python bench.py --config $c --no-cpu-baseline --no-aux 2>gpurun_out/r5j_$c.err | tail -1 > gpurun_out/r5j_$c.json
  python - $c <<'PY'
import json, sys
c = sys.argv[1]
try:
    r = json.loads(open(f"gpurun_out/r5j_{c}.json").read())
    print("config", c, round(r["value"], 1), "pairs/s", round(r["ms_per_step"], 2), "ms  pcie", round(r["pcie_inclusive"]["value"], 1), r["pcie_inclusive"]["mode"], "in_flight", r["config"]["batches_in_flight"], r["config"]["prepared_batches_rotated"], flush=True)
except Exception as e:
    print(c, "FAILED", e, open(f"gpurun_out/r5j_{c}.err").read()[-800:], flush=True)
PY
done
