#!/bin/bash
# round 5, call D: workgroup caps of the heads / conv1 launches in the loop (persistent walks over the tiles), configs[1]; + the pool auto-selection test
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fit.py -m gpu -q -p no:cacheprovider -k "pool_affinity" 2>&1 | tail -3
run() {
  local name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-aux "$@" > gpurun_out/r5d_$name.json 2> gpurun_out/r5d_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    r = json.loads(open(f"gpurun_out/r5d_{n}.json").read().strip().split("\n")[-1])
    print(n, round(r["value"], 1), "pairs/s", round(r["ms_per_step"], 2), "ms", "ok", r["status_ok_fraction"], flush=True)
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/r5d_{n}.err").read()[-600:], flush=True)
PY
}
run base
run h1024 --heads-grid 1024
run h512 --heads-grid 512
run h256 --heads-grid 256
run h128 --heads-grid 128
run h2048 --heads-grid 2048
run c768 --conv1-grid 768
run c256 --conv1-grid 256
run h512_c768 --heads-grid 512 --conv1-grid 768
run base2
