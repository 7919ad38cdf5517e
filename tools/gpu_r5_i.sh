#!/bin/bash
# round 5, call I: three batches in flight after the state-reuse fix (wait for the previous batch's completion event, not its slot stream), every configuration
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -p no:cacheprovider -k "in_flight or rotating or split_forward or self_stream" 2>&1 | tail -3
run() {
  local name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-aux "$@" > gpurun_out/r5i_$name.json 2> gpurun_out/r5i_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    r = json.loads(open(f"gpurun_out/r5i_{n}.json").read().strip().split("\n")[-1])
    print(n, round(r["value"], 1), "pairs/s", round(r["ms_per_step"], 2), "ms", "ok", r["status_ok_fraction"], flush=True)
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/r5i_{n}.err").read()[-600:], flush=True)
PY
}
for c in 1 2 3 4; do
run c${c}_d2_b4 --config $c
run c${c}_d3_b4 --config $c --inflight 3
run c${c}_d3_b6 --config $c --inflight 3 --batches 6
done
run c1_kp_d2 --keypoint-mode reference
run c1_kp_d3 --keypoint-mode reference --inflight 3 --batches 6
run c3_kp_d2 --config 3 --keypoint-mode reference
run c3_kp_d3 --config 3 --keypoint-mode reference --inflight 3 --batches 6
run c4f16_d2 --config 4 --precision f16
run c4f16_d3 --config 4 --precision f16 --inflight 3 --batches 6
