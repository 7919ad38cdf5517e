#!/bin/bash
# round 5, call B: the split forward (two-part enqueue, chain on a third stream) -- bitwise tests, then the A/B matrix at configs[1]:
# batches in flight x split x CU-masked slot streams x hardware queues
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_scnet.py tests/test_gpu_pipeline.py -m gpu -q --maxfail 5 -p no:cacheprovider -k "two_parts or split_forward or self_stream_cache or batches_in_flight" > gpurun_out/r5b_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5b_tests.log
tail -15 gpurun_out/r5b_tests.log
run() {   # name, args...
  local name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-aux "$@" > gpurun_out/r5b_$name.json 2> gpurun_out/r5b_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    r = json.loads(open(f"gpurun_out/r5b_{n}.json").read().strip().split("\n")[-1])
    print(n, round(r["value"], 1), "pairs/s", round(r["ms_per_step"], 2), "ms", "ok", r["status_ok_fraction"], flush=True)
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/r5b_{n}.err").read()[-600:], flush=True)
PY
}
run base
run hwq8 --hw-queues 8
run d3 --inflight 3
run d3_hwq8 --inflight 3 --hw-queues 8
run split_d2 --split-forward 1
run split_d3 --split-forward 1 --inflight 3
run split_d3_hwq8 --split-forward 1 --inflight 3 --hw-queues 8
run split_d3_hwq8_mp0 --split-forward 1 --inflight 3 --hw-queues 8 --mid-priority 0
run cu64 --slot-cus 64
run cu128 --slot-cus 128
run cu32 --slot-cus 32
run split_d3_hwq8_cu64 --split-forward 1 --inflight 3 --hw-queues 8 --slot-cus 64
run split_d3_hwq8_cu128 --split-forward 1 --inflight 3 --hw-queues 8 --slot-cus 128
