#!/bin/bash
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  local name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-aux "$@" > gpurun_out/r5m_$name.json 2> gpurun_out/r5m_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    r = json.loads(open(f"gpurun_out/r5m_{n}.json").read().strip().split("\n")[-1])
    print(n, round(r["value"], 1), "pairs/s", round(r["ms_per_step"], 2), "ms", "ok", r["status_ok_fraction"], flush=True)
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/r5m_{n}.err").read()[-600:], flush=True)
PY
}
run c2_d3 --config 2
run c2_d4_hwq8 --config 2 --inflight 4 --batches 8 --hw-queues 8
run c2_d3_hwq8 --config 2 --hw-queues 8
run c2_d3_prio0 --config 2 --net-priority 0
run c2_d3_fc2 --config 2 --fit-cluster 2
run c4_d3 --config 4
run c4_d4_hwq8 --config 4 --inflight 4 --batches 8 --hw-queues 8
run c4_d3_hwq8 --config 4 --hw-queues 8
run c4_d3_prio0 --config 4 --net-priority 0
run c1_d3_prio0 --net-priority 0
run c1_d3
