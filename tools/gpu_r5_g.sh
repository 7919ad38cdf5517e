#!/bin/bash
# round 5, call G: configs[2] (N=400: the slot-stream chain costs 4 ms of a step there) under the loop switches
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  local name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-aux "$@" > gpurun_out/r5g_$name.json 2> gpurun_out/r5g_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    r = json.loads(open(f"gpurun_out/r5g_{n}.json").read().strip().split("\n")[-1])
    print(n, round(r["value"], 1), "pairs/s", round(r["ms_per_step"], 2), "ms", "ok", r["status_ok_fraction"], flush=True)
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/r5g_{n}.err").read()[-600:], flush=True)
PY
}
run c2_base --config 2
run c2_d3 --config 2 --inflight 3
run c2_d3_b6 --config 2 --inflight 3 --batches 6
run c2_prio0 --config 2 --net-priority 0
run c2_fc2 --config 2 --fit-cluster 2
run c2_fc4 --config 2 --fit-cluster 4
run c2_split_d3 --config 2 --split-forward 1 --inflight 3
run c2_d3_fc4 --config 2 --inflight 3 --fit-cluster 4
run c4_base --config 4
run c4_d3 --config 4 --inflight 3
run c4_prio0 --config 4 --net-priority 0
run c2_base2 --config 2
