#!/bin/bash
# round 5, call H: batches in flight x prepared batches rotated (a third batch in flight needs >= 6 rotating batches: with 4 a batch's buffers are re-entered too soon)
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  local name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-aux "$@" > gpurun_out/r5h_$name.json 2> gpurun_out/r5h_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    r = json.loads(open(f"gpurun_out/r5h_{n}.json").read().strip().split("\n")[-1])
    print(n, round(r["value"], 1), "pairs/s", round(r["ms_per_step"], 2), "ms", "ok", r["status_ok_fraction"], flush=True)
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/r5h_{n}.err").read()[-600:], flush=True)
PY
}
for c in 1 2; do
run c${c}_d2_b4 --config $c
run c${c}_d2_b6 --config $c --batches 6
run c${c}_d3_b6 --config $c --inflight 3 --batches 6
run c${c}_d3_b9 --config $c --inflight 3 --batches 9
run c${c}_d4_b8 --config $c --inflight 4 --batches 8
run c${c}_split_d3_b6 --config $c --inflight 3 --batches 6 --split-forward 1
run c${c}_d3_b6_s40 --config $c --inflight 3 --batches 6 --steps 40
done
run c1_d2_b4_again --config 1
