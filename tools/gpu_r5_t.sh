#!/bin/bash
# round 5, call T: fit_pair_kernel<512> capped at 168 VGPRs (3 waves per SIMD; 57 spilled registers) so that a conv wave fits beside two of its waves -- alone and in the loop
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
V=$GRAFT_REPO_ROOT/relativepose_amd/librelpose_hip_w3.so
rm -f /tmp/pose1.npy
echo "== default"; RELPOSE_POSE_DUMP=/tmp/pose1.npy timeout 300 python tools/matcher_time.py 1 32 1 2>&1 | grep -v amdgpu.ids | cut -c1-120
echo "== 168 VGPRs"; RELPOSE_POSE_DUMP=/tmp/pose1.npy RELPOSE_LIB_PATH=$V timeout 300 python tools/matcher_time.py 1 32 1 2>&1 | grep -v amdgpu.ids | cut -c1-120
run() {
  local name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-h2d --no-aux "$@" > gpurun_out/r5t_$name.json 2> gpurun_out/r5t_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    r = json.loads(open(f"gpurun_out/r5t_{n}.json").read().strip().split("\n")[-1])
    print(n, round(r["value"], 1), "pairs/s", round(r["ms_per_step"], 2), "ms", "ok", r["status_ok_fraction"], flush=True)
except Exception as e:
    print(n, "FAILED", e, open(f"gpurun_out/r5t_{n}.err").read()[-600:], flush=True)
PY
}
run base1
RELPOSE_LIB_PATH=$V run w3_1
run base2
RELPOSE_LIB_PATH=$V run w3_2
run c3_base --config 3
RELPOSE_LIB_PATH=$V run c3_w3 --config 3
