#!/bin/bash
# round 4: self-stream cache -- parity tests, then same-box A/B of the bench (cache on / off)
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_scnet.py tests/test_gpu_pipeline.py -q -m gpu -x -k "self_stream or zero_warp or pose_outputs or in_flight or rotating or interleaved" 2>&1 | tail -8
timeout 400 python bench.py --no-cpu-baseline --no-h2d > gpurun_out/r4_bench_cache.json 2> gpurun_out/r4_bench_cache.err
timeout 400 python bench.py --no-cpu-baseline --no-h2d --no-aux --no-self-cache > gpurun_out/r4_bench_nocache.json 2> gpurun_out/r4_bench_nocache.err
python - <<'PY'
import json
for f in ("cache", "nocache"):
    try:
        d = json.loads(open(f"gpurun_out/r4_bench_{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "failed", e); print(open(f"gpurun_out/r4_bench_{f}.err").read()[-2000:])
PY
