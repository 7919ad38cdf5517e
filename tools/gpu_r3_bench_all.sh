#!/bin/bash
# usage (GPU box): bash tools/gpu_r3_bench_all.sh -> the default bench line (as the driver runs it) + configs 2..4 (+ plain f16), one summary line each
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 500 python bench.py 2>&1 | tail -1 > gpurun_out/bench_cfg1.json
for c in 2 3 4; do timeout 500 python bench.py --config $c --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_cfg$c.json; done
timeout 500 python bench.py --config 4 --precision f16 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_cfg4_f16.json
python - <<'PY'
import json
for c in ("1", "2", "3", "4", "4_f16"):
    try:
        d = json.load(open(f"gpurun_out/bench_cfg{c}.json"))
    except Exception as e:
        print(c, "failed", e); continue
    r = d["roofline"]; g = d.get("roofline_geometry"); a = d.get("roofline_affinity", {})
    print(c, "pairs/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "conv frac", round(r["frac"], 3), "achieved", round(r["achieved"], 1), r["unit"],
          "pcie", round(d["pcie_inclusive"]["value"], 1) if d.get("pcie_inclusive") else None,
          "cpu", d.get("cpu_baseline", {}).get("value"),
          "affinity", {k: round(v["achieved"], 1) for k, v in a.items() if isinstance(v, dict) and "achieved" in v},
          "geometry", {k: round(v["achieved"], 1) for k, v in (g or {}).items() if isinstance(v, dict) and "achieved" in v})
PY
