#!/bin/bash
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in 2 3 4; do
  timeout 500 python bench.py --config $c --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_cfg$c.json
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_cfg$c.json'))
print($c, round(d['value'],1), round(d['ms_per_step'],2), round(d['roofline']['frac'],3), round(d['roofline']['achieved'],1), round(d['pcie_inclusive']['value'],1))
PY
done
