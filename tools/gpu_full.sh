#!/bin/bash
# full GPU validation: every -m gpu test, then the default bench line
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
timeout 400 python bench.py 2>&1 | tail -1 > gpurun_out/bench_last.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_last.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'], d.get('pcie_inclusive',{}).get('value'), d['cpu_baseline']['value'] if 'cpu_baseline' in d else None)
print(d['roofline_affinity']['at_batch_1024']['achieved'], d['roofline_affinity']['at_bench_batch']['achieved'])
PY
