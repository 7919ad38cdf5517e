#!/bin/bash
# end-of-round validation: smoke(), every -m gpu test, the default bench line (as the driver runs them)
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4
timeout 500 python bench.py 2>&1 | tail -1 > gpurun_out/bench_last.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_last.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'], d['pcie_inclusive']['value'], d['cpu_baseline']['value'])
print(d['roofline_affinity']['at_batch_1024']['achieved'], d['roofline_affinity']['at_bench_batch']['achieved'])
PY
