#!/bin/bash
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python bench.py --no-aux --no-cpu-baseline"
for i in 1 2; do timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('pcie_inclusive',{}).get('value'))"; done
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/ovl -o p -- $B --no-h2d > gpurun_out/ovl.log 2>&1
python tools/overlap.py gpurun_out/ovl/p_results.db 800 tl 2>&1 | cut -c1-1200
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_bench.py -x -q 2>&1 | tail -3
