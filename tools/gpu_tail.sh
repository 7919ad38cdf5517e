#!/bin/bash
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python bench.py --no-aux --no-cpu-baseline"
for e in "RELPOSE_NO_HEAD_OVERLAP=1" "RELPOSE_X=1" "RELPOSE_NO_HEAD_OVERLAP=1" "RELPOSE_X=1"; do
  env $e timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'], d['pcie_inclusive']['value'])"
done
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
