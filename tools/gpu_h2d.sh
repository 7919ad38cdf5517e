#!/bin/bash
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python bench.py --no-aux --no-cpu-baseline"
for e in "RELPOSE_X=0" "RELPOSE_BENCH_COPY_STREAM=1"; do
  env $e timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['pcie_inclusive']['value'])"
done
