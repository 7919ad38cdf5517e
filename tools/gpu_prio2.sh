#!/bin/bash
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in 1 2; do for e in "RELPOSE_NET_PRIO=0" "RELPOSE_NET_PRIO=-1"; do
  env $e timeout 300 python bench.py --config $c --no-aux --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg$c $e', round(d['value'],1), round(d['pcie_inclusive']['value'],1))"
done; done
