#!/bin/bash
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python bench.py --no-aux --no-cpu-baseline --no-h2d"
for i in 1 2; do timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
RELPOSE_LEGACY_PAIRS=1 timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('legacy pairs', d['value'], d['ms_per_step'])"
timeout 300 $B --config 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config2', d['value'], d['ms_per_step'])"
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
