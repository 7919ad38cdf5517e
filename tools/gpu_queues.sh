#!/bin/bash
# HW-queue experiment: does the stream -> hardware-queue mapping (GPU_MAX_HW_QUEUES, default 4) serialise the batch streams with the SCNet stream?
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="python bench.py --no-h2d --no-aux --no-cpu-baseline"
for q in "" 2 8 16; do
  echo "== GPU_MAX_HW_QUEUES=[$q]"
  if [ -z "$q" ]; then timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  else GPU_MAX_HW_QUEUES=$q timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; fi
done
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/ovl -o p -- $B > gpurun_out/ovl.log 2>&1
python tools/overlap.py gpurun_out/ovl/p_results.db 800 tl 2>&1 | cut -c1-1500
