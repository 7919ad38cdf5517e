#!/bin/bash
# round 4 affinity: parity tests, then HIP-event timing of the tile kernel (default = v2) and the first-generation kernel (tile_v1)
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ "$1" != "notest" ]; then
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_matcher.py -q -m gpu -x -k "affinity or stages" 2>&1 | tail -15
fi
timeout 300 python tools/affinity_pmc.py 1024 10 200 400 2>&1 | tail -4
echo "-- tile_v1"
RELPOSE_AFF_SEL=tile_v1 timeout 300 python tools/affinity_pmc.py 1024 10 200 400 2>&1 | tail -4
