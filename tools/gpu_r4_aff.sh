#!/bin/bash
# round 4 affinity: parity tests, then HIP-event timing of the tile kernel at B=1024 for the sparse (suncg sigmas) and the dense window (scannet sigmas)
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ "$1" != "notest" ]; then
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_matcher.py -q -m gpu -x -k "affinity or stages or stage_goldens" 2>&1 | tail -5
fi
timeout 300 python tools/affinity_pmc.py 1024 10 200 400 2>&1 | tail -4
echo "-- scannet sigmas (dense window)"
RELPOSE_AFF_PARAMS=scannet timeout 300 python tools/affinity_pmc.py 1024 10 200 2>&1 | tail -2
