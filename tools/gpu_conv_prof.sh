#!/bin/bash
# usage (on the GPU box): bash tools/gpu_conv_prof.sh TAG  -> SCNet parity tests + per-layer conv timing from a rocprofv3 kernel trace
TAG=${1:-x}
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_scnet.py -x -q 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof$TAG -o p -- python tools/scnet_only.py 64 3 > gpurun_out/prof$TAG.log 2>&1
python tools/kernel_stats.py gpurun_out/prof$TAG/p_results.db 64 2>&1 | tail -70
