#!/bin/bash
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_matcher.py tests/test_gpu_fit.py -x -q 2>&1 | tail -4
timeout 300 python tools/matcher_time.py 1 2>&1 | tail -6
timeout 300 python tools/matcher_time.py 2 2>&1 | tail -6
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/mt -o p -- python tools/matcher_time.py 1 > gpurun_out/mt.log 2>&1
python tools/kernel_stats.py gpurun_out/mt/p_results.db 64 2>&1 | grep -E "pair_|affinity|fit_pair" | head
