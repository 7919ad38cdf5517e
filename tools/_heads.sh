#!/bin/bash
export HSA_DISABLE_COREDUMP_ON_EXCEPTION=1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_scnet.py -q -x -k "vs_oracle and f32 or pose_outputs or batched" 2>&1 | tail -3
rocprofv3 --kernel-trace -d gpurun_out/prof_h -o p -- python tools/scnet_only.py 64 3 f32 > gpurun_out/prof_h.log 2>&1
python tools/kernel_stats.py gpurun_out/prof_h/p_results.db 64 2>&1 | grep -E "heads_kernel|resize_out|forward wall"
rm -rf gpurun_out/prof_h
for i in 1 2; do echo "bench: $(timeout 250 python bench.py --no-cpu-baseline --no-h2d --no-aux 2>&1 | tail -1 | cut -c90-150)"; done
