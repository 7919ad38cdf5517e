#!/bin/bash
# only the bench trace of tools/gpu_profiles.sh (kernel stats + stream overlap)
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
grep '"metric"' gpurun_out/prof_bench.log > gpurun_out/prof_bench.json
python tools/kernel_stats.py gpurun_out/prof_bench/bench_results.db > gpurun_out/bench_kernel_stats.txt 2>&1
python tools/overlap.py gpurun_out/prof_bench/bench_results.db 250 1 > gpurun_out/overlap.txt 2>&1
cut -c1-200 gpurun_out/prof_bench.json
