#!/bin/bash
# kernel stats + stream overlap of the bench loop only (no aux measurements): bash tools/gpu_r4_prof_bench.sh [bench args]
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --no-cpu-baseline --no-h2d --no-aux "$@" > gpurun_out/prof_bench.log 2>&1
grep '"metric"' gpurun_out/prof_bench.log | cut -c1-300
python tools/kernel_stats.py gpurun_out/prof_bench/bench_results.db > gpurun_out/r4_bench_kernel_stats.txt 2>&1
python tools/overlap.py gpurun_out/prof_bench/bench_results.db 250 1 > gpurun_out/r4_overlap.txt 2>&1
rm -rf gpurun_out/prof_bench
