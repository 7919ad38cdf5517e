#!/bin/bash
# round-4 evidence refresh after the fit changes: kernel stats + stream overlap + forward timeline of the bench loop, and the bench line of every BASELINE configuration
# (the per-layer trace and HBM counters of the SCNet forward are unchanged: tools/gpu_r4_evidence.sh)
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --no-cpu-baseline --no-h2d --no-aux > gpurun_out/prof_bench.log 2>&1
grep '"metric"' gpurun_out/prof_bench.log | cut -c1-400 > gpurun_out/r04_bench_under_profiler.txt
python tools/kernel_stats.py gpurun_out/prof_bench/bench_results.db > gpurun_out/r04_bench_kernel_stats.txt 2>&1
python tools/overlap.py gpurun_out/prof_bench/bench_results.db 250 1 > gpurun_out/r04_overlap.txt 2>&1
python tools/chain_timeline.py gpurun_out/prof_bench/bench_results.db 5 > gpurun_out/r04_forward_timeline.txt 2>&1
rm -rf gpurun_out/prof_bench
bash tools/gpu_r3_bench_all.sh > gpurun_out/r04_bench_summary.txt 2>&1
cat gpurun_out/bench_cfg1.json gpurun_out/bench_cfg2.json gpurun_out/bench_cfg3.json gpurun_out/bench_cfg4.json gpurun_out/bench_cfg4_f16.json > gpurun_out/r04_bench_configs.txt
cat gpurun_out/r04_bench_summary.txt
