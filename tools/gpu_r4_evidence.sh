#!/bin/bash
# round-4 evidence for profiles/: fp32 per-layer trace of one full forward, its HBM counters (FETCH_SIZE / WRITE_SIZE in separate passes),
# kernel stats + stream overlap of the bench loop, and the bench line of every BASELINE configuration.
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_fwd gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/prof_bench
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_fwd -o p -- python tools/scnet_only.py 64 3 > gpurun_out/prof_fwd.log 2>&1
python tools/kernel_stats.py gpurun_out/prof_fwd/p_results.db 64 > gpurun_out/r04_scnet_forward_layers.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o p -- python tools/scnet_only.py 64 2 > gpurun_out/pmc_$c.log 2>&1
done
python tools/pmc_summary.py gpurun_out > gpurun_out/r04_scnet_hbm_pmc.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --no-cpu-baseline --no-h2d --no-aux > gpurun_out/prof_bench.log 2>&1
grep '"metric"' gpurun_out/prof_bench.log | cut -c1-400 > gpurun_out/r04_bench_under_profiler.txt
python tools/kernel_stats.py gpurun_out/prof_bench/bench_results.db > gpurun_out/r04_bench_kernel_stats.txt 2>&1
python tools/overlap.py gpurun_out/prof_bench/bench_results.db 250 1 > gpurun_out/r04_overlap.txt 2>&1
python tools/chain_timeline.py gpurun_out/prof_bench/bench_results.db 5 > gpurun_out/r04_forward_timeline.txt 2>&1
rm -rf gpurun_out/prof_fwd gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/prof_bench
bash tools/gpu_r3_bench_all.sh > gpurun_out/r04_bench_summary.txt 2>&1
cat gpurun_out/bench_cfg1.json gpurun_out/bench_cfg2.json gpurun_out/bench_cfg3.json gpurun_out/bench_cfg4.json gpurun_out/bench_cfg4_f16.json > gpurun_out/r04_bench_configs.txt
tail -3 gpurun_out/r04_scnet_hbm_pmc.txt; tail -4 gpurun_out/r04_scnet_forward_layers.txt; cat gpurun_out/r04_bench_summary.txt
