#!/bin/bash
# Regenerates the measurements behind profiles/ (run on the GPU box via gpurun; summaries land in gpurun_out/):
#   1. rocprofv3 --kernel-trace --stats of the default bench command        -> gpurun_out/prof_bench (+ .txt summary)
#   2. the same trace's stream-overlap analysis                              -> gpurun_out/overlap.txt
#   3. per-layer conv timing of one forward at the bench batch                -> gpurun_out/conv_layers.txt
#   4. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, counters only) of tools/scnet_only.py -> gpurun_out/hbm_pmc.txt
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench gpurun_out/prof_fwd gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
grep '"metric"' gpurun_out/prof_bench.log > gpurun_out/prof_bench.json
python tools/kernel_stats.py gpurun_out/prof_bench/bench_results.db > gpurun_out/bench_kernel_stats.txt 2>&1
python tools/overlap.py gpurun_out/prof_bench/bench_results.db 250 1 > gpurun_out/overlap.txt 2>&1
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_fwd -o p -- python tools/scnet_only.py 64 3 > gpurun_out/prof_fwd.log 2>&1
python tools/kernel_stats.py gpurun_out/prof_fwd/p_results.db 64 > gpurun_out/conv_layers.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o p -- python tools/scnet_only.py 64 2 > gpurun_out/pmc_$c.log 2>&1
done
python tools/pmc_summary.py gpurun_out > gpurun_out/hbm_pmc.txt 2>&1
tail -3 gpurun_out/hbm_pmc.txt; tail -4 gpurun_out/conv_layers.txt; cat gpurun_out/overlap.txt | head -8; cut -c1-300 gpurun_out/prof_bench.json
ls gpurun_out/prof_bench | head
# 5. the matcher alone (HIP events) + its kernel breakdown
timeout 300 python tools/matcher_time.py 1 > gpurun_out/matcher_time.txt 2>&1
timeout 300 python tools/matcher_time.py 2 >> gpurun_out/matcher_time.txt 2>&1
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/mt -o p -- python tools/matcher_time.py 1 > gpurun_out/mt.log 2>&1
python tools/kernel_stats.py gpurun_out/mt/p_results.db 2>&1 | grep -E "pair_|affinity|fit_pair|kernel  " > gpurun_out/matcher_kernels.txt
tail -12 gpurun_out/matcher_time.txt; cat gpurun_out/matcher_kernels.txt
