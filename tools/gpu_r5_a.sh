#!/bin/bash
# round 5 (re-entry), call A: the full -m gpu suite (parity log), the headline bench line as the driver runs it, the bench loop under the profiler
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --maxfail 20 -p no:cacheprovider > gpurun_out/r5a_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5a_tests.log
tail -8 gpurun_out/r5a_tests.log
cat gpurun_out/parity/*.jsonl > gpurun_out/r05_parity_full_suite.jsonl
timeout 500 python bench.py > gpurun_out/r5a_bench_cfg1.json 2> gpurun_out/r5a_bench_cfg1.err; echo "bench rc=$?"
cut -c1-1500 gpurun_out/r5a_bench_cfg1.json
rm -rf gpurun_out/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --no-cpu-baseline --no-h2d --no-aux > gpurun_out/prof_bench.log 2>&1
grep '"metric"' gpurun_out/prof_bench.log | cut -c1-400 > gpurun_out/r05_bench_under_profiler.txt
python tools/kernel_stats.py gpurun_out/prof_bench/bench_results.db > gpurun_out/r05_bench_kernel_stats.txt 2>&1
python tools/overlap.py gpurun_out/prof_bench/bench_results.db 250 1 > gpurun_out/r05_overlap.txt 2>&1
python tools/chain_timeline.py gpurun_out/prof_bench/bench_results.db 5 > gpurun_out/r05_forward_timeline.txt 2>&1
rm -rf gpurun_out/prof_bench
head -30 gpurun_out/r05_bench_kernel_stats.txt
