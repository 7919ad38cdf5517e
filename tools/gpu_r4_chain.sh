#!/bin/bash
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_bench -o bench -- python bench.py --no-cpu-baseline --no-h2d --no-aux "$@" > gpurun_out/prof_bench.log 2>&1
python tools/chain_timeline.py gpurun_out/prof_bench/bench_results.db 6
rm -rf gpurun_out/prof_bench
