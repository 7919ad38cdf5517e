#!/bin/bash
# round 5, last call: the bench lines of every configuration with the final tree (profiles/r05_bench_configs.txt)
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_r3_bench_all.sh > gpurun_out/r05_bench_summary.txt 2>&1
timeout 500 python bench.py --keypoint-mode reference 2>&1 | tail -1 > gpurun_out/bench_cfg1_refkp.json
timeout 500 python bench.py --config 3 --keypoint-mode reference 2>&1 | tail -1 > gpurun_out/bench_cfg3_refkp.json
cat gpurun_out/bench_cfg1.json gpurun_out/bench_cfg2.json gpurun_out/bench_cfg3.json gpurun_out/bench_cfg4.json gpurun_out/bench_cfg4_f16.json gpurun_out/bench_cfg1_refkp.json gpurun_out/bench_cfg3_refkp.json > gpurun_out/r05_bench_configs.txt
cat gpurun_out/r05_bench_summary.txt | cut -c1-200
