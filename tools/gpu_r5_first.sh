#!/bin/bash
# round 5, first GPU call: the full -m gpu suite with the round's new tests (16-bit free-running fixtures, config4 full size, op plans,
# cache_primitives, batched keypoints), then the headline bench line and the per-level-keypoints variant.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q --maxfail 20 -p no:cacheprovider > gpurun_out/r5_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_tests.log
tail -30 gpurun_out/r5_tests.log
timeout 500 python bench.py > gpurun_out/r5_bench_cfg1.json 2> gpurun_out/r5_bench_cfg1.err; echo "bench rc=$?"
cut -c1-600 gpurun_out/r5_bench_cfg1.json
timeout 500 python bench.py --keypoint-mode reference > gpurun_out/r5_bench_cfg1_refkp.json 2> gpurun_out/r5_bench_cfg1_refkp.err; echo "bench refkp rc=$?"
cut -c1-600 gpurun_out/r5_bench_cfg1_refkp.json; tail -5 gpurun_out/r5_bench_cfg1_refkp.err
