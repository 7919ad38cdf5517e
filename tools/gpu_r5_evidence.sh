#!/bin/bash
# round-5 evidence for profiles/ (final code), part 1: fp32 per-layer trace of one full forward + its HBM counters (FETCH_SIZE / WRITE_SIZE in separate
# passes), the step's three forwards alone (per-kernel times of the cached plans), kernel stats + stream overlap + forward timeline of the bench loop,
# the f16x3 / f16 layer traces, and the bench line of every BASELINE configuration (+ the per-level-keypoints variant of configs[1] and [3]).
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=r05
rm -rf gpurun_out/prof_fwd gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/prof_bench gpurun_out/prof_alone
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_fwd -o p -- python tools/scnet_only.py 64 3 > gpurun_out/prof_fwd.log 2>&1
python tools/kernel_stats.py gpurun_out/prof_fwd/p_results.db 64 > gpurun_out/${R}_scnet_forward_layers.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o p -- python tools/scnet_only.py 64 2 > gpurun_out/pmc_$c.log 2>&1
done
python tools/pmc_summary.py gpurun_out > gpurun_out/${R}_scnet_hbm_pmc.txt 2>&1
{ timeout 300 python tools/loop_forwards.py 64 10 2>&1 | grep -v amdgpu.ids; } > gpurun_out/${R}_step_forwards_alone.txt
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_alone -o p -- python tools/loop_forwards.py 64 4 > gpurun_out/prof_alone.log 2>&1
python tools/kernel_stats.py gpurun_out/prof_alone/p_results.db >> gpurun_out/${R}_step_forwards_alone.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --no-cpu-baseline --no-h2d --no-aux > gpurun_out/prof_bench.log 2>&1
grep '"metric"' gpurun_out/prof_bench.log | cut -c1-400 > gpurun_out/${R}_bench_under_profiler.txt
python tools/kernel_stats.py gpurun_out/prof_bench/bench_results.db > gpurun_out/${R}_bench_kernel_stats.txt 2>&1
python tools/overlap.py gpurun_out/prof_bench/bench_results.db 250 1 > gpurun_out/${R}_overlap.txt 2>&1
python tools/chain_timeline.py gpurun_out/prof_bench/bench_results.db 5 > gpurun_out/${R}_forward_timeline.txt 2>&1
for P in f16x3 f16; do
  rm -rf gpurun_out/prof_$P
  timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_$P -o p -- python tools/scnet_only.py 64 3 $P > gpurun_out/prof_$P.log 2>&1
  python tools/kernel_stats.py gpurun_out/prof_$P/p_results.db 64 > gpurun_out/${R}_scnet_forward_layers_$P.txt 2>&1
  rm -rf gpurun_out/prof_$P
done
rm -rf gpurun_out/prof_fwd gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/prof_bench gpurun_out/prof_alone
bash tools/gpu_r3_bench_all.sh > gpurun_out/${R}_bench_summary.txt 2>&1
timeout 500 python bench.py --keypoint-mode reference 2>&1 | tail -1 > gpurun_out/bench_cfg1_refkp.json
timeout 500 python bench.py --config 3 --keypoint-mode reference 2>&1 | tail -1 > gpurun_out/bench_cfg3_refkp.json
cat gpurun_out/bench_cfg1.json gpurun_out/bench_cfg2.json gpurun_out/bench_cfg3.json gpurun_out/bench_cfg4.json gpurun_out/bench_cfg4_f16.json gpurun_out/bench_cfg1_refkp.json gpurun_out/bench_cfg3_refkp.json > gpurun_out/${R}_bench_configs.txt
tail -3 gpurun_out/${R}_scnet_hbm_pmc.txt; tail -4 gpurun_out/${R}_scnet_forward_layers.txt; head -3 gpurun_out/${R}_step_forwards_alone.txt; cat gpurun_out/${R}_bench_summary.txt
python - <<'PY'
import json
for c in ("1_refkp", "3_refkp"):
    try:
        d = json.load(open(f"gpurun_out/bench_cfg{c}.json")); print(c, round(d["value"], 1), d.get("roofline_keypoints", {}).get("ms_per_level"))
    except Exception as e:
        print(c, "failed", e)
PY
