#!/bin/bash
# round 5, call C: the pool variant of the affinity build (screen / exact / rank as separate launches): parity tests, timing vs the tile kernel, per-kernel trace
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fit.py tests/test_gpu_matcher.py -m gpu -q --maxfail 8 -p no:cacheprovider -k "affinity or stages or matcher" > gpurun_out/r5c_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c_tests.log
tail -25 gpurun_out/r5c_tests.log
for sel in tile pool; do
  echo "== $sel (suncg sigmas)"; RELPOSE_AFF_SEL=$sel timeout 300 python tools/affinity_pmc.py 1024 10 200 400 2>&1 | grep -v amdgpu.ids
done
for sel in tile pool; do
  echo "== $sel (scannet sigmas)"; RELPOSE_AFF_PARAMS=scannet RELPOSE_AFF_SEL=$sel timeout 300 python tools/affinity_pmc.py 1024 10 200 2>&1 | grep -v amdgpu.ids
done
rm -rf gpurun_out/prof_aff
RELPOSE_AFF_SEL=pool timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_aff -o aff -- python tools/affinity_pmc.py 1024 5 200 > gpurun_out/prof_aff.log 2>&1
python tools/kernel_stats.py gpurun_out/prof_aff/aff_results.db 2>&1 | grep -E "aff3|affinity|fill|kernel  " | cut -c1-160
rm -rf gpurun_out/prof_aff
