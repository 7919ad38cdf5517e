#!/bin/bash
# A/B of runtime switches on ONE box: bash tools/gpu_ab.sh "VAR1=a VAR2=b" "VAR1=c" ...   (each arg = one environment)
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for envs in "$@"; do
  i=$((i+1))
  echo "== [$envs]"
  env $envs timeout 300 rocprofv3 --kernel-trace -d gpurun_out/ab$i -o p -- python tools/scnet_only.py 64 3 > gpurun_out/ab$i.log 2>&1
  python tools/kernel_stats.py gpurun_out/ab$i/p_results.db 64 2>&1 | grep -E "^conv[2-4]|^deconv[2-5]|conv total"
done
