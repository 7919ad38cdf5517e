#!/bin/bash
# usage: bash tools/gpu_ablate.sh "NAME1 NAME2" [grep pattern]   (on the GPU box; needs relativepose_amd/librelpose_hip_NAME.so built beforehand)
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
PAT=${2:-"^conv2|^conv3|^deconv3|^deconv2|conv total|conv1"}
for a in $1; do
  cp relativepose_amd/librelpose_hip.so /tmp/keep.so
  cp relativepose_amd/librelpose_hip_$a.so relativepose_amd/librelpose_hip.so
  timeout 300 rocprofv3 --kernel-trace -d gpurun_out/abl_$a -o p -- python tools/scnet_only.py 64 3 > gpurun_out/abl_$a.log 2>&1
  cp /tmp/keep.so relativepose_amd/librelpose_hip.so
  echo "== $a"; python tools/kernel_stats.py gpurun_out/abl_$a/p_results.db 64 2>&1 | grep -E "$PAT"
done
