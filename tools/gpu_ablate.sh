#!/bin/bash
# usage: bash tools/gpu_ablate.sh "5 3 2"   (on the GPU box; needs librelpose_hip_ablateN.so built beforehand)
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for a in $1; do
  cp relativepose_amd/librelpose_hip.so /tmp/keep.so
  cp relativepose_amd/librelpose_hip_ablate$a.so relativepose_amd/librelpose_hip.so
  timeout 300 rocprofv3 --kernel-trace -d gpurun_out/abl$a -o p -- python tools/scnet_only.py 64 3 > gpurun_out/abl$a.log 2>&1
  cp /tmp/keep.so relativepose_amd/librelpose_hip.so
  echo "== RP_ABLATE=$a"; python tools_prof.py gpurun_out/abl$a/p_results.db 64 2>&1 | grep -E "^conv2|^conv3|^deconv3|^deconv2|conv total"
done
