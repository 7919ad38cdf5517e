#!/bin/bash
# round 5, call N: the split-K rule of the mid-size layers (deconv4 / 5 / 6, conv5 / 6: want >= 3000 tiles per launch) -- experiments build, full forward alone
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
XP=$GRAFT_REPO_ROOT/relativepose_amd/librelpose_hip_xp.so
i=0
for envs in "X=0" "RELPOSE_WANT_TILES=1500" "RELPOSE_WANT_TILES=1000" "RELPOSE_WANT_TILES=6000" "RELPOSE_WANT_TILES64=1024" "RELPOSE_WANT_TILES64=4096" "RELPOSE_WANT_TILES=1500 RELPOSE_WANT_TILES64=1024"; do
  i=$((i+1))
  echo "== [$envs]"
  rm -rf gpurun_out/ab$i
  env $envs RELPOSE_LIB_PATH=$XP timeout 300 rocprofv3 --kernel-trace -d gpurun_out/ab$i -o p -- python tools/scnet_only.py 64 3 > gpurun_out/ab$i.log 2>&1
  python tools/kernel_stats.py gpurun_out/ab$i/p_results.db 64 2>&1 | grep -E "^conv[4-9]|^deconv[4-9]|conv total" | cut -c1-120
  rm -rf gpurun_out/ab$i
done
