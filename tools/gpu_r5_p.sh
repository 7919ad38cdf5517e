#!/bin/bash
# round 5, call P: fit_pair_kernel at 512 / 768 / 1024 threads (experiments build), N = 400 (configs[2]) and N = 200
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
XP=$GRAFT_REPO_ROOT/relativepose_amd/librelpose_hip_xp.so
rm -f /tmp/pose2.npy /tmp/pose1.npy
for t in 1024 768 512; do
  echo "== N=400, $t threads"; RELPOSE_POSE_DUMP=/tmp/pose2.npy RELPOSE_FIT_THREADS=$t RELPOSE_LIB_PATH=$XP timeout 300 python tools/matcher_time.py 2 32 1 2>&1 | grep -v amdgpu.ids | cut -c1-330
done
for t in 512 768; do
  echo "== N=200, $t threads"; RELPOSE_POSE_DUMP=/tmp/pose1.npy RELPOSE_FIT_THREADS=$t RELPOSE_LIB_PATH=$XP timeout 300 python tools/matcher_time.py 1 32 1 2>&1 | grep -v amdgpu.ids | cut -c1-330
done
