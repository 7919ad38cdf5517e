#!/bin/bash
# usage (GPU box): bash tools/gpu_r3_affinity_pmc.sh TAG [B] -> rocprofv3 passes over tools/affinity_pmc.py (counters in their own runs,
# --kernel-trace only beside them), one set of passes per keypoint count; summary in gpurun_out/TAG_affinity_pmc.txt
TAG=${1:-aff}; B=${2:-1024}
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/affinity_pmc.py $B 10 ${NS:-200 400} > gpurun_out/${TAG}_events.txt 2>&1
cat gpurun_out/${TAG}_events.txt > gpurun_out/${TAG}_affinity_pmc.txt
for N in ${NS:-200 400}; do
  T=${TAG}N$N
  run() { name=$1; shift; rm -rf gpurun_out/${T}_$name; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/${T}_$name -o p -- python tools/affinity_pmc.py $B 2 $N > gpurun_out/${T}_$name.log 2>&1; }
  run fetch FETCH_SIZE
  run write WRITE_SIZE
  run sqa SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_MFMA GRBM_GUI_ACTIVE
  run sqb SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  run sqc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64
  { echo "######## B=$B N=$N"; python tools/affinity_pmc_summary.py gpurun_out $T; } >> gpurun_out/${TAG}_affinity_pmc.txt 2>&1
  tail -2 gpurun_out/${T}_sqa.log
  rm -rf gpurun_out/${T}_fetch gpurun_out/${T}_write gpurun_out/${T}_sqa gpurun_out/${T}_sqb gpurun_out/${T}_sqc
done
cat gpurun_out/${TAG}_affinity_pmc.txt
