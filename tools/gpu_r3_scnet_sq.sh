#!/bin/bash
# usage (GPU box): bash tools/gpu_r3_scnet_sq.sh TAG [precision] -> SQ counter passes (MFMA busy, waits, LDS conflicts, occupancy) of one SCNet forward at 64 images
TAG=${1:-sq}; PREC=${2:-f32}
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { name=$1; shift; rm -rf gpurun_out/${TAG}_$name; timeout 400 rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/${TAG}_$name -o p -- python tools/scnet_only.py 64 2 $PREC > gpurun_out/${TAG}_$name.log 2>&1; }
run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE
{ echo "# rocprofv3 --pmc (own runs, --kernel-trace only) -- python tools/scnet_only.py 64 2 $PREC ; last forward; percentages = fractions of SQ_WAVE_CYCLES";
  python tools/sq_summary.py gpurun_out/${TAG}_a/p_results.db; echo "# LDS pass"; python tools/sq_summary.py gpurun_out/${TAG}_b/p_results.db; } > gpurun_out/${TAG}_scnet_sq.txt 2>&1
rm -rf gpurun_out/${TAG}_a gpurun_out/${TAG}_b
cat gpurun_out/${TAG}_scnet_sq.txt | cut -c1-400
