#!/bin/bash
# Round-6 GPU work, one parameterised script (on the GPU box: bash tools/gpu_r6.sh <stage> [args]); results under gpurun_out/.
#   split      : parity of the exact-product split modes (bf16x9 / bf16x6) + their per-layer traces + configs[1] bench lines per precision
#   layers P   : per-layer trace of one full forward in precision P (f32 | bf16x9 | bf16x6 | f16x3 | f16)
#   bench ARGS : one bench line (python bench.py ARGS), last line kept in gpurun_out/bench_last.json
#   suite      : the whole -m gpu suite + its parity log
#   sq P       : SQ counter passes (MFMA busy, waits, LDS) of one forward in precision P -> gpurun_out/r06_scnet_sq_pmc_P.txt
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
stage=$1; shift
layers() {   # $1 = precision
  rm -rf gpurun_out/prof_$1
  timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_$1 -o p -- python tools/scnet_only.py 64 3 $1 > gpurun_out/prof_$1.log 2>&1
  python tools/kernel_stats.py gpurun_out/prof_$1/p_results.db 64 > gpurun_out/r06_scnet_forward_layers_$1.txt 2>&1
  rm -rf gpurun_out/prof_$1
  tail -24 gpurun_out/r06_scnet_forward_layers_$1.txt
}
case $stage in
split)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_chain.hip -o /tmp/mfma_bf16_chain 2>/dev/null && /tmp/mfma_bf16_chain > gpurun_out/r06_mfma_bf16_chain.txt 2>&1
  cat gpurun_out/r06_mfma_bf16_chain.txt
  timeout 900 python -m pytest tests/test_gpu_scnet.py -x -q -k "layers_and_output_vs_oracle or zero_warp or self_stream or precision" 2>&1 | tail -5
  timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -k "16bit" 2>&1 | tail -5
  for P in bf16x9 bf16x6; do layers $P; done
  for P in bf16x9 bf16x6 f16x3 f32; do
    timeout 500 python bench.py --precision $P --no-cpu-baseline --no-h2d 2>&1 | tail -1 > gpurun_out/bench_cfg1_$P.json
    python - "$P" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/bench_cfg1_{sys.argv[1]}.json")); r = d["roofline"]
    print(sys.argv[1], "pairs/s", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), "conv frac", round(r["frac"], 3), "achieved", round(r["achieved"], 1),
          "in_loop", round(r["in_loop"]["frac"], 3), "alone ms", round(r["in_loop"]["forwards_alone"]["ms_per_step"], 2))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
  done
  ;;
layers) layers $1 ;;
ab)    # A/B of library variants on the per-layer trace: bash tools/gpu_r6.sh ab PREC name1 [name2 ...]  (relativepose_amd/librelpose_hip_<name>.so; "main" = the product library)
  P=$1; shift
  for v in "$@"; do
    if [ "$v" = main ]; then unset RELPOSE_LIB_PATH; else export RELPOSE_LIB_PATH=$GRAFT_REPO_ROOT/relativepose_amd/librelpose_hip_$v.so; fi
    echo "=== variant $v ($P)"
    rm -rf gpurun_out/prof_ab
    timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_ab -o p -- python tools/scnet_only.py 64 3 $P > gpurun_out/prof_ab.log 2>&1
    python tools/kernel_stats.py gpurun_out/prof_ab/p_results.db 64 > gpurun_out/r06_ab_${v}_$P.txt 2>&1
    rm -rf gpurun_out/prof_ab
    tail -21 gpurun_out/r06_ab_${v}_$P.txt | cut -c1-110
  done
  unset RELPOSE_LIB_PATH
  ;;
bench) timeout 900 python bench.py "$@" 2>&1 | tail -1 | tee gpurun_out/bench_last.json | cut -c1-600 ;;
sq)
  P=$1
  run() { name=$1; shift; rm -rf gpurun_out/sq_$name; timeout 400 rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/sq_$name -o p -- python tools/scnet_only.py 64 2 $P > gpurun_out/sq_$name.log 2>&1; }
  run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
  run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE
  { echo "# rocprofv3 --pmc (own runs, --kernel-trace only) -- python tools/scnet_only.py 64 2 $P ; last forward; percentages = fractions of SQ_WAVE_CYCLES";
    python tools/sq_summary.py gpurun_out/sq_a/p_results.db; echo "# LDS pass"; python tools/sq_summary.py gpurun_out/sq_b/p_results.db; } > gpurun_out/r06_scnet_sq_pmc_$P.txt 2>&1
  rm -rf gpurun_out/sq_a gpurun_out/sq_b
  cut -c1-330 gpurun_out/r06_scnet_sq_pmc_$P.txt
  ;;
suite)
  timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee gpurun_out/r06_gpu_suite_summary.txt
  ;;
esac
