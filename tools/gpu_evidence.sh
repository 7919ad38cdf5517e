#!/bin/bash
# The ONE GPU-side script of this repo (round 6 folded the 49 gpu_*.sh one-offs of rounds 1-5 into it; they live in the git history).
# On the GPU box:  bash tools/gpu_evidence.sh <stage> [args]     -- every stage writes under gpurun_out/; what is to be judged is copied to profiles/.
#   suite              the whole `-m gpu` test suite                                   -> r06_gpu_suite_summary.txt (+ parity/*.jsonl)
#   layers P           per-layer trace of one full forward (64 images) in precision P  -> r06_scnet_forward_layers_P.txt
#   hbm P              FETCH_SIZE / WRITE_SIZE passes of the same forward              -> r06_scnet_hbm_pmc_P.txt
#   sq P               SQ passes (MFMA busy, waits, LDS) of the same forward           -> r06_scnet_sq_pmc_P.txt
#   alone P            the three forwards of a pipeline step alone (+ kernel stats)     -> r06_step_forwards_alone_P.txt
#   profbench ARGS     bench.py under rocprofv3 --kernel-trace --stats                 -> r06_bench_under_profiler.txt, r06_bench_kernel_stats.txt, r06_overlap.txt, r06_forward_timeline.txt
#   bench ARGS         one bench line (python bench.py ARGS)                           -> bench_last.json
#   configs            the bench line of every BASELINE configuration + variants       -> r06_bench_configs.txt, r06_bench_summary.txt
#   matcher            the matcher alone at configs[1] / configs[2] sizes              -> r06_matcher.txt
#   affinity [SEL]     SQ / HBM counters of the affinity build at B = 1024, N = 200    -> r06_affinity_pmc.txt
#   ab P V1 [V2 ...]   per-layer A/B of library variants (tools/build_variant.py; "main" = the product library)
#   split              round 6's first call: parity of bf16x9 / bf16x6, their layer traces, bench lines per precision
# (rocprofv3: counters in their own runs with --kernel-trace only, never combined with --sys-trace; cd /tmp + TMPDIR=/tmp first.)
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=r06
stage=$1; shift
layers() {   # $1 = precision
  rm -rf gpurun_out/prof_$1
  timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_$1 -o p -- python tools/scnet_only.py 64 3 $1 > gpurun_out/prof_$1.log 2>&1
  python tools/kernel_stats.py gpurun_out/prof_$1/p_results.db 64 > gpurun_out/${R}_scnet_forward_layers_$1.txt 2>&1
  rm -rf gpurun_out/prof_$1
  tail -24 gpurun_out/${R}_scnet_forward_layers_$1.txt
}
benchline() {   # $1 = output file, rest = bench args
  o=$1; shift
  timeout 900 python bench.py "$@" 2>&1 | tail -1 > $o
}
summarise() {   # print one summary line per bench json given
  python - "$@" <<'PY'
import json, sys
for p in sys.argv[1:]:
    try:
        d = json.load(open(p))
    except Exception as e:
        print(p, "failed", e); continue
    r = d.get("roofline") or {}; a = d.get("roofline_affinity") or {}; il = r.get("in_loop") or {}
    print(p.split("/")[-1], "| pairs/s", round(d["value"], 1), "| ms/step", round(d["ms_per_step"], 2), "| prec", d["config"]["conv_precision"],
          "| conv frac", round(r.get("frac", 0), 3), "achieved", round(r.get("achieved", 0), 1), "of", round(r.get("peak", 0), 1), "| in_loop", round(il.get("frac", 0), 3),
          "| alone ms", round((il.get("forwards_alone") or {}).get("ms_per_step", 0), 2), "| pcie", round(d["pcie_inclusive"]["value"], 1) if d.get("pcie_inclusive") else None,
          "| cpu", (d.get("cpu_baseline") or {}).get("value"), "| affinity hbm", round(a.get("frac", 0), 3), "valu", round((a.get("valu") or {}).get("frac", 0), 3))
PY
}
case $stage in
suite)
  timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee gpurun_out/${R}_gpu_suite_summary.txt
  ;;
layers) layers $1 ;;
hbm)
  P=$1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$c
    timeout 600 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o p -- python tools/scnet_only.py 64 2 $P > gpurun_out/pmc_$c.log 2>&1
  done
  python tools/pmc_summary.py gpurun_out > gpurun_out/${R}_scnet_hbm_pmc_$P.txt 2>&1
  rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
  tail -3 gpurun_out/${R}_scnet_hbm_pmc_$P.txt
  ;;
sq)
  P=$1
  run() { name=$1; shift; rm -rf gpurun_out/sq_$name; timeout 400 rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/sq_$name -o p -- python tools/scnet_only.py 64 2 $P > gpurun_out/sq_$name.log 2>&1; }
  run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
  run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE
  { echo "# rocprofv3 --pmc (own runs, --kernel-trace only) -- python tools/scnet_only.py 64 2 $P ; last forward; percentages = fractions of SQ_WAVE_CYCLES";
    python tools/sq_summary.py gpurun_out/sq_a/p_results.db; echo "# LDS pass"; python tools/sq_summary.py gpurun_out/sq_b/p_results.db; } > gpurun_out/${R}_scnet_sq_pmc_$P.txt 2>&1
  rm -rf gpurun_out/sq_a gpurun_out/sq_b
  cut -c1-330 gpurun_out/${R}_scnet_sq_pmc_$P.txt
  ;;
alone)
  P=$1
  { timeout 300 python tools/loop_forwards.py 64 10 $P 2>&1 | grep -v amdgpu.ids; } > gpurun_out/${R}_step_forwards_alone_$P.txt
  rm -rf gpurun_out/prof_alone
  timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_alone -o p -- python tools/loop_forwards.py 64 4 $P > gpurun_out/prof_alone.log 2>&1
  python tools/kernel_stats.py gpurun_out/prof_alone/p_results.db >> gpurun_out/${R}_step_forwards_alone_$P.txt 2>&1
  rm -rf gpurun_out/prof_alone
  head -12 gpurun_out/${R}_step_forwards_alone_$P.txt
  ;;
profbench)
  rm -rf gpurun_out/prof_bench
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --no-cpu-baseline --no-h2d --no-aux "$@" > gpurun_out/prof_bench.log 2>&1
  grep '"metric"' gpurun_out/prof_bench.log | cut -c1-400 > gpurun_out/${R}_bench_under_profiler.txt
  python tools/kernel_stats.py gpurun_out/prof_bench/bench_results.db > gpurun_out/${R}_bench_kernel_stats.txt 2>&1
  python tools/overlap.py gpurun_out/prof_bench/bench_results.db 250 1 > gpurun_out/${R}_overlap.txt 2>&1
  python tools/chain_timeline.py gpurun_out/prof_bench/bench_results.db 5 > gpurun_out/${R}_forward_timeline.txt 2>&1
  rm -rf gpurun_out/prof_bench
  cat gpurun_out/${R}_bench_under_profiler.txt; head -14 gpurun_out/${R}_bench_kernel_stats.txt; tail -6 gpurun_out/${R}_overlap.txt
  ;;
bench) timeout 900 python bench.py "$@" 2>&1 | tail -1 | tee gpurun_out/bench_last.json | cut -c1-600 ;;
configs)
  benchline gpurun_out/bench_cfg1.json
  for c in 0 2 3 4; do benchline gpurun_out/bench_cfg$c.json --config $c --no-cpu-baseline; done
  benchline gpurun_out/bench_cfg1_f32.json --precision f32 --no-cpu-baseline
  benchline gpurun_out/bench_cfg1_bf16x9.json --precision bf16x9 --no-cpu-baseline
  benchline gpurun_out/bench_cfg1_f16x3.json --precision f16x3 --no-cpu-baseline
  benchline gpurun_out/bench_cfg4_f16.json --config 4 --precision f16 --no-cpu-baseline
  benchline gpurun_out/bench_cfg1_refkp.json --keypoint-mode reference --no-cpu-baseline
  L="gpurun_out/bench_cfg1.json gpurun_out/bench_cfg0.json gpurun_out/bench_cfg2.json gpurun_out/bench_cfg3.json gpurun_out/bench_cfg4.json gpurun_out/bench_cfg1_f32.json gpurun_out/bench_cfg1_bf16x9.json gpurun_out/bench_cfg1_f16x3.json gpurun_out/bench_cfg4_f16.json gpurun_out/bench_cfg1_refkp.json"
  cat $L > gpurun_out/${R}_bench_configs.txt
  summarise $L | tee gpurun_out/${R}_bench_summary.txt
  ;;
matcher)
  { timeout 300 python tools/matcher_time.py 1; timeout 300 python tools/matcher_time.py 2; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_matcher.txt
  for c in 1 2; do
    rm -rf gpurun_out/mt$c
    timeout 300 rocprofv3 --kernel-trace -d gpurun_out/mt$c -o p -- python tools/matcher_time.py $c > gpurun_out/mt$c.log 2>&1
    python tools/kernel_stats.py gpurun_out/mt$c/p_results.db 2>&1 | grep -E "pair_|affinity|fit_pair|kernel  " >> gpurun_out/${R}_matcher.txt
    rm -rf gpurun_out/mt$c
  done
  cat gpurun_out/${R}_matcher.txt
  ;;
affinity)
  export RELPOSE_AFF_SEL=${1:-tile}; B=1024; N=200; T=aff
  timeout 300 python tools/affinity_pmc.py $B 10 $N > gpurun_out/${R}_affinity_pmc.txt 2>&1
  run() { name=$1; shift; rm -rf gpurun_out/${T}_$name; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/${T}_$name -o p -- python tools/affinity_pmc.py $B 2 $N > gpurun_out/${T}_$name.log 2>&1; }
  run fetch FETCH_SIZE
  run write WRITE_SIZE
  run sqa SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_MFMA GRBM_GUI_ACTIVE
  run sqb SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  run sqc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64
  { echo "######## B=$B N=$N kernel=$RELPOSE_AFF_SEL"; python tools/affinity_pmc_summary.py gpurun_out $T; } >> gpurun_out/${R}_affinity_pmc.txt 2>&1
  rm -rf gpurun_out/${T}_fetch gpurun_out/${T}_write gpurun_out/${T}_sqa gpurun_out/${T}_sqb gpurun_out/${T}_sqc
  cut -c1-260 gpurun_out/${R}_affinity_pmc.txt
  ;;
ab)
  P=$1; shift
  for v in "$@"; do
    if [ "$v" = main ]; then unset RELPOSE_LIB_PATH; else export RELPOSE_LIB_PATH=$GRAFT_REPO_ROOT/relativepose_amd/librelpose_hip_$v.so; fi
    echo "=== variant $v ($P)"
    rm -rf gpurun_out/prof_ab
    timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_ab -o p -- python tools/scnet_only.py 64 3 $P > gpurun_out/prof_ab.log 2>&1
    python tools/kernel_stats.py gpurun_out/prof_ab/p_results.db 64 > gpurun_out/${R}_ab_${v}_$P.txt 2>&1
    rm -rf gpurun_out/prof_ab
    tail -21 gpurun_out/${R}_ab_${v}_$P.txt | cut -c1-110
  done
  unset RELPOSE_LIB_PATH
  ;;
split)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_chain.hip -o /tmp/mfma_bf16_chain 2>/dev/null && /tmp/mfma_bf16_chain > gpurun_out/${R}_mfma_bf16_chain.txt 2>&1
  cat gpurun_out/${R}_mfma_bf16_chain.txt
  timeout 900 python -m pytest tests/test_gpu_scnet.py -x -q -k "layers_and_output_vs_oracle or zero_warp or self_stream or precision" 2>&1 | tail -5
  timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -k "16bit" 2>&1 | tail -5
  for P in bf16x9 bf16x6; do layers $P; done
  for P in bf16x9 bf16x6 f16x3 f32; do benchline gpurun_out/bench_cfg1_$P.json --precision $P --no-cpu-baseline --no-h2d; done
  summarise gpurun_out/bench_cfg1_bf16x9.json gpurun_out/bench_cfg1_bf16x6.json gpurun_out/bench_cfg1_f16x3.json gpurun_out/bench_cfg1_f32.json
  ;;
*) echo "unknown stage $stage"; exit 2 ;;
esac
