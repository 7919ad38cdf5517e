#!/bin/bash
# usage (GPU box): bash tools/gpu_r4_fit_pmc.sh TAG -> SQ counter passes over the fit kernel alone (one workgroup per pair, B=32; tools/matcher_time.py c 32 1),
# counters in their own runs with --kernel-trace only
TAG=${1:-fitpmc}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/${TAG}_fit_pmc.txt
for c in 1 2; do
  run() { name=$1; shift; rm -rf gpurun_out/${TAG}_$name; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/${TAG}_$name -o p -- python tools/matcher_time.py $c 32 1 > gpurun_out/${TAG}_$name.log 2>&1; }
  run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
  run b SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE
  python - $TAG $c >> gpurun_out/${TAG}_fit_pmc.txt <<'PY'
import os, sys
sys.path.insert(0, "tools")
from kernel_stats import pmc, short
tag, c = sys.argv[1], sys.argv[2]
print(f"######## config {c}: fit_pair_kernel, one workgroup per pair, B=32 (per-launch averages; percentages of SQ_WAVE_CYCLES)")
for ps in "ab":
    agg = {}
    for (did, kn, st, en), v in pmc(f"gpurun_out/{tag}_{ps}/p_results.db").items():
        if "fit_pair_kernel" not in kn: continue
        a = agg.setdefault("fit", {"n": 0, "us": 0.0}); a["n"] += 1; a["us"] += (en - st) / 1e3
        for k, x in v.items(): a[k] = a.get(k, 0.0) + x
    for k, a in agg.items():
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        line = f"[{ps}] {a['us'] / a['n']:8.1f} us/launch ({a['n']} launches)"
        for cn in sorted(a):
            if cn in ("n", "us"): continue
            line += f"  {cn.replace('SQ_', '')}={a[cn] / a['n']:.4g}" + (f"({a[cn] / wc * 100:.1f}%)" if wc and cn.startswith("SQ_") and not cn.startswith("SQ_INSTS") and cn not in ("SQ_WAVE_CYCLES", "SQ_WAVES") else "")
        print(line)
PY
  rm -rf gpurun_out/${TAG}_a gpurun_out/${TAG}_b
done
cat gpurun_out/${TAG}_fit_pmc.txt
