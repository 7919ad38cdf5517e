#!/bin/bash
# round 5: scheduling experiment -- unused LDS on the head / tail launches of a forward so that they only enter CUs in the drain of the other batch's conv launches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { tag=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-aux --no-h2d "$@" > gpurun_out/r5_pad_$tag.json 2> gpurun_out/r5_pad_$tag.err
  python - <<PY
import json
try:
    r=json.loads(open("gpurun_out/r5_pad_$tag.json").read().strip().split("\n")[-1]); print("$tag", round(r["value"],1), round(r["ms_per_step"],2))
except Exception as e: print("$tag FAILED", e)
PY
}
run base
run head6 --head-lds-pad 6
run tail8 --tail-lds-pad 8
run both --head-lds-pad 6 --tail-lds-pad 8
run base2
run head12 --head-lds-pad 12
run cfg2_base --config 2
run cfg2_both --config 2 --head-lds-pad 6 --tail-lds-pad 8
