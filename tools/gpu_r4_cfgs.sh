#!/bin/bash
# tests of the cache plans, then bench lines (no aux) for the given configs with and without the self-stream cache: bash tools/gpu_r4_cfgs.sh "1 4" [notest]
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
if [ "$2" != "notest" ]; then timeout 900 python -m pytest tests/test_gpu_scnet.py tests/test_gpu_pipeline.py -q -m gpu -x 2>&1 | tail -3; fi
for c in $1; do
  for extra in "" "--no-self-cache"; do
    timeout 300 python bench.py --config $c --no-cpu-baseline --no-h2d --no-aux $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config', d['config']['baseline_config_index'], 'cache' if d['config']['self_stream_cache'] else 'nocache', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms/step')"
  done
done
