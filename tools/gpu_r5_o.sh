#!/bin/bash
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_scnet.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline --no-aux 2>/dev/null | tail -1 | cut -c1-330
