#!/bin/bash
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bench.py -m gpu -q -p no:cacheprovider -k eight_rank_plumbing -x 2>&1 | tail -60
