#!/bin/bash
# kernel-trace durations of the affinity kernels at B=1024 (args: N list)
cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/afftrace
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/afftrace -o p -- python tools/affinity_pmc.py 1024 5 "$@" > gpurun_out/afftrace.log 2>&1
tail -4 gpurun_out/afftrace.log
python tools/kernel_stats.py gpurun_out/afftrace/p_results.db | grep -E "affinity|kernel " 
rm -rf gpurun_out/afftrace
