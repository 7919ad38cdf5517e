cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for P in f16x3 f16; do
rocprofv3 --kernel-trace -d gpurun_out/prof_$P -o p -- python tools/scnet_only.py 64 3 $P > gpurun_out/prof_$P.log 2>&1
python tools/kernel_stats.py gpurun_out/prof_$P/p_results.db 64 > gpurun_out/conv_layers_$P.txt 2>&1
rm -rf gpurun_out/prof_$P
done
cat gpurun_out/conv_layers_f16x3.txt | tail -32
