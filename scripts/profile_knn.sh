# rocprofv3 kernel trace of the kNN sweep alone (randn inputs): per-kernel times -> gpurun_out/r3k/prof/knn_kernel_stats.csv
mkdir -p gpurun_out/r3k
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3k/prof -o knn -- python $GRAFT_REPO_ROOT/bench.py --workload knn --knn-inputs randn --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/r3k/knn.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3k/err.txt
cd $GRAFT_REPO_ROOT; tail -c 600 gpurun_out/r3k/knn.json; head -30 gpurun_out/r3k/prof/knn_kernel_stats.csv
