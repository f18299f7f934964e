cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 300 python tools/bench_train_graph.py 9 casred 2>&1 | tail -1
SMVS_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_ab/wg2xcd.so timeout 300 python tools/bench_train_graph.py 9 casred 2>&1 | tail -1
done
SMVS_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_ab/wg2xcd.so timeout 600 python -m pytest tests/test_hip_end_to_end.py -q -k "wgrad or weight_gradient" 2>&1 | tail -1
