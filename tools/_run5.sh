cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_costreg_training.py -x -q 2>&1 | tail -25
timeout 900 python -m pytest tests/test_hip_end_to_end.py -x -q -k "training or train" 2>&1 | tail -5
timeout 300 python tools/bench_train_graph.py 9 casmvs 2>&1 | tail -1
timeout 300 python tools/bench_train_graph.py 9 casred 2>&1 | tail -1
SMVS_TRAIN_FEATNET_NATIVE=0 timeout 300 python tools/bench_train_graph.py 9 casmvs 2>&1 | tail -1
SMVS_TRAIN_FEATNET_NATIVE=0 timeout 300 python tools/bench_train_graph.py 9 casred 2>&1 | tail -1
