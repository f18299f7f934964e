cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_costreg_training.py -q -k "reference or featnet" 2>&1 | grep -E "AssertionError:|passed|failed" | head
SMVS_TRAIN_FEATNET_NATIVE=0 timeout 900 python -m pytest tests/test_costreg_training.py -q -k "reference" 2>&1 | grep -E "AssertionError:|passed|failed" | head
timeout 300 python tools/bench_train_graph.py 9 casmvs 2>&1 | tail -1
