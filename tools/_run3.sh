cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_costreg_training.py -x -q 2>&1 | tail -25
python tools/bench_train_step.py 5 casmvs 2>&1 | tail -1
python tools/bench_train_step.py 5 ucs 2>&1 | tail -1
SMVS_TRAIN_COMPOSITE_MASK=256 python tools/bench_train_step.py 5 casmvs 2>&1 | tail -1
timeout 300 python tools/bench_train_graph.py 9 casmvs 2>&1 | tail -2
timeout 300 python tools/bench_train_graph.py 9 casred 2>&1 | tail -1
