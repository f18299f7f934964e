cd $GRAFT_REPO_ROOT
python tools/bench_train_step.py 3 casmvs pinhole 64,32,8 2>&1 | tail -1
timeout 300 python tools/bench_train_graph.py 9 casmvs pinhole 64,32,8 2>&1 | tail -1
SMVS_TRAIN_COMPOSITE_MASK=320 SMVS_TRAIN_FEATNET_NATIVE=0 timeout 200 python tools/bench_train_step.py 2 casmvs pinhole 64,32,8 2>&1 | tail -1
