cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_costreg_training.py -q -s 2>&1 | grep -E "worst|entry|norm of|running statistics|passed|failed|Error" | head -30
bash tools/profile_train_step.sh casmvs 2>&1 | grep -E "training step|wgrad|kernel time"
timeout 300 python tools/bench_train_graph.py 9 casmvs 2>&1 | tail -1
