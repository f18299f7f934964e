cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > gpurun_out/full_gpu_tests.txt
tail -5 gpurun_out/full_gpu_tests.txt
bash tools/profile_train_step.sh casmvs 2>&1 | tail -32
timeout 300 python tools/bench_train_graph.py 9 casred 2>&1 | tail -1
timeout 300 python tools/bench_train_graph.py 9 ucs 2>&1 | tail -1
