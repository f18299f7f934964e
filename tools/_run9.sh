cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_costreg_training.py -q -s -k "reference" 2>&1 | grep -E "worst|entry|norm of|running statistics|passed|failed" | head -30
timeout 900 python -m pytest tests/test_costreg_training.py -q 2>&1 | tail -2
SMVS_WGRAD3_WAVES=1024 true
python bench.py > gpurun_out/bench_default_new.json 2> gpurun_out/bench_default_new.err; tail -c 1500 gpurun_out/bench_default_new.json
