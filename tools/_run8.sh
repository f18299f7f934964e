cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_costreg_training.py -q -k reference > gpurun_out/ref3d.txt 2>&1
grep -E "entry \(|norm of|running statistics|passed|failed" gpurun_out/ref3d.txt | head -40
bash tools/profile_train_step.sh casmvs 2>&1 | grep -E "training step|wgrad|kernel time"
timeout 600 python tools/bench_bwd.py 2>&1 | head -4
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_end_to_end.py -q -k "backward or grad" 2>&1 | tail -2
