cd $GRAFT_REPO_ROOT
(timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6) > gpurun_out/full_gpu_tests.txt; tail -3 gpurun_out/full_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/bench.err; tail -c 300 gpurun_out/r05_bench_default.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_flags.json 2>> gpurun_out/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_bench_driver_flags.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'])
PY
