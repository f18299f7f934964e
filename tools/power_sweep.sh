#!/bin/bash
# On the GPU box: sustained run of each gpurun_ab build with power / sclk samples.  tools/power_sweep.sh base x4 ...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  SMVS_LIB_PATH=$REPO/gpurun_ab/$v.so python $REPO/bench.py --no-cpu-baseline --no-extra ${PS_WORKLOAD:+--workload $PS_WORKLOAD} --steps ${PS_STEPS:-6000} --warmup 10 > /tmp/ps_$v.out 2>&1 &
  pid=$!
  sleep ${PS_DELAY:-5}
  p=""; c=""
  for i in 1 2 3 4; do
    s=$(rocm-smi --showpower --showclocks 2>/dev/null)
    p="$p $(echo "$s" | grep -o 'Power (W): [0-9.]*' | grep -o '[0-9.]*$')"
    c="$c $(echo "$s" | grep sclk | grep -o '([0-9]*Mhz' | tr -d '(Mhz')"
    sleep 0.4
  done
  wait $pid
  echo "$v: $(tail -1 /tmp/ps_$v.out | grep -o 'ms_per_step": [0-9.]*')  W:$p  sclk:$c"
done
