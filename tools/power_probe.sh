#!/bin/bash
# On the GPU box: sample power / clocks while a command keeps the GPU busy.  tools/power_probe.sh <tag> <cmd...>
tag=$1; shift
"$@" > /tmp/pp_$tag.out 2>&1 &
pid=$!
sleep ${PP_DELAY:-6}
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' ' ; echo
  sleep 0.7
done
wait $pid
echo "[$tag] $(tail -1 /tmp/pp_$tag.out | cut -c1-300)"
