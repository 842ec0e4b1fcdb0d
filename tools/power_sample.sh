#!/bin/bash
# Board power and shader clock while a command runs: bash tools/power_sample.sh <out.txt> <command...>   (rocm-smi sampled every 0.5 s)
out=$1; shift
"$@" > /dev/null 2>&1 &
pid=$!
: > $out
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|Max Graphics" | tr '\n' ' ' >> $out
  echo >> $out
  sleep 0.5
done
wait $pid
