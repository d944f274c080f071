#!/bin/bash
# samples rocm-smi clocks / power while a bench variant loops (diagnostics): usage bash benchmarks/clock_probe.sh <tag> [env...]
TAG=$1; shift
O=gpurun_out; mkdir -p $O
( env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --no-also --no-pipeline --steps 40000 --probe-steps 10 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err ) &
BP=$!
sleep 12
for i in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power|fclk" | head -6 >> $O/${TAG}_smi.txt
  echo "--" >> $O/${TAG}_smi.txt
  sleep 1
done
wait $BP
cat $O/${TAG}_smi.txt | tail -14
