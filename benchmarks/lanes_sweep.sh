#!/bin/bash
# Lane-count A/B on one box: the shard sizes of an N = 2 / 4 / 8 run and the whole model, 3 and 4 lanes alternating.
# usage (via gpurun): bash benchmarks/lanes_sweep.sh > gpurun_out/lanes_sweep.txt
set -u
for rep in 1 2 3; do
  for n in 8 4 2 1; do
    for l in 3 4; do
      if [ $n = 1 ]; then a=""; else a="--shard-of $n --force-dist"; fi
      ms=$(timeout -k 10 200 python bench.py $a --lanes $l --no-also --no-cpu-baseline --steps 400 --warmup 40 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
      echo "rep $rep shard 1/$n lanes $l ms_per_step $ms"
    done
  done
done
