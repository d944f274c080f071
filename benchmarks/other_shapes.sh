#!/bin/bash
# The "other shapes" table of DESIGN.md 3: 1e8 rows, steady state, every result fetched.
# usage (via gpurun): bash benchmarks/other_shapes.sh > gpurun_out/other_shapes.txt
set -u
run() { # label, bench arguments
  local label=$1; shift
  timeout -k 10 300 python bench.py --no-also --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-34s kernel %d  ms/step %.4f  serial %.4f  kernels(median) %.4f  cut-sets/s %.3g  frac %.3f  canonical %.3f' % ('$label', d['config']['kernel_ran'], d['ms_per_step'], d['serial']['ms_per_step'], r['median_kernel_ms'], d['value'], r['frac'], r['canonical_equivalent_frac']))"
}
run "E=1e3 avg 4"        --events 1000 --avg-nnz 4
run "E=1e3 avg 6 cut_off 0" --events 1000 --avg-nnz 6 --cut-off 0
run "E=2048 avg 6"       --events 2048 --avg-nnz 6
run "E=1e4 avg 6"        --events 10000 --avg-nnz 6
run "E=5e4 avg 6"        --events 50000 --avg-nnz 6
run "E=1.1e5 avg 6"      --events 110000 --avg-nnz 6
run "E=1.5e5 avg 6"      --events 150000 --avg-nnz 6
run "E=2e5 avg 6"        --events 200000 --avg-nnz 6
run "E=260096 avg 6"     --events 260096 --avg-nnz 6
run "E=2^18 avg 6"       --events 262144 --avg-nnz 6
