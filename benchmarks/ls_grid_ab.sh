#!/bin/bash
# Two workgroups of the row-stream filter per CU where their LDS allows (table <= 64 KB)? usage (via gpurun): bash benchmarks/ls_grid_ab.sh
set -u
export CANOPY_AMD_LIB=$PWD/benchmarks/bin/libcanopy_A.so   # bash benchmarks/build_ls_variant.sh A -DCANOPY_ABLATIONS
for e in 1000 10000 50000; do for g in 0 512; do for rep in 1 2; do
  CANOPY_LS_GRID=$g timeout -k 10 200 python bench.py --events $e --no-also --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('E $e grid $g: step %.4f serial %.4f kernels %.4f' % (d['ms_per_step'], d['serial']['ms_per_step'], d['roofline']['median_kernel_ms']))"
done; done; done
