#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by a few per cent): benchmarks/bin/libcanopy_A.so and _B.so,
# alternated. usage (via gpurun): bash benchmarks/ab_libs.sh <tag> [bench.py arguments]
set -u
TAG=${1:-ab}; shift
O=gpurun_out; mkdir -p $O
for rep in 1 2 3; do
  for v in A B; do
    CANOPY_AMD_LIB=$PWD/benchmarks/bin/libcanopy_$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-also "$@" > $O/${TAG}_${v}_$rep.json 2> $O/${TAG}_${v}_$rep.err || echo "$v $rep rc=$?"
  done
done
python3 - <<PY
import json,glob
for v in "AB":
    for f in sorted(glob.glob("$O/${TAG}_%s_*.json" % v)):
        j=json.load(open(f)); r=j["roofline"]
        print(v, "ms/step %.4f serial %.4f kernels(median) %s" % (j["ms_per_step"], j.get("serial",{}).get("ms_per_step",0), r.get("kernel_ms")), j["result"]["n_kept"], j["result"]["rare_event"])
PY
