#!/bin/bash
# Kernel times of two library builds on one box: rocprofv3 --kernel-trace --stats over a serial bench run each (A, B, A, B).
# usage (via gpurun): bash benchmarks/ab_kernel_stats.sh <tag> [bench.py arguments]
set -u
TAG=${1:-abk}; shift
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for v in A B; do
    export CANOPY_AMD_LIB=$R/benchmarks/bin/libcanopy_$v.so
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${v}_$rep -o st -- python3 $R/bench.py --no-cpu-baseline --no-also --no-pipeline "$@" > $O/${TAG}_${v}_$rep.log 2>&1 || echo "$v $rep rc=$?"
    python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/${TAG}_${v}_$rep/st_kernel_stats.csv")))
for r in rows:
    if any(k in r["Name"] for k in ("k_fr_filter","k_fr_refine","k_finalize","k_fx2","k_bs<")):
        print("$v $rep %-50s calls %5s avg %8.1f us min %8.1f" % (r["Name"].split("(")[0][-50:], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
  done
done
