#!/bin/bash
# Collects the round's evidence on the GPU box: rocprofv3 kernel stats of the default bench command, then
# the bench line itself. usage (via gpurun): bash benchmarks/evidence.sh <tag>  -> gpurun_out/<tag>_*
set -eu
TAG=${1:-ev}
R=$PWD
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_stats" -o fr -- \
  python3 "$R/bench.py" --no-cpu-baseline > "$R/gpurun_out/${TAG}_stats.log" 2>&1
cd "$R"
timeout -k 10 500 python bench.py > "gpurun_out/${TAG}_bench.json" 2> "gpurun_out/${TAG}_bench.err"
