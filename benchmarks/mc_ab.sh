#!/bin/bash
# Monte-Carlo kernel A/B runs on the GPU box (via gpurun): usage bash benchmarks/mc_ab.sh <tag>
set -u
export TAG=${1:-mc}
O=gpurun_out
mkdir -p $O
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 200 python benchmarks/bench_mc.py "$@" > $O/${TAG}_$name.json 2> $O/${TAG}_$name.err || echo "$name rc=$?"; }
run base X=1 --
run nohash CANOPY_MC_DEBUG=1 --
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/%s_*.json" % os.environ.get("TAG","mc"))):
    try:
        j=json.load(open(f)); print(os.path.basename(f), "kernel_ms %.2f ms/step %.2f samples/s %.3e hits %d" % (j["kernel_ms"], j["ms_per_step"], j["value"], j["hits"]))
    except Exception as e:
        print(f, "ERR", e)
PY
