#!/bin/bash
# Round-2 A/B runs on the GPU box (via gpurun): usage bash benchmarks/r2_ab.sh <tag>
set -u
export TAG=${1:-ab}
O=gpurun_out
mkdir -p $O
run() { # name, env..., -- args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > $O/${TAG}_$name.json 2> $O/${TAG}_$name.err || echo "$name rc=$?"
}
# (edit the list below for the experiment at hand: name, environment, --, bench.py arguments)
run main X=1 -- --steps 200 --no-also
run main_serial X=1 -- --steps 60 --no-also --no-pipeline
run main_32bit_stream CANOPY_FR_NOP24=1 -- --steps 200 --no-also
run cut_off_0 X=1 -- --steps 60 --no-also --cut-off 0
run shard8 X=1 -- --steps 400 --no-also --shard-of 8
run shard8_rccl X=1 -- --steps 400 --no-also --shard-of 8 --force-dist
run e2e5 X=1 -- --no-also --events 200000 --steps 100
run config2 X=1 -- --no-also --rows 100000 --events 1000 --avg-nnz 4 --steps 1000
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/%s_*.json" % os.environ.get("TAG","ab"))):
    try:
        j=json.load(open(f)); r=j["roofline"]
        print(os.path.basename(f), "ms/step %.4f k_ms %.4f/%.4f pass %.4f frac %.3f kern %d" % (j["ms_per_step"], r["avg_kernel_ms"], r["min_kernel_ms"], r["avg_pass_ms"], r["frac"], j["config"]["kernel_ran"]), j["result"]["n_kept"], j["result"]["rare_event"], "prepare_s", j.get("model_prepare_s"), "also", j.get("also"))
    except Exception as e:
        print(f, "ERR", e)
PY
