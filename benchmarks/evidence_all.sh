#!/bin/bash
# Everything profiles/ holds for a round, in one gpurun call: bench + shard rehearsals + kernel stats (evidence.sh), the
# PMC passes of the headline pass and of the cut_off = 0 pass with their traffic files, the batch and Monte-Carlo benches.
# usage (via gpurun): bash benchmarks/evidence_all.sh <tag>
set -u
TAG=${1:-ev}
R=$PWD
O=$R/gpurun_out
mkdir -p "$O"
bash benchmarks/evidence.sh "$TAG"
bash benchmarks/pmc.sh "$O/${TAG}_pmc" --no-also > "$O/${TAG}_pmc_summary_fr.txt" 2>&1
cd "$R"
python3 benchmarks/make_traffic.py "$O/${TAG}_pmc" "$O/${TAG}_traffic.json" 100000000 100000 1e-8 5 ${ROUND:-3} > "$O/${TAG}_traffic.log" 2>&1
bash benchmarks/pmc.sh "$O/${TAG}_pmc_fx" --no-also --cut-off 0 > "$O/${TAG}_pmc_summary_fx.txt" 2>&1
cd "$R"
python3 benchmarks/make_traffic.py "$O/${TAG}_pmc_fx" "$O/${TAG}_traffic_fx.json" 100000000 100000 0 6 ${ROUND:-3} > "$O/${TAG}_traffic_fx.log" 2>&1
timeout -k 10 400 python benchmarks/bench_batch.py > "$O/${TAG}_batch_bench.jsonl" 2> "$O/${TAG}_batch_bench.err" || echo "batch rc=$?"
timeout -k 10 200 python benchmarks/bench_mc.py > "$O/${TAG}_bench_mc.json" 2> "$O/${TAG}_bench_mc.err" || echo "mc rc=$?"
PMC_SCRIPT=benchmarks/bench_mc.py PMC_ARGS="--steps 1" bash benchmarks/pmc.sh "$O/${TAG}_pmc_mc" > "$O/${TAG}_pmc_summary_mc.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_stats_mc" -o mc -- python3 "$R/benchmarks/bench_mc.py" --steps 2 > "$O/${TAG}_stats_mc.log" 2>&1 || echo "mc stats rc=$?"
cd "$R"
bash benchmarks/bench_n2_shim.sh 2 > "$O/${TAG}_bench_n2_shim.json" 2> "$O/${TAG}_bench_n2_shim.err" || echo "n2 shim rc=$?"
echo evidence_all done
