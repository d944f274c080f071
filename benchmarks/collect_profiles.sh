#!/bin/bash
# Copies what benchmarks/evidence_all.sh <tag> left under gpurun_out/ into profiles/ under the round's names.
# usage: bash benchmarks/collect_profiles.sh r03
set -eu
T=${1:-r03}
O=gpurun_out
cp $O/${T}_stats/fr_kernel_stats.csv profiles/${T}_kernel_stats.csv
cp $O/${T}_stats_serial/fr_kernel_stats.csv profiles/${T}_kernel_stats_serial.csv
cp $O/${T}_stats_mc/mc_kernel_stats.csv profiles/${T}_mc_kernel_stats.csv
cp $O/${T}_traffic.json profiles/traffic.json
cp $O/${T}_traffic_fx.json profiles/traffic_fx.json
cp $O/${T}_pmc_summary_fr.txt profiles/${T}_pmc_summary_fr.txt
cp $O/${T}_pmc_summary_mc.txt profiles/${T}_pmc_summary_mc.txt
cp $O/${T}_bench.json profiles/${T}_bench_n1.json
cp $O/${T}_bench_mc.json profiles/${T}_bench_mc_n1.json
cp $O/${T}_batch_bench.jsonl profiles/${T}_batch_bench.jsonl
cp $O/${T}_n2_rehearsal.json profiles/${T}_bench_n2_launcher_rehearsal.json
[ -s $O/${T}_bench_n2_shim.json ] && cp $O/${T}_bench_n2_shim.json profiles/${T}_bench_n2_shim_rehearsal.json
for f in shard2 shard2_serial shard4 shard4_serial shard8 shard8_serial shard8_rccl shard8_rccl_serial serial; do python3 -c "
import json
d=json.load(open('$O/${T}_$f.json')); r=d['roofline']
print(json.dumps({'run':'$f','ms_per_step':d['ms_per_step'],'serial_ms_per_step':d['serial']['ms_per_step'],'rows_this_rank':d['config']['rows_this_rank'],'pipelined':r['pipelined'],'lanes':r['lanes'],'median_kernel_ms':r['median_kernel_ms'],'median_pass_ms':r['median_pass_ms'],'collective_ms':r['collective_ms'],'verification':r['verification'],'results_fetched_per_step':d['config']['results_fetched_per_step'],'sharding':d['config']['sharding']}))"; done > profiles/${T}_shard_rehearsal.jsonl
echo collected
