#!/bin/bash
# Collects a round's evidence on the GPU box: rocprofv3 kernel stats of the default bench command, the bench line
# itself, the strong-scaling shard rehearsals (1/2, 1/4, 1/8 of the model on this one GPU, with and without pass
# pipelining, with the library's RCCL all-reduce at world size 1). usage (via gpurun): bash benchmarks/evidence.sh <tag>
set -u
TAG=${1:-ev}
R=$PWD
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_stats" -o fr -- \
  python3 "$R/bench.py" --no-cpu-baseline > "$O/${TAG}_stats.log" 2>&1 || echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_stats_serial" -o fr -- \
  python3 "$R/bench.py" --no-cpu-baseline --no-pipeline --no-also > "$O/${TAG}_stats_serial.log" 2>&1 || echo "serial stats rc=$?"
cd "$R"
timeout -k 10 500 python bench.py > "$O/${TAG}_bench.json" 2> "$O/${TAG}_bench.err" || echo "bench rc=$?"
# the N > 1 launch path from a bare shell: two ranks on this one GPU (RCCL refuses duplicate devices: the torch / gloo fallback runs)
timeout -k 10 300 python bench.py --gpus 2 --rehearse-backend gloo --steps 50 --warmup 5 > "$O/${TAG}_n2_rehearsal.json" 2> "$O/${TAG}_n2_rehearsal.err" || echo "n2 rc=$?"
for n in 2 4 8; do
  timeout -k 10 200 python bench.py --shard-of $n --no-also --no-cpu-baseline > "$O/${TAG}_shard$n.json" 2> "$O/${TAG}_shard$n.err" || echo "shard$n rc=$?"
  timeout -k 10 200 python bench.py --shard-of $n --no-also --no-cpu-baseline --no-pipeline > "$O/${TAG}_shard${n}_serial.json" 2> "$O/${TAG}_shard${n}_serial.err" || echo "shard$n serial rc=$?"
done
timeout -k 10 200 python bench.py --shard-of 8 --force-dist --no-cpu-baseline > "$O/${TAG}_shard8_rccl.json" 2> "$O/${TAG}_shard8_rccl.err" || echo "rccl rc=$?"
timeout -k 10 200 python bench.py --shard-of 8 --force-dist --no-cpu-baseline --no-pipeline > "$O/${TAG}_shard8_rccl_serial.json" 2> "$O/${TAG}_shard8_rccl_serial.err" || echo "rccl serial rc=$?"
timeout -k 10 200 python bench.py --no-also --no-cpu-baseline --no-pipeline > "$O/${TAG}_serial.json" 2> "$O/${TAG}_serial.err" || echo "serial rc=$?"
echo evidence done
