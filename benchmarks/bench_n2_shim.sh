#!/bin/bash
# bench.py at N = 2 (or N=$1) on ONE GPU through the library's own collective path, with the RCCL test double of
# tests/rccl_shim preloaded (real RCCL refuses two ranks on one device): the launcher, the id broadcast, canopy_comm_init_rank,
# the pipelined-burst verification, ticketed timed steps and per-rank statistics with real concurrency between ranks.
# (Timing under the shim says nothing about RCCL: its collectives are host functions.)
set -u
N=${1:-2}
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/shim && /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -shared -Wl,-soname,librccl.so.1 -o /tmp/shim/librccl.so.1 $R/tests/rccl_shim/rccl_shim.cpp -lrt 2>/dev/null
LD_PRELOAD=/tmp/shim/librccl.so.1 timeout -k 10 400 python $R/bench.py --gpus $N --rehearse-backend gloo --rows ${ROWS:-8000000} --steps 100 --warmup 10 --no-cpu-baseline --watchdog-s 300
