#!/usr/bin/env python3
"""Monte-Carlo throughput (BASELINE config 5: 1e9 Bernoulli samples through an AND/OR gate DAG,
bit-packed u64). Prints one JSON line: samples/s, event-draws/s, gate-evals/s. Compute/RNG bound: the
roofline that applies is the vector-ALU rate, not HBM. Multi-GPU: the sample range shards across
ranks (first_sample), hits add up exactly; run under torch.distributed.run for N>1."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    sys.stdout.flush()
    json_fd = os.dup(1)  # native libraries (RCCL's version banner) write to fd 1: keep stdout to the one JSON line
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1_000_000_000)
    ap.add_argument("--events", type=int, default=100)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--width", type=int, default=250)
    ap.add_argument("--fan-in", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--force-dist", action="store_true", help="world size 1 through the collective path (rehearsal on one GPU)")
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import canopy_amd
    from canopy_amd.quantifier import Dag
    from canopy_amd.synthetic import layered_dag
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("gloo")  # control plane only: the exchange itself is libcanopy's RCCL all-reduce
    ctx = canopy_amd.Context(device=local_rank)
    comm, collective = None, "none (single GPU)"
    if use_dist:
        from canopy_amd.distributed import library_comm
        err = ""
        try:
            comm = library_comm(ctx)
        except Exception as e:  # noqa: BLE001
            err = repr(e)
        flag = torch.tensor([1 if comm is not None else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:  # every rank falls back together
            if comm is not None:
                comm.close(); comm = None
            collective = "torch.distributed all_reduce of the hit counts (gloo) [fallback: %s]" % err
        else:
            collective = "ncclAllReduce of [hits, trials] issued by libcanopy.so (canopy_comm)"
    dag = layered_dag(20250704, args.events, args.layers, args.width, args.fan_in, atleast_every=5)
    p = np.random.default_rng(20250704).uniform(0.002, 0.1, args.events)
    d = Dag(ctx, dag)
    d.set_probabilities(p)
    per_rank = (args.samples // world) // 64 * 64
    d.run(20250704, per_rank, first_sample=rank * per_rank)  # warm-up (also spins the clocks up: 1e9 samples)
    d.run(20250704, per_rank, first_sample=rank * per_rank)
    t0 = time.perf_counter()
    ms = 0.0
    for _ in range(args.steps):
        r = d.run(20250704, per_rank, first_sample=rank * per_rank)
        ms += r.elapsed_kernel_ms
    elapsed = time.perf_counter() - t0
    if comm is not None:
        total_hits = int(comm.mc_fetch().hits)  # every rank holds the sum: the run ended with the library's all-reduce
    else:
        hits = torch.tensor([r.hits], dtype=torch.int64)
        if use_dist:
            dist.all_reduce(hits)  # the single exchange: one u64 sum
        total_hits = int(hits.item())
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        n_gates = len(dag["gate_type"]); total = per_rank * world
        os.write(json_fd, (json.dumps({"metric": "Monte-Carlo samples/sec", "value": total * args.steps / elapsed, "unit": "samples/s",
                          "n_gpus": world, "steps": args.steps, "ms_per_step": elapsed * 1e3 / args.steps,
                          "kernel_ms": ms / args.steps, "dtype": "u64 bit-packed", "data": "synthetic",
                          "config": {"workload": "layered AND/OR/K-of-N DAG", "samples": total, "n_events": args.events,
                                     "n_gates": n_gates, "fan_in": args.fan_in, "layers": args.layers},
                          "event_draws_per_s": total * args.events * args.steps / elapsed,
                          "gate_evals_per_s": total * n_gates * args.steps / elapsed,
                          "estimate": total_hits / total, "hits": total_hits, "collective": collective}) + "\n").encode())
    if comm is not None:
        comm.close()
    d.close(); ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
