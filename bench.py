#!/usr/bin/env python3
"""bench.py -- quantified cut-sets/sec of the hot path on N MI355X (BASELINE.json metric).

A "step" is one quantification pass (row kernel + final reduce, + the single RCCL all-reduce of
the shard partials at N>1) over the synthetic CSR resident in HBM. Workload at every N: each rank
holds ROWS_PER_GPU = 1e8 cut sets x 1e5 basic events, avg 6 nnz/row, fp64 (BASELINE config 3 per
GPU; at N GPUs the model is N*1e8 rows, row-sharded -> weak scaling). Rank r owns global rows
[r*1e8, (r+1)*1e8) of the same counter-based family (seed 20250704).

Launch: `python bench.py --gpus 1 --steps K --warmup W`, or for N>1
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...`.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20250704
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy rate)


def cpu_baseline(ctx, canopy_amd, n_events, avg, p, sample_rows):
    """The CPU oracle timed on this box's host cores on a bounded sample of the same workload
    (rows [0, sample_rows) of the family). Build-authored baseline ("port"): the reference has no
    CPU quantifier to time (reference src/canopy.cpp:1-31 holds no code)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_py
    cores = len(os.sched_getaffinity(0))
    row_ptr, col = oracle_py.synth_csr(SEED, sample_rows, n_events, avg, n_threads=cores)
    times = []
    for _ in range(3):
        q = oracle_py.quantify(row_ptr, col, p, cut_off=1e-8, limit_order=20, n_threads=cores, want_rows=False)
        times.append(q.elapsed_s)
    t_all = sorted(times)[1]
    one_rows = max(sample_rows // 10, 1)
    q1 = oracle_py.quantify(row_ptr[:one_rows + 1], col[:row_ptr[one_rows]], p, cut_off=1e-8, limit_order=20,
                            n_threads=1, want_rows=False)
    # parity gate on the same sample: GPU (through the C ABI) vs oracle
    m = ctx.synthetic(SEED, sample_rows, n_events, avg)
    m.set_probabilities(p)
    g = ctx.quantify(m, canopy_amd.default_params())
    m.close()
    ok = (g.n_kept == int(q.n_kept) and g.argmax_row == int(q.argmax_row)
          and abs(g.rare_event - q.rare_event) <= 1e-12 * q.rare_event and abs(g.mcub - q.mcub) <= 1e-12 * q.mcub)
    return {
        "value": sample_rows / t_all, "unit": "cut-sets/s", "cores": cores, "kind": "port",
        "sample": "rows [0,%d) of the same synthetic family (E=%d, avg %d nnz/row), median of 3 passes on %d threads; "
                  "1 thread: %.3g cut-sets/s on %d rows" % (sample_rows, n_events, avg, cores, one_rows / q1.elapsed_s,
                                                          one_rows),
        "gpu_matches_oracle_on_sample": bool(ok),
    }


def main():
    # Contract: rank 0 prints ONE JSON line on stdout. Native libraries write there too (RCCL prints a version banner
    # at communicator init), so file descriptor 1 points at stderr for the whole run and the line goes to the saved fd.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup", type=int, default=40,
                    help="untimed passes right after the model build, before the warm-up steps: the GPU idles while the model "
                         "is generated and its first ~20 launches run through a clock / power transient (filter launch 450 us "
                         "decaying to the steady 380 us: profiles/r01_filter_launch_sequence.txt)")
    ap.add_argument("--rows-per-gpu", type=int, default=100_000_000)
    ap.add_argument("--events", type=int, default=100_000)
    ap.add_argument("--avg-nnz", type=int, default=6)
    ap.add_argument("--cut-off", type=float, default=1e-8)
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, else a CANOPY_KERNEL_* id")
    ap.add_argument("--cpu-sample-rows", type=int, default=20_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-backend", default="",
                    help="rehearsal only: process-group backend other than nccl (gloo) with every rank on GPU "
                         "LOCAL_RANK %% device_count -- lets a one-GPU box run the N>1 code path")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearse the N>1 code path (process group, gather buffer, all-reduce) at world size 1")
    args = ap.parse_args()

    import torch  # first: one HIP runtime shared in-process
    import torch.distributed as dist
    import canopy_amd
    from canopy_amd.distributed import ShardCombiner

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the quantification path")
    if args.rehearse_backend:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.rehearse_backend:
            dist.init_process_group(args.rehearse_backend)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm

    # One stream carries the kernels AND the collective: a dedicated (non-default) torch stream whose
    # raw hipStream_t is handed to the library (a NULL handle would make the library create its own).
    tstream = torch.cuda.Stream(device=local_rank) if use_dist else None
    ctx = canopy_amd.Context(device=local_rank, stream=tstream.cuda_stream if tstream is not None else None)
    p = canopy_amd.synthetic_probabilities(SEED, args.events)
    t0 = time.time()
    model = ctx.synthetic(SEED, args.rows_per_gpu, args.events, args.avg_nnz, first_row=rank * args.rows_per_gpu)
    model.set_probabilities(p)
    info = model.info()
    t_build = time.time() - t0
    prm = canopy_amd.default_params(cut_off=args.cut_off, limit_order=20, kernel=args.kernel)
    t0 = time.time()
    model.prepare(args.kernel)  # one-off derived layout (wave-slab stream) for the variant that will run
    ctx.synchronize()
    t_prepare = time.time() - t0
    info = model.info()
    comb = ShardCombiner(ctx=ctx) if use_dist else None

    def step():
        ctx.quantify_async(model, prm)
        if comb is not None:
            with torch.cuda.stream(tstream):
                comb.all_reduce(force=args.force_dist)  # K5: the one collective, same stream as the kernels

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.spinup):  # setup, like the model build: not a warm-up step, not timed
        ctx.quantify_async(model, prm)
    ctx.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    ctx.timing_begin(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    k_ms_sum, tot_ms_sum, n_timed = ctx.timing_end()

    if comb is not None:
        with torch.cuda.stream(tstream):
            result = comb.result()
    else:
        result = canopy_amd.combine_partials([ctx.fetch_partial()])

    variant = ctx.last_kernel_variant()  # what AUTO resolved to, reported by the library
    kernel_label = {1: "k_rows (lane per row, L2 gather)", 2: "k_rows (lane per row, LDS table)",
                    3: "k_ws (wave-slab stream)", 4: "k_vs (value scatter)",
                    5: "k_fr_filter_q8 + k_fr_refine (quantised conservative filter, then exact refine)"
                    if info.n_events <= 112 * 1024 else "k_fr_filter + k_fr_refine (filter, then exact refine)"}[variant]
    if rank == 0:
        total_rows = info.n_rows * world
        ms_per_step = elapsed * 1e3 / args.steps
        k_ms = k_ms_sum / max(n_timed, 1)
        achieved = info.algorithmic_bytes / (k_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes committed under profiles/ (benchmarks/pmc.sh): only quoted
        # when they were collected for the kernel and workload that just ran
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("variant") == variant and tj.get("rows") == info.n_rows and tj.get("n_events") == info.n_events \
                        and tj.get("cut_off") == args.cut_off:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "quantified cut-sets/sec", "value": total_rows * args.steps / elapsed, "unit": "cut-sets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_passes": args.spinup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "synthetic CSR %.0e cut-sets x %.0e basic events per GPU, avg %d nnz/row, fp64, "
                            "rare-event + MCUB, cut_off %g" % (info.n_rows, info.n_events, args.avg_nnz, args.cut_off),
                "rows_per_gpu": info.n_rows, "nnz_per_gpu": info.nnz, "n_events": info.n_events,
                "sharding": "row-sharded, %d rank(s), one all-reduce of %d B" % (world, world * 64) if world > 1
                else "single GPU, no collective",
                "seed": SEED, "kernel": args.kernel, "kernel_ran": variant,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": kernel_label,
                "algorithmic_bytes_per_launch": info.algorithmic_bytes,
                "bytes_per_cut_set": info.algorithmic_bytes / max(info.n_rows, 1),
                "stream_bytes_per_launch": info.stream_bytes,
                "streamed_gbs": info.stream_bytes / (k_ms * 1e-3) / 1e9 if (info.stream_bytes and variant == 5) else None,
                "avg_kernel_ms": k_ms, "avg_pass_ms": tot_ms_sum / max(n_timed, 1), "launches_timed": n_timed,
            },
            "result": {"rare_event": result.rare_event, "mcub": result.mcub, "n_kept": result.n_kept,
                       "argmax_row": result.argmax_row},
            "model_build_s": t_build, "model_prepare_s": t_prepare,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ctx, canopy_amd, args.events, args.avg_nnz, p, args.cpu_sample_rows)
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    model.close()
    if comb is not None:
        comb.detach()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
