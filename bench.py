#!/usr/bin/env python3
"""bench.py -- quantified cut-sets/sec of the hot path on N MI355X (BASELINE.json metric).

A "step" is one quantification pass over the synthetic CSR resident in HBM: row kernels + final reduce and,
at N > 1, the single RCCL all-reduce of the shard partials -- issued by libcanopy.so itself right behind the final
reduce (canopy_comm_*, no Python between kernel and collective). The K timed steps run un-instrumented, with the
library's pass pipelining on (the latency-bound tail of step i -- exact pass, final reduce, all-reduce -- overlaps the
HBM-bound row kernel of step i+1: canopy_ctx_set_pipeline; --no-pipeline for the serial form). Per-kernel durations
(roofline, median / min per pass) come from a second, instrumented run of serial passes right after the timed region.

Workload. N = 1: BASELINE config 3 -- 1e8 cut sets x 1e5 basic events, avg 6 nnz/row, fp64.
N > 1 (default, `--scaling strong`): BASELINE config 4 -- the SAME 1e8-row model row-sharded N ways,
contiguous nnz-balanced shards (canopy_shard_rows' rule); rank r generates only its own shard of the counter-based
family (seed 20250704). `--scaling weak` is the opt-in: every rank holds --rows rows of an N x --rows model.

Launch: `python bench.py --gpus 1 --steps K --warmup W`, or for N>1
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...`.
torch.distributed (gloo) is the control plane only: it hands out libcanopy's RCCL id, runs the barriers and the
MAX over ranks of the timed region. Prints ONE JSON line on rank 0.
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


def traffic_for(variant, info, cut_off):
    """HBM bytes per launch from the PMC passes committed under profiles/ (benchmarks/pmc.sh + make_traffic.py): only
    quoted when they were collected for the kernel variant and workload that just ran."""
    for name in ("traffic.json", "traffic_fx.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            tj = json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        if tj.get("variant") == variant and tj.get("rows") == info.n_rows and tj.get("n_events") == info.n_events \
                and tj.get("cut_off") == cut_off:
            return tj.get("hbm_bytes_per_launch")
    return None


KERNEL_LABELS = {1: "k_rows (lane per row, L2 gather)", 2: "k_rows (lane per row, LDS table)",
                 3: "k_ws (wave-slab stream, exact CSR-order sums)", 4: "k_vs (value scatter)",
                 5: "k_fr_filter_q8 + k_fr_refine (quantised conservative filter, then exact refine)",
                 6: "k_fx2 + k_fr_refine (order-independent fixed-point sums, exact refine of the undecided rows)"}


def stats(xs):
    xs = sorted(xs)
    n = len(xs)
    if n == 0:
        return None, None, None
    med = xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])
    return sum(xs) / n, med, xs[0]


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
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong (default): --rows is the WHOLE model, sharded over the ranks (BASELINE config 4); "
                         "weak: --rows per rank")
    ap.add_argument("--rows", "--rows-per-gpu", dest="rows", type=int, default=100_000_000)
    ap.add_argument("--events", type=int, default=100_000)
    ap.add_argument("--avg-nnz", type=int, default=6)
    ap.add_argument("--cut-off", type=float, default=1e-8)
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, else a CANOPY_KERNEL_* id")
    ap.add_argument("--cpu-sample-rows", type=int, default=20_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the cut_off = 0 leg (N = 1 only)")
    ap.add_argument("--also-steps", type=int, default=30)
    ap.add_argument("--collective", choices=("lib", "torch"), default="lib",
                    help="lib: canopy_comm_* (ncclAllReduce issued by libcanopy.so); torch: torch.distributed all_reduce")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="A/B: passes strictly one after the other on one stream (canopy_ctx_set_pipeline(0))")
    ap.add_argument("--probe-steps", type=int, default=50, help="instrumented serial passes after the timed region")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="single-GPU rehearsal of a strong-scaling shard: run shard 0 of an N-way split of --rows")
    ap.add_argument("--rehearse-backend", default="",
                    help="rehearsal only: every rank on GPU LOCAL_RANK %% device_count (a one-GPU box running the N>1 code "
                         "path); the torch fallback collective then uses this backend (gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearse the N>1 code path (communicator, gather slots, all-reduce) at world size 1")
    args = ap.parse_args()

    import torch  # first: one HIP runtime shared in-process
    import torch.distributed as dist
    import canopy_amd
    from canopy_amd.distributed import ShardCombiner, library_comm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the quantification path")
    if args.rehearse_backend:  # several ranks on one GPU: the library's communicator is still tried first (RCCL may refuse
        local_rank = local_rank % torch.cuda.device_count()  # duplicate devices: then the fallback below is what is rehearsed)
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("gloo")  # control plane only

    # One stream carries the kernels AND the collective.
    tstream = torch.cuda.Stream(device=local_rank) if use_dist else None
    ctx = canopy_amd.Context(device=local_rank, stream=tstream.cuda_stream if tstream is not None else None)

    # ---- the collective: libcanopy's own RCCL communicator; torch's only if that cannot be set up ----
    comm, comb, collective, torch_group = None, None, "none (single GPU)", None
    if use_dist:
        err = ""
        if args.collective == "lib":
            try:
                comm = library_comm(ctx)
            except Exception as e:  # noqa: BLE001 -- reported in the line, then the fallback
                err = repr(e)
            flag = torch.tensor([1 if comm is not None else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:  # some rank failed: every rank falls back together
                if comm is not None:
                    comm.close()
                    comm = None
                sys.stderr.write("[bench] canopy_comm_init_rank failed (%s): falling back to torch.distributed\n" % err)
        if comm is not None:
            collective = "ncclAllReduce of %d B issued by libcanopy.so (canopy_comm, RCCL %d)" % (world * 64, comm.rccl_version())
        else:
            backend = args.rehearse_backend or "nccl"
            torch_group = dist.new_group(backend=backend)
            comb = ShardCombiner(ctx=ctx, group=torch_group)
            collective = "torch.distributed all_reduce of %d B (%s)%s" % (world * 64, backend,
                                                                         " [fallback: %s]" % err if err else "")

    # ---- the model: this rank's shard ----
    p = canopy_amd.synthetic_probabilities(SEED, args.events)
    t0 = time.time()
    n_shards = world if args.shard_of <= 0 else args.shard_of
    if args.scaling == "strong" and n_shards > 1:
        total_rows = args.rows
        bounds = canopy_amd.synthetic_shard_bounds(ctx, SEED, total_rows, args.avg_nnz, n_shards)
        k = rank if args.shard_of <= 0 else 0
        first_row, my_rows = int(bounds[k]), int(bounds[k + 1] - bounds[k])
    else:
        total_rows = args.rows * world
        first_row, my_rows = rank * args.rows, args.rows
    model = ctx.synthetic(SEED, my_rows, args.events, args.avg_nnz, first_row=first_row)
    model.set_probabilities(p)
    t_build = time.time() - t0
    prm = canopy_amd.default_params(cut_off=args.cut_off, limit_order=20, kernel=args.kernel)
    t0 = time.time()
    model.prepare(args.kernel)  # one-off derived layout for the variant that will run
    ctx.synchronize()
    t_prepare = time.time() - t0
    info = model.info()

    def step():
        ctx.quantify_async(model, prm)  # with a canopy_comm attached this ends with the all-reduce
        if comb is not None:
            with torch.cuda.stream(tstream):
                comb.all_reduce(force=args.force_dist)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    ctx.set_pipeline(not args.no_pipeline)
    pipeline_note = ""
    try:
        for _ in range(args.spinup):  # setup, like the model build: not a warm-up step, not timed
            step()
        ctx.synchronize()
    except canopy_amd._capi.CanopyError as e:  # e.g. an RCCL that refuses calls from several streams: serial passes instead
        if args.no_pipeline:
            raise
        sys.stderr.write("[bench] pipelined spin-up failed (%r): continuing with serial passes\n" % (e,))
        pipeline_note = "pipelining switched off after %r" % (e,)
        args.no_pipeline = True
        ctx.set_pipeline(False)
        for _ in range(args.spinup):
            step()
        ctx.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):  # the timed region: no per-pass events, nothing but the passes
        step()
    t_enqueued = time.perf_counter() - t0
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # instrumented probe: the same pass, serial, with HIP events around its kernels (per-kernel durations do not depend on
    # the pipelining; timing them inside the timed region would put event records between the overlapped halves)
    ctx.set_pipeline(False)
    ctx.timing_begin(args.probe_steps)
    for _ in range(args.probe_steps):
        step()
    fence()
    k_ms_sum, tot_ms_sum, n_timed = ctx.timing_end()
    launches = ctx.timing_launches(n_timed)
    ctx.set_pipeline(not args.no_pipeline)

    collective_ms = None
    if comm is not None:
        result = comm.fetch()
        collective_ms = comm.last_collective_ms()
    elif comb is not None:
        with torch.cuda.stream(tstream):
            result = comb.result()
    else:
        result = canopy_amd.combine_partials([ctx.fetch_partial()])
    ctx.fetch_partial()  # AUTO feedback + candidate count of the last pass
    cand = ctx.last_filter_candidates()

    variant = ctx.last_kernel_variant()  # what AUTO resolved to, reported by the library
    kernel_label = KERNEL_LABELS.get(variant, "?")
    if variant == 5 and info.n_events <= 112 * 1024 and info.stream_bytes < 3.6 * info.nnz + 8 * info.n_events and info.max_row_len <= 257:
        kernel_label = ("k_fr_filter_q8o + k_fr_refine (quantised conservative filter over the 24-bit slab stream, the slab's end step "
                        "folded into the next iterations; then exact refine)")
    if variant == 5 and info.n_events > 112 * 1024:
        kernel_label = "k_fr_filter_q8p + k_fr_refine (quantised filter, one column part of the table resident per sweep; then exact refine)"

    info = model.info()  # (before the all-rows leg adds its own layout to device_bytes)

    # ---- the pure-bandwidth leg, cut_off = 0 (every row through exp / log1p), same run, N = 1 ----
    also = None
    if world == 1 and not args.no_also and not args.force_dist:
        prm0 = canopy_amd.default_params(cut_off=0.0, limit_order=20, kernel=0)
        t0 = time.time()
        ctx.quantify_async(model, prm0)  # builds the layout of the all-rows kernel on first use
        ctx.synchronize()
        t_prep0 = time.time() - t0
        for _ in range(5):
            ctx.quantify_async(model, prm0)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.also_steps):
            ctx.quantify_async(model, prm0)
        ctx.synchronize()
        el0 = time.perf_counter() - t0
        ctx.set_pipeline(False)  # instrumented serial probe, as for the headline
        ctx.timing_begin(args.also_steps)
        for _ in range(args.also_steps):
            ctx.quantify_async(model, prm0)
        ctx.synchronize()
        _, _, n0 = ctx.timing_end()
        l0 = ctx.timing_launches(n0)
        ctx.set_pipeline(not args.no_pipeline)
        r0 = canopy_amd.combine_partials([ctx.fetch_partial()])
        v0 = ctx.last_kernel_variant()
        k_mean, k_med, k_min = stats([a for a, _ in l0])
        p_mean, p_med, p_min = stats([b for _, b in l0])
        also = {"cut_off_0": {
            "ms_per_step": el0 * 1e3 / args.also_steps, "steps": args.also_steps,
            "kernel_ran": v0, "kernel": KERNEL_LABELS.get(v0, "?"),
            "kernel_ms": {"mean": k_mean, "median": k_med, "min": k_min},
            "pass_ms": {"mean": p_mean, "median": p_med, "min": p_min},
            "frac": info.algorithmic_bytes / (k_med * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": traffic_for(v0, info, 0.0),
            "frac_real": (traffic_for(v0, info, 0.0) / (k_med * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic_for(v0, info, 0.0) else None,
            "cut_sets_per_s": info.n_rows * args.also_steps / el0,
            "n_kept": r0.n_kept, "rare_event": r0.rare_event, "mcub": r0.mcub, "layout_build_s": t_prep0,
        }}
    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        k_mean, k_med, k_min = stats([a for a, _ in launches])
        p_mean, p_med, p_min = stats([b for _, b in launches])
        achieved = info.algorithmic_bytes / (k_mean * 1e-3) / 1e9
        traffic = traffic_for(variant, info, args.cut_off)
        cand_frac = (cand[0] / cand[1]) if (cand and cand[1] and variant == 5) else None
        line = {
            "metric": "quantified cut-sets/sec", "value": total_rows * args.steps / elapsed, "unit": "cut-sets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_passes": args.spinup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "synthetic CSR %.3g cut-sets x %.0e basic events in all (%s: %.4g rows on this rank), avg %d nnz/row, "
                            "fp64, rare-event + MCUB, cut_off %g%s"
                            % (total_rows, info.n_events,
                               "one GPU" if world == 1 else "%s scaling over %d ranks" % (args.scaling, world),
                               info.n_rows, args.avg_nnz, args.cut_off,
                               "; %.2f %% of the rows pass the conservative filter into the exact pass, %.2f %% survive the cut-off"
                               % (100 * cand_frac, 100.0 * result.n_kept / max(total_rows if use_dist and not args.force_dist else info.n_rows, 1))
                               if cand_frac is not None else ""),
                "total_rows": total_rows, "rows_this_rank": info.n_rows, "first_row_this_rank": first_row,
                "nnz_this_rank": info.nnz, "n_events": info.n_events,
                "sharding": ("contiguous nnz-balanced row shards (canopy_shard_rows rule), %d rank(s); " % world + collective)
                if use_dist else "single GPU, no collective",
                "seed": SEED, "kernel": args.kernel, "kernel_ran": variant,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "frac_real": (traffic / (k_mean * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "note": "achieved/frac = SURVEY 8(d) ALGORITHMIC bytes (canonical CSR, 32 B per cut set) / kernel time: an "
                        "equivalent rate, not bytes moved -- the filter streams a re-encoding of the model (stream_bytes_per_launch: 18.1 B per cut set in the 24-bit form, 24.1 B otherwise), so frac may exceed 1; frac_real = "
                        "PMC-measured HBM bytes / kernel time / peak. Both hold while few rows pass the filter (config.workload); "
                        "see also.cut_off_0 for the all-rows case",
                "kernel": kernel_label,
                "algorithmic_bytes_per_launch": info.algorithmic_bytes,
                "bytes_per_cut_set": info.algorithmic_bytes / max(info.n_rows, 1),
                "stream_bytes_per_launch": info.stream_bytes,
                "streamed_gbs": info.stream_bytes / (k_mean * 1e-3) / 1e9 if (info.stream_bytes and variant == 5) else None,
                "avg_kernel_ms": k_mean, "median_kernel_ms": k_med, "min_kernel_ms": k_min,
                "avg_pass_ms": p_mean, "median_pass_ms": p_med, "min_pass_ms": p_min, "launches_timed": n_timed,
                "serial_pass_ms_note": "pass = row kernels + final reduce (+ all-reduce at N > 1), one pass at a time",
                "host_enqueue_ms_per_step": t_enqueued * 1e3 / args.steps,
                "last_collective_ms": collective_ms,
                "pipelined": not args.no_pipeline, "pipeline_note": pipeline_note or None,
                "probe": "kernel / pass ms: %d instrumented serial passes after the timed region" % n_timed,
            },
            "result": {"rare_event": result.rare_event, "mcub": result.mcub, "n_kept": result.n_kept,
                       "argmax_row": result.argmax_row},
            "model_build_s": t_build, "model_prepare_s": t_prepare, "device_bytes": info.device_bytes,
        }
        if also is not None:
            line["also"] = also
        if world == 1 and not args.no_cpu_baseline and not args.force_dist:
            line["cpu_baseline"] = cpu_baseline(ctx, canopy_amd, args.events, args.avg_nnz, p, args.cpu_sample_rows)
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    model.close()
    if comm is not None:
        comm.close()
    if comb is not None:
        comb.detach()
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
