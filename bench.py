#!/usr/bin/env python3
"""bench.py -- quantified cut-sets/sec of the hot path on N MI355X (BASELINE.json metric).

A "step" is one quantification pass over the synthetic CSR resident in HBM -- row kernels + final reduce and, at N > 1,
the single RCCL all-reduce of the shard partials, issued by libcanopy.so itself right behind the final reduce
(canopy_comm_*, no Python between kernel and collective) -- WITH its result delivered to the host: every step is
enqueued with a ticket (canopy_quantify_enqueue) and its record fetched (canopy_ticket_fetch /
canopy_comm_ticket_fetch) while up to two later steps are in flight. The library's pass pipelining is on (the
latency-bound tail of step i -- exact pass, final reduce, all-reduce -- overlaps the HBM-bound row kernel of step i+1;
--no-pipeline for the serial form). Per-kernel durations (roofline, median / min per pass) come from a second,
instrumented run of serial passes right after the timed region (HIP events on the stream the kernels run on).

Workload. N = 1: BASELINE config 3 -- 1e8 cut sets x 1e5 basic events, avg 6 nnz/row, fp64.
N > 1 (default, `--scaling strong`): BASELINE config 4 -- the SAME 1e8-row model row-sharded N ways,
contiguous nnz-balanced shards (canopy_shard_rows' rule); rank r generates only its own shard of the counter-based
family (seed 20250704). `--scaling weak` is the opt-in: every rank holds --rows rows of an N x --rows model.

Launch: `python bench.py --gpus N --steps K --warmup W` from a bare shell -- for N > 1 the script starts its N ranks
itself (N fresh child processes, spawned BEFORE this process touches torch or the GPU; rank 0's JSON line is relayed)
-- or `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment). torch.distributed (gloo) is the control plane only:
it hands out libcanopy's RCCL id, runs the barriers and the MAX over ranks of the timed region.
At N > 1 every rank first checks that a pipelined burst and a serial pass deliver bit-identical records; a mismatch
switches ALL ranks to serial passes (reported in roofline.pipeline_note). A watchdog turns a hang into a non-zero exit.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20250704
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy rate)


def parse_args(argv=None):
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
    ap.add_argument("--no-also", action="store_true", help="skip the extra legs (cut_off = 0, new p per step, selectivity)")
    ap.add_argument("--also-steps", type=int, default=30)
    ap.add_argument("--collective", choices=("lib", "torch"), default="lib",
                    help="lib: canopy_comm_* (ncclAllReduce issued by libcanopy.so); torch: torch.distributed all_reduce")
    ap.add_argument("--comm-mode", choices=("chained", "unordered", "own-stream"), default="chained",
                    help="how consecutive collectives of the communicator are ordered on the device (canopy_comm_set_mode)")
    ap.add_argument("--lanes", type=int, default=0,
                    help="lanes consecutive passes rotate through (2..4); 0 = what measured best: 4 (0.318 against 0.323 ms per step for the "
                         "whole model; 42.7 against 46.3 us at the 8-way shard size)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="A/B: passes strictly one after the other on one stream (canopy_ctx_set_pipeline(0))")
    ap.add_argument("--no-fetch", action="store_true", help="A/B: the timed steps are enqueued without fetching their records")
    ap.add_argument("--probe-steps", type=int, default=50, help="instrumented serial passes after the timed region")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="single-GPU rehearsal of a strong-scaling shard: run shard 0 of an N-way split of --rows")
    ap.add_argument("--rehearse-backend", default="",
                    help="rehearsal only: every rank on GPU LOCAL_RANK %% device_count (a one-GPU box running the N>1 code "
                         "path); the torch fallback collective then uses this backend (gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearse the N>1 code path (communicator, gather slots, all-reduce) at world size 1")
    ap.add_argument("--watchdog-s", type=float, default=900.0,
                    help="wall-clock limit of the whole run (launcher) and of each phase that can hang in a collective (ranks)")
    return ap.parse_args(argv)


# ---- launcher: N > 1 from a bare shell ---------------------------------------------------------------------------

def launch_ranks(args):
    """Start the N ranks as fresh child processes. This process has imported neither torch nor canopy_amd and never
    touches the GPU: nothing is re-executed from a process that has initialised it."""
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = subprocess.PIPE if r == 0 else sys.stderr  # rank 0 prints the line; nobody else writes to our stdout
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    deadline = time.time() + args.watchdog_s
    line = [b""]

    def read0():
        line[0] = procs[0].stdout.read()

    t = threading.Thread(target=read0, daemon=True)
    t.start()
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                rc = max(abs(c) for c in codes)
                break
            if any(c not in (None, 0) for c in codes):  # one rank died: the others would wait for it in a collective
                time.sleep(5.0)
                rc = next(abs(c) for c in codes if c not in (None, 0)) or 1
                sys.stderr.write("[bench] a rank exited with code %d: stopping the others\n" % rc)
                break
            if time.time() > deadline:
                sys.stderr.write("[bench] watchdog: the %d ranks did not finish within %.0f s: stopping them\n" % (n, args.watchdog_s))
                rc = 3
                break
            time.sleep(0.2)
    finally:
        for p in procs:  # exactly the processes started above
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    t.join(timeout=5)
    if line[0]:
        sys.stdout.write(line[0].decode())
        sys.stdout.flush()
    return rc


class Watchdog:
    """A phase that can hang in a collective (a rank that never issues its all-reduce, a mis-ordered one) ends the
    process with a message and a non-zero code instead of hanging the job."""

    def __init__(self, seconds, rank):
        self.seconds, self.rank, self.timer, self.what = seconds, rank, None, ""

    def arm(self, what):
        self.disarm()
        self.what = what

        def fire():
            sys.stderr.write("[bench] watchdog (rank %d): '%s' did not finish within %.0f s -- a stuck or mis-ordered "
                             "collective, or a rank that died; exiting with code 3\n" % (self.rank, self.what, self.seconds))
            sys.stderr.flush()
            os._exit(3)

        self.timer = threading.Timer(self.seconds, fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


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
                 6: "k_fx2 + k_fr_refine (order-independent fixed-point sums, exact refine of the undecided rows)",
                 7: "k_bs (band stream: workgroup-wide slabs, a barrier between column-sorted bands, exact CSR-order sums)"}


def stats(xs):
    xs = sorted(xs)
    n = len(xs)
    if n == 0:
        return None, None, None
    med = xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])
    return sum(xs) / n, med, xs[0]


def moved_bytes(variant, info, listed_rows):
    """Bytes one pass MUST move in the layout it actually reads (the byte figure restated for the
    layout in use), from the live numbers of this run:
      two-step passes  the slab stream the conservative kernel reads (canopy_model_info.stream_bytes: packed entries,
                       chunk offsets, the table once) + per row handed to the exact pass two 128-B lines of the canonical
                       CSR (its row_ptr pair and its col_idx span; rows are short, the lines random) + its 2-B list entry
                       written and read back;
      every other pass the canonical CSR (algorithmic bytes).
    """
    if variant == 5 and getattr(info, "stream_bytes_rows", 0):  # the row stream: per listed row the line of its columns and (shared by the item's candidates) the line of its position word
        return info.stream_bytes_rows + (listed_rows or 0) * (2 * 128 + 4)
    if variant == 5 and info.stream_bytes:
        return info.stream_bytes + (listed_rows or 0) * (2 * 128 + 4)
    if variant == 6 and info.stream_bytes_wide:  # the fixed-point pass streams the 32-bit form of the slab stream
        return info.stream_bytes_wide + (listed_rows or 0) * (2 * 128 + 4)
    if variant in (3, 7) and info.stream_bytes_exact:
        return info.stream_bytes_exact
    if variant in (3, 4) and info.stream_bytes:
        return info.stream_bytes
    return info.algorithmic_bytes


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))  # (before any import of torch / canopy_amd: this process stays off the GPU)

    # Contract: rank 0 prints ONE JSON line on stdout. Native libraries write there too (RCCL prints a version banner
    # at communicator init), so file descriptor 1 points at stderr for the whole run and the line goes to the saved fd.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch  # first: one HIP runtime shared in-process
    import torch.distributed as dist
    import canopy_amd
    from canopy_amd import _capi
    from canopy_amd.distributed import ShardCombiner, library_comm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the quantification path")
    if args.rehearse_backend:  # several ranks on one GPU: the library's communicator is still tried first (RCCL may refuse
        local_rank = local_rank % torch.cuda.device_count()  # duplicate devices: then the fallback below is what is rehearsed)
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    dog = Watchdog(args.watchdog_s, rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dog.arm("process group rendezvous")
        dist.init_process_group("gloo")  # control plane only
        dog.disarm()

    def all_ok(flag):
        """MIN over ranks of a local yes/no: every rank takes the same branch."""
        if not use_dist:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    # One stream carries the kernels AND the collective.
    tstream = torch.cuda.Stream(device=local_rank) if use_dist else None
    ctx = canopy_amd.Context(device=local_rank, stream=tstream.cuda_stream if tstream is not None else None)

    # ---- the collective: libcanopy's own RCCL communicator; torch's only if that cannot be set up ----
    comm, comb, collective, torch_group = None, None, "none (single GPU)", None
    if use_dist:
        err = ""
        if args.collective == "lib":
            dog.arm("canopy_comm_init_rank")
            try:
                comm = library_comm(ctx)
                comm.set_mode({"chained": _capi.COMM_CHAINED, "unordered": _capi.COMM_UNORDERED,
                               "own-stream": _capi.COMM_OWN_STREAM}[args.comm_mode])
            except Exception as e:  # noqa: BLE001 -- reported in the line, then the fallback
                err = repr(e)
                if comm is not None:
                    comm.close()
                    comm = None
            if not all_ok(comm is not None):  # some rank failed: every rank falls back together
                if comm is not None:
                    comm.close()
                    comm = None
                sys.stderr.write("[bench] canopy_comm_init_rank failed (%s): falling back to torch.distributed\n" % err)
            dog.disarm()
        if comm is not None:
            collective = "ncclAllReduce of %d B issued by libcanopy.so (canopy_comm, RCCL %d, collectives %s)" % (
                world * 64, comm.rccl_version(), args.comm_mode)
        else:
            # torch's own collective instead: RCCL ("nccl") first; if its communicator cannot be set up either (probed with
            # one all-reduce, the decision taken by all ranks together), gloo -- slow, but the run completes and says so
            backend = args.rehearse_backend or "nccl"
            dog.arm("torch.distributed fallback group (%s)" % backend)
            err2 = ""
            try:
                torch_group = dist.new_group(backend=backend)
                probe = torch.ones(1, dtype=torch.float64, device="cuda:%d" % local_rank)
                dist.all_reduce(probe, group=torch_group)
                torch.cuda.synchronize()
                ok_probe = int(probe.item()) == world
            except Exception as e:  # noqa: BLE001
                err2, ok_probe = repr(e), False
            if not all_ok(ok_probe):
                sys.stderr.write("[bench] torch.distributed backend %s failed (%s): falling back to gloo\n" % (backend, err2))
                backend = "gloo"
                torch_group = dist.new_group(backend=backend)
            dog.disarm()
            comb = ShardCombiner(ctx=ctx, group=torch_group)
            collective = "torch.distributed all_reduce of %d B (%s)%s" % (world * 64, backend,
                                                                         " [fallback: %s %s]" % (err, err2) if (err or err2) else "")

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

    def fetch_ticket(t):
        """The result of one step on the host: the combined records at N > 1, the shard record otherwise."""
        if comm is not None:
            return comm.ticket_partials(t)
        return [ctx.ticket_fetch(t)]

    def rec_tuple(parts):
        return tuple((q.rare_hi, q.rare_lo, q.log1p_hi, q.log1p_lo, q.n_kept, q.n_certain, q.max_logp, q.argmax_row) for q in parts)

    depth = {"n": 4}  # steps in flight: as many as lanes when pipelined, ONE otherwise (un-pipelined passes share a lane,
    #                    which keeps the record of its last ticketed pass only: fetch before the next enqueue)

    lanes = args.lanes if args.lanes >= 2 else 4

    def set_pipeline(on):
        ctx.set_pipeline(lanes if on else 0)
        depth["n"] = lanes if on else 1

    def run_steps(n, set_p=None, fetch=True, prm_=None):
        """n steps, each with its result fetched while up to two later steps are in flight (pipelined) or before the next
        is enqueued (serial). The torch-collective fallback has no tickets: its steps are enqueued only."""
        prm_ = prm if prm_ is None else prm_
        inflight, last = [], None
        for i in range(n):
            if set_p is not None:
                model.set_probabilities(set_p(i))
            if comb is not None:
                ctx.quantify_async(model, prm_)
                with torch.cuda.stream(tstream):
                    comb.all_reduce(force=args.force_dist)
                continue
            if not fetch:
                ctx.quantify_async(model, prm_)
                continue
            inflight.append(ctx.enqueue(model, prm_))
            if len(inflight) == depth["n"]:
                last = fetch_ticket(inflight.pop(0))
        while inflight:
            last = fetch_ticket(inflight.pop(0))
        return last

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    fetch_steps = not args.no_fetch
    set_pipeline(not args.no_pipeline)
    pipeline_note = ""

    # ---- N > 1: a pipelined burst must deliver what a serial pass delivers, record for record, on every rank ----
    verify = None
    if use_dist and comm is not None and not args.no_pipeline:
        dog.arm("pipelined-burst verification")
        ok_local, why = True, ""
        try:
            set_pipeline(False)
            serial_rec = rec_tuple(fetch_ticket(ctx.enqueue(model, prm)))
            set_pipeline(True)
            tickets, burst = [], []
            for i in range(9):  # more than two rounds over the lanes, as many in flight as there are lanes
                tickets.append(ctx.enqueue(model, prm))
                if len(tickets) == lanes:
                    burst.append(rec_tuple(fetch_ticket(tickets.pop(0))))
            while tickets:
                burst.append(rec_tuple(fetch_ticket(tickets.pop(0))))
            bad = [i for i, b in enumerate(burst) if b != serial_rec]
            if bad:
                ok_local, why = False, "pipelined passes %s of 9 delivered records that differ from the serial pass" % bad
            if len(serial_rec) != world:
                ok_local, why = False, "expected %d records, got %d" % (world, len(serial_rec))
        except _capi.CanopyError as e:
            ok_local, why = False, repr(e)
        everyone = all_ok(ok_local)
        dog.disarm()
        verify = {"pipelined_equals_serial": bool(everyone), "burst": 9}
        if not everyone:
            pipeline_note = "pipelining switched off on every rank: " + (why or "another rank reported a mismatch")
            sys.stderr.write("[bench] rank %d: %s\n" % (rank, pipeline_note))
            args.no_pipeline = True
            set_pipeline(False)

    dog.arm("spin-up + warm-up")
    run_steps(args.spinup, fetch=fetch_steps)  # setup, like the model build: not a warm-up step, not timed
    ctx.synchronize()
    run_steps(args.warmup, fetch=fetch_steps)
    fence()
    dog.arm("timed region")
    t0 = time.perf_counter()
    last_parts = run_steps(args.steps, fetch=fetch_steps)  # the timed region: the passes and the fetch of every result
    t_enqueued = time.perf_counter() - t0
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed_own = time.perf_counter() - t0
    elapsed = elapsed_own
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- the serial form, un-instrumented, same fetch discipline: what a strictly one-at-a-time caller sees ----
    dog.arm("serial leg")
    serial_steps = max(10, min(args.steps, 50))
    set_pipeline(False)
    run_steps(5, fetch=fetch_steps)
    fence()
    t0 = time.perf_counter()
    run_steps(serial_steps, fetch=fetch_steps)
    ctx.synchronize()
    torch.cuda.synchronize()
    serial_own = time.perf_counter() - t0
    serial_elapsed = serial_own
    if use_dist:
        dist.barrier()
        t = torch.tensor([serial_elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        serial_elapsed = float(t.item())

    # instrumented probe: the same pass, serial, with HIP events around its kernels on the stream they run on (per-kernel
    # durations do not depend on the pipelining; timing them inside the timed region would put event records between the
    # overlapped halves). One pass at a time at N > 1, so that every pass's collective time can be read.
    dog.arm("instrumented probe")
    coll_ms = []
    ctx.timing_begin(args.probe_steps)
    for _ in range(args.probe_steps):
        if comb is not None:
            run_steps(1)
        else:
            ctx.quantify_async(model, prm)
            if comm is not None:
                coll_ms.append(comm.last_collective_ms())
    fence()
    k_ms_sum, tot_ms_sum, n_timed = ctx.timing_end()
    launches = ctx.timing_launches(n_timed)
    set_pipeline(not args.no_pipeline)
    dog.disarm()

    if comm is not None:
        result = comm.fetch()
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
    if variant == 5 and getattr(info, "stream_bytes_rows", 0):
        kernel_label = ("k_fr_filter_ls + k_fr_refine (quantised conservative filter over the ROW STREAM -- rows sorted by length inside every "
                        "8192-row slab, one row per lane, 12 .. 18-bit columns, no row ids; lists of positions -- then exact refine of the "
                        "listed rows through their position words: row id, row start, row length)")

    info = model.info()  # (before the all-rows leg adds its own layout to device_bytes)
    k_mean, k_med, k_min = stats([a for a, _ in launches])
    p_mean, p_med, p_min = stats([b for _, b in launches])
    _, c_med, _ = stats(coll_ms)
    c_max = max(coll_ms) if coll_ms else None

    # per-rank numbers for rank 0's line: a miss of the scaling target can be attributed
    mine = {"rank": rank, "rows": info.n_rows, "nnz": info.nnz, "ms_per_step": elapsed_own * 1e3 / args.steps,
            "serial_ms_per_step": serial_own * 1e3 / serial_steps, "median_kernel_ms": k_med, "median_pass_ms": p_med,
            "collective_ms_median": c_med, "collective_ms_max": c_max,
            "host_enqueue_ms_per_step": t_enqueued * 1e3 / args.steps}
    per_rank = [mine]
    if use_dist:
        box = [None] * world
        dist.all_gather_object(box, mine)
        per_rank = box

    # ---- extra legs, N = 1 only: the pure-bandwidth pass, a new p per step, the survivor cliff ----
    also = None
    if world == 1 and not args.no_also and not args.force_dist:
        also = {}
        base_p = p

        def timed_leg(n, prm_, set_p=None):
            run_steps(3, set_p=set_p, prm_=prm_)
            ctx.synchronize()
            t0_ = time.perf_counter()
            last_ = run_steps(n, set_p=set_p, prm_=prm_)
            ctx.synchronize()
            return time.perf_counter() - t0_, last_

        # (1) new probabilities every step: host log + upload + pass + fetch, nothing drained
        import numpy as np
        rng = np.random.default_rng(7)
        variants_p = [np.clip(base_p * np.exp(0.05 * rng.standard_normal(base_p.shape)), 0.0, 1.0) for _ in range(8)]
        n_newp = max(args.also_steps, 150)  # (a short leg measures the ramp after the previous leg, not the steady state)
        run_steps(20, set_p=lambda i: variants_p[i % 8])
        el_p, last_p = timed_leg(n_newp, prm, set_p=lambda i: variants_p[i % 8])
        logs = [np.log(v) for v in variants_p]

        def set_log(i):
            model.set_log_probabilities(logs[i % 8])
            return None

        def run_log_steps(n):
            inflight, last_ = [], None
            for i in range(n):
                set_log(i)
                inflight.append(ctx.enqueue(model, prm))
                if len(inflight) == depth["n"]:
                    last_ = fetch_ticket(inflight.pop(0))
            while inflight:
                last_ = fetch_ticket(inflight.pop(0))
            return last_

        run_log_steps(20)
        ctx.synchronize()
        t0 = time.perf_counter()
        run_log_steps(n_newp)
        ctx.synchronize()
        el_l = time.perf_counter() - t0
        t0 = time.perf_counter()
        for i in range(8):
            model.set_probabilities(variants_p[i])
        host_set_ms = (time.perf_counter() - t0) * 1e3 / 8
        ctx.synchronize()
        model.set_probabilities(base_p)
        also["new_p"] = {
            "ms_per_step": el_p * 1e3 / n_newp, "steps": n_newp,
            "ms_per_step_log_input": el_l * 1e3 / n_newp,
            "set_probabilities_host_ms": host_set_ms,
            "cut_sets_per_s": info.n_rows * n_newp / el_p,
            "what": "every step: canopy_model_set_probabilities with a DIFFERENT vector (host log of %d values on the library's worker "
                    "threads + upload) -> canopy_quantify_enqueue -> canopy_ticket_fetch of the step two back; wall clock, host time "
                    "included; _log_input: canopy_model_set_log_probabilities instead" % info.n_events,
            "n_kept_last": int(last_p[0].n_kept),
        }

        # (2) SURVEY 8(d): the pure-bandwidth leg, cut_off = 0 (every row through exp / log1p)
        prm0 = canopy_amd.default_params(cut_off=0.0, limit_order=20, kernel=0)
        t0 = time.time()
        ctx.quantify_async(model, prm0)  # builds the layout of the all-rows kernel on first use
        ctx.synchronize()
        t_prep0 = time.time() - t0
        el0, _ = timed_leg(args.also_steps, prm0)
        set_pipeline(False)  # instrumented serial probe, as for the headline
        ctx.timing_begin(args.also_steps)
        for _ in range(args.also_steps):
            ctx.quantify_async(model, prm0)
        ctx.synchronize()
        _, _, n0 = ctx.timing_end()
        l0 = ctx.timing_launches(n0)
        set_pipeline(not args.no_pipeline)
        r0 = canopy_amd.combine_partials([ctx.fetch_partial()])
        cand0 = ctx.last_filter_candidates()
        v0 = ctx.last_kernel_variant()
        info0 = model.info()
        k0_mean, k0_med, k0_min = stats([a for a, _ in l0])
        p0_mean, p0_med, p0_min = stats([b for _, b in l0])
        moved0 = moved_bytes(v0, info0, cand0[0] if cand0 else 0)
        tr0 = traffic_for(v0, info, 0.0)
        also["cut_off_0"] = {
            "ms_per_step": el0 * 1e3 / args.also_steps, "steps": args.also_steps,
            "kernel_ran": v0, "kernel": KERNEL_LABELS.get(v0, "?"),
            "kernel_ms": {"mean": k0_mean, "median": k0_med, "min": k0_min},
            "pass_ms": {"mean": p0_mean, "median": p0_med, "min": p0_min},
            "frac": moved0 / (k0_med * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "moved_bytes_per_launch": moved0,
            "canonical_equivalent_frac": info.algorithmic_bytes / (k0_med * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": tr0,
            "frac_pmc": (tr0 / (k0_med * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr0 else None,
            "cut_sets_per_s": info.n_rows * args.also_steps / el0,
            "n_kept": r0.n_kept, "rare_event": r0.rare_event, "mcub": r0.mcub, "layout_build_s": t_prep0,
        }

        # (2b) K1 as literally stated: every cut set's s_i written out (40 B per cut set: CSR in, 8 B out), the exact CSR-order kernel
        prm_r = canopy_amd.default_params(cut_off=args.cut_off, limit_order=20, flags=_capi.QUANT_ROW_LOGP, kernel=0)
        t0 = time.time()
        ctx.quantify_async(model, prm_r)  # builds the band stream on first use
        ctx.synchronize()
        t_prep_r = time.time() - t0
        for _ in range(3):
            ctx.quantify_async(model, prm_r)
        ctx.synchronize()
        ctx.timing_begin(10)
        for _ in range(10):
            ctx.quantify_async(model, prm_r)
        ctx.synchronize()
        _, _, n_r = ctx.timing_end()
        l_r = ctx.timing_launches(n_r)
        _, kr_med, kr_min = stats([a for a, _ in l_r])
        v_r = ctx.last_kernel_variant()
        bytes_r = info.algorithmic_bytes + 8 * info.n_rows
        info_r = model.info()
        moved_r = moved_bytes(v_r, info_r, 0) + 8 * info.n_rows  # the stream the kernel reads + the s_i it writes
        also["row_logp"] = {
            "kernel_ran": v_r, "kernel": KERNEL_LABELS.get(v_r, "?"), "kernel_ms": {"median": kr_med, "min": kr_min},
            "frac": moved_r / (kr_med * 1e-3) / 1e9 / HBM_PEAK_GBS, "moved_bytes_per_launch": moved_r,
            "canonical_bytes_per_launch": bytes_r, "canonical_frac": bytes_r / (kr_med * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "cut_sets_per_s": info.n_rows / (kr_med * 1e-3), "layout_build_s": t_prep_r,
            "what": "CANOPY_QUANT_ROW_LOGP: the per-cut-set log-probabilities themselves (bit-exact CSR-order fp64 sums) written to HBM "
                    "(SURVEY 8(d): 40 B per cut set canonical); bound by the LDS fp64 adds and the table gathers of the band stream "
                    "(DESIGN.md 3, benchmarks/microbench8.hip)",
        }

        # (3) the survivor cliff: the same pass at cut-offs more and more rows survive
        sel = []
        for co in (1e-9, 1e-10, 1e-11, 1e-12, 1e-13):  # ~2, 3.6, 5.7, 7.9, 10.5 % survivors (past 10 % listed AUTO hands over to FIXED_SUM)
            prm_s = canopy_amd.default_params(cut_off=co, limit_order=20, kernel=0)
            ctx.quantify_async(model, prm_s)
            ctx.fetch_partial()  # AUTO's feedback for this cut-off (it may leave filter/refine for the fixed-point pass)
            n_sel = max(10, args.also_steps // 3)
            el_s, last_s = timed_leg(n_sel, prm_s)
            ctx.quantify_async(model, prm_s)
            ctx.fetch_partial()
            cs = ctx.last_filter_candidates()
            sel.append({"cut_off": co, "survivors_frac": float(last_s[0].n_kept) / info.n_rows,
                        "listed_frac": (cs[0] / cs[1]) if cs and cs[1] else None,
                        "ms_per_step": el_s * 1e3 / n_sel, "kernel_ran": ctx.last_kernel_variant()})
        also["selectivity"] = sel

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        traffic = traffic_for(variant, info, args.cut_off)
        listed = cand[0] if (cand and variant in (5, 6)) else 0
        cand_frac = (cand[0] / cand[1]) if (cand and cand[1] and variant == 5) else None
        moved = moved_bytes(variant, info, listed)
        achieved = moved / (k_med * 1e-3) / 1e9  # bytes the pass moves in the layout it reads / median kernel time
        kept_total = result.n_kept
        line = {
            "metric": "quantified cut-sets/sec", "value": total_rows * args.steps / elapsed, "unit": "cut-sets/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_passes": args.spinup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "synthetic CSR %.3g cut-sets x %.0e basic events in all (%s: %.4g rows on this rank), avg %d nnz/row, "
                            "fp64, rare-event + MCUB, cut_off %g; every step's result fetched by the host%s"
                            % (total_rows, info.n_events,
                               "one GPU" if world == 1 else "%s scaling over %d ranks" % (args.scaling, world),
                               info.n_rows, args.avg_nnz, args.cut_off,
                               "; %.2f %% of the rows pass the conservative filter into the exact pass, %.2f %% survive the cut-off"
                               % (100 * cand_frac, 100.0 * kept_total / max(total_rows if use_dist and not args.force_dist else info.n_rows, 1))
                               if cand_frac is not None else ""),
                "total_rows": total_rows, "rows_this_rank": info.n_rows, "first_row_this_rank": first_row,
                "nnz_this_rank": info.nnz, "n_events": info.n_events,
                "sharding": ("contiguous nnz-balanced row shards (canopy_shard_rows rule), %d rank(s); " % world + collective)
                if use_dist else "single GPU, no collective",
                "seed": SEED, "kernel": args.kernel, "kernel_ran": variant,
                "results_fetched_per_step": bool(fetch_steps and comb is None),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "frac_pmc": (traffic / (k_med * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "canonical_equivalent_gbs": info.algorithmic_bytes / (k_med * 1e-3) / 1e9,
                "canonical_equivalent_frac": info.algorithmic_bytes / (k_med * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "achieved / frac = bytes the pass MUST move in the layout it reads (moved_bytes_per_launch: the stream "
                        "the conservative kernel reads -- stream_bytes_per_launch: the row stream, else the slab stream -- plus two "
                        "128-B lines and a list entry per row handed to the exact pass, from this run's live count) / MEDIAN kernel time of the "
                        "instrumented serial probe / 8 TB/s; traffic / frac_pmc = the PMC-measured HBM bytes of the same workload "
                        "(profiles/traffic*.json, rocprofv3 --pmc passes) over the same time; canonical_equivalent_* = SURVEY 8(d)'s "
                        "32 B-per-cut-set figure of the canonical CSR over the same time: an equivalent rate, not bytes moved (it "
                        "exceeds 1 because the stream is a narrower re-encoding). All hold while few rows pass the filter "
                        "(config.workload, also.selectivity); also.cut_off_0 is the all-rows pass",
                "kernel": kernel_label,
                "moved_bytes_per_launch": moved,
                "algorithmic_bytes_per_launch": info.algorithmic_bytes,
                "bytes_per_cut_set": moved / max(info.n_rows, 1),
                "canonical_bytes_per_cut_set": info.algorithmic_bytes / max(info.n_rows, 1),
                "stream_bytes_per_launch": (getattr(info, "stream_bytes_rows", 0) if variant == 5 else 0) or info.stream_bytes,
                "rows_to_exact_pass": listed,
                "avg_kernel_ms": k_mean, "median_kernel_ms": k_med, "min_kernel_ms": k_min,
                "avg_pass_ms": p_mean, "median_pass_ms": p_med, "min_pass_ms": p_min, "launches_timed": n_timed,
                "serial_pass_ms_note": "pass = row kernels + final reduce (+ all-reduce at N > 1), one pass at a time",
                "host_enqueue_ms_per_step": t_enqueued * 1e3 / args.steps,
                "collective_ms": {"median": c_med, "max": c_max, "passes": len(coll_ms)} if coll_ms else None,
                "pipelined": not args.no_pipeline, "lanes": lanes if not args.no_pipeline else 1, "pipeline_note": pipeline_note or None,
                "verification": verify,
                "probe": "kernel / pass ms: %d instrumented serial passes after the timed region (a pass that runs alone reads the row stream "
                         "with non-temporal loads -- its exact pass then finds its working set in cache: 37 instead of 45 us --, the "
                         "pipelined steps of the timed region read it with ordinary loads, which is faster when two passes share the "
                         "memory system; the filter itself takes the same time either way)" % n_timed,
            },
            "serial": {"ms_per_step": serial_elapsed * 1e3 / serial_steps, "steps": serial_steps,
                       "value": total_rows * serial_steps / serial_elapsed,
                       "what": "the same steps with pipelining off: one pass at a time, its result fetched before the next is "
                               "enqueued -- what a caller that never overlaps passes sees"},
            "per_rank": per_rank,
            "result": {"rare_event": result.rare_event, "mcub": result.mcub, "n_kept": result.n_kept,
                       "argmax_row": result.argmax_row},
            "model_build_s": t_build, "model_prepare_s": t_prepare, "device_bytes": info.device_bytes,
        }
        if last_parts is not None:
            lr = canopy_amd.combine_partials(list(last_parts))
            line["result"]["last_timed_step_n_kept"] = lr.n_kept
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
        dog.arm("teardown")
        dist.barrier()
        dist.destroy_process_group()
        dog.disarm()


if __name__ == "__main__":
    main()
