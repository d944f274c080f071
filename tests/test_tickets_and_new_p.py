"""Round 3: one result per pass without serialising the passes (tickets), a new probability vector per pass without
draining the lanes (the model's ring of probability sets), caller-supplied log-probabilities, the communicator's
ordering modes, and the 8 nnz-balanced shards of the full 1e8-row model through canopy_comm (-m gpu).

Reference anchor: every analysis starts from fresh BasicEvent::p() values (src/mef/openpsa/event/event.h:93-96); the
reference holds no implementation and no fixture for this path (parity unpinned): the checker is oracle/."""
import os

import numpy as np
import pytest

SEED = 20250704
TOL = 1e-12


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def _threads():
    return max(1, min(len(os.sched_getaffinity(0)), 64))


def _same(a, b):
    return (a.rare_hi, a.rare_lo, a.log1p_hi, a.log1p_lo, a.n_kept, a.n_certain, a.max_logp, a.argmax_row) == \
           (b.rare_hi, b.rare_lo, b.log1p_hi, b.log1p_lo, b.n_kept, b.n_certain, b.max_logp, b.argmax_row)


def _perturbed(p, k):
    """A different probability vector per step, deterministic."""
    rng = np.random.default_rng(1000 + k)
    return np.clip(p * np.exp(0.3 * rng.standard_normal(p.shape)), 0.0, 1.0)


@pytest.mark.gpu
def test_tickets_new_p_every_pass_equals_serial_passes(gpu_ctx, oracle):
    """12 steps of (new p, ticketed pipelined pass), up to three in flight, against the same 12 steps run strictly one
    after the other with a blocking fetch -- records bit-identical -- and three of them against the oracle."""
    import canopy_amd
    n, e = 3_000_000, 20_000
    model = gpu_ctx.synthetic(SEED, n, e, 6)
    base = canopy_amd.synthetic_probabilities(SEED, e)
    prm = canopy_amd.default_params(cut_off=1e-9)
    ps = [_perturbed(base, k) for k in range(12)]
    # serial: set, pass, blocking fetch
    gpu_ctx.set_pipeline(False)
    serial = []
    for p in ps:
        model.set_probabilities(p)
        gpu_ctx.quantify_async(model, prm)
        serial.append(gpu_ctx.fetch_partial())
    assert gpu_ctx.last_kernel_variant() == 5
    # pipelined: up to three (new p, pass) in flight, fetched by ticket in enqueue order
    gpu_ctx.set_pipeline(True)
    got, inflight = [], []
    for p in ps:
        model.set_probabilities(p)
        inflight.append(gpu_ctx.enqueue(model, prm))
        if len(inflight) == 3:
            got.append(gpu_ctx.ticket_fetch(inflight.pop(0)))
    while inflight:
        got.append(gpu_ctx.ticket_fetch(inflight.pop(0)))
    assert len(got) == 12
    for a, b in zip(got, serial):
        assert _same(a, b)
    row_ptr, col = model.download()
    for k in (0, 5, 11):
        want = oracle.quantify(row_ptr, col, ps[k], cut_off=1e-9, n_threads=_threads(), want_rows=False)
        r = canopy_amd.combine_partials([got[k]])
        assert r.n_kept == int(want.n_kept) and r.argmax_row == int(want.argmax_row) and r.max_logp == want.max_logp
        assert rel(r.rare_event, want.rare_event) < TOL and rel(r.mcub, want.mcub) < TOL
    model.close()


@pytest.mark.gpu
def test_ticket_rules(gpu_ctx):
    """A ticket stays fetchable (any order, repeatedly) until 64 later passes have been enqueued, then expires; ticket_done
    polls; the legacy fetch still delivers the last pass; un-pipelined passes in a row keep their records too."""
    import canopy_amd
    from canopy_amd import _capi
    n, e = 1_200_000, 5000
    model = gpu_ctx.synthetic(SEED, n, e, 6)
    model.set_probabilities(canopy_amd.synthetic_probabilities(SEED, e))
    prm = canopy_amd.default_params(cut_off=1e-9)
    gpu_ctx.set_pipeline(True)
    t = [gpu_ctx.enqueue(model, prm) for _ in range(9)]
    assert len(set(t)) == 9 and all(x > 0 for x in t)
    recs = [gpu_ctx.ticket_fetch(x) for x in reversed(t)]  # any order, lanes long since reused
    assert all(_same(recs[0], r) for r in recs[1:])
    assert _same(gpu_ctx.ticket_fetch(t[0]), recs[0])  # and again
    assert gpu_ctx.ticket_done(t[8])
    last = gpu_ctx.fetch_partial()
    assert _same(last, recs[-1])
    with pytest.raises(_capi.CanopyError):
        gpu_ctx.ticket_fetch(987654321)
    # un-pipelined variants carry tickets too (they share lane 0): three in a row, each with a different cut-off
    cuts = (1e-9, 1e-7, 0.0)
    tk = [gpu_ctx.enqueue(model, canopy_amd.default_params(cut_off=c, kernel=_capi.KERNEL_ROWS)) for c in cuts]
    rr = [gpu_ctx.ticket_fetch(x) for x in tk]
    assert rr[0].n_kept == recs[0].n_kept and rr[0].argmax_row == recs[0].argmax_row and rr[0].max_logp == recs[0].max_logp
    assert rr[0].n_kept > rr[1].n_kept > 0 and rr[2].n_kept == n
    # 64 later passes (un-ticketed ones count) expire a ticket
    for _ in range(64 - len(tk) - 1):
        gpu_ctx.quantify_async(model, prm)
    assert gpu_ctx.ticket_fetch(t[8]).n_kept == recs[0].n_kept  # t[8] + 1 (legacy fetch: no pass) ... still inside the ring
    gpu_ctx.quantify_async(model, prm)
    gpu_ctx.quantify_async(model, prm)
    with pytest.raises(_capi.CanopyError) as err:
        gpu_ctx.ticket_fetch(t[8])
    assert err.value.status == _capi.ERR_STATE
    gpu_ctx.synchronize()
    model.close()


@pytest.mark.gpu
def test_set_log_probabilities_and_validation(gpu_ctx, oracle):
    import canopy_amd
    from canopy_amd import _capi
    n, e = 200_000, 3000
    row_ptr, col = oracle.synth_csr(SEED, n, e, 6, n_threads=4)
    p = oracle.synth_probabilities(SEED, e)
    p[7] = 0.0   # log -> -inf
    p[11] = 1.0  # log -> 0
    model = gpu_ctx.upload(row_ptr, col, e)
    prm = canopy_amd.default_params(cut_off=1e-10, flags=_capi.QUANT_ROW_LOGP)
    model.set_probabilities(p)
    a = gpu_ctx.quantify(model, prm)
    rows_a = gpu_ctx.fetch_row_logp(model)
    with np.errstate(divide="ignore"):
        model.set_log_probabilities(np.log(p))  # numpy's log is glibc's on this platform for these values? compare through the oracle
    b = gpu_ctx.quantify(model, prm)
    rows_b = gpu_ctx.fetch_row_logp(model)
    want = oracle.quantify(row_ptr, col, p, cut_off=1e-10)
    assert np.array_equal(rows_a, want.row_logp)
    assert a.n_kept == int(want.n_kept) and a.argmax_row == int(want.argmax_row)
    # the log route is the same chain of additions on caller-supplied logarithms (numpy's log may differ from glibc's by an ulp)
    with np.errstate(divide="ignore"):
        lp = np.log(p)
    for i in range(0, n, 997):
        seq = 0.0
        first = True
        for c in col[row_ptr[i]:row_ptr[i + 1]]:
            seq = lp[c] if first else seq + lp[c]
            first = False
        assert rows_b[i] == seq or (np.isinf(rows_b[i]) and np.isinf(seq))
    assert abs(b.n_kept - a.n_kept) <= 2  # (an ulp of log can move a row across the cut-off; the sums agree far inside 1e-9)
    assert rel(b.rare_event, a.rare_event) < 1e-9
    for bad in (np.full(e, 0.1), np.full(e, np.nan)):
        with pytest.raises(_capi.CanopyError) as err:
            model.set_log_probabilities(bad)
        assert err.value.status == _capi.ERR_INVALID_ARGUMENT
    bad_p = p.copy()
    bad_p[5] = 1.5
    with pytest.raises(_capi.CanopyError) as err:
        model.set_probabilities(bad_p)
    assert err.value.status == _capi.ERR_INVALID_ARGUMENT
    # a rejected vector leaves the model's current probabilities in place
    c = gpu_ctx.quantify(model, prm)
    assert (c.rare_event, c.n_kept) == (b.rare_event, b.n_kept)
    model.close()


@pytest.mark.gpu
def test_probability_ring_wraps_without_passes_and_with_row_kernels(gpu_ctx, oracle):
    """More set_probabilities calls than the ring has sets, with and without passes in between, across kernel variants
    (lane 0 passes have no completion event: the writer waits for the stream instead)."""
    import canopy_amd
    from canopy_amd import _capi
    n, e = 150_000, 1500
    row_ptr, col = oracle.synth_csr(SEED, n, e, 4, n_threads=4)
    base = oracle.synth_probabilities(SEED, e)
    model = gpu_ctx.upload(row_ptr, col, e)
    for k in range(9):  # no pass in between: only the last one counts
        model.set_probabilities(_perturbed(base, k))
    prm = canopy_amd.default_params(cut_off=1e-10, kernel=_capi.KERNEL_ROWS)
    got = gpu_ctx.quantify(model, prm)
    want = oracle.quantify(row_ptr, col, _perturbed(base, 8), cut_off=1e-10, want_rows=False)
    assert got.n_kept == int(want.n_kept) and got.argmax_row == int(want.argmax_row) and rel(got.rare_event, want.rare_event) < TOL
    recs = []
    for k in range(10):  # un-pipelined passes queue up on lane 0 while the ring wraps
        model.set_probabilities(_perturbed(base, 20 + k))
        gpu_ctx.quantify_async(model, canopy_amd.default_params(cut_off=1e-10, kernel=_capi.KERNEL_ROWS_LDS if k % 2 else _capi.KERNEL_ROWS))
        if k in (3, 9):
            recs.append((k, gpu_ctx.fetch_partial()))
    for k, rec in recs:
        want = oracle.quantify(row_ptr, col, _perturbed(base, 20 + k), cut_off=1e-10, want_rows=False)
        r = canopy_amd.combine_partials([rec])
        assert r.n_kept == int(want.n_kept) and r.argmax_row == int(want.argmax_row) and rel(r.rare_event, want.rare_event) < TOL
    model.close()


@pytest.mark.gpu
def test_two_contexts_share_one_model(canopy_lib, oracle):
    """Quantification takes the model as const: two contexts (two host threads) run passes on one model at once; the
    lazily built slab stream and the 8-bit table are built once, under the model's mutex."""
    import threading
    import canopy_amd
    n, e = 1_500_000, 8000
    a = canopy_amd.Context(device=0)
    b = canopy_amd.Context(device=0)
    model = a.synthetic(SEED, n, e, 6)
    model.set_probabilities(canopy_amd.synthetic_probabilities(SEED, e))
    a.synchronize()
    prm = canopy_amd.default_params(cut_off=1e-9)
    out = {}

    def run(name, ctx):
        recs = []
        for _ in range(6):
            recs.append(ctx.ticket_fetch(ctx.enqueue(model, prm)))
        out[name] = recs

    ta, tb = threading.Thread(target=run, args=("a", a)), threading.Thread(target=run, args=("b", b))
    ta.start(); tb.start(); ta.join(); tb.join()
    ref = out["a"][0]
    assert all(_same(ref, r) for r in out["a"] + out["b"])
    # a writer now has readers of two contexts to wait for (device-wide wait), then both contexts read the new set
    p2 = _perturbed(canopy_amd.synthetic_probabilities(SEED, e), 3)
    model.set_probabilities(p2)
    ra, rb = a.ticket_fetch(a.enqueue(model, prm)), b.ticket_fetch(b.enqueue(model, prm))
    assert _same(ra, rb) and not _same(ra, ref)
    row_ptr, col = model.download()
    want = oracle.quantify(row_ptr, col, p2, cut_off=1e-9, n_threads=_threads(), want_rows=False)
    r = canopy_amd.combine_partials([ra])
    assert r.n_kept == int(want.n_kept) and r.argmax_row == int(want.argmax_row) and rel(r.rare_event, want.rare_event) < TOL
    model.close()
    b.close()
    a.close()


@pytest.mark.gpu
def test_comm_modes_and_comm_tickets_world_1(gpu_ctx):
    """The three orderings of consecutive collectives give identical records; canopy_comm_ticket_fetch delivers every pass of
    a pipelined burst; a context freed before its communicator takes it along (no use-after-free)."""
    import canopy_amd
    from canopy_amd import _capi
    n, e = 2_000_000, 10_000
    model = gpu_ctx.synthetic(SEED, n, e, 6)
    base = canopy_amd.synthetic_probabilities(SEED, e)
    prm = canopy_amd.default_params(cut_off=1e-9)
    model.set_probabilities(base)
    plain = gpu_ctx.ticket_fetch(gpu_ctx.enqueue(model, prm))
    comm = canopy_amd.Comm(gpu_ctx, canopy_amd.Comm.unique_id(), 1, 0)
    try:
        for mode in (_capi.COMM_CHAINED, _capi.COMM_UNORDERED, _capi.COMM_OWN_STREAM, _capi.COMM_CHAINED):
            comm.set_mode(mode)
            tickets = [gpu_ctx.enqueue(model, prm) for _ in range(3)]
            for t in tickets:
                parts = comm.ticket_partials(t)
                assert len(parts) == 1 and _same(parts[0], plain)
                r = comm.ticket_fetch(t)
                assert r.n_kept == int(plain.n_kept)
            for k in range(7):  # un-fetched burst, then the legacy fetch of the last pass
                gpu_ctx.quantify_async(model, prm)
            assert _same(comm.partials()[0], plain)
        # new p per pass through the communicator
        want = []
        for k in range(5):
            model.set_probabilities(_perturbed(base, k))
            want.append(gpu_ctx.enqueue(model, prm))
            if k >= 2:
                comm.ticket_partials(want[k - 2])
    finally:
        comm.close()
    model.close()
    # context freed first: the communicator is detached (every call on it fails cleanly), its handle is still the caller's to free
    ctx2 = canopy_amd.Context(device=0)
    comm2 = canopy_amd.Comm(ctx2, canopy_amd.Comm.unique_id(), 1, 0)
    ctx2.close()
    res = _capi.QuantResult()
    import ctypes as C
    assert _capi.lib().canopy_comm_fetch(comm2.handle, 0, C.byref(res), None) == _capi.ERR_INVALID_ARGUMENT
    assert _capi.lib().canopy_comm_set_mode(comm2.handle, _capi.COMM_CHAINED) == _capi.ERR_INVALID_ARGUMENT
    comm2.close()  # releases the RCCL communicator and the buffers; no context is touched


@pytest.mark.gpu
def test_config4_eight_shards_of_the_full_model_through_canopy_comm(gpu_ctx, oracle):
    """BASELINE config 4 on the one GPU this box has: the 8 nnz-balanced shards of the SAME 1e8-row model
    (canopy_synthetic_shard_bounds), one after the other, each through a world-1 canopy_comm (final reduce -> slot buffer
    -> ncclAllReduce -> published records), folded with canopy_combine_partials in shard order, against the oracle over
    the whole model: integers exact, sums within 1e-12."""
    import canopy_amd
    total, e, avg = 100_000_000, 100_000, 6
    p = canopy_amd.synthetic_probabilities(SEED, e)
    bounds = canopy_amd.synthetic_shard_bounds(gpu_ctx, SEED, total, avg, 8)
    assert bounds[0] == 0 and bounds[8] == total and all(bounds[k] < bounds[k + 1] for k in range(8))
    prm = canopy_amd.default_params(cut_off=1e-8)
    parts, nnz = [], []
    comm = canopy_amd.Comm(gpu_ctx, canopy_amd.Comm.unique_id(), 1, 0)
    try:
        for k in range(8):
            m = gpu_ctx.synthetic(SEED, int(bounds[k + 1] - bounds[k]), e, avg, first_row=int(bounds[k]))
            m.set_probabilities(p)
            t = [gpu_ctx.enqueue(m, prm) for _ in range(3)]  # a pipelined burst per shard: every lane, every record
            recs = [comm.ticket_partials(x)[0] for x in t]
            assert _same(recs[0], recs[1]) and _same(recs[0], recs[2])
            assert gpu_ctx.last_kernel_variant() == 5
            parts.append(recs[0])
            nnz.append(m.info().nnz)
            m.close()
    finally:
        comm.close()
    assert max(nnz) - min(nnz) <= 2 * (2 * avg - 2)  # nnz-balanced: within a row of each other
    got = canopy_amd.combine_partials(parts)
    # the oracle over the whole model, shard by shard (memory), folded the same way by ITS combine
    o_parts = []
    for k in range(8):
        first, rows = int(bounds[k]), int(bounds[k + 1] - bounds[k])
        rp, col = oracle.synth_csr(SEED, rows, e, avg, n_threads=_threads(), first_row=first)
        o_parts.append(oracle.quantify(rp, col, p, cut_off=1e-8, n_threads=_threads(), want_rows=False, row_offset=first))
        del rp, col
    want = oracle.combine([q.res for q in o_parts])  # the oracle's own fold of its shard results
    assert got.n_kept == int(want.n_kept)
    assert got.argmax_row == int(want.argmax_row) and got.max_logp == want.max_logp
    assert rel(got.rare_event, want.rare_event) < TOL and rel(got.mcub, want.mcub) < TOL
    for k in range(8):
        assert int(parts[k].n_kept) == int(o_parts[k].n_kept)
        assert rel(parts[k].rare_hi + parts[k].rare_lo, o_parts[k].rare_event) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [2, 4])
def test_lane_count_is_a_runtime_knob(gpu_ctx, lanes):
    """canopy_ctx_set_pipeline(n): n lanes rotate, n tickets stay valid, a new probability vector per pass still reads the
    set it was enqueued with (the ring of sets is larger than the largest rotation); records equal the serial ones."""
    import canopy_amd
    n, e = 1_500_000, 12_000
    model = gpu_ctx.synthetic(SEED, n, e, 6)
    base = canopy_amd.synthetic_probabilities(SEED, e)
    prm = canopy_amd.default_params(cut_off=1e-9)
    ps = [_perturbed(base, 40 + k) for k in range(2 * lanes + 3)]
    gpu_ctx.set_pipeline(0)
    serial = []
    for p in ps:
        model.set_probabilities(p)
        serial.append(gpu_ctx.ticket_fetch(gpu_ctx.enqueue(model, prm)))
    gpu_ctx.set_pipeline(lanes)
    try:
        tickets, got = [], []
        for p in ps:
            model.set_probabilities(p)
            tickets.append(gpu_ctx.enqueue(model, prm))
            if len(tickets) == lanes:
                got.append(gpu_ctx.ticket_fetch(tickets.pop(0)))
        while tickets:
            got.append(gpu_ctx.ticket_fetch(tickets.pop(0)))
        assert len(got) == len(serial) and all(_same(a, b) for a, b in zip(got, serial))
        # tickets outlive the reuse of their lane
        t = [gpu_ctx.enqueue(model, prm) for _ in range(2 * lanes + 1)]
        for x in t:
            assert _same(gpu_ctx.ticket_fetch(x), serial[-1])
    finally:
        gpu_ctx.set_pipeline(1)  # the default rotation
    model.close()
