"""K5 inside the library (canopy_comm_*: the RCCL all-reduce issued by libcanopy.so), the strong-scaling shard
bounds, captured-pass invalidation rules, and the full-size headline config against the oracle (-m gpu)."""
import ctypes as C
import os

import numpy as np
import pytest

SEED = 20250704
TOL = 1e-12


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def _threads():
    return max(1, min(len(os.sched_getaffinity(0)), 64))


@pytest.mark.gpu
def test_library_collective_world_1(gpu_ctx, oracle):
    """canopy_comm_init_rank at world size 1 (all a one-GPU box can host): the pass ends with a real ncclAllReduce
    issued by the library; results equal the detached pass bit for bit; canopy_quantify returns the combined record;
    detaching restores single-shard behaviour."""
    import canopy_amd
    from canopy_amd import _capi
    n, e = 300_000, 5000
    row_ptr, col = oracle.synth_csr(SEED, n, e, 6, n_threads=4)
    p = oracle.synth_probabilities(SEED, e)
    want = oracle.quantify(row_ptr, col, p, cut_off=1e-10, row_offset=777)
    model = gpu_ctx.upload(row_ptr, col, e, row_offset=777)
    model.set_probabilities(p)
    prm = canopy_amd.default_params(cut_off=1e-10)
    plain = gpu_ctx.quantify(model, prm)
    comm = canopy_amd.Comm(gpu_ctx, canopy_amd.Comm.unique_id(), 1, 0)
    try:
        assert comm.rccl_version() >= 20000
        got = gpu_ctx.quantify(model, prm)  # combined over the (one) rank
        assert (got.rare_event, got.mcub, got.n_kept, got.argmax_row, got.max_logp) == \
               (plain.rare_event, plain.mcub, plain.n_kept, plain.argmax_row, plain.max_logp)
        gpu_ctx.quantify_async(model, prm)
        again = comm.fetch()
        parts = comm.partials()
        assert len(parts) == 1 and int(parts[0].n_kept) == int(want.n_kept)
        assert (again.rare_event, again.n_kept, again.argmax_row) == (got.rare_event, got.n_kept, got.argmax_row)
        assert got.n_kept == int(want.n_kept) and got.argmax_row == int(want.argmax_row) and got.max_logp == want.max_logp
        assert rel(got.rare_event, want.rare_event) < TOL and rel(got.mcub, want.mcub) < TOL
        # a captured pass would silently drop the collective: refused while a communicator is attached
        with pytest.raises(_capi.CanopyError) as err:
            gpu_ctx.graph_capture(model, prm)
        assert err.value.status == _capi.ERR_UNSUPPORTED
        # a second communicator on the same context is a call-order error
        with pytest.raises(_capi.CanopyError) as err:
            canopy_amd.Comm(gpu_ctx, canopy_amd.Comm.unique_id(), 1, 0)
        assert err.value.status == _capi.ERR_STATE
    finally:
        comm.close()
    after = gpu_ctx.quantify(model, prm)
    assert (after.rare_event, after.n_kept) == (plain.rare_event, plain.n_kept)
    model.close()


@pytest.mark.gpu
def test_library_collective_single_process_shape(gpu_ctx, oracle):
    """canopy_comm_init_all + canopy_comm_quantify_all_async (one process driving N devices, collectives grouped):
    exercised with the one device this box has."""
    import canopy_amd
    from canopy_amd import _capi
    lib = _capi.lib()
    n, e = 120_000, 900
    model = gpu_ctx.synthetic(SEED, n, e, 4)
    p = canopy_amd.synthetic_probabilities(SEED, e)
    model.set_probabilities(p)
    prm = canopy_amd.default_params()
    plain = gpu_ctx.quantify(model, prm)
    ctxs = (C.c_void_p * 1)(gpu_ctx.handle)
    comms = (C.c_void_p * 1)()
    _capi.check(lib.canopy_comm_init_all(ctxs, 1, comms), "canopy_comm_init_all", gpu_ctx.handle)
    try:
        models = (C.c_void_p * 1)(model.handle)
        _capi.check(lib.canopy_comm_quantify_all_async(comms, models, 1, C.byref(prm)), "canopy_comm_quantify_all_async",
                    gpu_ctx.handle)
        res = _capi.QuantResult()
        _capi.check(lib.canopy_comm_fetch(comms[0], 0, C.byref(res), None), "canopy_comm_fetch", gpu_ctx.handle)
        assert (res.rare_event, res.mcub, res.n_kept, res.argmax_row) == (plain.rare_event, plain.mcub, plain.n_kept,
                                                                          plain.argmax_row)
    finally:
        lib.canopy_comm_free(comms[0])
    model.close()


@pytest.mark.gpu
def test_monte_carlo_collective_world_1(gpu_ctx):
    import canopy_amd
    dag = {"n_events": 4, "gate_type": [1, 0], "gate_min": [0, 0], "arg_ptr": [0, 2, 4], "args": [0, 1, 4, 2], "top": 5}
    d = canopy_amd.quantifier.Dag(gpu_ctx, dag)
    d.set_probabilities(np.array([0.3, 0.2, 0.5, 0.9]))
    alone = d.run(11, 1_000_000)
    comm = canopy_amd.Comm(gpu_ctx, canopy_amd.Comm.unique_id(), 1, 0)
    try:
        d.run_async(11, 1_000_000)
        both = comm.mc_fetch()
        assert both.hits == alone.hits and both.num_trials == 1_000_000
    finally:
        comm.close()
    d.close()


@pytest.mark.gpu
def test_synthetic_shard_bounds_match_the_host_rule(gpu_ctx, oracle):
    import canopy_amd
    n = 300_000
    row_ptr, _ = oracle.synth_csr(SEED, n, 1000, 6, n_threads=4)
    for shards in (1, 2, 3, 8):
        assert np.array_equal(canopy_amd.synthetic_shard_bounds(gpu_ctx, SEED, n, 6, shards), oracle.shard_rows(row_ptr, shards))
    # the shards of the family generated one by one are the whole, record for record
    p = canopy_amd.synthetic_probabilities(SEED, 1000)
    whole = gpu_ctx.synthetic(SEED, n, 1000, 6)
    whole.set_probabilities(p)
    want = gpu_ctx.quantify(whole, canopy_amd.default_params(cut_off=1e-10))
    whole.close()
    b = canopy_amd.synthetic_shard_bounds(gpu_ctx, SEED, n, 6, 8)
    parts = []
    for k in range(8):
        m = gpu_ctx.synthetic(SEED, int(b[k + 1] - b[k]), 1000, 6, first_row=int(b[k]))
        m.set_probabilities(p)
        gpu_ctx.quantify_async(m, canopy_amd.default_params(cut_off=1e-10))
        parts.append(gpu_ctx.fetch_partial())
        m.close()
    got = canopy_amd.combine_partials(parts)
    assert (got.n_kept, got.argmax_row, got.max_logp) == (want.n_kept, want.argmax_row, want.max_logp)
    assert rel(got.rare_event, want.rare_event) < TOL and rel(got.mcub, want.mcub) < TOL


@pytest.mark.gpu
def test_captured_pass_and_eager_passes_with_other_cut_offs(gpu_ctx, oracle):
    """A captured FILTER_REFINE pass rewrites the model's quantised filter table with ITS cut-off on every replay;
    eager passes with another cut-off in between must not trust the cached table (the filter would stop
    being conservative and drop survivors silently)."""
    import canopy_amd
    from canopy_amd import _capi
    n, e = 1_200_000, 30_000
    row_ptr, col = oracle.synth_csr(SEED, n, e, 6, n_threads=_threads())
    p = oracle.synth_probabilities(SEED, e)
    want8 = oracle.quantify(row_ptr, col, p, cut_off=1e-8, n_threads=_threads(), want_rows=False)
    want4 = oracle.quantify(row_ptr, col, p, cut_off=1e-4, n_threads=_threads(), want_rows=False)
    assert want8.n_kept > want4.n_kept
    model = gpu_ctx.upload(row_ptr, col, e)
    model.set_probabilities(p)
    prm4 = canopy_amd.default_params(cut_off=1e-4, kernel=_capi.KERNEL_FILTER_REFINE)
    prm8 = canopy_amd.default_params(cut_off=1e-8, kernel=_capi.KERNEL_FILTER_REFINE)
    g = gpu_ctx.graph_capture(model, prm4)
    try:
        for _ in range(2):
            a = gpu_ctx.quantify(model, prm8)
            assert a.n_kept == int(want8.n_kept) and a.argmax_row == int(want8.argmax_row)
            gpu_ctx.graph_launch(g)
            b = canopy_amd.combine_partials([gpu_ctx.fetch_partial()])
            assert b.n_kept == int(want4.n_kept) and rel(b.rare_event, want4.rare_event) < TOL
            c = gpu_ctx.quantify(model, prm8)  # the eager pass right after a replay: table must be rebuilt
            assert c.n_kept == int(want8.n_kept) and rel(c.rare_event, want8.rare_event) < TOL
    finally:
        gpu_ctx.graph_free(g)
        model.close()


@pytest.mark.gpu
def test_captured_pass_is_refused_after_scratch_reallocation(canopy_lib):
    """A graph holds raw pointers to the context's scratch: once a later pass made the context replace any of it,
    canopy_graph_launch returns CANOPY_ERR_STATE instead of replaying into freed memory."""
    import canopy_amd
    from canopy_amd import _capi
    with canopy_amd.Context(device=0) as ctx:
        pa = canopy_amd.synthetic_probabilities(SEED, 900)
        a = ctx.synthetic(SEED, 50_000, 900, 4)
        a.set_probabilities(pa)
        prm = canopy_amd.default_params(flags=_capi.QUANT_KEEP_MASK)
        g = ctx.graph_capture(a, prm)
        ctx.graph_launch(g)
        first = canopy_amd.combine_partials([ctx.fetch_partial()])
        b = ctx.synthetic(SEED, 400_000, 900, 4)  # larger: keep mask and partial records are reallocated
        b.set_probabilities(pa)
        ctx.quantify(b, prm)
        with pytest.raises(_capi.CanopyError) as err:
            ctx.graph_launch(g)
        assert err.value.status == _capi.ERR_STATE
        ctx.graph_free(g)
        g = ctx.graph_capture(a, prm)  # capturing again is the documented remedy
        ctx.graph_launch(g)
        again = canopy_amd.combine_partials([ctx.fetch_partial()])
        assert (again.rare_event, again.n_kept) == (first.rare_event, first.n_kept)
        ctx.graph_free(g)
        a.close()
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cut_off", [1e-8, 0.0])
def test_config3_full_size_against_the_oracle(gpu_ctx, oracle, cut_off):
    """BASELINE config 3 at FULL size (1e8 cut sets x 1e5 events, avg 6 nnz/row), the pass AUTO picks, against the
    oracle on the box's host threads: integers exact, sums <= 1e-12. cut_off 1e-8 = the headline (filter/refine);
    cut_off 0 = the all-rows path."""
    import canopy_amd
    n, e, avg = 100_000_000, 100_000, 6
    p = oracle.synth_probabilities(SEED, e)
    model = gpu_ctx.synthetic(SEED, n, e, avg)
    model.set_probabilities(p)
    got = gpu_ctx.quantify(model, canopy_amd.default_params(cut_off=cut_off, limit_order=20))
    variant = gpu_ctx.last_kernel_variant()
    model.close()
    row_ptr, col = oracle.synth_csr(SEED, n, e, avg, n_threads=_threads())
    want = oracle.quantify(row_ptr, col, p, cut_off=cut_off, limit_order=20, n_threads=_threads(), want_rows=False)
    del row_ptr, col
    assert got.n_kept == int(want.n_kept), variant
    assert got.argmax_row == int(want.argmax_row) and got.max_logp == want.max_logp, variant
    assert rel(got.rare_event, want.rare_event) < TOL and rel(got.mcub, want.mcub) < TOL, variant
    if cut_off == 0.0:
        assert got.n_kept == n


@pytest.mark.gpu
def test_pipelined_passes_equal_serial_passes(canopy_lib, oracle):
    """canopy_ctx_set_pipeline: consecutive sum-only passes rotate through three lanes and overlap on the device. What
    comes back is the record of the LAST enqueued pass, bit for bit what the serial context gives -- also when the
    probabilities change between passes, when a keep-mask pass (un-pipelined) is interleaved, and for the all-rows
    fixed-point pass."""
    import canopy_amd
    from canopy_amd import _capi
    n, e = 1_300_000, 30_000
    row_ptr, col = oracle.synth_csr(SEED, n, e, 6, n_threads=_threads())
    p = oracle.synth_probabilities(SEED, e)
    scales = [1.0, 0.5, 2.0, 1.5, 0.25, 3.0, 1.0, 0.7]
    out = {}
    for pipe in (True, False):
        with canopy_amd.Context(device=0) as ctx:
            ctx.set_pipeline(pipe)
            model = ctx.upload(row_ptr, col, e)
            recs = []
            for k, sc in enumerate(scales):
                model.set_probabilities(np.minimum(1.0, p * sc))
                for _ in range(1 + k % 3):  # several passes in flight per probability vector
                    ctx.quantify_async(model, canopy_amd.default_params())
                r = canopy_amd.combine_partials([ctx.fetch_partial()])
                recs.append((r.rare_event, r.mcub, r.n_kept, r.argmax_row, r.max_logp))
                if k == 3:  # an un-pipelined pass in between: exact keep mask of this vector
                    ctx.quantify_async(model, canopy_amd.default_params(flags=_capi.QUANT_KEEP_MASK))
                    mask = ctx.fetch_keep_mask(model)
                    assert int(np.unpackbits(mask.view(np.uint8)).sum()) == r.n_kept
                if k == 5:  # the all-rows pass rides the lanes too
                    for _ in range(4):
                        ctx.quantify_async(model, canopy_amd.default_params(cut_off=0.0))
                    r0 = canopy_amd.combine_partials([ctx.fetch_partial()])
                    assert r0.n_kept == n and ctx.last_kernel_variant() == _capi.KERNEL_FIXED_SUM
                    recs.append((r0.rare_event, r0.mcub, r0.n_kept, r0.argmax_row, r0.max_logp))
            out[pipe] = recs
            model.close()
    assert out[True] == out[False]
    want = oracle.quantify(row_ptr, col, np.minimum(1.0, p * scales[-1]), n_threads=_threads(), want_rows=False)
    last = out[True][-1]
    assert last[2] == int(want.n_kept) and last[3] == int(want.argmax_row) and last[4] == want.max_logp
    assert rel(last[0], want.rare_event) < TOL and rel(last[1], want.mcub) < TOL
