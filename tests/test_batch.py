"""N4 -- batched quantification (K probability vectors against one model) through the C ABI.
Definition under test: trial k of canopy_quantify_batch == canopy_quantify after
canopy_model_set_probabilities(p_trials[k]) == the CPU oracle on p_trials[k]: integer outputs and max_logp
bit-exact, rare-event / MCUB <= 1e-12 relative. Parity is unpinned by the reference (it has no
uncertainty-analysis code: only the knobs at src/mef/openpsa/settings.h:118-130,281-295)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 20250704
TOL = 1e-12


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def trials_from(p, n_trials, sigma, seed):
    """Log-normal multipliers around the point values (error-factor style), clipped to [0, 1]."""
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n_trials, p.size))
    return np.minimum(1.0, p[None, :] * np.exp(sigma * z))


def same(got, want):
    assert got.n_kept == int(want.n_kept)
    assert got.argmax_row == int(want.argmax_row)
    if int(want.n_kept):
        assert got.max_logp == want.max_logp
    if math.isfinite(want.rare_event) and want.rare_event > 0:
        assert rel(got.rare_event, want.rare_event) < TOL, (got.rare_event, want.rare_event)
        assert rel(got.mcub, want.mcub) < TOL, (got.mcub, want.mcub)
    else:
        assert got.rare_event == want.rare_event and got.mcub == want.mcub


@pytest.mark.parametrize("n_rows,n_events,avg,n_trials,cut_off,limit_order", [
    (30_000, 500, 4, 13, 1e-8, 20),     # K not a multiple of 8
    (30_000, 500, 4, 3, 0.0, 20),       # no cut-off: every row is a candidate
    (50_000, 3000, 6, 8, 1e-10, 5),     # limit_order bites
    (1, 10, 4, 1, 1e-8, 20),
])
def test_batch_matches_oracle_and_single_passes(gpu_ctx, oracle, n_rows, n_events, avg, n_trials, cut_off, limit_order):
    import canopy_amd
    row_ptr, col = oracle.synth_csr(SEED, n_rows, n_events, avg)
    p = canopy_amd.synthetic_probabilities(SEED, n_events)
    pt = trials_from(p, n_trials, 0.7, 11)
    model = gpu_ctx.upload(row_ptr, col, n_events, row_offset=1000)
    prm = canopy_amd.default_params(cut_off=cut_off, limit_order=limit_order)
    res, info = gpu_ctx.quantify_batch(model, pt, prm)
    assert len(res) == n_trials and 0 <= info.n_candidates <= n_rows
    for k in range(n_trials):
        want = oracle.quantify(row_ptr, col, pt[k], cut_off=cut_off, limit_order=limit_order, n_threads=2, row_offset=1000,
                               want_rows=False)
        same(res[k], want)
        model.set_probabilities(pt[k])
        single = gpu_ctx.quantify(model, prm)
        assert (single.n_kept, single.argmax_row, single.max_logp) == (res[k].n_kept, res[k].argmax_row, res[k].max_logp)
        assert rel(single.rare_event, res[k].rare_event) < TOL or single.rare_event == res[k].rare_event
    model.close()


def test_batch_large_shard_uses_the_filter_pass(gpu_ctx, oracle):
    """2e6 rows x 1e5 events: the envelope pass is FILTER_REFINE (AUTO), candidates are a small fraction."""
    import canopy_amd
    n_rows, n_events, avg, n_trials = 2_000_000, 100_000, 6, 16
    p = canopy_amd.synthetic_probabilities(SEED, n_events)
    pt = trials_from(p, n_trials, 0.3, 5)
    model = gpu_ctx.synthetic(SEED, n_rows, n_events, avg)
    prm = canopy_amd.default_params()
    res, info = gpu_ctx.quantify_batch(model, pt, prm)
    assert 0 < info.n_candidates < n_rows // 4
    row_ptr, col = oracle.synth_csr(SEED, n_rows, n_events, avg, n_threads=4)
    for k in (0, 7, 15):
        want = oracle.quantify(row_ptr, col, pt[k], cut_off=1e-8, limit_order=20, n_threads=4, want_rows=False)
        same(res[k], want)
    for k in range(n_trials):
        model.set_probabilities(pt[k])
        single = gpu_ctx.quantify(model, prm)
        assert (single.n_kept, single.argmax_row, single.max_logp) == (res[k].n_kept, res[k].argmax_row, res[k].max_logp)
        assert rel(single.rare_event, res[k].rare_event) < TOL and rel(single.mcub, res[k].mcub) < TOL
    assert max(r.n_kept for r in res) <= info.n_candidates  # every trial's survivors lie inside the envelope's
    model.close()


def test_batch_log_input_and_extreme_probabilities(gpu_ctx, oracle):
    import canopy_amd
    rows = [[0, 1], [1, 2, 3], [4], [0, 4], [2], []]
    row_ptr = np.zeros(len(rows) + 1, dtype=np.int64)
    row_ptr[1:] = np.cumsum([len(r) for r in rows])
    col = np.array([c for r in rows for c in r], dtype=np.int32)
    pt = np.array([[0.5, 0.25, 1.0, 0.0, 1e-3],
                   [1.0, 1.0, 1.0, 1.0, 1.0],
                   [0.0, 0.0, 0.0, 0.0, 0.0],
                   [1e-3, 2e-3, 3e-3, 4e-3, 5e-3]])
    model = gpu_ctx.upload(row_ptr, col, 5)
    prm = canopy_amd.default_params(cut_off=1e-8)
    res, _ = gpu_ctx.quantify_batch(model, pt, prm)
    with np.errstate(divide="ignore"):
        res_log, _ = gpu_ctx.quantify_batch(model, np.log(pt), prm, log_input=True)
    for k in range(len(pt)):
        want = oracle.quantify(row_ptr, col, pt[k], cut_off=1e-8, limit_order=20, want_rows=False)
        same(res[k], want)
        assert (res_log[k].n_kept, res_log[k].argmax_row, res_log[k].rare_event, res_log[k].mcub) == \
               (res[k].n_kept, res[k].argmax_row, res[k].rare_event, res[k].mcub)
    model.close()


def test_batch_argument_errors(gpu_ctx):
    import canopy_amd
    from canopy_amd import _capi
    model = gpu_ctx.synthetic(SEED, 1000, 100, 4)
    bad = np.full((2, 100), 0.5)
    bad[1, 7] = 1.5
    with pytest.raises(_capi.CanopyError) as e:
        gpu_ctx.quantify_batch(model, bad)
    assert e.value.status == _capi.ERR_INVALID_ARGUMENT
    with pytest.raises(_capi.CanopyError) as e:
        gpu_ctx.quantify_batch(model, np.full((2, 100), 0.5), canopy_amd.default_params(flags=_capi.QUANT_ROW_LOGP))
    assert e.value.status == _capi.ERR_UNSUPPORTED
    with pytest.raises(ValueError):
        gpu_ctx.quantify_batch(model, np.full((2, 99), 0.5))
    # the model's own probabilities are untouched by a batch
    p = canopy_amd.synthetic_probabilities(SEED, 100)
    model.set_probabilities(p)
    before = gpu_ctx.quantify(model)
    gpu_ctx.quantify_batch(model, np.full((3, 100), 0.5))
    after = gpu_ctx.quantify(model)
    assert (before.rare_event, before.n_kept, before.argmax_row) == (after.rare_event, after.n_kept, after.argmax_row)
    model.close()


def test_batch_records_into_a_caller_device_buffer(gpu_ctx):
    """The multi-GPU route: the trial kernels write this rank's records into the all-gather slot."""
    import canopy_amd
    from canopy_amd.distributed import BatchCombiner
    n_events, n_trials = 800, 11
    p = canopy_amd.synthetic_probabilities(SEED, n_events)
    pt = trials_from(p, n_trials, 0.4, 2)
    model = gpu_ctx.synthetic(SEED, 60_000, n_events, 5, first_row=7_000_000)
    prm = canopy_amd.default_params()
    want, _ = gpu_ctx.quantify_batch(model, pt, prm)
    comb = BatchCombiner(n_trials, ctx=gpu_ctx)
    gpu_ctx.quantify_batch_async(model, pt, prm, device_out=comb.device_out_ptr())
    gpu_ctx.synchronize()
    comb.all_gather()
    got = comb.results()
    for g, w in zip(got, want):
        assert (g.rare_event, g.mcub, g.n_kept, g.argmax_row, g.max_logp) == (w.rare_event, w.mcub, w.n_kept, w.argmax_row,
                                                                              w.max_logp)
        assert w.argmax_row >= 7_000_000
    model.close()
