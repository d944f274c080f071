"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.
Bar: integer/index outputs and per-row s_i bit-exact; rare-event / MCUB <= 1e-12 relative."""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 20250704
TOL = 1e-12  # north_star: top-event probability within 1e-12 relative of the CPU reference
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quantify_mpmath.json")


def csr(rows):
    row_ptr = np.zeros(len(rows) + 1, dtype=np.int64)
    for i, r in enumerate(rows):
        row_ptr[i + 1] = row_ptr[i] + len(r)
    return row_ptr, np.array([c for r in rows for c in r], dtype=np.int32)


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def variants(n_events, max_row_len=10, cut_off=0.0):
    from canopy_amd import _capi
    v = [_capi.KERNEL_ROWS]
    if n_events <= 12288:
        v.append(_capi.KERNEL_ROWS_LDS)
    if n_events <= (1 << 19):
        v.append(_capi.KERNEL_WAVE_STREAM)
    if n_events <= (1 << 17):
        v.append(_capi.KERNEL_BAND_STREAM)
    if n_events <= (1 << 18) and max_row_len <= 4095:
        v.append(_capi.KERNEL_VALUE_SCATTER)
    if 1 <= n_events <= (1 << 18) and cut_off > 0:
        v.append(_capi.KERNEL_FILTER_REFINE)  # sum-only variant (no per-row output)
    if 1 <= n_events <= (1 << 18) and max_row_len <= 63:
        v.append(_capi.KERNEL_FIXED_SUM)  # sum-only variant, any cut-off
    return v


def check_against_oracle(ctx, oracle, row_ptr, col, p, n_events, cut_off, limit_order, row_offset=0, threads=4):
    from canopy_amd import _capi, default_params
    want = oracle.quantify(row_ptr, col, p, cut_off=cut_off, limit_order=limit_order, n_threads=threads,
                           row_offset=row_offset)
    model = ctx.upload(row_ptr, col, n_events, row_offset=row_offset)
    model.set_probabilities(p)
    max_len = int(np.diff(row_ptr).max()) if len(row_ptr) > 1 else 0
    for kern in variants(n_events, max_len, cut_off):
        if kern in (_capi.KERNEL_FILTER_REFINE, _capi.KERNEL_FIXED_SUM):
            got = ctx.quantify(model, default_params(cut_off=cut_off, limit_order=limit_order,
                                                     flags=_capi.QUANT_KEEP_MASK, kernel=kern))
            assert got.n_kept == int(want.n_kept) and got.argmax_row == int(want.argmax_row), kern
            if want.n_kept:
                assert got.max_logp == want.max_logp
            if len(row_ptr) > 1:
                assert np.array_equal(ctx.fetch_keep_mask(model), want.keep_mask), kern
            if math.isfinite(want.rare_event) and want.rare_event > 0:
                assert rel(got.rare_event, want.rare_event) < TOL and rel(got.mcub, want.mcub) < TOL
            else:
                assert got.rare_event == want.rare_event and got.mcub == want.mcub
            got2 = ctx.quantify(model, default_params(cut_off=cut_off, limit_order=limit_order, kernel=kern))
            assert (got2.rare_event, got2.mcub, got2.n_kept, got2.argmax_row) == (got.rare_event, got.mcub, got.n_kept,
                                                                                  got.argmax_row)
            continue
        prm = default_params(cut_off=cut_off, limit_order=limit_order,
                             flags=_capi.QUANT_ROW_LOGP | _capi.QUANT_KEEP_MASK, kernel=kern)
        try:
            got = ctx.quantify(model, prm)
        except _capi.CanopyError as e:
            # the explicit-only VALUE_SCATTER layout is placed greedily and may refuse a model (a column every row
            # contains): CANOPY_ERR_UNSUPPORTED, as canopy.h documents for an explicitly requested variant
            if kern == _capi.KERNEL_VALUE_SCATTER and e.status == _capi.ERR_UNSUPPORTED:
                continue
            # the band stream refuses a slab that would need more than 8192 bands (rows of thousands of entries)
            if kern == _capi.KERNEL_BAND_STREAM and e.status == _capi.ERR_UNSUPPORTED and max_len > 2000:
                continue
            raise
        assert got.n_kept == int(want.n_kept), kern
        assert got.argmax_row == int(want.argmax_row), kern
        if want.n_kept:
            assert got.max_logp == want.max_logp, kern
        if len(row_ptr) > 1:
            assert np.array_equal(ctx.fetch_row_logp(model), want.row_logp), kern  # bit-exact s_i
            assert np.array_equal(ctx.fetch_keep_mask(model), want.keep_mask), kern
        if math.isfinite(want.rare_event) and want.rare_event > 0:
            assert rel(got.rare_event, want.rare_event) < TOL, (kern, got.rare_event, want.rare_event)
            assert rel(got.mcub, want.mcub) < TOL, (kern, got.mcub, want.mcub)
        else:
            assert got.rare_event == want.rare_event and got.mcub == want.mcub
        # sum-only pass (no per-row outputs) gives the same scalars
        got2 = ctx.quantify(model, default_params(cut_off=cut_off, limit_order=limit_order, kernel=kern))
        assert (got2.rare_event, got2.mcub, got2.n_kept, got2.argmax_row) == (got.rare_event, got.mcub, got.n_kept,
                                                                              got.argmax_row)
    model.close()
    return want


# ---- known answers and edge cases ------------------------------------------------------------------

def test_known_answers(gpu_ctx, oracle):
    w = check_against_oracle(gpu_ctx, oracle, *csr([[0, 1, 2]]), np.array([0.1, 0.2, 0.3]), 3, 0.0, 20)
    assert rel(w.rare_event, 0.006) < 1e-15
    check_against_oracle(gpu_ctx, oracle, *csr([[0], [1], [2], [3]]), np.array([1e-3, 2e-3, 3e-3, 4e-3]), 4, 0.0, 20)
    w = check_against_oracle(gpu_ctx, oracle, *csr([[0, 1], [0, 2], [1, 2]]), np.full(3, 0.1), 3, 0.0, 20)
    assert rel(w.mcub, 0.029701) < 1e-12 and w.argmax_row == 0


def test_edge_cases(gpu_ctx, oracle):
    # empty model
    check_against_oracle(gpu_ctx, oracle, np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.array([0.5]), 1,
                         1e-8, 20)
    # empty rows, certain (p=1) and impossible (p=0) events
    check_against_oracle(gpu_ctx, oracle, *csr([[], [0], [], [1, 2], [3]]), np.array([0.5, 1.0, 1.0, 0.0]), 4, 0.0, 20)
    check_against_oracle(gpu_ctx, oracle, *csr([[3], [0, 3]]), np.array([0.5, 1.0, 1.0, 0.0]), 4, 1e-8, 20)
    # cut-off boundary: p == cut_off is kept, just below is dropped
    cut = 1e-8
    check_against_oracle(gpu_ctx, oracle, *csr([[0], [1], [2], [3, 4]]),
                         np.array([cut, cut * (1 - 1e-12), cut * (1 + 1e-12), 1e-4, 1e-4]), 5, cut, 20)
    # limit_order
    check_against_oracle(gpu_ctx, oracle, *csr([[0, 1, 2], [0, 1], [3, 4, 5], [5]]), np.full(6, 0.5), 6, 0.0, 2)


def test_ragged_and_long_rows(gpu_ctx, oracle):
    rng = np.random.default_rng(7)
    n_events = 5000
    rows = []
    for i in range(3000):
        ln = int(rng.choice([0, 1, 2, 5, 17, 40, 300, 4000], p=[.05, .2, .3, .3, .1, .03, .015, .005]))
        rows.append(np.sort(rng.choice(n_events, size=ln, replace=False)).tolist())
    p = rng.uniform(0.3, 1.0, size=n_events)  # long rows must not all underflow
    check_against_oracle(gpu_ctx, oracle, *csr(rows), p, n_events, 1e-30, 0)
    check_against_oracle(gpu_ctx, oracle, *csr(rows), p, n_events, 1e-6, 20)


def test_row_counts_around_slab_and_wave_sizes(gpu_ctx, oracle):
    p = oracle.synth_probabilities(SEED, 1000)
    for n in (1, 63, 64, 65, 255, 256, 257, 1023, 4097):
        row_ptr, col = oracle.synth_csr(SEED, n, 1000, 4)
        check_against_oracle(gpu_ctx, oracle, row_ptr, col, p, 1000, 0.0, 20)


def test_invalid_models_and_arguments_are_rejected(gpu_ctx):
    from canopy_amd import _capi
    bad = [
        (np.array([0, 2, 1], dtype=np.int64), np.array([0, 1], dtype=np.int32)),   # row_ptr not monotone
        (np.array([0, 2], dtype=np.int64), np.array([1, 0], dtype=np.int32)),      # not ascending
        (np.array([0, 2], dtype=np.int64), np.array([1, 1], dtype=np.int32)),      # duplicate
        (np.array([0, 1], dtype=np.int64), np.array([7], dtype=np.int32)),         # out of range
        (np.array([0, 1], dtype=np.int64), np.array([-1], dtype=np.int32)),        # negative
        (np.array([0, 5, 2], dtype=np.int64), np.array([0, 1], dtype=np.int32)),   # row_ptr past nnz
    ]
    for row_ptr, col in bad:
        with pytest.raises(_capi.CanopyError) as e:
            gpu_ctx.upload(row_ptr, col, 3)
        assert e.value.status == _capi.ERR_INVALID_MODEL
    m = gpu_ctx.upload(*csr([[0]]), 1)
    with pytest.raises(_capi.CanopyError) as e:
        gpu_ctx.quantify(m)  # probabilities not set
    assert e.value.status == _capi.ERR_STATE
    with pytest.raises(_capi.CanopyError) as e:
        m.set_probabilities(np.array([1.5]))
    assert e.value.status == _capi.ERR_INVALID_ARGUMENT
    m.close()


# ---- golden vectors ------------------------------------------------------------------------------------

@pytest.mark.parametrize("idx", range(6))
def test_against_mpmath_golden(gpu_ctx, idx):
    from canopy_amd import default_params
    with open(GOLD) as f:
        g = json.load(f)
    c = g["cases"][idx]
    model = gpu_ctx.synthetic(g["seed"], c["n_rows"], g["n_events"], c["avg"], first_row=c["first_row"])
    model.set_probabilities(np.array([float.fromhex(x) for x in g["p_%s_hex" % c["p"]]]))
    e = c["expected"]
    for kern in variants(g["n_events"], 10, c["cut_off"]):
        r = gpu_ctx.quantify(model, default_params(cut_off=c["cut_off"], limit_order=c["limit_order"], kernel=kern))
        assert r.n_kept == e["n_kept"] and r.argmax_row == e["argmax_row"]
        assert rel(r.rare_event, float(e["rare"])) < 1e-12 and rel(r.mcub, float(e["mcub"])) < 1e-12
    model.close()


# ---- BASELINE configs -----------------------------------------------------------------------------------

def test_synthetic_generator_matches_oracle(gpu_ctx, oracle):
    for (n, e, avg, first) in ((100000, 1000, 4, 0), (50000, 100000, 6, 0), (4097, 100000, 6, 99990000),
                               (1000, 64, 17, 5)):
        m = gpu_ctx.synthetic(SEED, n, e, avg, first_row=first)
        row_ptr, col = m.download()
        o_rp, o_col = oracle.synth_csr(SEED, n, e, avg, first_row=first, n_threads=4)
        assert np.array_equal(row_ptr, o_rp) and np.array_equal(col, o_col)
        info = m.info()
        assert info.nnz == o_rp[-1] and info.row_offset == first and info.max_row_len == 2 * avg - 2
        assert info.algorithmic_bytes == (n + 1) * 8 + info.nnz * 4 + e * 8
        m.close()


@pytest.mark.parametrize("cut_off", [1e-8, 0.0])
def test_config2_full_parity(gpu_ctx, oracle, cut_off):
    """BASELINE config 2: 1e5 cut sets x 1e3 events, avg 4 nnz/row."""
    row_ptr, col = oracle.synth_csr(SEED, 100000, 1000, 4, n_threads=4)
    p = oracle.synth_probabilities(SEED, 1000)
    check_against_oracle(gpu_ctx, oracle, row_ptr, col, p, 1000, cut_off, 20)


def test_config3_shape_reduced_full_parity(gpu_ctx, oracle):
    """Config-3 shape (1e5 events, avg 6) at 2e6 rows: full per-row comparison against the oracle."""
    n = 2_000_000
    row_ptr, col = oracle.synth_csr(SEED, n, 100000, 6, n_threads=8)
    p = oracle.synth_probabilities(SEED, 100000)
    check_against_oracle(gpu_ctx, oracle, row_ptr, col, p, 100000, 1e-12, 20, threads=8)
    check_against_oracle(gpu_ctx, oracle, row_ptr[:300001] , col[:row_ptr[300000]], p, 100000, 0.0, 5, threads=8)


def test_config3_full_size_properties(gpu_ctx, oracle):
    """BASELINE config 3 at full size (1e8 x 1e5, avg 6): size-independent properties.
    - sampled row windows: s_i and the keep mask bit-exact against the oracle (counter-based rows)
    - linearity: two half-models (row ranges) combine to the whole: integers exact, sums <= 1e-12
    - monotonicity: n_kept(cut_off=0) == R; raising the cut-off only removes rows
    """
    from canopy_amd import _capi, combine_partials, default_params
    n, e, avg = 100_000_000, 100000, 6
    p = oracle.synth_probabilities(SEED, e)
    model = gpu_ctx.synthetic(SEED, n, e, avg)
    model.set_probabilities(p)
    prm = default_params(cut_off=1e-14, limit_order=20, flags=_capi.QUANT_ROW_LOGP | _capi.QUANT_KEEP_MASK)
    whole = gpu_ctx.quantify(model, prm)
    for first in (0, 12_345_600, 49_999_936, 99_990_016):
        cnt = 9984
        o_rp, o_col = oracle.synth_csr(SEED, cnt, e, avg, first_row=first, n_threads=4)
        want = oracle.quantify(o_rp, o_col, p, cut_off=1e-14, limit_order=20)
        assert np.array_equal(gpu_ctx.fetch_row_logp(model, first, cnt), want.row_logp)
        assert np.array_equal(gpu_ctx.fetch_keep_mask(model, first // 64, cnt // 64), want.keep_mask)
    part_whole = gpu_ctx.fetch_partial()
    all_rows = gpu_ctx.quantify(model, default_params(cut_off=0.0, limit_order=0))
    assert all_rows.n_kept == n and all_rows.n_kept >= whole.n_kept
    assert 0.05 < all_rows.rare_event < 0.12  # ~0.08 at config 3
    assert all_rows.mcub < all_rows.rare_event
    model.close()
    parts = []
    kept = 0
    for first, cnt in ((0, 37_000_000), (37_000_000, 63_000_000)):
        m = gpu_ctx.synthetic(SEED, cnt, e, avg, first_row=first)
        m.set_probabilities(p)
        gpu_ctx.quantify_async(m, default_params(cut_off=1e-14, limit_order=20))
        parts.append(gpu_ctx.fetch_partial())
        kept += int(parts[-1].n_kept)
        m.close()
    both = combine_partials(parts)
    assert both.n_kept == whole.n_kept == kept and both.argmax_row == whole.argmax_row
    assert both.max_logp == whole.max_logp
    assert rel(both.rare_event, whole.rare_event) < TOL and rel(both.mcub, whole.mcub) < TOL
    assert part_whole.n_kept == whole.n_kept


@pytest.mark.parametrize("n_events", [114_689, 200_000, 260_096, 260_097, 262_144])
def test_filter_refine_wide_event_space(gpu_ctx, oracle, n_events):
    """Beyond 112 K events the 8-bit filter table no longer fits LDS. Up to 260 096 (round 3) the table is kept at four bits per
    event and the pass stays a single sweep (114 689 = the first such size, odd; 260 096 = the last: the table fills the 160 KiB
    to the byte); beyond that the columns are cut into parts, one part of the 8-bit table resident per sweep, the row sums carried
    between sweeps (260 097 = the first size with parts, 2^18 = the largest event space of the slab stream: 18 column bits). Every variant against the oracle,
    FILTER_REFINE and FIXED_SUM included."""
    row_ptr, col = oracle.synth_csr(SEED, 300000, n_events, 6, n_threads=4)
    p = oracle.synth_probabilities(SEED, n_events)
    p[::7] = np.minimum(1.0, p[::7] * 3e3)  # some likely events: candidates with entries in every part
    check_against_oracle(gpu_ctx, oracle, row_ptr, col, p, n_events, 1e-10, 20, threads=4)


def test_gather_slot_buffer_single_rank(gpu_ctx, oracle):
    """K5 plumbing on the GPU: the final-reduce kernel writes this rank's record into its slot of the
    all-reduce buffer and zeroes the others (here world=3, rank=1 emulated without a process group)."""
    import torch
    from canopy_amd import _capi, combine_partials, default_params
    row_ptr, col = oracle.synth_csr(SEED, 30000, 1000, 4)
    p = oracle.synth_probabilities(SEED, 1000)
    model = gpu_ctx.upload(row_ptr, col, 1000, row_offset=777)
    model.set_probabilities(p)
    buf = torch.full((3 * _capi.PARTIAL_DOUBLES,), 7.0, dtype=torch.float64, device="cuda")
    _capi.check(_capi.lib().canopy_ctx_set_gather_buffer(gpu_ctx.handle, buf.data_ptr(), 3, 1), "set_gather")
    try:
        want = gpu_ctx.quantify(model, default_params(cut_off=1e-9))
        gpu_ctx.synchronize()
        host = buf.cpu().numpy()
    finally:
        _capi.lib().canopy_ctx_set_gather_buffer(gpu_ctx.handle, None, 0, 0)
    assert np.all(host[:8] == 0.0) and np.all(host[16:] == 0.0)
    got = combine_partials([_capi.Partial(*host[k * 8:(k + 1) * 8].tolist()) for k in range(3)])
    assert (got.rare_event, got.mcub, got.n_kept, got.argmax_row) == (want.rare_event, want.mcub, want.n_kept,
                                                                     want.argmax_row)
    o = oracle.quantify(row_ptr, col, p, cut_off=1e-9, row_offset=777)
    assert got.n_kept == int(o.n_kept) and got.argmax_row == int(o.argmax_row)
    model.close()


def test_auto_leaves_filter_refine_when_most_rows_survive(gpu_ctx, oracle):
    """AUTO feedback: with a cut-off that keeps most rows the filter flags > 20 % of them; the next AUTO
    pass on that model runs the all-rows fixed-point pass instead. Results are the oracle's either way."""
    from canopy_amd import _capi, default_params
    n_events, n_rows = 5000, 1_200_000  # AUTO considers filter/refine from 1e6 rows on
    row_ptr, col = oracle.synth_csr(SEED, n_rows, n_events, 4, n_threads=4)
    p = oracle.synth_probabilities(SEED, n_events)
    want = oracle.quantify(row_ptr, col, p, cut_off=1e-30, limit_order=20, n_threads=4, want_rows=False)
    assert want.n_kept > 0.5 * n_rows
    model = gpu_ctx.upload(row_ptr, col, n_events)
    model.set_probabilities(p)
    ran = []
    for _ in range(3):
        got = gpu_ctx.quantify(model, default_params(cut_off=1e-30))
        assert got.n_kept == int(want.n_kept) and got.argmax_row == int(want.argmax_row)
        assert rel(got.rare_event, want.rare_event) < TOL and rel(got.mcub, want.mcub) < TOL
        ran.append(gpu_ctx.last_kernel_variant())
    # first call: filter + a refine pass over nearly every row; its candidate count then steers AUTO away
    # (to the fixed-point all-rows pass: no per-row output was asked for)
    assert ran == [_capi.KERNEL_FILTER_REFINE, _capi.KERNEL_FIXED_SUM, _capi.KERNEL_FIXED_SUM]
    # with the default cut-off few rows survive and AUTO stays with filter/refine
    for _ in range(2):
        gpu_ctx.quantify(model, default_params())
    assert gpu_ctx.last_kernel_variant() == _capi.KERNEL_FILTER_REFINE
    model.close()


def test_filter_refine_rows_on_the_cut_off_boundary(gpu_ctx, oracle):
    """Adversarial for the conservative filter: the cut-off is placed exactly ON some rows' s_i (and one
    ulp around it), with rows of 1..40 entries and probabilities spanning 1e-300..1. A filter that is not
    strictly conservative would lose survivors here; count, argmax and the keep mask must match the oracle."""
    from canopy_amd import _capi, default_params
    rng = np.random.default_rng(11)
    n_events, n_rows = 3000, 60000
    p = 10.0 ** rng.uniform(-6, 0, size=n_events)
    p[:5] = [1.0, 1e-300, 1e-12, 0.5, 0.0]
    rows = []
    for i in range(n_rows):
        ln = int(rng.integers(1, 41)) if i % 7 else int(rng.integers(1, 4))
        rows.append(np.sort(rng.choice(n_events, size=ln, replace=False)).tolist())
    row_ptr, col = csr(rows)
    base = oracle.quantify(row_ptr, col, p, cut_off=0.0, limit_order=0)
    finite = base.row_logp[np.isfinite(base.row_logp)]
    model = gpu_ctx.upload(row_ptr, col, n_events)
    model.set_probabilities(p)
    for q in (0.5, 0.9, 0.99, 0.999):
        s_star = float(np.quantile(finite, q))
        s_star = float(finite[np.argmin(np.abs(finite - s_star))])  # exactly some row's s_i
        for cut in (math.exp(s_star), np.nextafter(math.exp(s_star), 0.0), np.nextafter(math.exp(s_star), 1.0)):
            if not (0.0 < cut <= 1.0):
                continue
            want = oracle.quantify(row_ptr, col, p, cut_off=float(cut), limit_order=0)
            for kern in (_capi.KERNEL_FILTER_REFINE, _capi.KERNEL_WAVE_STREAM, _capi.KERNEL_ROWS):
                got = gpu_ctx.quantify(model, default_params(cut_off=float(cut), limit_order=0, flags=_capi.QUANT_KEEP_MASK,
                                                             kernel=kern))
                assert got.n_kept == int(want.n_kept) and got.argmax_row == int(want.argmax_row), (q, cut, kern)
                assert np.array_equal(gpu_ctx.fetch_keep_mask(model), want.keep_mask), (q, cut, kern)
                assert rel(got.rare_event, want.rare_event) < TOL
    model.close()


def test_captured_pass_replays_like_an_eager_one(gpu_ctx, oracle):
    """canopy_graph_capture / canopy_graph_launch: same record as the eager pass, also after the probabilities changed
    between replays, for a one-kernel variant and for filter/refine (whose quantised table is rebuilt inside the graph)."""
    import canopy_amd
    from canopy_amd import _capi
    for n_rows, n_events, avg, flags in ((60_000, 900, 4, _capi.QUANT_KEEP_MASK), (1_100_000, 30_000, 6, 0)):
        p = canopy_amd.synthetic_probabilities(SEED, n_events)
        model = gpu_ctx.synthetic(SEED, n_rows, n_events, avg)
        model.set_probabilities(p)
        prm = canopy_amd.default_params(flags=flags)
        g = gpu_ctx.graph_capture(model, prm)
        try:
            for scale in (1.0, 0.5, 2.0):
                model.set_probabilities(np.minimum(1.0, p * scale))
                want = gpu_ctx.quantify(model, prm)
                want_mask = gpu_ctx.fetch_keep_mask(model) if flags else None
                gpu_ctx.graph_launch(g)
                got = canopy_amd.combine_partials([gpu_ctx.fetch_partial()])
                assert (got.rare_event, got.mcub, got.n_kept, got.argmax_row, got.max_logp) == \
                       (want.rare_event, want.mcub, want.n_kept, want.argmax_row, want.max_logp)
                if flags:
                    assert np.array_equal(gpu_ctx.fetch_keep_mask(model), want_mask)
        finally:
            gpu_ctx.graph_free(g)
            model.close()


def test_fixed_sum_decides_integers_exactly(gpu_ctx, oracle):
    """FIXED_SUM sums every row in 2^-47 fixed point and recomputes in exact CSR order only the rows that error could
    decide differently. Adversarial for it: rows sitting exactly on / one ulp around the cut-off, many rows tied for
    the argmax (ties -> lowest row), events with p = 0 (clamped table entries), p = 1, p so small that |s| passes the
    accuracy budget, and a limit_order below the longest row."""
    from canopy_amd import _capi, default_params
    rng = np.random.default_rng(77)
    n_events, n_rows = 3000, 40_000
    p = 10.0 ** rng.uniform(-6, -0.5, size=n_events)
    p[0], p[1], p[2], p[3] = 0.0, 1.0, 1e-200, 5e-324
    best = [int(np.argmax(p[4:])) + 4]
    rows = []
    for i in range(n_rows):
        if 20_000 <= i < 20_500:
            rows.append(best)  # the global maximum, tied 500 times (ties resolve to the lowest row)
            continue
        r = np.sort(rng.choice(np.arange(4, n_events), size=int(rng.integers(1, 9)), replace=False)).tolist()
        if i % 97 == 0:
            r = [i // 97 % 4] + r  # the special events (smallest ids: the row stays ascending)
        rows.append(r)
    row_ptr, col = csr(rows)
    full = oracle.quantify(row_ptr, col, p, cut_off=0.0, limit_order=0, n_threads=4)
    assert int(full.argmax_row) <= 20_000 and full.max_logp == np.log(p[best[0]])
    finite = full.row_logp[np.isfinite(full.row_logp)]
    pivots = [float(np.exp(x)) for x in np.quantile(finite, [0.3, 0.6, 0.9])]
    pivots += [float(np.nextafter(pv, 0.0)) for pv in pivots[:2]] + [float(np.nextafter(pivots[0], 1.0))]
    model = gpu_ctx.upload(row_ptr, col, n_events)
    model.set_probabilities(p)
    for cut in [0.0, 1e-300] + pivots:
        for lim in (0, 3, 20):
            want = oracle.quantify(row_ptr, col, p, cut_off=cut, limit_order=lim, n_threads=4)
            got = gpu_ctx.quantify(model, default_params(cut_off=cut, limit_order=lim, kernel=_capi.KERNEL_FIXED_SUM,
                                                         flags=_capi.QUANT_KEEP_MASK))
            assert got.n_kept == int(want.n_kept) and got.argmax_row == int(want.argmax_row), (cut, lim)
            if want.n_kept and want.argmax_row >= 0:
                assert got.max_logp == want.max_logp, (cut, lim)
            assert np.array_equal(gpu_ctx.fetch_keep_mask(model), want.keep_mask), (cut, lim)
            if want.rare_event > 0:
                assert rel(got.rare_event, want.rare_event) < TOL and rel(got.mcub, want.mcub) < TOL, (cut, lim)
            else:
                assert got.rare_event == want.rare_event and got.mcub == want.mcub
    model.close()
