"""GPU parity on model shapes chosen to stress the slab-stream builder and the slab passes: a column every row contains,
all entries of a slab in one column bucket, rows of the longest length the fixed-point pass takes, the last column of
the widest event space, empty rows between full ones. Every kernel variant against the oracle (tests/test_gpu_parity.py's
checker: integers and per-row sums bit-exact, rare-event / MCUB within 1e-12)."""
import numpy as np
import pytest

from test_gpu_parity import check_against_oracle

pytestmark = pytest.mark.gpu


def csr_from_rows(rows):
    row_ptr = np.zeros(len(rows) + 1, dtype=np.int64)
    row_ptr[1:] = np.cumsum([len(r) for r in rows])
    col = np.fromiter((c for r in rows for c in r), dtype=np.int32, count=int(row_ptr[-1]))
    return row_ptr, col


def random_rows(rng, n_rows, lo, hi, max_len, always=()):
    rows = []
    for _ in range(n_rows):
        k = int(rng.integers(0, max_len + 1))
        extra = rng.choice(np.arange(lo, hi), size=min(k, hi - lo), replace=False) if k else np.empty(0, dtype=np.int64)
        rows.append(sorted(set(int(c) for c in extra) | set(always)))
    return rows


@pytest.mark.parametrize("cut_off", [0.0, 1e-6])
def test_a_column_in_every_row(gpu_ctx, oracle, cut_off):
    rng = np.random.default_rng(11)
    n_events = 5000
    rows = random_rows(rng, 20011, 1, n_events, 6, always=(0,))
    row_ptr, col = csr_from_rows(rows)
    p = 10.0 ** rng.uniform(-3, -0.5, size=n_events)
    p[0] = 0.9
    check_against_oracle(gpu_ctx, oracle, row_ptr, col, p, n_events, cut_off, 20)


@pytest.mark.parametrize("cut_off", [0.0, 1e-8])
def test_all_entries_in_one_column_bucket(gpu_ctx, oracle, cut_off):
    rng = np.random.default_rng(12)
    n_events = 100_000
    rows = random_rows(rng, 17000, 500, 600, 5)  # 100 of 1e5 columns: one of the builder's 512 buckets takes everything
    row_ptr, col = csr_from_rows(rows)
    p = 10.0 ** rng.uniform(-4, -1, size=n_events)
    check_against_oracle(gpu_ctx, oracle, row_ptr, col, p, n_events, cut_off, 20)


@pytest.mark.parametrize("cut_off", [0.0, 1e-30])
def test_rows_of_63_entries(gpu_ctx, oracle, cut_off):
    rng = np.random.default_rng(13)
    n_events = 64
    rows = [sorted(int(c) for c in rng.choice(n_events, size=63, replace=False)) for _ in range(9000)]
    rows += [[], [5], list(range(63))]
    row_ptr, col = csr_from_rows(rows)
    p = rng.uniform(0.3, 1.0, size=n_events)
    check_against_oracle(gpu_ctx, oracle, row_ptr, col, p, n_events, cut_off, 0)


def test_last_column_of_the_widest_event_space(gpu_ctx, oracle):
    rng = np.random.default_rng(14)
    n_events = 1 << 18
    rows = random_rows(rng, 8200, 0, n_events - 1, 4, always=(n_events - 1,))
    rows[100] = []
    row_ptr, col = csr_from_rows(rows)
    p = 10.0 ** rng.uniform(-3, -1, size=n_events)
    p[n_events - 1] = 0.5
    check_against_oracle(gpu_ctx, oracle, row_ptr, col, p, n_events, 1e-9, 20)


def test_empty_rows_between_full_slabs(gpu_ctx, oracle):
    rng = np.random.default_rng(15)
    n_events = 3000
    rows = random_rows(rng, 8192, 0, n_events, 8) + [[] for _ in range(8192)] + random_rows(rng, 700, 0, n_events, 8)
    row_ptr, col = csr_from_rows(rows)
    p = 10.0 ** rng.uniform(-2, -0.3, size=n_events)
    check_against_oracle(gpu_ctx, oracle, row_ptr, col, p, n_events, 1e-4, 20)


def test_24_bit_stream_exception_slab_and_rebuild(gpu_ctx, oracle):
    """The quantised filter's 24-bit stream: three full slabs + a sparse last one (an exception slab: all its rows go to the
    exact pass), the 32-bit stream rebuilt on demand by the fixed-point pass, and a sparse model that keeps the 32-bit form."""
    from canopy_amd import _capi, default_params
    n_events, n_rows = 50_000, 3 * 8192 + 300
    row_ptr, col = oracle.synth_csr(20250704, n_rows, n_events, 6, n_threads=4)
    p = oracle.synth_probabilities(20250704, n_events)
    model = gpu_ctx.upload(row_ptr, col, n_events)
    model.set_probabilities(p)
    model.prepare(_capi.KERNEL_FILTER_REFINE)
    nnz = int(row_ptr[-1])
    # a model with a row stream leaves the column-sorted chunks to the first pass that reads them
    assert model.info().stream_bytes == 0 and model.info().stream_bytes_rows > 0
    model.prepare(_capi.KERNEL_FIXED_SUM)
    assert 0 < model.info().stream_bytes < 3.3 * nnz + 8 * n_events + 4096  # 3 B per entry (+ table, offsets), not 4
    for cut_off in (1e-8, 1e-5):
        want = oracle.quantify(row_ptr, col, p, cut_off=cut_off, limit_order=20, n_threads=4, want_rows=False)
        for kern in (_capi.KERNEL_FILTER_REFINE, _capi.KERNEL_FIXED_SUM, _capi.KERNEL_FILTER_REFINE):
            got = gpu_ctx.quantify(model, default_params(cut_off=cut_off, kernel=kern))
            assert got.n_kept == int(want.n_kept) and got.argmax_row == int(want.argmax_row), (cut_off, kern)
            assert abs(got.rare_event - want.rare_event) <= 1e-12 * want.rare_event
            assert abs(got.mcub - want.mcub) <= 1e-12 * want.mcub
    model.close()
    # 2 entries per row over 1e5 events: a chunk of 256 entries spans ~1560 columns -> the model keeps the 32-bit form
    n_events = 100_000
    row_ptr, col = oracle.synth_csr(7, n_rows, n_events, 2, n_threads=4)
    p = oracle.synth_probabilities(20250704, n_events)
    model = gpu_ctx.upload(row_ptr, col, n_events)
    model.set_probabilities(p)
    model.prepare(_capi.KERNEL_FIXED_SUM)  # (builds the chunks: the filter itself would read the row stream)
    assert model.info().stream_bytes >= 4 * int(row_ptr[-1])
    want = oracle.quantify(row_ptr, col, p, cut_off=1e-3, limit_order=20, n_threads=4, want_rows=False)
    got = gpu_ctx.quantify(model, default_params(cut_off=1e-3, kernel=_capi.KERNEL_FILTER_REFINE))
    assert got.n_kept == int(want.n_kept) and got.argmax_row == int(want.argmax_row)
    assert abs(got.rare_event - want.rare_event) <= 1e-12 * want.rare_event
    model.close()


@pytest.mark.parametrize("long_len", [257, 258])
def test_packed_row_sums_do_not_carry_into_the_neighbour(gpu_ctx, oracle, long_len):
    """The folded filter packs two rows' u16 sums into one LDS word: a row of 257 entries that all clamp at 255 sums to
    exactly 65535 (no carry into the row beside it); 258 entries take the plain filter. Every second row is such a row, the
    rows between them survive the cut-off with one likely event."""
    rng = np.random.default_rng(21)
    n_events = 600
    rows = []
    for i in range(9000):
        if i % 2 == 0:
            rows.append(sorted(int(c) for c in rng.choice(np.arange(100, n_events), size=long_len, replace=False)))
        else:
            rows.append([int(rng.integers(0, 100))])
    row_ptr, col = csr_from_rows(rows)
    p = np.full(n_events, 1e-300)  # -log p far beyond the table's range: every such entry clamps at 255
    p[:100] = rng.uniform(0.2, 0.9, size=100)
    from canopy_amd import _capi, default_params
    want = oracle.quantify(row_ptr, col, p, cut_off=1e-8, limit_order=0, n_threads=4, want_rows=False)
    assert int(want.n_kept) == 4500
    model = gpu_ctx.upload(row_ptr, col, n_events)
    model.set_probabilities(p)
    got = gpu_ctx.quantify(model, default_params(cut_off=1e-8, limit_order=0, kernel=_capi.KERNEL_FILTER_REFINE))
    assert got.n_kept == 4500 and got.argmax_row == int(want.argmax_row)
    assert abs(got.rare_event - want.rare_event) <= 1e-12 * want.rare_event
    model.close()


def test_pipelined_passes_over_alternating_models(gpu_ctx, oracle):
    """Consecutive asynchronous passes share three lanes of scratch: two models of different sizes and cut-offs in
    alternation (scratch of a lane regrows between passes, the quantised table is rebuilt per model), fetched at irregular
    points, must give what each model gives alone."""
    from canopy_amd import combine_partials, default_params
    specs = [(600_000, 100_000, 6, 1e-8), (1_300_000, 3_000, 5, 1e-6), (20_000, 500, 4, 1e-8)]
    models, wants, prms = [], [], []
    for k, (n_rows, n_events, avg, cut) in enumerate(specs):
        row_ptr, col = oracle.synth_csr(100 + k, n_rows, n_events, avg, n_threads=4)
        p = oracle.synth_probabilities(100 + k, n_events)
        m = gpu_ctx.upload(row_ptr, col, n_events)
        m.set_probabilities(p)
        models.append(m)
        prms.append(default_params(cut_off=cut))
        wants.append(oracle.quantify(row_ptr, col, p, cut_off=cut, limit_order=20, n_threads=4, want_rows=False))
    gpu_ctx.set_pipeline(True)
    rng = np.random.default_rng(5)
    try:
        last = None
        for i in range(120):
            last = int(rng.integers(0, len(models)))
            gpu_ctx.quantify_async(models[last], prms[last])
            if rng.random() < 0.2:
                got = combine_partials([gpu_ctx.fetch_partial()])
                w = wants[last]
                assert got.n_kept == int(w.n_kept) and got.argmax_row == int(w.argmax_row), (i, last)
                assert abs(got.rare_event - w.rare_event) <= 1e-12 * w.rare_event
        got = combine_partials([gpu_ctx.fetch_partial()])
        assert got.n_kept == int(wants[last].n_kept)
    finally:
        gpu_ctx.synchronize()
        for m in models:
            m.close()


def test_band_stream_is_auto_for_per_row_output_and_takes_long_rows(gpu_ctx, oracle):
    """The band stream (kernels_bs.hip): AUTO's exact all-rows kernel from 2e6 rows (E <= 131072) -- every s_i and the keep
    mask of a 3e6-row model (184 slabs, the last one partial) bit for bit, limit_order below the longest row (the epilogue
    reads row lengths); rows of up to 300 entries (the builder's plain forward rule beyond 32 entries; a row needs as many
    bands as it has entries) and a column every row contains; beyond 2^17 basic events the variant is refused."""
    import canopy_amd
    from canopy_amd import _capi, default_params
    seed, n, e = 20250704, 3_000_000, 50_000
    model = gpu_ctx.synthetic(seed, n, e, 6)
    p = canopy_amd.synthetic_probabilities(seed, e)
    model.set_probabilities(p)
    row_ptr, col = model.download()
    assert model.info().stream_bytes_exact == 0
    model.prepare(_capi.KERNEL_BAND_STREAM)  # (the layout can be built ahead of the first pass)
    per_row = model.info().stream_bytes_exact / n
    assert 24.0 < per_row < 30.0, per_row  # 4 B per entry, avg 6 entries per row, a few per cent of padding
    for cut_off, limit in ((1e-8, 20), (0.0, 7)):
        want = oracle.quantify(row_ptr, col, p, cut_off=cut_off, limit_order=limit, n_threads=4)
        got = gpu_ctx.quantify(model, default_params(cut_off=cut_off, limit_order=limit, flags=_capi.QUANT_ROW_LOGP | _capi.QUANT_KEEP_MASK))
        assert gpu_ctx.last_kernel_variant() == _capi.KERNEL_BAND_STREAM
        assert got.n_kept == int(want.n_kept) and got.argmax_row == int(want.argmax_row) and got.max_logp == want.max_logp
        assert abs(got.rare_event - want.rare_event) <= 1e-12 * want.rare_event and abs(got.mcub - want.mcub) <= 1e-12 * want.mcub
        assert np.array_equal(gpu_ctx.fetch_row_logp(model), want.row_logp)
        assert np.array_equal(gpu_ctx.fetch_keep_mask(model), want.keep_mask)
    model.close()
    # ONE row of 1500 entries somewhere in those 3e6: only its slab gets the extra bands, AUTO still takes the band stream
    rng = np.random.default_rng(77)
    i = 1_234_567
    long_row = np.sort(rng.choice(e, size=1500, replace=False)).astype(np.int32)
    lens = np.diff(row_ptr)
    lens[i] = 1500
    rp2 = np.zeros(n + 1, dtype=np.int64)
    rp2[1:] = np.cumsum(lens)
    col2 = np.concatenate([col[:row_ptr[i]], long_row, col[row_ptr[i + 1]:]])
    model = gpu_ctx.upload(rp2, col2, e)
    model.set_probabilities(p)
    want = oracle.quantify(rp2, col2, p, cut_off=1e-8, limit_order=0, n_threads=4)
    got = gpu_ctx.quantify(model, default_params(cut_off=1e-8, limit_order=0, flags=_capi.QUANT_ROW_LOGP))
    assert gpu_ctx.last_kernel_variant() == _capi.KERNEL_BAND_STREAM
    assert got.n_kept == int(want.n_kept) and got.argmax_row == int(want.argmax_row)
    assert np.array_equal(gpu_ctx.fetch_row_logp(model), want.row_logp)
    model.close()
    # long rows, a hot column
    rows = random_rows(rng, 40_000, 0, 3000, 300, always=(1500,))
    rp, ci = csr_from_rows(rows)
    pp = 10.0 ** rng.uniform(-3, 0, size=3000)
    check_against_oracle(gpu_ctx, oracle, rp, ci, pp, 3000, 1e-30, 0)
    # too wide for the 17-bit columns of the band stream: an explicit request is refused, AUTO takes another kernel
    wide = gpu_ctx.synthetic(seed, 50_000, 140_000, 6)
    wide.set_probabilities(canopy_amd.synthetic_probabilities(seed, 140_000))
    with pytest.raises(_capi.CanopyError) as err:
        gpu_ctx.quantify(wide, default_params(kernel=_capi.KERNEL_BAND_STREAM, flags=_capi.QUANT_ROW_LOGP))
    assert err.value.status == _capi.ERR_UNSUPPORTED
    gpu_ctx.quantify(wide, default_params(flags=_capi.QUANT_ROW_LOGP))
    assert gpu_ctx.last_kernel_variant() != _capi.KERNEL_BAND_STREAM
    wide.close()


@pytest.mark.parametrize("n_events", [3000, 10_000, 40_000, 70_000, 120_000, 200_000])
def test_row_stream_of_the_quantised_filter(gpu_ctx, oracle, n_events):
    """The quantised filter's row stream (kernels_ls.hip): rows sorted by length inside every slab, columns of 12 / 14 / 16 /
    17 bits (8-bit table) and 17 / 18 bits (4-bit table, more than 112 K events), lists of positions that the exact pass
    translates. Two steps of four slabs, a third with one full slab and a partial
    one; lengths 0 .. 12 with a few rows of 40, 300 and 1000 entries (the class of 255 and more); the best row twice
    (the argmax tie goes to the lower row although the list does not ascend in rows); keep mask and sums against the
    oracle for every variant."""
    from canopy_amd import _capi
    rng = np.random.default_rng(77 + n_events)
    n_rows = 9 * 8192 + 777
    rows = random_rows(rng, n_rows, 0, n_events, 12)
    for r, k in ((5, 40), (8191, 300), (8192, 1000), (40_000, 300), (n_rows - 1, 40)):
        rows[r] = sorted(int(c) for c in rng.choice(n_events, size=k, replace=False))
    p = 10.0 ** rng.uniform(-3.0, -0.3, size=n_events)
    best = [int(np.argmax(p))]
    rows[60_000] = best
    rows[123] = best      # the same (largest) probability twice: row 123 is the argmax
    rows[9000] = []       # an empty cut set is kept whatever the cut-off
    row_ptr, col = csr_from_rows(rows)
    model = gpu_ctx.upload(row_ptr, col, n_events)
    model.prepare(_capi.KERNEL_FILTER_REFINE)
    info = model.info()
    assert info.stream_bytes_rows > 0
    # (bits / 8) bytes per entry + the padding of the items that hold the long rows + descriptors, masks, the table once
    bits = 12 if n_events + 2 <= 4096 else 14 if n_events + 2 <= 16384 else 16 if n_events + 2 <= 65536 else 17 if n_events + 2 <= 131072 else 18
    assert info.stream_bytes_rows < (bits / 8 + 0.1) * int(row_ptr[-1]) + 64 * (1000 + 300 + 300 + 40 + 40) * 2.3 + 8 * n_events + 65536
    model.close()
    for cut_off in (1e-6, 1e-3):
        check_against_oracle(gpu_ctx, oracle, row_ptr, col, p, n_events, cut_off, 20)
    check_against_oracle(gpu_ctx, oracle, row_ptr, col, p, n_events, 1e-4, 3)  # limit_order below most rows


def test_row_stream_gives_way_to_the_chunks_when_one_long_row_would_pad_it(gpu_ctx, oracle):
    """An item of the row stream is as long as its longest row: one row of 5000 entries among 20000 rows of two would make
    the stream seven times the model's entries. Such a model keeps the column-sorted chunks (stream_bytes_rows == 0) and
    gives the same results."""
    from canopy_amd import _capi
    rng = np.random.default_rng(5)
    n_events = 20_000
    rows = [sorted(int(c) for c in rng.choice(n_events, size=2, replace=False)) for _ in range(20_000)]
    rows[777] = sorted(int(c) for c in rng.choice(n_events, size=5000, replace=False))
    row_ptr, col = csr_from_rows(rows)
    p = 10.0 ** rng.uniform(-3.0, -0.3, size=n_events)
    model = gpu_ctx.upload(row_ptr, col, n_events)
    model.prepare(_capi.KERNEL_FILTER_REFINE)
    info = model.info()
    assert info.stream_bytes_rows == 0 and info.stream_bytes > 0
    model.close()
    check_against_oracle(gpu_ctx, oracle, row_ptr, col, p, n_events, 1e-4, 20)
