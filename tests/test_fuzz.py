"""Seeded fuzz: random model shapes through every kernel variant, the batch path and the importance pass, against
the oracle. Shapes the fixed tests do not hit: 1..3 basic events, single rows, empty rows in runs, row counts just
around the slab / wave-slab sizes, probabilities exactly 0 and 1, cut-offs from 0 to 1, small limit orders."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-12


def random_model(rng):
    n_events = int(rng.choice([1, 2, 3, 7, 64, 65, 1000, 2049, 5000, 20000, 120000, 200000, 230000, 262144]))  # (the last four: the 4-bit table / column parts)
    n_rows = int(rng.choice([1, 2, 63, 64, 65, 4095, 4096, 4097, 8191, 8192, 8193, 16385, 30011]))
    max_len = int(min(n_events, rng.choice([1, 2, 4, 10, 25])))
    lens = rng.integers(0 if rng.random() < 0.3 else 1, max_len + 1, size=n_rows)
    if rng.random() < 0.3:  # a run of empty rows
        a = int(rng.integers(0, n_rows))
        lens[a:a + int(rng.integers(1, 200))] = 0
    row_ptr = np.zeros(n_rows + 1, dtype=np.int64)
    row_ptr[1:] = np.cumsum(lens)
    col = np.empty(int(row_ptr[-1]), dtype=np.int32)
    for i in range(n_rows):
        if lens[i]:
            col[row_ptr[i]:row_ptr[i + 1]] = np.sort(rng.choice(n_events, size=int(lens[i]), replace=False))
    kind = rng.integers(0, 3)
    if kind == 0:
        p = 10.0 ** rng.uniform(-8, 0, size=n_events)
    elif kind == 1:
        p = rng.uniform(0.0, 1.0, size=n_events)
    else:
        p = 10.0 ** rng.uniform(-4, -1, size=n_events)
    if rng.random() < 0.4 and n_events > 2:
        p[rng.integers(0, n_events)] = 1.0
        p[rng.integers(0, n_events)] = 0.0
    cut_off = float(rng.choice([0.0, 1e-300, 1e-12, 1e-8, 1e-3, 0.5, 1.0]))
    limit_order = int(rng.choice([0, 1, 3, 20]))
    return n_events, row_ptr, col, p, cut_off, limit_order


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def agree(got, want):
    assert got.n_kept == int(want.n_kept) and got.argmax_row == int(want.argmax_row)
    if int(want.n_kept):
        assert got.max_logp == want.max_logp
    for g, w in ((got.rare_event, want.rare_event), (got.mcub, want.mcub)):
        if math.isfinite(w) and w > 0:
            assert rel(g, w) < TOL, (g, w)
        else:
            assert g == w or (math.isnan(g) and math.isnan(w)), (g, w)


@pytest.mark.parametrize("seed", range(int(os.environ.get("CANOPY_FUZZ_SEEDS", "24"))))
def test_random_models_every_variant(gpu_ctx, oracle, seed):
    from canopy_amd import _capi, default_params
    rng = np.random.default_rng(1000 + seed)
    n_events, row_ptr, col, p, cut_off, limit_order = random_model(rng)
    n_rows = len(row_ptr) - 1
    want = oracle.quantify(row_ptr, col, p, cut_off=cut_off, limit_order=limit_order, n_threads=2, row_offset=5)
    model = gpu_ctx.upload(row_ptr, col, n_events, row_offset=5)
    model.set_probabilities(p)
    max_len = int(np.diff(row_ptr).max())
    kernels = [_capi.KERNEL_AUTO, _capi.KERNEL_ROWS, _capi.KERNEL_WAVE_STREAM]
    if n_events <= 12288:
        kernels.append(_capi.KERNEL_ROWS_LDS)
    if max_len <= 4095:
        kernels.append(_capi.KERNEL_VALUE_SCATTER)
    if n_events <= (1 << 17):
        kernels.append(_capi.KERNEL_BAND_STREAM)
    for kern in kernels:
        got = gpu_ctx.quantify(model, default_params(cut_off=cut_off, limit_order=limit_order, kernel=kern,
                                                     flags=_capi.QUANT_ROW_LOGP | _capi.QUANT_KEEP_MASK))
        agree(got, want)
        assert np.array_equal(gpu_ctx.fetch_row_logp(model), want.row_logp), kern
        assert np.array_equal(gpu_ctx.fetch_keep_mask(model), want.keep_mask), kern
    sum_only = [_capi.KERNEL_FILTER_REFINE] if cut_off > 0 else []
    if max_len <= 63:
        sum_only.append(_capi.KERNEL_FIXED_SUM)  # any cut-off, 0 included
    for kern in sum_only:
        got = gpu_ctx.quantify(model, default_params(cut_off=cut_off, limit_order=limit_order, kernel=kern,
                                                     flags=_capi.QUANT_KEEP_MASK))
        agree(got, want)
        assert np.array_equal(gpu_ctx.fetch_keep_mask(model), want.keep_mask), kern
    # importance contributions behind AUTO
    gpu_ctx.quantify(model, default_params(cut_off=cut_off, limit_order=limit_order, flags=_capi.QUANT_IMPORTANCE))
    contrib = gpu_ctx.fetch_importance(model)
    ref = oracle.importance(row_ptr, col, p, cut_off=cut_off, limit_order=limit_order)
    assert np.array_equal(contrib == 0, ref == 0)
    nz = (ref != 0) & np.isfinite(ref)
    if nz.any():
        assert np.max(np.abs(contrib[nz] - ref[nz]) / ref[nz]) < TOL
    # batch: three perturbed vectors
    pt = np.clip(p[None, :] * np.exp(0.5 * rng.standard_normal((3, n_events))), 0.0, 1.0)
    res, _ = gpu_ctx.quantify_batch(model, pt, default_params(cut_off=cut_off, limit_order=limit_order))
    for k in range(3):
        agree(res[k], oracle.quantify(row_ptr, col, pt[k], cut_off=cut_off, limit_order=limit_order, row_offset=5,
                                      want_rows=False))
    model.close()
    assert n_rows >= 1


@pytest.mark.parametrize("seed", range(int(os.environ.get("CANOPY_FUZZ_SEEDS", "24")) // 2))
def test_random_ticketed_sequences(gpu_ctx, oracle, seed):
    """Random sequences of (new p or new log p or the old one, a new cut-off or the old one, a ticketed pass) over TWO random
    models on 2..4 lanes, records fetched in enqueue order with a random number of passes outstanding: every record against
    the oracle's for the vector and cut-off ITS pass was enqueued with."""
    import canopy_amd
    from canopy_amd import default_params
    rng = np.random.default_rng(9000 + seed)
    lanes = int(rng.integers(2, 5))
    models = []
    for _ in range(2):
        n_events, row_ptr, col, p, cut_off, limit_order = random_model(rng)
        m = gpu_ctx.upload(row_ptr, col, n_events)
        m.set_probabilities(p)
        models.append(dict(m=m, n_events=n_events, row_ptr=row_ptr, col=col, p=p.copy(), limit_order=limit_order, cut_off=cut_off))
    gpu_ctx.set_pipeline(lanes)
    try:
        inflight = []  # (ticket, expected)

        def fetch_one():
            t, want = inflight.pop(0)
            agree(canopy_amd.combine_partials([gpu_ctx.ticket_fetch(t)]), want)

        for step in range(int(rng.integers(6, 16))):
            d = models[int(rng.integers(0, 2))]
            u = rng.random()
            if u < 0.4:  # a new probability vector
                d["p"] = np.clip(d["p"] * np.exp(0.4 * rng.standard_normal(d["n_events"])), 0.0, 1.0)
                d["m"].set_probabilities(d["p"])
            elif u < 0.6:  # caller-supplied logs: the library must take them as they are
                d["p"] = np.clip(d["p"] * np.exp(0.4 * rng.standard_normal(d["n_events"])), 1e-300, 1.0)
                d["m"].set_log_probabilities(np.array([math.log(x) for x in d["p"]]))  # (math.log = libm's log, the oracle's)
            if rng.random() < 0.3:
                d["cut_off"] = float(rng.choice([0.0, 1e-12, 1e-8, 1e-3]))
            want = oracle.quantify(d["row_ptr"], d["col"], d["p"], cut_off=d["cut_off"], limit_order=d["limit_order"], n_threads=2,
                                   want_rows=False)
            inflight.append((gpu_ctx.enqueue(d["m"], default_params(cut_off=d["cut_off"], limit_order=d["limit_order"])), want))
            while len(inflight) >= lanes or (inflight and rng.random() < 0.3):
                fetch_one()
        while inflight:
            fetch_one()
    finally:
        gpu_ctx.set_pipeline(True)
        for d in models:
            d["m"].close()


@pytest.mark.parametrize("seed", range(max(2, int(os.environ.get("CANOPY_FUZZ_SEEDS", "24")) // 4)))
def test_random_medium_models_every_variant(gpu_ctx, oracle, seed):
    """Models of 5e4 .. 5e5 rows of the synthetic family (several slabs of every layout, full bands and chunks, a partial last
    slab) with random width, row length, probabilities and filters: every variant against the oracle, per-row sums and keep
    masks bit for bit."""
    from test_gpu_parity import check_against_oracle
    rng = np.random.default_rng(7000 + seed)
    n_rows = int(rng.integers(50_000, 500_000))
    n_events = int(rng.choice([300, 3000, 30_000, 100_000, 131_072, 200_000]))
    avg = int(rng.choice([2, 3, 6, 9]))
    row_ptr, col = oracle.synth_csr(int(rng.integers(1, 1 << 30)), n_rows, n_events, avg, n_threads=4)
    kind = rng.integers(0, 3)
    p = 10.0 ** rng.uniform(-8, 0, size=n_events) if kind == 0 else (rng.uniform(0.0, 1.0, size=n_events) if kind == 1 else 10.0 ** rng.uniform(-3, -1, size=n_events))
    if rng.random() < 0.3:
        p[rng.integers(0, n_events)] = 1.0
        p[rng.integers(0, n_events)] = 0.0
    cut_off = float(rng.choice([0.0, 1e-12, 1e-8, 1e-4]))
    limit_order = int(rng.choice([0, 3, 20]))
    check_against_oracle(gpu_ctx, oracle, row_ptr, col, p, n_events, cut_off, limit_order, row_offset=int(rng.integers(0, 1000)))


def random_dag(rng):
    from canopy_amd import _capi as c
    n_events = int(rng.choice([1, 2, 5, 17, 64, 80]))
    n_gates = int(rng.choice([0, 1, 3, 20, 120, 300]))
    gate_type, gate_min, arg_ptr, args = [], [], [0], []
    for g in range(n_gates):
        n_prev = n_events + g
        t = int(rng.choice([c.GATE_AND, c.GATE_OR, c.GATE_ATLEAST, c.GATE_XOR, c.GATE_NOT, c.GATE_NAND, c.GATE_NOR, c.GATE_NULL]))
        if t in (c.GATE_NOT, c.GATE_NULL):
            k = 1
        elif t == c.GATE_ATLEAST:
            k = int(min(n_prev, rng.choice([2, 3, 7, 8, 30, 63])))
        else:
            k = int(min(n_prev, rng.choice([1, 2, 4, 8, 9, 17, 40, 70])))
        # bias towards recent nodes so that deep chains appear, but keep events reachable
        pool = np.arange(n_prev)
        w = np.where(pool >= n_prev - 10, 8.0, 1.0)
        a = rng.choice(pool, size=k, replace=False, p=w / w.sum())
        neg = rng.random(k) < 0.2
        gate_type.append(t)
        gate_min.append(int(rng.integers(0, k + 2)) if t == c.GATE_ATLEAST else 0)
        args.extend(int(~x) if n else int(x) for x, n in zip(a, neg))
        arg_ptr.append(len(args))
    top = int(rng.integers(0, n_events + n_gates)) if rng.random() < 0.3 else n_events + n_gates - 1
    dag = dict(n_events=n_events, gate_type=np.array(gate_type, dtype=np.int32), gate_min=np.array(gate_min, dtype=np.int32),
               arg_ptr=np.array(arg_ptr, dtype=np.int32), args=np.array(args, dtype=np.int32), top=top)
    p = rng.uniform(0.0, 1.0, n_events) if rng.random() < 0.5 else 10.0 ** rng.uniform(-4, -0.5, n_events)
    if n_events > 2 and rng.random() < 0.4:
        p[0], p[1] = 1.0, 0.0
    return dag, p


@pytest.mark.parametrize("seed", range(int(os.environ.get("CANOPY_FUZZ_SEEDS", "24"))))
def test_random_dags_monte_carlo_bit_exact(gpu_ctx, oracle, seed):
    from canopy_amd.quantifier import Dag
    rng = np.random.default_rng(5000 + seed)
    dag, p = random_dag(rng)
    d = Dag(gpu_ctx, dag)
    d.set_probabilities(p)
    for n, first in ((int(rng.integers(1, 200)), 0), (64 * int(rng.integers(1, 40)) + int(rng.integers(0, 64)), 64 * int(rng.integers(0, 1000))),
                     (40_000, 128)):
        got = d.run(seed, n, first_sample=first)
        want = oracle.mc(dag, p, seed, n, first_sample=first, n_threads=2)
        assert got.hits == want, (n, first, got.hits, want)
    d.close()
