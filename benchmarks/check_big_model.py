import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import canopy_amd, oracle_py
SEED, N, E, AVG = 20250704, int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000, 100_000, 6
cores = len(os.sched_getaffinity(0))
p = canopy_amd.synthetic_probabilities(SEED, E)
ctx = canopy_amd.Context(device=0)
m = ctx.synthetic(SEED, N, E, AVG); m.set_probabilities(p)
got = ctx.quantify(m, canopy_amd.default_params())
print("gpu", got.n_kept, got.argmax_row, got.rare_event, got.mcub, "variant", ctx.last_kernel_variant(), flush=True)
t0 = time.time()
rp, col = oracle_py.synth_csr(SEED, N, E, AVG, n_threads=cores)
print("oracle synth s", time.time() - t0, "nnz", int(rp[-1]), flush=True)
t0 = time.time()
w = oracle_py.quantify(rp, col, p, n_threads=cores, want_rows=False)
print("oracle quantify s", time.time() - t0, int(w.n_kept), int(w.argmax_row), w.rare_event, w.mcub, flush=True)
ok = got.n_kept == int(w.n_kept) and got.argmax_row == int(w.argmax_row) and got.max_logp == w.max_logp \
     and abs(got.rare_event - w.rare_event) <= 1e-12 * w.rare_event and abs(got.mcub - w.mcub) <= 1e-12 * w.mcub
print("MATCH" if ok else "MISMATCH", "(cut_off 1e-8, AUTO)", flush=True)
# the same pass timed: ticketed steps, every record fetched, four lanes
ctx.set_pipeline(4)
prm = canopy_amd.default_params()
for _ in range(8):
    ctx.quantify_async(m, prm)
ctx.synchronize()
inflight, n_steps, same = [], 40, True
t0 = time.time()
for _ in range(n_steps):
    inflight.append(ctx.enqueue(m, prm))
    if len(inflight) == 4:
        r = canopy_amd.combine_partials([ctx.ticket_fetch(inflight.pop(0))]); same &= r.n_kept == got.n_kept and r.rare_event == got.rare_event
for t in inflight:
    r = canopy_amd.combine_partials([ctx.ticket_fetch(t)]); same &= r.n_kept == got.n_kept and r.rare_event == got.rare_event
ctx.synchronize()
dt = (time.time() - t0) / n_steps
print("gpu cut_off 1e-8: ms per step %.3f (every result fetched, 4 lanes), %.3g cut-sets/s, every step's record %s the first pass's"
      % (dt * 1e3, N / dt, "==" if same else "!="), flush=True)
ctx.set_pipeline(0)
# the same model with cut_off = 0: every row through exp / log1p (AUTO -> the fixed-point pass + exact refine of the contenders)
prm0 = canopy_amd.default_params(cut_off=0.0)
got0 = ctx.quantify(m, prm0)
for _ in range(3):
    ctx.quantify_async(m, prm0)
ctx.synchronize()
t0 = time.time()
for _ in range(5):
    ctx.quantify_async(m, prm0)
ctx.synchronize()
dt0 = (time.time() - t0) / 5
print("gpu cut_off 0:", got0.n_kept, got0.argmax_row, got0.rare_event, got0.mcub, "variant", ctx.last_kernel_variant(), "ms per pass %.3f" % (dt0 * 1e3), flush=True)
t0 = time.time()
w0 = oracle_py.quantify(rp, col, p, cut_off=0.0, n_threads=cores, want_rows=False)
print("oracle quantify (cut_off 0) s", time.time() - t0, int(w0.n_kept), int(w0.argmax_row), w0.rare_event, w0.mcub, flush=True)
ok0 = got0.n_kept == int(w0.n_kept) and got0.argmax_row == int(w0.argmax_row) and got0.max_logp == w0.max_logp \
      and abs(got0.rare_event - w0.rare_event) <= 1e-12 * w0.rare_event and abs(got0.mcub - w0.mcub) <= 1e-12 * w0.mcub
print("MATCH" if ok0 else "MISMATCH", "(cut_off 0)")
# K1 as stated on the same model: every s_i written (AUTO -> the band stream), all of them against the oracle's, bit for bit
from canopy_amd import _capi
prm_r = canopy_amd.default_params(flags=_capi.QUANT_ROW_LOGP | _capi.QUANT_KEEP_MASK)
got_r = ctx.quantify(m, prm_r)
for _ in range(2):
    ctx.quantify_async(m, prm_r)
ctx.synchronize()
t0 = time.time()
for _ in range(3):
    ctx.quantify_async(m, prm_r)
ctx.synchronize()
dt_r = (time.time() - t0) / 3
rows_gpu = ctx.fetch_row_logp(m)
mask_gpu = ctx.fetch_keep_mask(m)
print("gpu per-row output: variant", ctx.last_kernel_variant(), "ms per pass %.3f" % (dt_r * 1e3), got_r.n_kept, got_r.argmax_row, flush=True)
t0 = time.time()
wr = oracle_py.quantify(rp, col, p, n_threads=cores, want_rows=True)
print("oracle quantify with rows s", time.time() - t0, flush=True)
ok_r = np.array_equal(rows_gpu, wr.row_logp) and np.array_equal(mask_gpu, wr.keep_mask) and got_r.n_kept == int(wr.n_kept)
print("MATCH" if ok_r else "MISMATCH", "(every s_i and the keep mask, %d rows)" % len(rows_gpu), flush=True)
info = m.info()
print("device_bytes", info.device_bytes, "rows", info.n_rows, "nnz", info.nnz, "stream_bytes_exact", info.stream_bytes_exact)
