import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, canopy_amd, oracle_py
SEED=20250704
n, e = 100_000_000, 100_000
t0=time.time(); rp, col = oracle_py.synth_csr(SEED, n, e, 6, n_threads=len(os.sched_getaffinity(0))); print("synth s %.2f" % (time.time()-t0), flush=True)
ctx = canopy_amd.Context(device=0)
for validate in (True, False, True):
    t0=time.time(); m = ctx.upload(rp, col, e, validate=validate); ctx.synchronize(); t1=time.time()-t0
    m.set_probabilities(canopy_amd.synthetic_probabilities(SEED, e))
    t0=time.time(); m.prepare(0); ctx.synchronize(); t2=time.time()-t0
    t0=time.time(); r = ctx.quantify(m, canopy_amd.default_params()); t3=time.time()-t0
    print("upload (validate=%s) %.3f s = %.1f GB/s; prepare %.3f s; first quantify %.4f s; n_kept %d" % (validate, t1, (rp.nbytes+col.nbytes)/t1/1e9, t2, t3, r.n_kept), flush=True)
    m.close()
