"""Driver for benchmarks/pmc.sh: three per-row-output passes of the band stream at config 3 (PMC_SCRIPT=benchmarks/bs_pmc_driver.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import canopy_amd
from canopy_amd import _capi
SEED = 20250704
ctx = canopy_amd.Context(device=0)
m = ctx.synthetic(SEED, 100_000_000, 100_000, 6)
m.set_probabilities(canopy_amd.synthetic_probabilities(SEED, 100_000))
prm = canopy_amd.default_params(cut_off=1e-8, flags=_capi.QUANT_ROW_LOGP, kernel=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for _ in range(3):
    ctx.quantify_async(m, prm)
ctx.synchronize()
print("ran", ctx.last_kernel_variant())
