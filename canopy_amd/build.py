"""Build canopy_amd/libcanopy.so (hand-written HIP for gfx950 + the C ABI of include/canopy.h).

In-tree, explicit hipcc: the .so travels with the repo snapshot to the GPU box.
hipcc cross-compiles gfx950 without a GPU, so this also is the CPU-side "does it build" check.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcanopy.so")

SOURCES = ["canopy_api.hip", "kernels_rows.hip", "kernels_build.hip", "kernels_ws.hip", "kernels_vs.hip", "kernels_mc.hip", "kernels_fr.hip", "mef_lowering.cpp"]
HEADERS = ["canopy_device.h", "canopy_internal.h"]

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",          # s_i must be a chain of plain IEEE adds (bit-exact vs the oracle)
    "-munsafe-fp-atomics",        # LDS fp64 add as one ds_add_f64 (IEEE add; bit-exactness is tested)
    "-fvisibility=hidden", "-DCANOPY_BUILD", "-DCANOPY_SHARED",
]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "canopy.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
