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

SOURCES = ["canopy_api.hip", "quantify_pass.hip", "host_pool.cpp", "kernels_rows.hip", "kernels_build.hip", "kernels_ws.hip", "kernels_bs.hip", "kernels_ls.hip", "kernels_vs.hip", "kernels_mc.hip", "kernels_fr.hip",
           "kernels_batch.hip", "comm_rccl.hip", "mef_lowering.cpp"]
HEADERS = ["canopy_device.h", "canopy_internal.h", "canopy_guard.h"]

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",          # s_i must be a chain of plain IEEE adds (bit-exact vs the oracle)
    "-munsafe-fp-atomics",        # LDS fp64 add as one ds_add_f64 (IEEE add; bit-exactness is tested)
    "-fvisibility=hidden", "-DCANOPY_BUILD", "-DCANOPY_SHARED",
]


OBJ_DIR = os.path.join(HERE, "build")


def _deps_common():
    return [os.path.join(CSRC, f) for f in HEADERS] + [os.path.join(ROOT, "include", "canopy.h"), os.path.abspath(__file__)]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """One object per source (compiled in parallel, only when stale), then one link.

    CANOPY_ABLATIONS=1 in the environment of THIS script compiles the tuning switches and timing ablations of
    benchmarks/ in (-DCANOPY_ABLATIONS); the default library reads no environment variable at run time."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    common = _deps_common()
    flags = [f for f in FLAGS if f != "-shared"]
    if os.environ.get("CANOPY_ABLATIONS", "") not in ("", "0"):
        flags.append("-DCANOPY_ABLATIONS")
    stamp = os.path.join(OBJ_DIR, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True  # another flag set built the objects that are there
    jobs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, src + ".o")
        objs.append(obj)
        if force or _newer(obj, [os.path.join(CSRC, src)] + common):
            cmd = [hipcc] + flags + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            jobs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, pr in jobs if pr.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    if force or jobs or _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden"] + objs + ["-ldl", "-o", LIB]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
