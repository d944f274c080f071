"""CPU-side checks of the drop-in boundary: libcanopy.so builds (hipcc cross-compiles gfx950),
loads, exports every symbol include/canopy.h declares, refuses to run without a GPU (no CPU
fallback), and its host-only entry points agree with the oracle."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 20250704


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "canopy.h")).read()
    return sorted(set(re.findall(r"CANOPY_API[^;(]*?\b(canopy_\w+)\s*\(", text)))


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "canopy.h"\nint main(void){canopy_quant_params p; (void)p; return 0;}\n')
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    "-c", str(src), "-o", str(tmp_path / "t.o")], check=True)


def test_cpp_header_compiles(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include "canopy-cpp.h"\nint main(){ canopy::quant_params p; (void)p; return 0; }\n')
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                    "-o", str(tmp_path / "t.o")], check=True)


def test_library_exports_every_declared_symbol(canopy_lib):
    from canopy_amd import _capi
    syms = declared_symbols()
    assert len(syms) >= 25
    out = subprocess.run(["nm", "-D", "--defined-only", _capi.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (canopy_\w+)", out))
    assert set(syms) <= exported, sorted(set(syms) - exported)
    assert set(syms) == set(_capi.SIGNATURES), sorted(set(syms) ^ set(_capi.SIGNATURES))


def test_library_does_not_link_the_oracle(canopy_lib):
    from canopy_amd import _capi
    out = subprocess.run(["readelf", "-d", _capi.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", _capi.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "oracle_" not in syms
    for dirpath, _, files in os.walk(os.path.join(ROOT, "canopy_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                # comments may cite the oracle; code must never include, import, link or open it
                for pat in (r'#\s*include\s*[<"][^>"]*oracle', r"^\s*(import|from)\s+oracle", r"libcanopy_oracle",
                            r"oracle_(quantify|synth|combine|draw|shard)\w*\s*\("):
                    assert not re.search(pat, text, re.M), (f, pat)


def test_status_strings_and_version(canopy_lib):
    assert canopy_lib.canopy_status_string(0) == b"ok"
    assert canopy_lib.canopy_status_string(-4) == b"no usable HIP device"
    v = [C.c_int() for _ in range(3)]
    canopy_lib.canopy_version(*[C.byref(x) for x in v])
    assert (v[0].value, v[1].value) == (0, 1)


def test_default_params_mirror_reference_settings(canopy_lib):
    from canopy_amd import default_params
    p = default_params()
    assert p.cut_off == 1e-8 and p.limit_order == 20 and p.approximation == 3  # settings.h:323,332


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


def test_no_cpu_fallback_without_gpu(canopy_lib):
    if not _no_gpu():
        pytest.skip("a GPU is visible")
    import canopy_amd
    with pytest.raises(canopy_amd._capi.CanopyError) as e:
        canopy_amd.Context(device=0)
    assert e.value.status == canopy_amd._capi.ERR_NO_DEVICE


def test_shard_rows_matches_oracle(canopy_lib, oracle):
    from canopy_amd import shard_rows
    row_ptr, _ = oracle.synth_csr(SEED, 20000, 1000, 6)
    for n in (1, 2, 3, 8, 64):
        assert np.array_equal(shard_rows(row_ptr, n), oracle.shard_rows(row_ptr, n))
    empty = np.zeros(11, dtype=np.int64)
    assert np.array_equal(shard_rows(empty, 4), [0, 2, 5, 7, 10])
    # sliced row_ptr (row_ptr[0] != 0)
    assert np.array_equal(shard_rows(row_ptr[5000:15001], 4), oracle.shard_rows(row_ptr[5000:15001], 4))


def test_synthetic_probabilities_match_oracle(canopy_lib, oracle):
    from canopy_amd import synthetic_probabilities
    for e in (1000, 100000):
        assert np.array_equal(synthetic_probabilities(SEED, e), oracle.synth_probabilities(SEED, e))


def test_combine_partials_matches_oracle(canopy_lib, oracle):
    from canopy_amd import _capi, combine_partials
    row_ptr, col = oracle.synth_csr(SEED, 30000, 1000, 4)
    p = oracle.synth_probabilities(SEED, 1000)
    whole = oracle.quantify(row_ptr, col, p, cut_off=1e-10)
    b = oracle.shard_rows(row_ptr, 4)
    parts, oparts = [], []
    for k in range(4):
        rp = row_ptr[b[k]:b[k + 1] + 1]
        q = oracle.quantify(rp - rp[0], col[rp[0]:rp[-1]], p, cut_off=1e-10, row_offset=int(b[k])).res
        oparts.append(q)
        parts.append(_capi.Partial(q.rare_hi, q.rare_lo, q.log1p_hi, q.log1p_lo, q.n_kept, q.n_certain, q.max_logp,
                                   q.argmax_row))
    r = combine_partials(parts)
    oc = oracle.combine(oparts)
    assert r.n_kept == whole.n_kept and r.argmax_row == whole.argmax_row and r.max_logp == whole.max_logp
    assert r.rare_event == oc.rare_event and r.mcub == oc.mcub  # same fold, same order: bitwise
    assert abs(r.rare_event - whole.rare_event) <= 1e-14 * whole.rare_event
    # clamp flag and the certain-event rule
    big = _capi.Partial(3.0, 0.0, -1.0, 0.0, 5.0, 0.0, -0.1, 2.0)
    assert combine_partials([big], _capi.QUANT_CLAMP).rare_event == 1.0
    assert combine_partials([big]).rare_event == 3.0
    one = _capi.Partial(1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0)
    assert combine_partials([one]).mcub == 1.0
    none = _capi.Partial(0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -np.inf, -1.0)
    r = combine_partials([none, none])
    assert r.n_kept == 0 and r.argmax_row == -1 and r.rare_event == 0.0 and r.mcub == 0.0


def test_default_library_carries_no_tuning_switches(canopy_lib):
    """Round 3: the tuning switches and the wrong-result timing ablations of benchmarks/ (CANOPY_*_DEBUG, CANOPY_FR_*, ...) are
    compiled in only with -DCANOPY_ABLATIONS; the library the product ships reads no environment variable (a stray variable
    must never change a quantification). The names must not even occur in the binary, nor `getenv` among its imports."""
    from canopy_amd import _capi
    blob = open(_capi.LIB_PATH, "rb").read()
    for name in (b"CANOPY_FR_DEBUG", b"CANOPY_WS_DEBUG", b"CANOPY_VS_DEBUG", b"CANOPY_MC_DEBUG", b"CANOPY_NO_PIPELINE",
                 b"CANOPY_FR_", b"CANOPY_FX_", b"CANOPY_MC_", b"CANOPY_COMM_OWN_STREAM"):
        assert name not in blob, name
    undefined = subprocess.run(["nm", "-D", "--undefined-only", _capi.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert " getenv" not in undefined and "secure_getenv" not in undefined
