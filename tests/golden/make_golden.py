#!/usr/bin/env python3
"""Generate tests/golden/quantify_mpmath.json -- the vectors that pin the CPU oracle.

The reference holds no golden vector for this path (its tests/ is a 0-byte CMakeLists.txt and
there is no quantifier to run), so the oracle is pinned independently instead:
  * the synthetic CSR is rebuilt here in PURE PYTHON integers (no C, no numpy arithmetic) and its
    sha256 is recorded -> pins the counter-based generator of oracle/ and of the HIP library;
  * rare-event and MCUB are recomputed with mpmath at 50 significant digits directly from the
    definition (P_i = prod p_j; rare = sum P_i; mcub = 1 - prod(1 - P_i)), treating the fp64 inputs
    p_j as exact rationals -> pins the oracle's arithmetic (log-domain, compensated) to <= 1e-13.
Run:  python tests/golden/make_golden.py      (needs mpmath; ~1 min)
"""
import hashlib
import json
import os
import struct
import sys

import mpmath as mp

mp.mp.dps = 50
M = (1 << 64) - 1
SEED = 20250704


def mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def draw(seed, stream, index, t):
    key = mix64(seed ^ ((stream * 0xD1B54A32D192ED03) & M))
    return mix64((mix64(key ^ index) + t) & M)


def bounded(h, n):
    return ((h >> 32) * n) >> 32


def synth_csr(seed, first_row, n_rows, n_events, avg):
    row_ptr, cols = [0], []
    for i in range(first_row, first_row + n_rows):
        ln = 2 + bounded(draw(seed, 1, i, 0), 2 * avg - 3)
        row, t = [], 0
        while len(row) < ln:
            c = bounded(draw(seed, 2, i, t), n_events)
            t += 1
            if c not in row:
                row.append(c)
        cols.extend(sorted(row))
        row_ptr.append(len(cols))
    return row_ptr, cols


def uniform(seed, j):
    return (draw(seed, 3, j, 0) >> 11) * 2.0 ** -53


def csr_sha(row_ptr, cols):
    h = hashlib.sha256()
    h.update(struct.pack("<%dq" % len(row_ptr), *row_ptr))
    h.update(struct.pack("<%di" % len(cols), *cols))
    return h.hexdigest()


def quantify_mp(row_ptr, cols, p, cut_off, limit_order, row_offset):
    pm = [mp.mpf(x) for x in p]
    rare, prod, n_kept = mp.mpf(0), mp.mpf(1), 0
    best, arg, min_gap = None, -1, None
    for i in range(len(row_ptr) - 1):
        b, e = row_ptr[i], row_ptr[i + 1]
        P = mp.mpf(1)
        for k in range(b, e):
            P *= pm[cols[k]]
        if cut_off > 0:
            gap = abs(mp.log(P) - mp.log(mp.mpf(cut_off))) if P > 0 else mp.inf
            min_gap = gap if min_gap is None or gap < min_gap else min_gap
        keep = (cut_off <= 0 or P >= mp.mpf(cut_off)) and (limit_order == 0 or (e - b) <= limit_order)
        if not keep:
            continue
        n_kept += 1
        rare += P
        prod *= (1 - P)
        if best is None or P > best:
            best, arg = P, row_offset + i
    return dict(rare=mp.nstr(rare, 30), mcub=mp.nstr(1 - prod, 30), n_kept=n_kept, argmax_row=arg,
                max_logp=mp.nstr(mp.log(best), 30) if best is not None else "-inf",
                min_log_gap_to_cutoff=mp.nstr(min_gap, 5) if min_gap is not None else None)


def main():
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.pow.restype = ctypes.c_double
    libm.pow.argtypes = [ctypes.c_double, ctypes.c_double]
    E = 1000
    u = [uniform(SEED, j) for j in range(E)]
    p_low = [libm.pow(10.0, -8.0 + 5.0 * x) for x in u]   # the benchmark's family (glibc pow)
    p_high = [0.05 + 0.9 * x for x in u]                  # plain fp64 arithmetic: P_i not tiny
    cases = [
        dict(name="c2_like_nocut", first_row=0, n_rows=2000, avg=4, p="low", cut_off=0.0, limit_order=0),
        dict(name="c2_like_cut1e-8", first_row=0, n_rows=2000, avg=4, p="low", cut_off=1e-8, limit_order=20),
        dict(name="c3_like_avg6", first_row=0, n_rows=20000, avg=6, p="low", cut_off=1e-12, limit_order=5),
        dict(name="offset_rows", first_row=123456789, n_rows=3000, avg=6, p="low", cut_off=0.0, limit_order=20),
        dict(name="high_p_mcub", first_row=0, n_rows=300, avg=4, p="high", cut_off=1e-3, limit_order=20),
        dict(name="high_p_saturating", first_row=7, n_rows=64, avg=2, p="high", cut_off=0.0, limit_order=0),
    ]
    out = dict(seed=SEED, n_events=E, p_low_hex=[x.hex() for x in p_low], p_high_hex=[x.hex() for x in p_high],
               digits=mp.mp.dps, cases=[])
    for c in cases:
        row_ptr, cols = synth_csr(SEED, c["first_row"], c["n_rows"], E, c["avg"])
        res = quantify_mp(row_ptr, cols, p_low if c["p"] == "low" else p_high, c["cut_off"], c["limit_order"],
                          c["first_row"])
        c = dict(c, nnz=len(cols), csr_sha256=csr_sha(row_ptr, cols), expected=res)
        out["cases"].append(c)
        print(c["name"], res, file=sys.stderr)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "quantify_mpmath.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
