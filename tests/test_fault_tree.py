"""K0 -- Open-PSA MEF fault tree -> cut sets / gate DAG (config 1 plumbing, host logic).
The reference has no example model and no lowering, so the answers are pinned by brute force: the
minimal cut sets of a coherent tree are the minimal TRUE assignments of its truth table."""
import itertools
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLE = os.path.join(ROOT, "examples", "two_of_three_plus_or.xml")


def eval_dag(dag, bits):
    from canopy_amd import _capi as c
    val = list(bits)
    for g, t in enumerate(dag["gate_type"]):
        a = [val[x] if x >= 0 else 1 - val[~x] for x in dag["args"][dag["arg_ptr"][g]:dag["arg_ptr"][g + 1]]]
        if t == c.GATE_AND: r = int(all(a))
        elif t == c.GATE_OR: r = int(any(a))
        elif t == c.GATE_ATLEAST: r = int(sum(a) >= dag["gate_min"][g])
        elif t == c.GATE_XOR: r = sum(a) % 2
        elif t == c.GATE_NOT: r = 1 - a[0]
        elif t == c.GATE_NAND: r = 1 - int(all(a))
        elif t == c.GATE_NOR: r = 1 - int(any(a))
        else: r = a[0]
        val.append(r)
    return val[dag["top"]]


def brute_force(dag, p):
    n = dag["n_events"]
    true_sets, prob = [], 0.0
    for bits in itertools.product((0, 1), repeat=n):
        if eval_dag(dag, bits):
            true_sets.append(frozenset(j for j in range(n) if bits[j]))
            prob += math.prod(p[j] if bits[j] else 1 - p[j] for j in range(n))
    minimal = [s for s in true_sets if not any(t < s for t in true_sets)]
    return sorted(sorted(s) for s in minimal), prob


def csr_rows(row_ptr, col):
    return sorted(list(col[row_ptr[i]:row_ptr[i + 1]]) for i in range(len(row_ptr) - 1))


def test_example_tree_end_to_end_on_cpu(canopy_lib, oracle):
    from canopy_amd.fault_tree import FaultTree
    ft = FaultTree(path=EXAMPLE)
    assert ft.top_gate_name == "TOP" and ft.info.n_basic_events == 7 and ft.info.n_house_events == 1
    names = ft.basic_event_names
    p = ft.probabilities()
    assert dict(zip(names, p)) == {"pumpA": 0.1, "pumpB": 0.1, "pumpC": 0.1, "power": 0.02, "valve1": 0.05,
                                   "valve2": 0.05, "leak": 1e-3}
    row_ptr, col = ft.cut_sets()
    got = sorted(sorted(names[c] for c in r) for r in csr_rows(row_ptr, col))
    assert got == sorted([["leak"], ["pumpA", "pumpB"], ["pumpA", "pumpC"], ["pumpB", "pumpC"], ["power", "valve1"],
                          ["power", "valve2"]])
    for i in range(len(row_ptr) - 1):
        assert np.all(np.diff(col[row_ptr[i]:row_ptr[i + 1]]) > 0)  # ascending: valid for canopy_model_upload
    want_sets, exact = brute_force(ft.dag(), p)
    assert csr_rows(row_ptr, col) == want_sets
    q = oracle.quantify(row_ptr, col, p, cut_off=0.0)
    assert abs(q.rare_event - 0.033) < 1e-15
    assert q.rare_event >= q.mcub >= exact and abs(exact - 0.0308655046) < 1e-9
    # limit_order drops nothing here at 2, keeps only {leak} at 1
    assert len(ft.cut_sets(limit_order=2)[0]) == 7 and csr_rows(*ft.cut_sets(limit_order=1)) == [[names.index("leak")]]
    # Monte-Carlo on the exported DAG agrees with the exact probability
    hits = oracle.mc(ft.dag(), p, 5, 2_000_000, n_threads=4)
    assert abs(hits / 2e6 - exact) < 5 * math.sqrt(exact * (1 - exact) / 2e6)


MEF = """<opsa-mef><define-fault-tree name='t'>%s</define-fault-tree><model-data>%s</model-data></opsa-mef>"""


def events(names, p=0.1):
    return "".join("<define-basic-event name='%s'><float value='%g'/></define-basic-event>" % (n, p) for n in names)


def test_house_events_fold_as_constants(canopy_lib):
    from canopy_amd.fault_tree import FaultTree
    gates = ("<define-gate name='top'><or><gate name='g1'/><gate name='g2'/></or></define-gate>"
             "<define-gate name='g1'><and><basic-event name='a'/><house-event name='h'/></and></define-gate>"
             "<define-gate name='g2'><and><basic-event name='b'/><basic-event name='c'/></and></define-gate>")
    on = FaultTree(MEF % (gates, events("abc") + "<define-house-event name='h'><constant value='true'/></define-house-event>"))
    off = FaultTree(MEF % (gates, events("abc") + "<define-house-event name='h'><constant value='false'/></define-house-event>"))
    assert csr_rows(*on.cut_sets()) == [[0], [1, 2]]
    assert csr_rows(*off.cut_sets()) == [[1, 2]]
    # a house event that decides the top event outright
    const = FaultTree(MEF % ("<define-gate name='top'><or><basic-event name='a'/><house-event name='h'/></or></define-gate>",
                             events("a") + "<define-house-event name='h'><constant value='true'/></define-house-event>"))
    row_ptr, col = const.cut_sets()
    assert list(row_ptr) == [0, 0] and const.info.reserved == 2  # one empty cut set: TOP is TRUE


@pytest.mark.parametrize("k,n", [(1, 3), (2, 4), (3, 5), (5, 5)])
def test_atleast_expansion_matches_truth_table(canopy_lib, k, n):
    from canopy_amd.fault_tree import FaultTree
    names = [chr(ord("a") + i) for i in range(n)]
    gate = "<define-gate name='top'><atleast min='%d'>%s</atleast></define-gate>" % (
        k, "".join("<basic-event name='%s'/>" % x for x in names))
    ft = FaultTree(MEF % (gate, events(names, 0.2)))
    sets, _ = brute_force(ft.dag(), ft.probabilities())
    assert csr_rows(*ft.cut_sets()) == sets and len(sets) == math.comb(n, k)


def test_nested_tree_with_shared_events_is_minimised(canopy_lib):
    from canopy_amd.fault_tree import FaultTree
    gates = ("<define-gate name='top'><and><gate name='x'/><gate name='y'/></and></define-gate>"
             "<define-gate name='x'><or><basic-event name='a'/><basic-event name='b'/></or></define-gate>"
             "<define-gate name='y'><or><basic-event name='a'/><gate name='z'/></or></define-gate>"
             "<define-gate name='z'><and><basic-event name='b'/><basic-event name='c'/></and></define-gate>")
    ft = FaultTree(MEF % (gates, events("abc")))
    sets, _ = brute_force(ft.dag(), ft.probabilities())
    assert csr_rows(*ft.cut_sets()) == sets == [[0], [1, 2]]  # {a}, {b,c}; {a,b}, {a,b,c} are absorbed


def test_non_coherent_tree_goes_to_the_dag_only(canopy_lib, oracle):
    from canopy_amd import _capi
    from canopy_amd.fault_tree import FaultTree
    gates = ("<define-gate name='top'><and><basic-event name='a'/><gate name='n'/></and></define-gate>"
             "<define-gate name='n'><not><basic-event name='b'/></not></define-gate>")
    ft = FaultTree(MEF % (gates, events("ab", 0.3)))
    with pytest.raises(_capi.CanopyError) as e:
        ft.cut_sets()
    assert e.value.status == _capi.ERR_UNSUPPORTED
    _, exact = brute_force(ft.dag(), ft.probabilities())
    assert abs(exact - 0.3 * 0.7) < 1e-15
    hits = oracle.mc(ft.dag(), ft.probabilities(), 3, 1_000_000)
    assert abs(hits / 1e6 - 0.21) < 5 * math.sqrt(0.21 * 0.79 / 1e6)


@pytest.mark.parametrize("body,data,needle", [
    ("<define-gate name='top'><or><gate name='g'/></or></define-gate>", events("a"), "never defined"),
    ("<define-gate name='top'><or><basic-event name='a'/></or></define-gate>", "", "no probability"),
    ("<define-gate name='t1'><or><basic-event name='a'/></or></define-gate>"
     "<define-gate name='t2'><or><basic-event name='a'/></or></define-gate>", events("a"), "more than one top"),
    ("<define-gate name='g1'><or><gate name='g2'/></or></define-gate>"
     "<define-gate name='g2'><or><gate name='g1'/></or></define-gate>", events("a"), "cycle"),
    ("<define-gate name='top'><atleast min='4'><basic-event name='a'/><basic-event name='b'/></atleast></define-gate>",
     events("ab"), "min out of range"),
    ("<define-gate name='top'><or><basic-event name='a'/></or></define-gate>",
     "<define-basic-event name='a'><float value='1.5'/></define-basic-event>", "outside [0,1]"),
])
def test_malformed_models_are_rejected_with_a_message(canopy_lib, body, data, needle):
    from canopy_amd import _capi
    from canopy_amd.fault_tree import FaultTree
    with pytest.raises(_capi.CanopyError) as e:
        FaultTree(MEF % (body, data))
    assert e.value.status == _capi.ERR_INVALID_MODEL and needle in str(e.value)


def test_xml_reader_tolerates_comments_declarations_and_quotes(canopy_lib):
    from canopy_amd import _capi
    from canopy_amd.fault_tree import FaultTree
    text = ('<?xml version="1.0"?>\n<!DOCTYPE opsa-mef>\n<!-- c --><opsa-mef>\n <define-fault-tree name="t">'
            '<label>text is ignored</label><define-gate name="top"> <or> <basic-event name=\'a\' /><!-- x -->'
            '<basic-event name="b"/></or></define-gate></define-fault-tree>' + "<model-data>" + events("ab") +
            "</model-data></opsa-mef>")
    assert csr_rows(*FaultTree(text).cut_sets()) == [[0], [1]]
    for broken in ("<opsa-mef><define-gate name='x'>", "<other/>", "<opsa-mef><a></b></opsa-mef>", ""):
        with pytest.raises(_capi.CanopyError):
            FaultTree(broken)


@pytest.mark.gpu
def test_example_tree_through_the_gpu_paths(gpu_ctx, oracle):
    """Config 1 end to end on the device: MEF -> CSR -> canopy_quantify, and MEF -> DAG -> canopy_mc_run."""
    from canopy_amd import default_params
    from canopy_amd.fault_tree import FaultTree
    from canopy_amd.quantifier import Dag
    ft = FaultTree(path=EXAMPLE)
    p = ft.probabilities()
    row_ptr, col = ft.cut_sets()
    model = gpu_ctx.upload(row_ptr, col, ft.info.n_basic_events)
    model.set_probabilities(p)
    got = gpu_ctx.quantify(model, default_params(cut_off=0.0))
    want = oracle.quantify(row_ptr, col, p, cut_off=0.0)
    assert got.n_kept == 6 and got.argmax_row == int(want.argmax_row)
    assert abs(got.rare_event - want.rare_event) <= 1e-12 * want.rare_event
    assert abs(got.mcub - want.mcub) <= 1e-12 * want.mcub
    model.close()
    d = Dag(gpu_ctx, ft.dag())
    d.set_probabilities(p)
    r = d.run(7, 3_000_000)
    assert r.hits == oracle.mc(ft.dag(), p, 7, 3_000_000, n_threads=4)
    assert abs(r.estimate - 0.0308655046) < 5 * r.std_error
    d.close()


@pytest.mark.gpu
def test_cpp_example_runs_end_to_end(canopy_lib, tmp_path):
    """The C++ consumer of include/canopy-cpp.h: built against libcanopy.so, run on the device."""
    import json
    import subprocess
    exe = tmp_path / "quantify_mef"
    lib_dir = os.path.join(ROOT, "canopy_amd")
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "quantify_mef.cpp"), "-L", lib_dir, "-lcanopy",
                    "-Wl,-rpath," + lib_dir, "-o", str(exe)], check=True)
    out = subprocess.run([str(exe), EXAMPLE], check=True, capture_output=True, text=True).stdout
    r = json.loads(out)
    assert r["cut_sets"] == 6 and r["basic_events"] == 7 and r["top"] == "TOP"
    assert abs(r["rare_event"] - 0.033) < 1e-15 and abs(r["mcub"] - 0.03260898707329901) < 1e-15
    assert abs(r["monte_carlo"] - 0.0308655046) < 5 * r["mc_std_error"]
    lo, mid, hi = r["batch_rare_event"]  # trials at 0.5x, 1x, 2x the point probabilities (canopy_quantify_batch)
    assert mid == r["rare_event"] and lo < mid < hi


@pytest.mark.gpu
def test_cpp_benchmark_driver_matches_the_python_path(canopy_lib, gpu_ctx, tmp_path):
    """benchmarks/bench_quantify.cpp (the native driver over canopy-cpp.h) on a 2e6-row model: same integers and
    sums as the Python path through the same ABI."""
    import json
    import subprocess
    import canopy_amd
    exe = tmp_path / "bench_quantify"
    lib_dir = os.path.join(ROOT, "canopy_amd")
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "benchmarks", "bench_quantify.cpp"), "-L", lib_dir, "-lcanopy",
                    "-Wl,-rpath," + lib_dir, "-o", str(exe)], check=True)
    out = subprocess.run([str(exe), "2000000", "100000", "6", "3", "1"], check=True, capture_output=True, text=True).stdout
    r = json.loads(out)
    model = gpu_ctx.synthetic(20250704, 2_000_000, 100_000, 6)
    model.set_probabilities(canopy_amd.synthetic_probabilities(20250704, 100_000))
    want = gpu_ctx.quantify(model)
    model.close()
    assert (r["n_kept"], r["argmax_row"], r["rare_event"], r["mcub"]) == (want.n_kept, want.argmax_row, want.rare_event, want.mcub)
    assert r["kernel_ran"] == 5 and r["value"] > 0 and 0 < r["roofline"]["frac"] < 2


def random_coherent_mef(rng):
    """A random coherent tree (and / or / atleast, shared sub-gates, shared events, a house event now and then) as
    Open-PSA MEF text; gate g may only refer to gates with a larger index -> acyclic; gate 0 is the top."""
    n_events = int(rng.integers(2, 11))
    n_gates = int(rng.integers(1, 9))
    ev = ["e%d" % j for j in range(n_events)]
    houses = {"h_true": "true", "h_false": "false"} if rng.random() < 0.4 else {}
    text_gates = []
    for g in range(n_gates):
        pool = ev + ["g%d" % k for k in range(g + 1, n_gates)] + list(houses)
        k = int(rng.integers(2, min(5, len(pool)) + 1))
        args = list(rng.choice(pool, size=k, replace=False))
        if g + 1 < n_gates and "g%d" % (g + 1) not in args:
            args[0] = "g%d" % (g + 1)  # keeps every gate inside the top's cone
        kind = rng.choice(["and", "or", "atleast"])
        inner = "".join(ref(a) for a in args)
        if kind == "atleast":
            body = "<atleast min='%d'>%s</atleast>" % (int(rng.integers(1, len(args) + 1)), inner)
        else:
            body = "<%s>%s</%s>" % (kind, inner, kind)
        text_gates.append("<define-gate name='g%d'>%s</define-gate>" % (g, body))
    data = events(ev, 0.1) + "".join("<define-house-event name='%s'><constant value='%s'/></define-house-event>" % kv
                                     for kv in houses.items())
    return MEF % ("".join(text_gates), data)


def ref(name):
    if name.startswith("g"):
        return "<gate name='%s'/>" % name
    if name.startswith("h_"):
        return "<house-event name='%s'/>" % name
    return "<basic-event name='%s'/>" % name


@pytest.mark.parametrize("seed", range(60))
def test_random_coherent_trees_against_truth_tables(canopy_lib, seed):
    """MOCUS-style lowering vs brute force: the minimal cut sets are the minimal TRUE assignments of the truth table
    of the exported gate DAG (which itself is checked against the written tree by the fixed cases above)."""
    from canopy_amd import _capi
    from canopy_amd.fault_tree import FaultTree
    rng = np.random.default_rng(77 + seed)
    text = random_coherent_mef(rng)
    try:
        ft = FaultTree(xml_text=text)
    except _capi.CanopyError as e:  # a tree whose top folds to a constant has no gate left: legal rejection
        assert e.status in (_capi.ERR_INVALID_MODEL, _capi.ERR_UNSUPPORTED), text
        return
    try:
        dag = ft.dag()
    except _capi.CanopyError as e:  # the house events decide the top event: constant FALSE (no cut set) or TRUE (the empty one)
        assert e.status == _capi.ERR_UNSUPPORTED
        row_ptr, col = ft.cut_sets()
        assert len(row_ptr) - 1 in (0, 1) and len(col) == 0
        return
    p = ft.probabilities()
    want_sets, _ = brute_force(dag, p)
    row_ptr, col = ft.cut_sets()
    assert csr_rows(row_ptr, col) == want_sets, text
    for i in range(len(row_ptr) - 1):
        assert np.all(np.diff(col[row_ptr[i]:row_ptr[i + 1]]) > 0)
    lo = ft.cut_sets(limit_order=2)
    assert csr_rows(*lo) == [s for s in want_sets if len(s) <= 2]


@pytest.mark.gpu
def test_c_example_ticket_sequence(canopy_lib, gpu_ctx, tmp_path):
    """examples/ticket_sequence.c: a C99 consumer of canopy.h built with gcc -- a new probability vector per pass, results by
    ticket with up to three passes in flight -- against the same sequence through the Python binding, one pass at a time."""
    import json
    import subprocess
    import numpy as np
    import canopy_amd
    exe = tmp_path / "ticket_sequence"
    lib_dir = os.path.join(ROOT, "canopy_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "ticket_sequence.c"), "-L", lib_dir, "-lcanopy", "-Wl,-rpath," + lib_dir, "-lm",
                    "-o", str(exe)], check=True)
    rows, events, n = 1_500_000, 30_000, 9
    out = subprocess.run([str(exe), str(rows), str(events), str(n)], check=True, capture_output=True, text=True).stdout
    got = json.loads(out)["results"]
    assert len(got) == n
    model = gpu_ctx.synthetic(20250704, rows, events, 6)
    base = canopy_amd.synthetic_probabilities(20250704, events)
    for k in range(n):
        model.set_probabilities(np.minimum(1.0, base * (0.5 + k / (n - 1))))
        want = gpu_ctx.quantify(model)
        assert (got[k]["n_kept"], got[k]["rare_event"], got[k]["mcub"]) == (want.n_kept, want.rare_event, want.mcub), k
    model.close()
