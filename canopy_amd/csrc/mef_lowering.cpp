// mef_lowering.cpp -- K0: Open-PSA MEF (fault-tree subset) -> gate DAG and minimal cut sets. Host only.
//
// Written from the Open-PSA Model Exchange Format itself; no text of the reference's GPL-3
// src/mef/openpsa is used. What is mirrored from the reference is behaviour only: house events act as
// Boolean constants (event.h:34-57), the top event is the unreferenced gate (fault_tree.h:151-186),
// a basic event's probability is its constant expression's value (event.h:93-96), connective names
// are those of event.h:160-162 ("and", "or", "atleast", "xor", "not", "nand", "nor", "null").
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <new>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "canopy.h"
#include "canopy_guard.h"

namespace {

// ---- a minimal XML reader: elements, attributes, text ignored, comments / PIs / DOCTYPE skipped ----
struct XmlNode {
    std::string name;
    std::map<std::string, std::string> attr;
    std::vector<XmlNode> kids;
};

struct XmlParser {
    const char* p;
    const char* end;
    explicit XmlParser(const char* b, size_t n) : p(b), end(b + n) {}
    [[noreturn]] void fail(const std::string& m) { throw std::runtime_error("XML: " + m); }
    bool starts(const char* s) const { size_t n = strlen(s); return (size_t)(end - p) >= n && memcmp(p, s, n) == 0; }
    void skip_ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    void skip_until(const char* s) {
        size_t n = strlen(s);
        while (p < end && !starts(s)) ++p;
        if (p >= end) fail(std::string("unterminated construct, expected ") + s);
        p += n;
    }
    void skip_misc() { // whitespace, text, comments, processing instructions, DOCTYPE
        for (;;) {
            while (p < end && *p != '<') ++p;
            if (p >= end) return;
            if (starts("<!--")) skip_until("-->");
            else if (starts("<?")) skip_until("?>");
            else if (starts("<!")) skip_until(">");
            else return;
        }
    }
    std::string name_tok() {
        const char* b = p;
        while (p < end && (isalnum((unsigned char)*p) || *p == '-' || *p == '_' || *p == ':' || *p == '.')) ++p;
        if (p == b) fail("name expected");
        return std::string(b, p);
    }
    static constexpr int kMaxDepth = 256; // untrusted input: the recursion below must not exhaust the stack
    int depth = 0;
    struct DepthGuard { int& d; explicit DepthGuard(int& x) : d(x) { ++d; } ~DepthGuard() { --d; } };
    XmlNode element() { // p at '<' of a start tag
        DepthGuard guard(depth);
        if (depth > kMaxDepth) fail("elements nested deeper than 256 levels");
        ++p;
        XmlNode n;
        n.name = name_tok();
        for (;;) {
            skip_ws();
            if (p >= end) fail("unterminated tag <" + n.name);
            if (starts("/>")) { p += 2; return n; }
            if (*p == '>') { ++p; break; }
            std::string key = name_tok();
            skip_ws();
            if (p >= end || *p != '=') fail("'=' expected after attribute " + key);
            ++p;
            skip_ws();
            if (p >= end || (*p != '"' && *p != '\'')) fail("quoted value expected for attribute " + key);
            const char q = *p++;
            const char* b = p;
            while (p < end && *p != q) ++p;
            if (p >= end) fail("unterminated attribute value");
            n.attr[key] = std::string(b, p);
            ++p;
        }
        for (;;) { // children until the matching end tag
            skip_misc();
            if (p >= end) fail("missing </" + n.name + ">");
            if (starts("</")) {
                p += 2;
                std::string close = name_tok();
                if (close != n.name) fail("</" + close + "> closes <" + n.name + ">");
                skip_ws();
                if (p >= end || *p != '>') fail("'>' expected");
                ++p;
                return n;
            }
            n.kids.push_back(element());
        }
    }
    XmlNode document() {
        skip_misc();
        if (p >= end) fail("no root element");
        return element();
    }
};

// ---- the fault tree ---------------------------------------------------------------------------------
enum NodeKind { kBasic, kHouse, kGate };
struct Ref { NodeKind kind; int index; bool neg; };
struct Gate { std::string name; int type = CANOPY_GATE_NULL; int min = 0; std::vector<Ref> args; bool defined = false; };

} // namespace

struct canopy_ft {
    std::vector<std::string> basic_names;
    std::vector<double> p;
    std::vector<bool> basic_defined;
    std::vector<std::string> house_names;
    std::vector<bool> house_value;
    std::vector<Gate> gates;
    std::map<std::string, int> basic_ix, house_ix, gate_ix;
    int top = -1;
    // folded DAG (house events removed, topological order)
    std::vector<int> d_type, d_min, d_ptr, d_args;
    std::vector<int> dag_of_gate; // gate index -> dag gate index, -1 constant false, -2 constant true
    int d_top = 0;
    bool top_const = false, top_const_value = false;
};

namespace {

int connective_of(const std::string& n) {
    if (n == "and") return CANOPY_GATE_AND;
    if (n == "or") return CANOPY_GATE_OR;
    if (n == "atleast") return CANOPY_GATE_ATLEAST;
    if (n == "xor") return CANOPY_GATE_XOR;
    if (n == "not") return CANOPY_GATE_NOT;
    if (n == "nand") return CANOPY_GATE_NAND;
    if (n == "nor") return CANOPY_GATE_NOR;
    return -1;
}

struct Builder {
    canopy_ft& ft;
    explicit Builder(canopy_ft& f) : ft(f) {}

    int gate_id(const std::string& name) {
        auto it = ft.gate_ix.find(name);
        if (it != ft.gate_ix.end()) return it->second;
        ft.gates.push_back(Gate());
        ft.gates.back().name = name;
        return ft.gate_ix[name] = (int)ft.gates.size() - 1;
    }
    int basic_id(const std::string& name) {
        auto it = ft.basic_ix.find(name);
        if (it != ft.basic_ix.end()) return it->second;
        ft.basic_names.push_back(name);
        ft.p.push_back(0.0);
        ft.basic_defined.push_back(false);
        return ft.basic_ix[name] = (int)ft.basic_names.size() - 1;
    }
    int house_id(const std::string& name) {
        auto it = ft.house_ix.find(name);
        if (it != ft.house_ix.end()) return it->second;
        ft.house_names.push_back(name);
        ft.house_value.push_back(false);
        return ft.house_ix[name] = (int)ft.house_names.size() - 1;
    }
    static const std::string& need(const XmlNode& n, const char* key) {
        auto it = n.attr.find(key);
        if (it == n.attr.end()) throw std::runtime_error("<" + n.name + "> lacks attribute '" + key + "'");
        return it->second;
    }
    // an argument of a formula: <gate name/>, <basic-event name/>, <house-event name/>, <event name [type]/>,
    // or <not> around one of them
    Ref arg_ref(const XmlNode& n, std::vector<std::pair<std::string, Ref*>>& untyped) {
        (void)untyped;
        if (n.name == "gate") return Ref{kGate, gate_id(need(n, "name")), false};
        if (n.name == "basic-event") return Ref{kBasic, basic_id(need(n, "name")), false};
        if (n.name == "house-event") return Ref{kHouse, house_id(need(n, "name")), false};
        if (n.name == "event") {
            auto t = n.attr.find("type");
            const std::string& nm = need(n, "name");
            if (t != n.attr.end()) {
                if (t->second == "gate") return Ref{kGate, gate_id(nm), false};
                if (t->second == "house-event") return Ref{kHouse, house_id(nm), false};
                return Ref{kBasic, basic_id(nm), false};
            }
            // untyped: resolved after all definitions are known (gate > house event > basic event)
            return Ref{kBasic, -1 - (int)pending.size(), (pending.push_back(nm), false)};
        }
        if (n.name == "not") {
            if (n.kids.size() != 1) throw std::runtime_error("<not> needs exactly one argument");
            Ref r = arg_ref(n.kids[0], untyped);
            r.neg = !r.neg;
            return r;
        }
        throw std::runtime_error("nested formula <" + n.name + "> inside a gate is not supported: name it as its own gate");
    }
    std::vector<std::string> pending; // names of untyped <event> references

    void define_gate(const XmlNode& n) {
        const int g = gate_id(need(n, "name"));
        Gate& G = ft.gates[(size_t)g];
        if (G.defined) throw std::runtime_error("gate '" + G.name + "' is defined twice");
        const XmlNode* f = nullptr;
        for (const XmlNode& k : n.kids)
            if (k.name != "label" && k.name != "attributes") { if (f) throw std::runtime_error("gate '" + G.name + "' has more than one formula"); f = &k; }
        if (!f) throw std::runtime_error("gate '" + G.name + "' has no formula");
        std::vector<std::pair<std::string, Ref*>> untyped;
        std::vector<Ref> args;
        int type = connective_of(f->name), mn = 0;
        if (type < 0) { // a bare reference: pass-through ("null")
            type = CANOPY_GATE_NULL;
            args.push_back(arg_ref(*f, untyped));
        } else if (type == CANOPY_GATE_NOT && f->kids.size() == 1 && connective_of(f->kids[0].name) < 0) {
            args.push_back(arg_ref(f->kids[0], untyped));
        } else {
            if (type == CANOPY_GATE_ATLEAST) mn = atoi(need(*f, "min").c_str());
            for (const XmlNode& k : f->kids) args.push_back(arg_ref(k, untyped));
        }
        Gate& G2 = ft.gates[(size_t)g]; // gate_id may have grown the vector
        G2.type = type; G2.min = mn; G2.args = args; G2.defined = true;
        if (args.empty()) throw std::runtime_error("gate '" + G2.name + "' has no arguments");
        if ((type == CANOPY_GATE_NOT || type == CANOPY_GATE_NULL) && args.size() != 1) throw std::runtime_error("gate '" + G2.name + "': one argument expected");
        if (type == CANOPY_GATE_ATLEAST && (mn < 1 || mn > (int)args.size())) throw std::runtime_error("gate '" + G2.name + "': atleast min out of range");
    }
    void define_basic(const XmlNode& n) {
        const int j = basic_id(need(n, "name"));
        for (const XmlNode& k : n.kids) {
            if (k.name == "float" || k.name == "int") {
                const double v = atof(need(k, "value").c_str());
                if (!(v >= 0.0 && v <= 1.0)) throw std::runtime_error("basic event '" + ft.basic_names[(size_t)j] + "': probability outside [0,1]");
                ft.p[(size_t)j] = v;
                ft.basic_defined[(size_t)j] = true;
            } else if (k.name != "label" && k.name != "attributes") {
                throw std::runtime_error("basic event '" + ft.basic_names[(size_t)j] + "': only constant <float value=.../> expressions are supported");
            }
        }
    }
    void define_house(const XmlNode& n) {
        const int h = house_id(need(n, "name"));
        for (const XmlNode& k : n.kids)
            if (k.name == "constant") ft.house_value[(size_t)h] = (need(k, "value") == "true");
    }
    void walk(const XmlNode& n) {
        for (const XmlNode& k : n.kids) {
            if (k.name == "define-gate") define_gate(k);
            else if (k.name == "define-basic-event") define_basic(k);
            else if (k.name == "define-house-event") define_house(k);
            else if (k.name == "define-fault-tree" || k.name == "define-component" || k.name == "model-data") walk(k);
        }
    }
    void resolve_untyped() {
        for (Gate& G : ft.gates)
            for (Ref& r : G.args)
                if (r.kind == kBasic && r.index < 0) {
                    const std::string& nm = pending[(size_t)(-1 - r.index)];
                    if (ft.gate_ix.count(nm)) { r.kind = kGate; r.index = ft.gate_ix[nm]; }
                    else if (ft.house_ix.count(nm)) { r.kind = kHouse; r.index = ft.house_ix[nm]; }
                    else r.index = basic_id(nm);
                }
    }
};

// fold house events, order gates topologically, detect cycles
void finalize(canopy_ft& ft) {
    const int n = (int)ft.gates.size();
    if (n == 0) throw std::runtime_error("the model defines no gate");
    for (const Gate& G : ft.gates)
        if (!G.defined) throw std::runtime_error("gate '" + G.name + "' is referenced but never defined");
    for (size_t j = 0; j < ft.basic_names.size(); ++j)
        if (!ft.basic_defined[j]) throw std::runtime_error("basic event '" + ft.basic_names[j] + "' has no probability");
    std::vector<int> refd((size_t)n, 0);
    for (const Gate& G : ft.gates) for (const Ref& r : G.args) if (r.kind == kGate) refd[(size_t)r.index] = 1;
    int top = -1;
    for (int g = 0; g < n; ++g) if (!refd[(size_t)g]) { if (top >= 0) throw std::runtime_error("more than one top gate ('" + ft.gates[(size_t)top].name + "', '" + ft.gates[(size_t)g].name + "')"); top = g; }
    if (top < 0) throw std::runtime_error("no top gate: every gate is referenced by another one (cycle)");
    ft.top = top;
    // depth-first post-order from the top = topological order; state 1 = on the stack (cycle)
    std::vector<int> state((size_t)n, 0);
    ft.dag_of_gate.assign((size_t)n, -3);
    const int E = (int)ft.basic_names.size();
    struct Frame { int g; size_t k; };
    std::vector<Frame> st;
    st.push_back({top, 0});
    state[(size_t)top] = 1;
    ft.d_ptr.assign(1, 0);
    while (!st.empty()) {
        Frame& f = st.back();
        const Gate& G = ft.gates[(size_t)f.g];
        if (f.k < G.args.size()) {
            const Ref& r = G.args[f.k++];
            if (r.kind == kGate) {
                if (state[(size_t)r.index] == 1) throw std::runtime_error("cycle through gate '" + ft.gates[(size_t)r.index].name + "'");
                if (state[(size_t)r.index] == 0) { state[(size_t)r.index] = 1; st.push_back({r.index, 0}); }
            }
            continue;
        }
        // all arguments ready: fold constants
        std::vector<int> a; // dag node ids (negative = complemented)
        int n_true = 0, n_false = 0;
        for (const Ref& r : G.args) {
            int cst = -1; // -1 not a constant, 0 false, 1 true
            int id = 0;
            if (r.kind == kHouse) cst = ft.house_value[(size_t)r.index] ? 1 : 0;
            else if (r.kind == kBasic) id = r.index;
            else { const int dg = ft.dag_of_gate[(size_t)r.index]; if (dg == -1) cst = 0; else if (dg == -2) cst = 1; else id = E + dg; }
            if (cst >= 0) { if ((cst == 1) != r.neg) ++n_true; else ++n_false; }
            else a.push_back(r.neg ? ~id : id);
        }
        int type = G.type, mn = G.min, result = -3; // -1 false, -2 true, -3 a real gate
        bool invert = false;
        if (type == CANOPY_GATE_NAND) { type = CANOPY_GATE_AND; invert = true; }
        if (type == CANOPY_GATE_NOR) { type = CANOPY_GATE_OR; invert = true; }
        if (type == CANOPY_GATE_NOT) { type = CANOPY_GATE_NULL; invert = true; }
        if (type == CANOPY_GATE_AND) { if (n_false) result = -1; else if (a.empty()) result = -2; }
        else if (type == CANOPY_GATE_OR) { if (n_true) result = -2; else if (a.empty()) result = -1; }
        else if (type == CANOPY_GATE_ATLEAST) { mn -= n_true; if (mn <= 0) result = -2; else if (mn > (int)a.size()) result = -1; }
        else if (type == CANOPY_GATE_XOR) { if (a.empty()) result = (n_true & 1) ? -2 : -1; else if (n_true & 1) invert = !invert; }
        else { if (a.empty()) result = n_true ? -2 : -1; } // NULL
        if (result != -3) {
            if (invert) result = (result == -1) ? -2 : -1;
            ft.dag_of_gate[(size_t)f.g] = result;
        } else {
            int out_type = type, out_min = mn;
            if (invert) { // express the inversion with the complement connectives
                if (type == CANOPY_GATE_AND) out_type = CANOPY_GATE_NAND;
                else if (type == CANOPY_GATE_OR) out_type = CANOPY_GATE_NOR;
                else if (type == CANOPY_GATE_NULL) out_type = CANOPY_GATE_NOT;
                else if (type == CANOPY_GATE_XOR) { a[0] = ~a[0]; } // flipping one input flips a parity
                else throw std::runtime_error("internal: inverted atleast");
            }
            ft.dag_of_gate[(size_t)f.g] = (int)ft.d_type.size();
            ft.d_type.push_back(out_type);
            ft.d_min.push_back(out_type == CANOPY_GATE_ATLEAST ? out_min : 0);
            for (int x : a) ft.d_args.push_back(x);
            ft.d_ptr.push_back((int)ft.d_args.size());
        }
        state[(size_t)f.g] = 2;
        st.pop_back();
    }
    const int dt = ft.dag_of_gate[(size_t)top];
    if (dt < 0) { ft.top_const = true; ft.top_const_value = (dt == -2); ft.d_top = 0; }
    else ft.d_top = E + dt;
}

// ---- MOCUS-style expansion --------------------------------------------------------------------------
typedef std::vector<int> CutSet; // sorted node ids of the folded DAG (events < E, gates >= E)

void combinations(const std::vector<int>& a, int k, size_t start, std::vector<int>& cur, std::vector<std::vector<int>>& out) {
    if ((int)cur.size() == k) { out.push_back(cur); return; }
    for (size_t i = start; i < a.size(); ++i) { cur.push_back(a[i]); combinations(a, k, i + 1, cur, out); cur.pop_back(); }
}

std::vector<CutSet> cut_sets_of(const canopy_ft& ft, int limit_order) {
    const int E = (int)ft.basic_names.size();
    std::vector<CutSet> result;
    if (ft.top_const) { if (ft.top_const_value) result.push_back(CutSet()); return result; }
    for (int t : ft.d_type)
        if (t != CANOPY_GATE_AND && t != CANOPY_GATE_OR && t != CANOPY_GATE_ATLEAST && t != CANOPY_GATE_NULL)
            throw std::invalid_argument("the tree is not coherent (negation / xor): cut sets are not defined, use the Monte-Carlo DAG");
    for (int x : ft.d_args) if (x < 0) throw std::invalid_argument("the tree has complemented arguments: use the Monte-Carlo DAG");
    std::set<CutSet> done;
    std::set<CutSet> seen; // partial sets already queued: shared sub-gates reach the same set along many paths
    std::vector<CutSet> work(1, CutSet(1, ft.d_top));
    seen.insert(work[0]);
    size_t guard = 0;
    // a partial set that already contains a finished cut set can only yield supersets of it: prune
    auto dominated = [&](const CutSet& n) {
        if (done.size() > 4096) return false; // keep the check cheap on big outputs
        for (const CutSet& m : done)
            if (m.size() <= n.size() && std::includes(n.begin(), n.end(), m.begin(), m.end())) return true;
        return false;
    };
    while (!work.empty()) {
        if (++guard > 4000000 || done.size() > 2000000) throw std::length_error("cut-set expansion exceeds the small-tree budget");
        CutSet cs = work.back();
        work.pop_back();
        size_t gi = cs.size();
        for (size_t i = 0; i < cs.size(); ++i) if (cs[i] >= E) { gi = i; break; }
        if (gi == cs.size()) { if (!dominated(cs)) done.insert(cs); continue; }
        const int g = cs[gi] - E;
        cs.erase(cs.begin() + (long)gi);
        std::vector<int> a(ft.d_args.begin() + ft.d_ptr[(size_t)g], ft.d_args.begin() + ft.d_ptr[(size_t)g + 1]);
        std::vector<std::vector<int>> alts;
        const int type = ft.d_type[(size_t)g];
        if (type == CANOPY_GATE_AND || type == CANOPY_GATE_NULL) alts.push_back(a);
        else if (type == CANOPY_GATE_OR) for (int x : a) alts.push_back(std::vector<int>(1, x));
        else { std::vector<int> cur; combinations(a, ft.d_min[(size_t)g], 0, cur, alts); }
        for (const std::vector<int>& alt : alts) {
            CutSet n = cs;
            n.insert(n.end(), alt.begin(), alt.end());
            std::sort(n.begin(), n.end());
            n.erase(std::unique(n.begin(), n.end()), n.end());
            int n_events = 0;
            for (int x : n) n_events += (x < E);
            if (limit_order > 0 && n_events > limit_order) continue; // only grows from here
            if (!seen.insert(n).second || dominated(n)) continue;
            work.push_back(n);
        }
    }
    // minimise: drop supersets (sets sorted by size, then lexicographically -> deterministic row order)
    std::vector<CutSet> all(done.begin(), done.end());
    std::stable_sort(all.begin(), all.end(), [](const CutSet& x, const CutSet& y) { return x.size() != y.size() ? x.size() < y.size() : x < y; });
    for (const CutSet& c : all) {
        bool minimal = true;
        for (const CutSet& m : result)
            if (m.size() < c.size() && std::includes(c.begin(), c.end(), m.begin(), m.end())) { minimal = false; break; }
        if (minimal) result.push_back(c);
    }
    return result;
}

void set_err(char* err, size_t n, const std::string& m) {
    if (err && n) { snprintf(err, n, "%s", m.c_str()); }
}

} // namespace

extern "C" {

canopy_status canopy_ft_parse(const char* xml_text, size_t len, canopy_ft** out, char* err, size_t err_len) try {
    if (!xml_text || !out) return CANOPY_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    canopy_ft* ft = new (std::nothrow) canopy_ft();
    if (!ft) return CANOPY_ERR_OUT_OF_MEMORY;
    try {
        XmlParser xp(xml_text, len);
        XmlNode root = xp.document();
        if (root.name != "opsa-mef") throw std::runtime_error("root element is <" + root.name + ">, expected <opsa-mef>");
        Builder b(*ft);
        b.walk(root);
        b.resolve_untyped();
        finalize(*ft);
    } catch (const std::exception& e) {
        set_err(err, err_len, e.what());
        delete ft;
        return CANOPY_ERR_INVALID_MODEL;
    }
    *out = ft;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_ft_parse_file(const char* path, canopy_ft** out, char* err, size_t err_len) try {
    if (!path || !out) return CANOPY_ERR_INVALID_ARGUMENT;
    FILE* f = fopen(path, "rb");
    if (!f) { set_err(err, err_len, std::string("cannot open ") + path); return CANOPY_ERR_INVALID_ARGUMENT; }
    std::string text;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
    fclose(f);
    return canopy_ft_parse(text.data(), text.size(), out, err, err_len);
} CANOPY_CATCH_ALL

void canopy_ft_free(canopy_ft* ft) { delete ft; }

canopy_status canopy_ft_get_info(const canopy_ft* ft, canopy_ft_info* info) try {
    if (!ft || !info) return CANOPY_ERR_INVALID_ARGUMENT;
    info->n_basic_events = (int32_t)ft->basic_names.size();
    info->n_gates = (int32_t)ft->d_type.size();
    info->n_house_events = (int32_t)ft->house_names.size();
    info->top_gate = ft->top;
    info->n_dag_args = (int32_t)ft->d_args.size();
    info->reserved = ft->top_const ? (ft->top_const_value ? 2 : 1) : 0; /* 0 regular, 1 top == false, 2 top == true */
    return CANOPY_OK;
} CANOPY_CATCH_ALL

const char* canopy_ft_basic_event_name(const canopy_ft* ft, int32_t j) {
    return (ft && j >= 0 && (size_t)j < ft->basic_names.size()) ? ft->basic_names[(size_t)j].c_str() : "";
}
const char* canopy_ft_gate_name(const canopy_ft* ft, int32_t g) {
    return (ft && g >= 0 && (size_t)g < ft->gates.size()) ? ft->gates[(size_t)g].name.c_str() : "";
}

canopy_status canopy_ft_probabilities(const canopy_ft* ft, double* p) try {
    if (!ft || (!p && !ft->p.empty())) return CANOPY_ERR_INVALID_ARGUMENT;
    for (size_t j = 0; j < ft->p.size(); ++j) p[j] = ft->p[j];
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_ft_dag(const canopy_ft* ft, int32_t* gate_type, int32_t* gate_min, int32_t* arg_ptr, int32_t* args, int32_t* top) try {
    if (!ft || !arg_ptr || !top) return CANOPY_ERR_INVALID_ARGUMENT;
    if (ft->top_const) return CANOPY_ERR_UNSUPPORTED; // the house events decide the top event: nothing to simulate
    for (size_t g = 0; g < ft->d_type.size(); ++g) { gate_type[g] = ft->d_type[g]; gate_min[g] = ft->d_min[g]; }
    for (size_t g = 0; g < ft->d_ptr.size(); ++g) arg_ptr[g] = ft->d_ptr[g];
    for (size_t k = 0; k < ft->d_args.size(); ++k) args[k] = ft->d_args[k];
    *top = ft->d_top;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_ft_cut_sets(const canopy_ft* ft, int32_t limit_order, int64_t* n_sets, int64_t* nnz, int64_t* row_ptr, int32_t* col_idx) try {
    if (!ft || !n_sets || !nnz || limit_order < 0) return CANOPY_ERR_INVALID_ARGUMENT;
    std::vector<CutSet> cs;
    try {
        cs = cut_sets_of(*ft, limit_order);
    } catch (const std::invalid_argument&) {
        return CANOPY_ERR_UNSUPPORTED;
    } catch (const std::exception&) {
        return CANOPY_ERR_UNSUPPORTED;
    }
    int64_t total = 0;
    for (const CutSet& c : cs) total += (int64_t)c.size();
    *n_sets = (int64_t)cs.size();
    *nnz = total;
    if (row_ptr && (col_idx || total == 0)) {
        int64_t k = 0;
        row_ptr[0] = 0;
        for (size_t i = 0; i < cs.size(); ++i) {
            for (int x : cs[i]) col_idx[k++] = x; // ascending basic-event indices
            row_ptr[i + 1] = k;
        }
    }
    return CANOPY_OK;
} CANOPY_CATCH_ALL

} // extern "C"
