// kernels_mc.hip -- K6: direct Monte-Carlo propagation through the AND/OR/K-of-N gate DAG with
// bit-packed samples (BASELINE config 5), gfx950 / wave64 only, plus its C ABI entry points.
//
// Reference side: only the vocabulary exists (Approximation::kMonteCarlo settings.h:19; num_trials,
// seed settings.h:324-325; Connective event.h:143-157; Formula args with complement formula.h:6-148).
//
// A workgroup (4 waves) owns a TILE of W consecutive 64-sample words; the node values of the tile live in
// LDS as node[id][W] u64. Per tile:
//   phase 1  Bernoulli words. Sample l of word w of event j owns a 53-bit uniform u whose bit (52-k) is
//            bit l of the plane R_k = mix64(ek_j + 64 w + k); TRUE iff u < ceil(p_j 2^53). A lane resolves
//            all 64 samples of one (event, word) item together, plane by plane from the most significant
//            bit: `undecided` lanes shrink by half per plane, so an item costs ~7 hashes instead of the
//            64 of a hash per sample. Lanes pull items in a strided loop and refill as they finish.
//   phase 2  level by level (workgroup barrier between levels), lane per (gate, word pair): bitwise
//            AND / OR / XOR / NOT on 16-byte node reads, K-of-N through a bit-sliced adder + comparator.
//            The gate descriptors are read once per tile (not once per word).
//   hits += popcount(top words & valid-sample mask). Integer arithmetic only -> bit-exact vs the oracle.
// Compute / RNG bound (vector ALU: 64-bit multiplies of the hash), not HBM bound: nothing is streamed.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <new>
#include <vector>

#include "canopy_device.h"
#include "canopy_internal.h"

#ifdef CANOPY_ABLATIONS // CANOPY_MC_DEBUG: timing ablations of benchmarks/mc_ab.sh (wrong results)
#define MC_ABL(bit) (a.abl & (bit))
#else
#define MC_ABL(bit) 0
#endif

using namespace canopy;

namespace {

constexpr uint64_t kStreamMc = 4;
constexpr int kMcThreads = 512; // 8 waves: one pass per DAG level for a few hundred gates x W/2 word pairs
constexpr int kMaxAtleastArgs = 63;

struct McArgs {
    const int4* desc;        // level-ordered gates: {type | K << 8, output node, arg begin, arg end}
    const int32_t* args;
    const int32_t* level_ptr;
    const uint64_t* thresh;
    int32_t n_events, n_gates, n_levels, top, n_args, n_slots;
    uint64_t key;          // stream_key(seed, kStreamMc)
    int64_t first_word;    // first_sample / 64
    int64_t n_words;       // ceil(num_trials / 64)
    int64_t num_trials;
    unsigned long long* hits;
    int32_t vote_planes; // 3 when every K-of-N gate has <= 7 arguments, else 6
    int32_t p1_lanes; // lanes of the workgroup that draw samples (a multiple of 64, <= kMcThreads)
    int32_t abl; // timing ablations (CANOPY_MC_DEBUG, wrong results): 1 = no hashing, 2 = no gate levels
};

template <int WL>
struct Words {
    uint64_t v[WL];
};

// the words of argument `a` (~id = complemented), XOR-ed with `flip` (all ones: complement once more)
template <int W, int WL>
__device__ __forceinline__ Words<WL> node_words(const uint64_t* node, int32_t a, int sub, uint64_t flip) {
    const int32_t sgn = a >> 31; // 0 or -1
    const int id = a ^ sgn;      // a >= 0 ? a : ~a
    const uint64_t m = (uint64_t)(int64_t)sgn ^ flip;
    Words<WL> r;
    const uint64_t* p = node + (size_t)id * W + sub * WL;
#pragma unroll
    for (int i = 0; i < WL; ++i) r.v[i] = p[i] ^ m;
    return r;
}

// bit-sliced counter of TRUE arguments, P planes (P = 3 counts to 7, P = 6 to 63)
template <int P>
__device__ __forceinline__ void vote_add(uint64_t (&c)[6], uint64_t carry) {
#pragma unroll
    for (int pl = 0; pl < P; ++pl) {
        const uint64_t tt = c[pl] & carry;
        c[pl] ^= carry;
        carry = tt;
    }
}
template <int P>
__device__ __forceinline__ uint64_t vote_ge(const uint64_t (&c)[6], int K) { // count >= K, per sample bit
    uint64_t gt = 0ULL, eq = ~0ULL;
#pragma unroll
    for (int pl = P - 1; pl >= 0; --pl) {
        const bool kb = ((K >> pl) & 1) != 0;
        gt |= kb ? 0ULL : (eq & c[pl]);
        eq &= kb ? c[pl] : ~c[pl];
    }
    return (K >> P) ? 0ULL : (gt | eq); // K beyond the counter's range: never reached
}

// DAG_LDS: gate descriptors, argument lists and the per-event draw keys / thresholds are copied into LDS
// once per workgroup, so the tile loop touches no global memory at all (with two waves per SIMD a chain of
// dependent global loads per level was the whole cost of the first version of this kernel).
template <int W, bool DAG_LDS>
__global__ __launch_bounds__(kMcThreads) void k_mc(McArgs a) {
    constexpr int WL = W >= 2 ? 2 : 1; // words per lane in phase 2
    constexpr int S = W / WL;          // lanes per gate
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* node = reinterpret_cast<uint64_t*>(smem); // [n_nodes][W]
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int n_nodes = a.n_slots; // live-range slots + the constant FALSE and TRUE nodes (argument padding)
    // LDS copies (DAG_LDS only): int4 desc[n_gates] | u64 ek[n_events] | u64 th[n_events] | i32 args[n_args]
    int4* s_desc = reinterpret_cast<int4*>(smem + (((size_t)n_nodes * W * 8 + 15) & ~(size_t)15));
    uint64_t* s_ek = reinterpret_cast<uint64_t*>(s_desc + a.n_gates);
    uint64_t* s_th = s_ek + a.n_events;
    int32_t* s_args = reinterpret_cast<int32_t*>(s_th + a.n_events);
    if (tid < 2 * W) node[(size_t)(a.n_slots - 2) * W + tid] = tid < W ? 0ULL : ~0ULL;
    if (DAG_LDS) {
        for (int i = tid; i < a.n_gates; i += kMcThreads) s_desc[i] = a.desc[i];
        for (int i = tid; i < a.n_args; i += kMcThreads) s_args[i] = a.args[i];
        for (int i = tid; i < a.n_events; i += kMcThreads) {
            s_ek[i] = mix64(a.key ^ (uint64_t)i);
            s_th[i] = a.thresh[i];
        }
    }
    __syncthreads();
    const int4* desc = DAG_LDS ? s_desc : a.desc;
    const int32_t* args = DAG_LDS ? s_args : a.args;
    unsigned long long my_hits = 0;
    const int64_t n_tiles = (a.n_words + W - 1) / W;
    const int n_items = a.n_events * W;
    for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int64_t w0 = t * W;
        // ---- phase 1: one lane resolves the 64 samples of an (event, word) item plane by plane
        {
            int item = tid, slot = 0, k = 0;
            uint64_t ctr = 0, T = 0, lt = 0, und = 0;
            bool active = tid < a.p1_lanes && item < n_items;
            bool fresh = active;
            while (__any(active)) {
                if (active) {
                    if (fresh) {
                        const int j = item / W, wi = item - j * W;
                        ctr = (DAG_LDS ? s_ek[j] : mix64(a.key ^ (uint64_t)j)) + (uint64_t)(a.first_word + w0 + wi) * 64u;
                        T = DAG_LDS ? s_th[j] : a.thresh[j];
                        const bool certain = (T >> 53) != 0 || MC_ABL(1); // p == 1
                        lt = certain ? ~0ULL : 0ULL;
                        und = certain ? 0ULL : ~0ULL;
                        k = 0;
                        slot = item;
                        fresh = false;
                    }
                    if (und != 0ULL && k < 53) {
                        const uint64_t r = mix64(ctr + (uint64_t)k);
                        const bool tb = ((T >> (52 - k)) & 1ULL) != 0;
                        lt |= tb ? (und & ~r) : 0ULL; // threshold bit 1, sample bit 0: below
                        und &= tb ? r : ~r;           // still equal to the threshold's prefix
                        ++k;
                    }
                    if (und == 0ULL || k == 53) { // k == 53 with lanes left: u == threshold, not below
                        node[slot] = lt;          // slot = j * W + wi = item
                        item += a.p1_lanes;
                        active = item < n_items;
                        fresh = active;
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 2: gates, level by level, lane per (gate, WL words)
        for (int lv = 0; lv < (MC_ABL(2) ? 0 : a.n_levels); ++lv) {
            const int g0 = a.level_ptr[lv], g1 = a.level_ptr[lv + 1];
            for (int it = tid; it < (g1 - g0) * S; it += kMcThreads) {
                const int g = g0 + it / S, sub = it % S;
                const int4 d = desc[g];
                const int type = d.x & 0xff, K = (d.x >> 8) & 0xff, b = d.z, e = d.w;
                // De Morgan: AND(x..) = ~OR(~x..). Every connective of the AND / OR families is ONE OR
                // accumulation over (argument ^ in_flip), XOR-ed with out_flip at the end; the host padded the
                // argument list to a multiple of 4 with the constant node that is neutral after in_flip, so
                // the loop carries no predicates. XOR and K-of-N gates (rare) add their own accumulators under
                // wave-uniform branches.
                const uint64_t in_flip = (d.x & (1 << 16)) ? ~0ULL : 0ULL, out_flip = (d.x & (1 << 17)) ? ~0ULL : 0ULL;
                const bool vote = type == CANOPY_GATE_ATLEAST, is_xor = type == CANOPY_GATE_XOR;
                const bool any_vote = __any(vote), any_xor = __any(is_xor);
                Words<WL> r;
                const int q_end = MC_ABL(8) ? b : e;
                if (!any_vote && !any_xor) { // the common wave: OR accumulation only, no branches in the loop
                    uint64_t ro[WL];
#pragma unroll
                    for (int i = 0; i < WL; ++i) ro[i] = 0ULL;
                    for (int q = b; q < q_end; q += 4) {
                        int32_t aa[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) aa[u] = args[q + u];
                        Words<WL> x[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) x[u] = node_words<W, WL>(node, aa[u], sub, in_flip);
#pragma unroll
                        for (int i = 0; i < WL; ++i) ro[i] |= (x[0].v[i] | x[1].v[i]) | (x[2].v[i] | x[3].v[i]);
                    }
#pragma unroll
                    for (int i = 0; i < WL; ++i) r.v[i] = ro[i] ^ out_flip;
                } else {
                    uint64_t ro[WL], rx[WL], c[WL][6];
#pragma unroll
                    for (int i = 0; i < WL; ++i) {
                        ro[i] = 0ULL; rx[i] = 0ULL;
#pragma unroll
                        for (int pl = 0; pl < 6; ++pl) c[i][pl] = 0ULL;
                    }
                    for (int q = b; q < q_end; q += 4) {
                        int32_t aa[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) aa[u] = args[q + u];
                        Words<WL> x[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) x[u] = node_words<W, WL>(node, aa[u], sub, in_flip);
#pragma unroll
                        for (int i = 0; i < WL; ++i) {
                            ro[i] |= (x[0].v[i] | x[1].v[i]) | (x[2].v[i] | x[3].v[i]);
                            rx[i] ^= (x[0].v[i] ^ x[1].v[i]) ^ (x[2].v[i] ^ x[3].v[i]);
                        }
                        if (any_vote) { // wave-uniform
                            if (a.vote_planes == 3) {
#pragma unroll
                                for (int u = 0; u < 4; ++u)
#pragma unroll
                                    for (int i = 0; i < WL; ++i) vote_add<3>(c[i], x[u].v[i]);
                            } else {
#pragma unroll
                                for (int u = 0; u < 4; ++u)
#pragma unroll
                                    for (int i = 0; i < WL; ++i) vote_add<6>(c[i], x[u].v[i]);
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < WL; ++i) {
                        uint64_t val = is_xor ? rx[i] : (ro[i] ^ out_flip);
                        if (any_vote) { // wave-uniform
                            const uint64_t ge = a.vote_planes == 3 ? vote_ge<3>(c[i], K) : vote_ge<6>(c[i], K);
                            if (vote) val = ge;
                        }
                        r.v[i] = val;
                    }
                }
                uint64_t* dst = node + (size_t)d.y * W + sub * WL;
#pragma unroll
                for (int i = 0; i < WL; ++i) dst[i] = r.v[i];
            }
            if (!MC_ABL(4)) __syncthreads();
        }
        // ---- top node
        if (tid < W) {
            const int64_t w = w0 + tid;
            const int64_t left = a.num_trials - w * 64;
            if (left > 0) {
                const uint64_t valid = left >= 64 ? ~0ULL : ((1ULL << left) - 1ULL);
                my_hits += (unsigned long long)__builtin_popcountll(node[(size_t)a.top * W + tid] & valid);
            }
        }
        __syncthreads(); // the next tile overwrites the node array
    }
    if (tid < kWave) { // only the first wave counted (W <= 64)
#pragma unroll
        for (int dlt = kWave / 2; dlt >= 1; dlt >>= 1) my_hits += __shfl_down(my_hits, dlt, kWave);
        if (lane == 0 && my_hits) atomicAdd(a.hits, my_hits); // integer: order-independent
    }
}

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            (void)hipGetLastError();                                                                    \
            return fail((ctx), _e == hipErrorOutOfMemory ? CANOPY_ERR_OUT_OF_MEMORY : CANOPY_ERR_HIP,   \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                             \
        }                                                                                               \
    } while (0)

void free_dag(canopy_dag* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    void* ptrs[] = {d->d_desc, d->d_args, d->d_level_ptr, d->d_thresh};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete d;
}

} // namespace

extern "C" {

canopy_status canopy_dag_upload(canopy_ctx* ctx, const canopy_dag_view* v, canopy_dag** out) try {
    if (!ctx || !v || !out) return CANOPY_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (v->n_events < 0 || v->n_gates < 0 || (v->n_gates > 0 && (!v->gate_type || !v->arg_ptr)))
        return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "bad dag view");
    const int32_t n_nodes = v->n_events + v->n_gates;
    if (v->top < 0 || v->top >= n_nodes) return fail(ctx, CANOPY_ERR_INVALID_MODEL, "top node id out of range");
    // validate on the host (the DAG is small: it is the model, not the sample stream)
    if (v->n_gates > 0 && v->arg_ptr[0] != 0) return fail(ctx, CANOPY_ERR_INVALID_MODEL, "arg_ptr[0] != 0");
    for (int32_t g = 0; g < v->n_gates; ++g) {
        const int32_t b = v->arg_ptr[g], e = v->arg_ptr[g + 1];
        const int32_t type = v->gate_type[g];
        if (e < b || (e > b && !v->args)) return fail(ctx, CANOPY_ERR_INVALID_MODEL, "arg_ptr not monotone");
        if (type < CANOPY_GATE_AND || type > CANOPY_GATE_NULL)
            return fail(ctx, CANOPY_ERR_UNSUPPORTED, "connective of gate " + std::to_string(g) + " is not supported");
        const int32_t n_args = e - b;
        if (((type == CANOPY_GATE_NOT || type == CANOPY_GATE_NULL) && n_args != 1) || n_args < 1)
            return fail(ctx, CANOPY_ERR_INVALID_MODEL, "gate " + std::to_string(g) + ": wrong number of arguments");
        if (type == CANOPY_GATE_ATLEAST) {
            // K above the argument count is legal (the gate is then constantly FALSE); the descriptor keeps 8 bits of K
            if (!v->gate_min || n_args > kMaxAtleastArgs || v->gate_min[g] < 0 || v->gate_min[g] > 255)
                return fail(ctx, CANOPY_ERR_UNSUPPORTED, "ATLEAST gate " + std::to_string(g) + ": need 0 <= K <= 255 and <= 63 arguments");
        }
        for (int32_t k = b; k < e; ++k) {
            const int32_t raw = v->args[k];
            const int32_t id = raw >= 0 ? raw : ~raw;
            if (id < 0 || id >= v->n_events + g) // an event, or a gate strictly before this one
                return fail(ctx, CANOPY_ERR_INVALID_MODEL, "gate " + std::to_string(g) + ": argument is not an earlier node");
        }
    }
    // Internal gate list. A lane evaluates one gate by walking its arguments one after the other, so a wide
    // gate (the OR over a whole layer at the top of a tree) would serialise hundreds of dependent LDS reads
    // in a single lane while the rest of the workgroup waits at the level barrier. AND / OR / XOR are
    // associative: gates of those families with more than kMaxFanIn arguments become a tree of helper gates
    // of at most kMaxFanIn arguments (NAND / NOR: helpers of the AND / OR family, negation at the root).
    // Same Boolean function, so the same sample bits; node ids are internal from here on.
    constexpr int32_t kMaxFanIn = 8;
    struct IGate { int32_t type, K; std::vector<int32_t> args; };
    std::vector<IGate> ig;
    ig.reserve((size_t)v->n_gates + 16);
    std::vector<int32_t> node_of((size_t)v->n_gates, 0); // external gate -> internal node id
    auto map_arg = [&](int32_t raw) {
        const int32_t id = raw >= 0 ? raw : ~raw;
        const int32_t m = id < v->n_events ? id : node_of[(size_t)(id - v->n_events)];
        return raw >= 0 ? m : ~m;
    };
    for (int32_t g = 0; g < v->n_gates; ++g) {
        const int32_t type = v->gate_type[g];
        std::vector<int32_t> cur;
        for (int32_t k = v->arg_ptr[g]; k < v->arg_ptr[g + 1]; ++k) cur.push_back(map_arg(v->args[k]));
        const bool assoc = type == CANOPY_GATE_AND || type == CANOPY_GATE_OR || type == CANOPY_GATE_XOR ||
                           type == CANOPY_GATE_NAND || type == CANOPY_GATE_NOR;
        if (assoc) {
            const int32_t base = type == CANOPY_GATE_NAND ? (int32_t)CANOPY_GATE_AND
                                 : type == CANOPY_GATE_NOR ? (int32_t)CANOPY_GATE_OR : type;
            while ((int32_t)cur.size() > kMaxFanIn) {
                std::vector<int32_t> next;
                for (size_t c0 = 0; c0 < cur.size(); c0 += kMaxFanIn) {
                    const size_t c1 = std::min(cur.size(), c0 + (size_t)kMaxFanIn);
                    if (c1 - c0 == 1) { next.push_back(cur[c0]); continue; }
                    ig.push_back({base, 0, std::vector<int32_t>(cur.begin() + (long)c0, cur.begin() + (long)c1)});
                    next.push_back(v->n_events + (int32_t)ig.size() - 1);
                }
                cur.swap(next);
            }
        }
        ig.push_back({type, (type == CANOPY_GATE_ATLEAST && v->gate_min) ? v->gate_min[g] : 0, cur});
        node_of[(size_t)g] = v->n_events + (int32_t)ig.size() - 1;
    }
    const int32_t n_igates = (int32_t)ig.size();
    int32_t max_vote_args = 0;
    for (const IGate& q : ig)
        if (q.type == CANOPY_GATE_ATLEAST) max_vote_args = std::max(max_vote_args, (int32_t)q.args.size());
    const int32_t top_node = v->top < v->n_events ? v->top : node_of[(size_t)(v->top - v->n_events)];
    std::vector<int32_t> level((size_t)n_igates, 0);
    int32_t n_levels = 0;
    for (int32_t g = 0; g < n_igates; ++g) {
        int32_t lv = 0;
        for (int32_t raw : ig[(size_t)g].args) {
            const int32_t id = raw >= 0 ? raw : ~raw;
            if (id >= v->n_events) lv = std::max(lv, level[(size_t)(id - v->n_events)] + 1);
        }
        level[(size_t)g] = lv;
        n_levels = std::max(n_levels, lv + 1);
    }
    // level-ordered gate list (stable inside a level)
    std::vector<int32_t> order((size_t)n_igates);
    for (int32_t g = 0; g < n_igates; ++g) order[(size_t)g] = g;
    // inside a level the K-of-N gates go last, so that most waves never run the vote counter
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
        if (level[(size_t)x] != level[(size_t)y]) return level[(size_t)x] < level[(size_t)y];
        return (ig[(size_t)x].type == CANOPY_GATE_ATLEAST) < (ig[(size_t)y].type == CANOPY_GATE_ATLEAST);
    });
    std::vector<int32_t> l_desc, l_args, level_ptr((size_t)n_levels + 1, 0);
    // LDS slots by liveness: a node's slot is recycled after the last level that reads it (the top node's
    // never), so a layered tree keeps its events plus about two layers resident instead of every gate --
    // which is what lets a tile hold twice as many sample words. A slot released after level L is handed out
    // at level L + 1 at the earliest: no gate of that level reads it, so writing it there is safe.
    const int32_t n_inodes = v->n_events + n_igates;
    std::vector<int32_t> last_use((size_t)n_inodes, -1), slot_of((size_t)n_inodes, -1);
    for (int32_t g = 0; g < n_igates; ++g)
        for (int32_t raw : ig[(size_t)g].args) {
            const int32_t id = raw >= 0 ? raw : ~raw;
            last_use[(size_t)id] = std::max(last_use[(size_t)id], level[(size_t)g]);
        }
    for (int32_t g = 0; g < n_igates; ++g) // a gate's slot lives at least through its own level
        last_use[(size_t)(v->n_events + g)] = std::max(last_use[(size_t)(v->n_events + g)], level[(size_t)g]);
    last_use[(size_t)top_node] = n_levels + 1; // read after the last level
    std::vector<std::vector<int32_t>> released((size_t)n_levels + 2);
    for (int32_t id = 0; id < n_inodes; ++id) {
        const int32_t lu = last_use[(size_t)id];
        if (lu <= n_levels) released[(size_t)(lu < 0 ? 0 : lu + 1)].push_back(id); // free from level lu + 1 on
    }
    for (int32_t j = 0; j < v->n_events; ++j) slot_of[(size_t)j] = j; // phase 1 writes event j to slot j
    int32_t n_slots = v->n_events;
    {
        std::vector<int32_t> free_slots; // kept sorted descending: pop_back() = smallest slot
        size_t oi = 0;
        for (int32_t lv = 0; lv < n_levels; ++lv) {
            for (int32_t id : released[(size_t)lv])
                if (slot_of[(size_t)id] >= 0) free_slots.push_back(slot_of[(size_t)id]);
            std::sort(free_slots.begin(), free_slots.end(), [](int32_t x, int32_t y) { return x > y; });
            for (; oi < order.size() && level[(size_t)order[oi]] == lv; ++oi) {
                int32_t sl;
                if (!free_slots.empty()) { sl = free_slots.back(); free_slots.pop_back(); }
                else sl = n_slots++;
                slot_of[(size_t)(v->n_events + order[oi])] = sl;
            }
        }
    }
    auto slot_arg = [&](int32_t raw) { return raw >= 0 ? slot_of[(size_t)raw] : ~slot_of[(size_t)~raw]; };
    const int32_t false_node = n_slots, true_node = n_slots + 1; // constants kept by the kernel
    for (int32_t g : order) {
        const int32_t b = (int32_t)l_args.size();
        const int32_t type = ig[(size_t)g].type;
        // AND family: arguments are complemented on the way in (in_flip), so its neutral padding is TRUE
        const bool in_flip = type == CANOPY_GATE_AND || type == CANOPY_GATE_NAND;
        const bool out_flip = type == CANOPY_GATE_AND || type == CANOPY_GATE_NOR || type == CANOPY_GATE_NOT;
        for (int32_t x : ig[(size_t)g].args) l_args.push_back(slot_arg(x));
        while ((l_args.size() - (size_t)b) % 4 != 0) l_args.push_back(in_flip ? true_node : false_node);
        l_desc.push_back(type | (ig[(size_t)g].K << 8) | ((int32_t)in_flip << 16) | ((int32_t)out_flip << 17));
        l_desc.push_back(slot_of[(size_t)(v->n_events + g)]);
        l_desc.push_back(b);
        l_desc.push_back((int32_t)l_args.size());
        level_ptr[(size_t)level[(size_t)g] + 1] += 1;
    }
    for (int32_t l = 0; l < n_levels; ++l) level_ptr[(size_t)l + 1] += level_ptr[(size_t)l];

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    canopy_dag* d = new (std::nothrow) canopy_dag();
    if (!d) return CANOPY_ERR_OUT_OF_MEMORY;
    d->device = ctx->device;
    d->n_events = v->n_events; d->n_gates = n_igates; d->n_levels = n_levels; d->top = slot_of[(size_t)top_node];
    d->n_slots = n_slots + 2; // + constant FALSE / TRUE
    d->n_args = (int32_t)l_args.size();
    d->vote_planes = max_vote_args <= 7 ? 3 : 6;
    auto up = [&](int32_t** dst, const std::vector<int32_t>& src) -> hipError_t {
        hipError_t e = hipMalloc(dst, sizeof(int32_t) * std::max<size_t>(src.size(), 1));
        if (e == hipSuccess && !src.empty())
            e = hipMemcpy(*dst, src.data(), sizeof(int32_t) * src.size(), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(reinterpret_cast<int32_t**>(&d->d_desc), l_desc);
    if (e == hipSuccess) e = up(&d->d_args, l_args);
    if (e == hipSuccess) e = up(&d->d_level_ptr, level_ptr);
    if (e == hipSuccess) e = hipMalloc(&d->d_thresh, sizeof(uint64_t) * (size_t)std::max(v->n_events, 1));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        free_dag(d);
        return fail(ctx, e == hipErrorOutOfMemory ? CANOPY_ERR_OUT_OF_MEMORY : CANOPY_ERR_HIP,
                    std::string("dag upload: ") + hipGetErrorString(e));
    }
    *out = d;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

void canopy_dag_free(canopy_dag* dag) { free_dag(dag); }

canopy_status canopy_dag_set_probabilities(canopy_ctx* ctx, canopy_dag* d, const double* p) try {
    if (!ctx || !d || (!p && d->n_events > 0)) return CANOPY_ERR_INVALID_ARGUMENT;
    std::vector<uint64_t> t((size_t)d->n_events);
    for (int32_t j = 0; j < d->n_events; ++j) {
        if (!(p[j] >= 0.0 && p[j] <= 1.0))
            return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "probability " + std::to_string(j) + " outside [0,1]");
        t[(size_t)j] = (uint64_t)ceil(p[j] * 9007199254740992.0); // p * 2^53 is exact; TRUE iff draw < threshold
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (d->n_events > 0)
        HIP_TRY(ctx, hipMemcpy(d->d_thresh, t.data(), sizeof(uint64_t) * t.size(), hipMemcpyHostToDevice));
    d->has_p = true;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

void canopy_mc_params_default(canopy_mc_params* p) {
    if (!p) return;
    p->seed = 0;           // reference settings.h:324
    p->first_sample = 0;
    p->num_trials = 1000;  // settings.h:325
    p->batch_size = 1;     // settings.h:326
    p->sample_size = 1;    // settings.h:327
}

canopy_status canopy_mc_run_async(canopy_ctx* ctx, const canopy_dag* d, const canopy_mc_params* prm) try {
    if (!ctx || !d || !prm) return CANOPY_ERR_INVALID_ARGUMENT;
    if (d->device != ctx->device) return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "dag lives on another device");
    if (!d->has_p) return fail(ctx, CANOPY_ERR_STATE, "canopy_dag_set_probabilities has not been called");
    if (prm->num_trials < 0 || prm->first_sample < 0 || (prm->first_sample & 63) != 0 || prm->batch_size < 0 || prm->sample_size < 0)
        return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "num_trials, batch_size, sample_size >= 0 and first_sample a non-negative multiple of 64");
    // samples of this call: num_trials batches of batch_size trials of sample_size samples (reference settings.h:325-327; 0 reads as 1)
    const __int128 wide = (__int128)prm->num_trials * (prm->batch_size > 0 ? prm->batch_size : 1) * (prm->sample_size > 0 ? prm->sample_size : 1);
    if (wide > ((__int128)1 << 62)) return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "num_trials * batch_size * sample_size exceeds 2^62");
    const int64_t total = (int64_t)wide;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_mc_hits) {
        HIP_TRY(ctx, hipMalloc(&ctx->d_mc_hits, sizeof(unsigned long long)));
        HIP_TRY(ctx, hipHostMalloc(&ctx->h_mc_hits, sizeof(unsigned long long)));
        HIP_TRY(ctx, hipEventCreate(&ctx->ev_mc[0]));
        HIP_TRY(ctx, hipEventCreate(&ctx->ev_mc[1]));
    }
    const size_t n_nodes = (size_t)d->n_slots; // LDS node slots (liveness-packed) incl. constant FALSE / TRUE
    // LDS plan: node[n_nodes][W] (+ the DAG structure when it fits). Prefer two workgroups per CU
    // (<= 78 KiB each) with the structure resident; then one workgroup; then structure left in global memory.
    const size_t dag_bytes = (size_t)d->n_gates * 16 + (size_t)d->n_events * 16 + (size_t)d->n_args * 4 + 16;
    int W = 0;
    bool dag_lds = true;
    for (size_t budget : {(size_t)78 * 1024, (size_t)156 * 1024}) {
        for (int w = 16; w >= 2 && !W; w >>= 1)
            if (n_nodes * (size_t)w * 8 + dag_bytes <= budget) W = w;
        if (W) break;
    }
    if (!W) {
        dag_lds = n_nodes * 8 + dag_bytes <= (size_t)156 * 1024;
        W = 1;
        if (!dag_lds)
            for (int w = 16; w >= 2; w >>= 1)
                if (n_nodes * (size_t)w * 8 <= (size_t)156 * 1024) { W = w; break; }
    }
    static const int w_env = CANOPY_TUNE_ENV("CANOPY_MC_W") ? atoi(CANOPY_TUNE_ENV("CANOPY_MC_W")) : 0; // tuning only
    if (dag_lds && (w_env == 1 || w_env == 2 || w_env == 4 || w_env == 8 || w_env == 16) &&
        n_nodes * (size_t)w_env * 8 + dag_bytes <= (size_t)156 * 1024)
        W = w_env;
    const size_t lds = std::max<size_t>(((n_nodes * (size_t)W * 8 + 15) & ~(size_t)15) + (dag_lds ? dag_bytes : 0), 16);
    if (lds > 160 * 1024) return fail(ctx, CANOPY_ERR_UNSUPPORTED, "gate DAG too large for the LDS node array (events + gates <= 20480)");
    McArgs a;
    a.desc = d->d_desc; a.args = d->d_args; a.level_ptr = d->d_level_ptr; a.thresh = d->d_thresh;
    a.n_events = d->n_events; a.n_gates = d->n_gates; a.n_levels = d->n_levels; a.top = d->top; a.n_args = d->n_args; a.n_slots = d->n_slots;
    a.key = stream_key(prm->seed, kStreamMc);
    a.first_word = prm->first_sample / 64;
    a.n_words = (total + 63) / 64;
    a.num_trials = total;
    a.hits = ctx->d_mc_hits;
    static const int mc_dbg = CANOPY_TUNE_ENV("CANOPY_MC_DEBUG") ? atoi(CANOPY_TUNE_ENV("CANOPY_MC_DEBUG")) : 0; // tuning only
    a.abl = mc_dbg;
    a.vote_planes = d->vote_planes;
    static const int p1_env = CANOPY_TUNE_ENV("CANOPY_MC_P1") ? atoi(CANOPY_TUNE_ENV("CANOPY_MC_P1")) : 0; // tuning only
    a.p1_lanes = (p1_env >= 64 && p1_env <= kMcThreads && p1_env % 64 == 0) ? p1_env : kMcThreads;
    const int64_t n_tiles = (a.n_words + W - 1) / W;
    const int64_t blocks_per_cu = std::max<int64_t>(1, std::min<int64_t>(4, (160 * 1024) / lds)); // 8 waves each
    int64_t grid = std::min<int64_t>(n_tiles, (int64_t)ctx->n_cu * blocks_per_cu);
    if (grid < 1) grid = 1;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_mc_hits, 0, sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_mc[0], ctx->stream));
// (the LDS attribute is per device: set on every launch, no process-wide flag)
#define MC_GO(WW, LL)                                                                                                      \
    do {                                                                                                                   \
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mc<WW, LL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((k_mc<WW, LL>), dim3((unsigned)grid), dim3(kMcThreads), lds, ctx->stream, a);                    \
    } while (0)
    if (dag_lds) {
        switch (W) {
            case 16: MC_GO(16, true); break;
            case 8: MC_GO(8, true); break;
            case 4: MC_GO(4, true); break;
            case 2: MC_GO(2, true); break;
            default: MC_GO(1, true); break;
        }
    } else {
        switch (W) {
            case 16: MC_GO(16, false); break;
            case 8: MC_GO(8, false); break;
            case 4: MC_GO(4, false); break;
            case 2: MC_GO(2, false); break;
            default: MC_GO(1, false); break;
        }
    }
#undef MC_GO
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->comm) { // shards of the sample range add: one u64 sum all-reduce of {hits, trials}
        canopy_status cst = comm_all_reduce_mc(ctx, ctx->d_mc_hits, (long long)total);
        if (cst != CANOPY_OK) return cst;
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_mc[1], ctx->stream));
    ctx->mc_last_trials = total;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_mc_fetch(canopy_ctx* ctx, canopy_mc_result* r) try {
    if (!ctx || !r) return CANOPY_ERR_INVALID_ARGUMENT;
    if (!ctx->d_mc_hits) return fail(ctx, CANOPY_ERR_STATE, "no Monte-Carlo pass has been enqueued");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_mc_hits, ctx->d_mc_hits, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_mc[0], ctx->ev_mc[1]));
    r->hits = (int64_t)*ctx->h_mc_hits;
    r->num_trials = ctx->mc_last_trials;
    r->estimate = r->num_trials > 0 ? (double)r->hits / (double)r->num_trials : 0.0;
    r->std_error = r->num_trials > 0 ? sqrt(r->estimate * (1.0 - r->estimate) / (double)r->num_trials) : 0.0;
    r->elapsed_kernel_ms = ms;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_mc_run(canopy_ctx* ctx, const canopy_dag* d, const canopy_mc_params* prm, canopy_mc_result* r) try {
    if (!r) return CANOPY_ERR_INVALID_ARGUMENT;
    canopy_status st = canopy_mc_run_async(ctx, d, prm);
    if (st != CANOPY_OK) return st;
    return canopy_mc_fetch(ctx, r);
} CANOPY_CATCH_ALL

} // extern "C"
