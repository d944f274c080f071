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
// Two kernels: k_mc above (any DAG that fits the LDS: descriptors and argument lists read per tile), and k_mc_reg (round 3:
// the lane's gate schedule in registers, the events of several tiles drawn together with ticketed items, a top OR folded
// into the gates it reads) for DAGs whose gates lower to at most four arguments -- 72 -> 45 ms per 1e9 samples at config 5.
#include <math.h>
#include <stdlib.h>
#include <string.h>

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
constexpr int kMcRegMaxLevels = 16; // k_mc_reg: levels and per-lane schedule slots it can hold
constexpr int kMcRegMaxSlots = 16;

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


// ---- k_mc_reg (round 3): the gate schedule in registers, several tiles hashed at once ----------------------------
// Two things bound k_mc above (profiles/r01_pmc_summary_mc.txt: 22 % vector-ALU issue, 59 % of the wave cycles waiting):
//   * phase 2 re-reads every gate's descriptor and argument list out of LDS for every tile, although a lane evaluates the
//     SAME (gate, word pair) items tile after tile: a third of the LDS bytes and all the index arithmetic of a level are
//     loop invariants. Here every lane loads its items ONCE -- per item four LDS byte addresses of the arguments (bit 0 =
//     complemented), the destination address and the flag word -- into registers (the DAG is lowered to gates of at most four
//     arguments for this kernel), and a tile's phase 2 is straight-line code: 4 x ds_read_b128, OR, ds_write_b128 per item.
//   * phase 1 hands a lane one or two (event, word) items of ~7 hash rounds each (a geometric tail): the wave runs to its
//     slowest lane, 57 % of the lanes busy. Here the events of T tiles are drawn together into a staging area (T x 800 items on
//     512 lanes at config 5: six per lane), and the tiles then take their turn through the gates.
// The top event, when it is an OR over gates (the usual shape: a tree's top), is not a gate at all here: the lanes OR the words
// of the gates it reads as they produce them, one cross-lane reduction per tile replaces the levels of its helper tree.
// Sample definition, node words, hit count: identical to k_mc and to the oracle, bit for bit.
struct McPlan { // per-lane schedule: slot k = item (tid + slot_j[k] * 512) of level slot_lv[k]; the same for every lane
    int32_t n_total;
    uint32_t barrier_mask;           // bit k: the slot opens a new level (workgroup barrier first)
    uint8_t slot_lv[kMcRegMaxSlots];
    uint8_t slot_j[kMcRegMaxSlots];
    int32_t top_folded, top_out_flip;
    int32_t T;                       // tiles whose events are drawn together (the staging area holds T x n_events x W words)
};

struct U64x2 { uint64_t a, b; };

template <int W, int NS>
__global__ __launch_bounds__(kMcThreads) void k_mc_reg(McArgs a, McPlan plan) {
    const int T = plan.T;
    constexpr int WL = 2;       // words per lane in phase 2
    constexpr int S = W / WL;   // lanes per gate
    static_assert(W >= 2 && (kMcThreads % S) == 0, "a lane keeps the same word pair in every slot");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid / kWave;
    const int n_nodes = a.n_slots;
    // LDS: node[n_nodes][W] | stage[T][n_events][W] | ek[n_events] | th[n_events] | red[waves][W]
    uint64_t* node = reinterpret_cast<uint64_t*>(smem);
    const size_t node_bytes = ((size_t)n_nodes * W * 8 + 15) & ~(size_t)15;
    uint64_t* stage = reinterpret_cast<uint64_t*>(smem + node_bytes);
    uint64_t* s_ek = stage + (size_t)T * a.n_events * W;
    uint64_t* s_th = s_ek + a.n_events;
    uint64_t* s_red = s_th + a.n_events;
    unsigned* s_ticket = reinterpret_cast<unsigned*>(s_red + (size_t)(kMcThreads / kWave) * W); // [W]
    const float inv_events = 1.0f / (float)(a.n_events > 0 ? a.n_events : 1);
    if (tid < 2 * W) node[(size_t)(a.n_slots - 2) * W + tid] = tid < W ? 0ULL : ~0ULL; // the constant FALSE / TRUE nodes
    for (int i = tid; i < a.n_events; i += kMcThreads) {
        s_ek[i] = mix64(a.key ^ (uint64_t)i);
        s_th[i] = a.thresh[i];
    }
    // ---- the lane's schedule, once: LDS byte addresses and flags of its items
    uint32_t sa[NS][4], sdst[NS], sflg[NS];
    const int sub = tid % S;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        sflg[k] = 0; sdst[k] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) sa[k][q] = 0;
        if (k < plan.n_total) {
            const int lv = plan.slot_lv[k];
            const int g0 = a.level_ptr[lv];
            const int n_it = (a.level_ptr[lv + 1] - g0) * S;
            const int it = tid + (int)plan.slot_j[k] * kMcThreads;
            if (it < n_it) {
                const int4 d = a.desc[g0 + it / S];
                uint32_t f = ((uint32_t)d.x << 1) | 1u; // bit 0 valid, 1..8 type, 9..16 K, 17 in_flip, 18 out_flip, 19 feeds the top
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int32_t aa = a.args[d.z + q]; // (padded to four by the lowering)
                    const int32_t sgn = aa >> 31;
                    sa[k][q] = (uint32_t)(((aa ^ sgn) * W + sub * WL) * 8);
                    f |= (uint32_t)(sgn & 1) << (20 + q); // bits 20..23: argument q is complemented
                }
                sdst[k] = (uint32_t)((d.y * W + sub * WL) * 8);
                sflg[k] = f;
            }
        }
    }
    __syncthreads();
    unsigned long long my_hits = 0;
    const int64_t n_tiles = (a.n_words + W - 1) / W;
    const int64_t n_super = (n_tiles + T - 1) / T;
    const int items_tile = a.n_events * W;
    for (int64_t st = blockIdx.x; st < n_super; st += gridDim.x) {
        const int64_t w_super = st * T * W; // first word of the super-tile
        // ---- phase 1: the (event, word) items of T tiles, one lane resolves the 64 samples of an item plane by plane.
        // A lane keeps ITS word of the tile (512 lanes = 64 rows of W words): item = (row, tid % W), row = t * n_events + j walks
        // down 64 rows at a time -- no division per item. The threshold travels shifted (its next bit is the sign bit).
        {
            constexpr int RSTEP = kMcThreads / W;
            const int wi = tid % W;
            const int n_rows = T * a.n_events;
            // rows are handed out: the first 64 by position, the rest by a ticket per word column (an item takes 1 .. 53 hash
            // rounds, ~7 on average: lanes that draw short items come back for more instead of waiting for the wave's longest)
            if (tid < W) s_ticket[tid] = RSTEP;
            __syncthreads();
            int row = tid / W, t = 0, j = 0;
            int next_row = 0; // the ticket of the item AFTER this one, drawn when this one starts: its LDS round trip hides under the item's hash rounds
            int k = 0;
            uint64_t ctr = 0, Tsh = 0, lt = 0, und = 0;
            bool active = row < n_rows;
            bool fresh = active;
            while (__any(active)) {
                if (active) {
                    if (fresh) {
                        next_row = (int)atomicAdd(&s_ticket[wi], 1u);
                        t = (int)(((float)row + 0.5f) * inv_events); // row = t * n_events + j (row < 2^23: the float quotient is off by one at most)
                        if (t * a.n_events > row) --t;
                        else if ((t + 1) * a.n_events <= row) ++t;
                        j = row - t * a.n_events;
                        ctr = s_ek[j] + (uint64_t)(a.first_word + w_super + (int64_t)t * W + wi) * 64u;
                        const uint64_t Tt = s_th[j];
                        const bool certain = (Tt >> 53) != 0 || MC_ABL(1); // p == 1
                        Tsh = Tt << 11;                                      // bit 52 of the threshold in bit 63
                        lt = certain ? ~0ULL : 0ULL;
                        und = certain ? 0ULL : ~0ULL;
                        k = 0;
                        fresh = false;
                    }
                    if (und != 0ULL && k < 53) {
                        const uint64_t r = mix64(ctr + (uint64_t)k);
                        const uint64_t tbm = (uint64_t)((int64_t)Tsh >> 63); // all ones: the threshold's bit of this plane is 1
                        const uint64_t un = und & (r ^ ~tbm);               // still equal to the threshold's prefix
                        lt |= (und ^ un) & tbm;                             // threshold bit 1, sample bit 0: below
                        und = un;
                        Tsh <<= 1;
                        ++k;
                    }
                    if (und == 0ULL || k == 53) { // k == 53 with lanes left: u == threshold, not below
                        stage[(size_t)row * W + wi] = lt; // stage[t][j][wi]
                        row = next_row;
                        active = row < n_rows;
                        fresh = active;
                    }
                }
            }
        }
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            // the tile's event words into the node array (slot j = event j)
            for (int i = tid; i < items_tile; i += kMcThreads) node[i] = stage[(size_t)t * items_tile + i];
            __syncthreads();
            uint64_t top0 = 0ULL, top1 = 0ULL;
            // ---- phase 2: the lane's items, straight out of its registers
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                if (k < plan.n_total && !MC_ABL(2)) { // (uniform)
                    if ((plan.barrier_mask >> k) & 1u) __syncthreads();
                    uint32_t f = sflg[k];
                    asm volatile("" : "+v"(f)); // (everything decoded from the flag word is decoded per tile: hoisted out of the tile loop
                                                //  the masks of 8 x 4 arguments alone would take 64 registers)
                    if (f & 1u) {
                        const uint32_t fin = 0u - ((f >> 17) & 1u), fout = 0u - ((f >> 18) & 1u);
                        const uint64_t out_flip = ((uint64_t)fout << 32) | fout;
                        uint64_t x0[4], x1[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const U64x2 v = *reinterpret_cast<const U64x2*>(smem + sa[k][q]);
                            const uint32_t mq = (0u - ((f >> (20 + q)) & 1u)) ^ fin;
                            const uint64_t m = ((uint64_t)mq << 32) | mq;
                            x0[q] = v.a ^ m; x1[q] = v.b ^ m;
                        }
                        uint64_t r0 = ((x0[0] | x0[1]) | (x0[2] | x0[3])) ^ out_flip;
                        uint64_t r1 = ((x1[0] | x1[1]) | (x1[2] | x1[3])) ^ out_flip;
                        const int type = (int)((f >> 1) & 0xffu);
                        const bool vote = type == CANOPY_GATE_ATLEAST, is_xor = type == CANOPY_GATE_XOR;
                        if (__any(vote || is_xor)) { // (the K-of-N gates sort last inside a level: most waves never come here)
                            if (is_xor) { r0 = (x0[0] ^ x0[1]) ^ (x0[2] ^ x0[3]); r1 = (x1[0] ^ x1[1]) ^ (x1[2] ^ x1[3]); }
                            if (vote) {
                                const int K = (int)((f >> 9) & 0xffu);
                                uint64_t c0[6] = {0, 0, 0, 0, 0, 0}, c1[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
                                for (int q = 0; q < 4; ++q) { vote_add<3>(c0, x0[q]); vote_add<3>(c1, x1[q]); }
                                r0 = vote_ge<3>(c0, K); r1 = vote_ge<3>(c1, K);
                            }
                        }
                        U64x2 o; o.a = r0; o.b = r1;
                        *reinterpret_cast<U64x2*>(smem + sdst[k]) = o;
                        if (f & (1u << 19)) { top0 |= r0; top1 |= r1; } // this gate is an argument of the (folded) top OR
                    }
                }
            }
            // ---- the top event's words of this tile
            const int64_t w0 = w_super + (int64_t)t * W;
            if (plan.top_folded) {
#pragma unroll
                for (int dlt = S; dlt < kWave; dlt <<= 1) { // lanes of equal tid % S hold the same word pair
                    top0 |= __shfl_xor(top0, dlt, kWave);
                    top1 |= __shfl_xor(top1, dlt, kWave);
                }
                if (lane < S) { s_red[wave * W + lane * WL] = top0; s_red[wave * W + lane * WL + 1] = top1; }
                __syncthreads();
                if (tid < W) {
                    uint64_t v = 0ULL;
#pragma unroll
                    for (int wv = 0; wv < kMcThreads / kWave; ++wv) v |= s_red[wv * W + tid];
                    if (plan.top_out_flip) v = ~v;
                    const int64_t left = a.num_trials - (w0 + tid) * 64;
                    if (left > 0) my_hits += (unsigned long long)__builtin_popcountll(v & (left >= 64 ? ~0ULL : ((1ULL << left) - 1ULL)));
                }
            } else {
                __syncthreads();
                if (tid < W) {
                    const int64_t left = a.num_trials - (w0 + tid) * 64;
                    if (left > 0)
                        my_hits += (unsigned long long)__builtin_popcountll(node[(size_t)a.top * W + tid] & (left >= 64 ? ~0ULL : ((1ULL << left) - 1ULL)));
                }
            }
            __syncthreads(); // the next tile overwrites the event slots (and s_red)
        }
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
    void* ptrs[] = {d->d_desc, d->d_args, d->d_level_ptr, d->d_thresh, d->d_reg_desc, d->d_reg_args, d->d_reg_level_ptr};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete d;
}

// The internal form of a gate DAG the kernels evaluate (host side, at upload).
struct Lowered {
    std::vector<int32_t> desc, args, level_ptr; // desc: 4 ints per gate {type | K << 8 | flags, output slot, arg begin, arg end}
    int32_t n_gates = 0, n_levels = 0, n_slots = 0, top_slot = -1, vote_planes = 6, max_args = 0;
    bool top_folded = false, top_out_flip = false;
};

// max_fan_in: AND / OR / XOR family gates with more arguments become trees of helper gates of at most that many.
// fold_top: when the top event is an OR / NOR gate over un-complemented gates that nothing else reads, it is NOT emitted:
// the gates it reads carry the flag bit 18 ("feeds the top") and the kernel ORs their words as it produces them.
void lower_dag(const canopy_dag_view* v, int32_t kMaxFanIn, bool fold_top, Lowered& L) {
    bool fold = false;
    int32_t top_gate = -1;
    if (fold_top && v->top >= v->n_events) {
        top_gate = v->top - v->n_events;
        const int32_t tt = v->gate_type[top_gate];
        fold = tt == CANOPY_GATE_OR || tt == CANOPY_GATE_NOR;
        for (int32_t k = v->arg_ptr[top_gate]; fold && k < v->arg_ptr[top_gate + 1]; ++k)
            if (v->args[k] < v->n_events) fold = false; // an event or a complemented argument
        for (int32_t g = 0; fold && g < v->n_gates; ++g)
            for (int32_t k = v->arg_ptr[g]; k < v->arg_ptr[g + 1]; ++k) {
                const int32_t raw = v->args[k];
                if ((raw >= 0 ? raw : ~raw) == v->top) { fold = false; break; } // somebody reads the top gate
            }
        if (fold) { L.top_folded = true; L.top_out_flip = tt == CANOPY_GATE_NOR; }
    }
    std::vector<char> feeds_top_ext((size_t)v->n_gates, 0);
    if (fold)
        for (int32_t k = v->arg_ptr[top_gate]; k < v->arg_ptr[top_gate + 1]; ++k) feeds_top_ext[(size_t)(v->args[k] - v->n_events)] = 1;
    // Internal gate list. A lane evaluates one gate by walking its arguments one after the other, so a wide
    // gate (the OR over a whole layer at the top of a tree) would serialise hundreds of dependent LDS reads
    // in a single lane while the rest of the workgroup waits at the level barrier. AND / OR / XOR are
    // associative: gates of those families with more than kMaxFanIn arguments become a tree of helper gates
    // of at most kMaxFanIn arguments (NAND / NOR: helpers of the AND / OR family, negation at the root).
    // Same Boolean function, so the same sample bits; node ids are internal from here on.
    struct IGate { int32_t type, K; std::vector<int32_t> args; bool feeds_top; };
    std::vector<IGate> ig;
    ig.reserve((size_t)v->n_gates + 16);
    std::vector<int32_t> node_of((size_t)v->n_gates, 0); // external gate -> internal node id
    auto map_arg = [&](int32_t raw) {
        const int32_t id = raw >= 0 ? raw : ~raw;
        const int32_t m = id < v->n_events ? id : node_of[(size_t)(id - v->n_events)];
        return raw >= 0 ? m : ~m;
    };
    for (int32_t g = 0; g < v->n_gates; ++g) {
        if (fold && g == top_gate) continue; // folded into the gates it reads
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
                    ig.push_back({base, 0, std::vector<int32_t>(cur.begin() + (long)c0, cur.begin() + (long)c1), false});
                    next.push_back(v->n_events + (int32_t)ig.size() - 1);
                }
                cur.swap(next);
            }
        }
        ig.push_back({type, (type == CANOPY_GATE_ATLEAST && v->gate_min) ? v->gate_min[g] : 0, cur, feeds_top_ext[(size_t)g] != 0});
        node_of[(size_t)g] = v->n_events + (int32_t)ig.size() - 1;
    }
    const int32_t n_igates = (int32_t)ig.size();
    int32_t max_vote_args = 0;
    for (const IGate& q : ig)
        if (q.type == CANOPY_GATE_ATLEAST) max_vote_args = std::max(max_vote_args, (int32_t)q.args.size());
    const int32_t top_node = fold ? -1 : (v->top < v->n_events ? v->top : node_of[(size_t)(v->top - v->n_events)]);
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
    if (top_node >= 0) last_use[(size_t)top_node] = n_levels + 1; // read after the last level
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
        l_desc.push_back(type | (ig[(size_t)g].K << 8) | ((int32_t)in_flip << 16) | ((int32_t)out_flip << 17) |
                         ((int32_t)ig[(size_t)g].feeds_top << 18));
        L.max_args = std::max(L.max_args, (int32_t)ig[(size_t)g].args.size());
        l_desc.push_back(slot_of[(size_t)(v->n_events + g)]);
        l_desc.push_back(b);
        l_desc.push_back((int32_t)l_args.size());
        level_ptr[(size_t)level[(size_t)g] + 1] += 1;
    }
    for (int32_t l = 0; l < n_levels; ++l) level_ptr[(size_t)l + 1] += level_ptr[(size_t)l];

    L.desc.swap(l_desc); L.args.swap(l_args); L.level_ptr.swap(level_ptr);
    L.n_gates = n_igates; L.n_levels = n_levels; L.n_slots = n_slots + 2; // + constant FALSE / TRUE
    L.top_slot = top_node >= 0 ? slot_of[(size_t)top_node] : -1;
    L.vote_planes = max_vote_args <= 7 ? 3 : 6;
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
    Lowered L;
    lower_dag(v, 8, false, L);
    Lowered R; // the register-schedule kernel: gates of at most four arguments, the top OR folded
    lower_dag(v, 4, true, R);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    canopy_dag* d = new (std::nothrow) canopy_dag();
    if (!d) return CANOPY_ERR_OUT_OF_MEMORY;
    d->device = ctx->device;
    d->n_events = v->n_events; d->n_gates = L.n_gates; d->n_levels = L.n_levels; d->top = L.top_slot;
    d->n_slots = L.n_slots;
    d->n_args = (int32_t)L.args.size();
    d->vote_planes = L.vote_planes;
    // the second form, for k_mc_reg: usable when no gate is left with more than four arguments (a wide K-of-N gate is not
    // associative: it stays whole, and the model runs through the generic kernel)
    d->reg_ok = R.max_args <= 4 && R.n_levels <= kMcRegMaxLevels && R.n_gates > 0;
    d->reg_n_gates = R.n_gates; d->reg_n_levels = R.n_levels; d->reg_n_slots = R.n_slots; d->reg_top = R.top_slot;
    d->reg_n_args = (int32_t)R.args.size();
    d->reg_top_folded = R.top_folded; d->reg_top_out_flip = R.top_out_flip;
    d->reg_level_gates.clear();
    for (int32_t l = 0; l < R.n_levels; ++l) d->reg_level_gates.push_back(R.level_ptr[(size_t)l + 1] - R.level_ptr[(size_t)l]);
    auto up = [&](int32_t** dst, const std::vector<int32_t>& src) -> hipError_t {
        hipError_t e = hipMalloc(dst, sizeof(int32_t) * std::max<size_t>(src.size(), 1));
        if (e == hipSuccess && !src.empty())
            e = hipMemcpy(*dst, src.data(), sizeof(int32_t) * src.size(), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(reinterpret_cast<int32_t**>(&d->d_desc), L.desc);
    if (e == hipSuccess) e = up(&d->d_args, L.args);
    if (e == hipSuccess) e = up(&d->d_level_ptr, L.level_ptr);
    if (e == hipSuccess && d->reg_ok) e = up(reinterpret_cast<int32_t**>(&d->d_reg_desc), R.desc);
    if (e == hipSuccess && d->reg_ok) e = up(&d->d_reg_args, R.args);
    if (e == hipSuccess && d->reg_ok) e = up(&d->d_reg_level_ptr, R.level_ptr);
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
    // ---- k_mc_reg when the model fits it: W = 8, at most 16 schedule slots per lane, two workgroups per CU
    static const bool reg_off = CANOPY_TUNE_ENV("CANOPY_MC_NOREG") != nullptr; // tuning only
    if (d->reg_ok && !reg_off) {
        constexpr int RW = 8, RS = RW / 2;
        McPlan plan;
        memset(&plan, 0, sizeof(plan));
        int n_total = 0;
        bool fits = true;
        for (int32_t lv = 0; lv < d->reg_n_levels && fits; ++lv) {
            const int items = d->reg_level_gates[(size_t)lv] * RS;
            const int spl = (items + kMcThreads - 1) / kMcThreads;
            for (int j = 0; j < spl; ++j) {
                if (n_total >= kMcRegMaxSlots) { fits = false; break; }
                plan.slot_lv[n_total] = (uint8_t)lv;
                plan.slot_j[n_total] = (uint8_t)j;
                if (j == 0 && lv > 0) plan.barrier_mask |= 1u << n_total;
                ++n_total;
            }
        }
        plan.n_total = n_total;
        plan.top_folded = d->reg_top_folded ? 1 : 0;
        plan.top_out_flip = d->reg_top_out_flip ? 1 : 0;
        const size_t fixed = (((size_t)d->reg_n_slots * RW * 8 + 15) & ~(size_t)15) + (size_t)d->n_events * 16 + (size_t)(kMcThreads / kWave) * RW * 8 + RW * 4 + 16;
        const size_t per_tile = (size_t)d->n_events * RW * 8;
        static const int t_env = CANOPY_TUNE_ENV("CANOPY_MC_T") ? atoi(CANOPY_TUNE_ENV("CANOPY_MC_T")) : 0; // tuning only
        int Tt = 0;
        for (int tt = 8; tt >= 1 && !Tt; --tt)
            if (fixed + per_tile * (size_t)tt <= (size_t)78 * 1024) Tt = tt; // two workgroups per CU
        if (!Tt && fixed + per_tile <= (size_t)156 * 1024) Tt = 1;           // one
        if (t_env >= 1 && t_env <= 16 && fixed + per_tile * (size_t)t_env <= (size_t)156 * 1024) Tt = t_env;
        plan.T = Tt;
        if (fits && Tt && n_total > 0) {
            McArgs ra = a;
            ra.desc = d->d_reg_desc; ra.args = d->d_reg_args; ra.level_ptr = d->d_reg_level_ptr;
            ra.n_gates = d->reg_n_gates; ra.n_levels = d->reg_n_levels; ra.top = d->reg_top; ra.n_args = d->reg_n_args; ra.n_slots = d->reg_n_slots;
            ra.n_words = (total + 63) / 64;
            const size_t rlds = fixed + per_tile * (size_t)Tt;
            const int64_t r_tiles = (ra.n_words + RW - 1) / RW;
            const int64_t r_super = (r_tiles + Tt - 1) / Tt;
            const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(2, (int64_t)((160 * 1024) / rlds)));
            int64_t rgrid = std::min<int64_t>(r_super, (int64_t)ctx->n_cu * per_cu);
            if (rgrid < 1) rgrid = 1;
#define MCR_GO(NSS)                                                                                                        \
            do {                                                                                                           \
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mc_reg<RW, NSS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                hipLaunchKernelGGL((k_mc_reg<RW, NSS>), dim3((unsigned)rgrid), dim3(kMcThreads), rlds, ctx->stream, ra, plan); \
            } while (0)
            if (n_total <= 8) MCR_GO(8);
            else if (n_total <= 12) MCR_GO(12);
            else MCR_GO(16);
#undef MCR_GO
            HIP_TRY(ctx, hipGetLastError());
            if (ctx->comm) { // shards of the sample range add: one u64 sum all-reduce of {hits, trials}
                canopy_status cst = comm_all_reduce_mc(ctx, ctx->d_mc_hits, (long long)total);
                if (cst != CANOPY_OK) return cst;
            }
            HIP_TRY(ctx, hipEventRecord(ctx->ev_mc[1], ctx->stream));
            ctx->mc_last_trials = total;
            return CANOPY_OK;
        }
    }
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
