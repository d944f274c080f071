// kernels_mc.hip -- K6: direct Monte-Carlo propagation through the AND/OR/K-of-N gate DAG with
// bit-packed samples (BASELINE config 5), gfx950 / wave64 only, plus its C ABI entry points.
//
// Reference side: only the vocabulary exists (Approximation::kMonteCarlo settings.h:19; num_trials,
// seed settings.h:324-325; Connective event.h:143-157; Formula args with complement formula.h:6-148).
//
// One wave = one 64-sample word at a time:
//   phase 1  lane l draws sample (64*w + l) of every basic event j: bit = (mix64(ek[j] + n) >> 11) < T[j];
//            __ballot packs the 64 lanes into the event's u64 word -> the wave's private LDS node array;
//   phase 2  level by level, lane per gate: a lane combines the 64-sample WORDS of its gate's arguments
//            with bitwise AND / OR / XOR / NOT, K-of-N through a bit-sliced adder + comparator. A gate
//            only reads nodes of earlier levels; LDS executes one wave's operations in issue order, so
//            no barrier is needed between levels;
//   hits += popcount(top word & valid-sample mask). Integer arithmetic only -> bit-exact vs the oracle.
// Compute / RNG bound (one 64-bit hash per event and sample), not HBM bound: nothing is streamed.
#include <math.h>

#include <algorithm>
#include <new>
#include <vector>

#include "canopy_device.h"
#include "canopy_internal.h"

using namespace canopy;

namespace {

constexpr uint64_t kStreamMc = 4;
constexpr int kMcWaves = 4; // waves per workgroup
constexpr int kMaxAtleastArgs = 63;

struct McArgs {
    const int32_t* gate_type;
    const int32_t* gate_min;
    const int32_t* gate_node;
    const int32_t* arg_ptr;
    const int32_t* args;
    const int32_t* level_ptr;
    const uint64_t* thresh;
    int32_t n_events, n_gates, n_levels, top;
    uint64_t key;          // stream_key(seed, kStreamMc)
    int64_t first_word;    // first_sample / 64
    int64_t n_words;       // ceil(num_trials / 64)
    int64_t num_trials;
    unsigned long long* hits;
};

__device__ __forceinline__ uint64_t node_word(const uint64_t* node, int32_t a) {
    return a >= 0 ? node[a] : ~node[~a];
}

__global__ __launch_bounds__(kMcWaves * kWave) void k_mc(McArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int n_nodes = a.n_events + a.n_gates;
    // per-event constants shared by the workgroup: the event's draw key and its threshold
    uint64_t* s_ek = reinterpret_cast<uint64_t*>(smem);
    uint64_t* s_th = s_ek + a.n_events;
    uint64_t* node = s_th + a.n_events + (size_t)wave * n_nodes;
    for (int j = tid; j < a.n_events; j += kMcWaves * kWave) {
        s_ek[j] = mix64(a.key ^ (uint64_t)j);
        s_th[j] = a.thresh[j];
    }
    __syncthreads();

    unsigned long long my_hits = 0; // lane 0 counts
    const int64_t GW = (int64_t)gridDim.x * kMcWaves;
    for (int64_t w = (int64_t)blockIdx.x * kMcWaves + wave; w < a.n_words; w += GW) {
        const uint64_t n = (uint64_t)(a.first_word + w) * 64u + (uint64_t)lane; // global sample index
        // ---- phase 1: Bernoulli bits of every basic event, 64 samples per ballot
        int j = 0;
        for (; j + 4 <= a.n_events; j += 4) { // 4 independent hashes in flight per lane
            uint64_t h[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) h[u] = mix64(s_ek[j + u] + n);
            unsigned long long m[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) m[u] = __ballot((h[u] >> 11) < s_th[j + u]);
            if (lane < 4) node[j + lane] = m[0] * (lane == 0) + m[1] * (lane == 1) + m[2] * (lane == 2) + m[3] * (lane == 3);
        }
        for (; j < a.n_events; ++j) {
            const uint64_t h = mix64(s_ek[j] + n);
            const unsigned long long m = __ballot((h >> 11) < s_th[j]);
            if (lane == 0) node[j] = m;
        }
        // ---- phase 2: gates, level by level, lane per gate
        for (int lv = 0; lv < a.n_levels; ++lv) {
            const int g0 = a.level_ptr[lv], g1 = a.level_ptr[lv + 1];
            for (int g = g0 + lane; g < g1; g += kWave) {
                const int type = a.gate_type[g];
                const int b = a.arg_ptr[g], e = a.arg_ptr[g + 1];
                uint64_t r;
                if (type == CANOPY_GATE_AND || type == CANOPY_GATE_NAND) {
                    r = ~0ULL;
                    for (int k = b; k < e; ++k) r &= node_word(node, a.args[k]);
                    if (type == CANOPY_GATE_NAND) r = ~r;
                } else if (type == CANOPY_GATE_OR || type == CANOPY_GATE_NOR) {
                    r = 0ULL;
                    for (int k = b; k < e; ++k) r |= node_word(node, a.args[k]);
                    if (type == CANOPY_GATE_NOR) r = ~r;
                } else if (type == CANOPY_GATE_XOR) {
                    r = 0ULL;
                    for (int k = b; k < e; ++k) r ^= node_word(node, a.args[k]);
                } else if (type == CANOPY_GATE_NOT) {
                    r = ~node_word(node, a.args[b]);
                } else if (type == CANOPY_GATE_NULL) {
                    r = node_word(node, a.args[b]);
                } else { // ATLEAST: bit-sliced count of TRUE arguments (<= 63 -> 6 planes), then count >= K
                    uint64_t c[6] = {0, 0, 0, 0, 0, 0};
                    for (int k = b; k < e; ++k) {
                        uint64_t carry = node_word(node, a.args[k]);
#pragma unroll
                        for (int p = 0; p < 6; ++p) {
                            const uint64_t t = c[p] & carry;
                            c[p] ^= carry;
                            carry = t;
                        }
                    }
                    const int K = a.gate_min[g];
                    uint64_t gt = 0ULL, eq = ~0ULL;
#pragma unroll
                    for (int p = 5; p >= 0; --p) {
                        if ((K >> p) & 1) eq &= c[p];
                        else { gt |= eq & c[p]; eq &= ~c[p]; }
                    }
                    r = gt | eq;
                }
                node[a.gate_node[g]] = r;
            }
        }
        // ---- top node
        if (lane == 0) {
            const int64_t left = a.num_trials - w * 64;
            const uint64_t valid = left >= 64 ? ~0ULL : ((1ULL << left) - 1ULL);
            my_hits += (unsigned long long)__builtin_popcountll(node[a.top] & valid);
        }
    }
    if (lane == 0 && my_hits) atomicAdd(a.hits, my_hits); // integer: order-independent
}

canopy_status fail(canopy_ctx* ctx, canopy_status st, const std::string& msg) {
    if (ctx) ctx->last_error = msg;
    return st;
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
    void* ptrs[] = {d->d_gate_type, d->d_gate_min, d->d_gate_node, d->d_arg_ptr, d->d_args, d->d_level_ptr, d->d_thresh};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete d;
}

} // namespace

extern "C" {

canopy_status canopy_dag_upload(canopy_ctx* ctx, const canopy_dag_view* v, canopy_dag** out) {
    if (!ctx || !v || !out) return CANOPY_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (v->n_events < 0 || v->n_gates < 0 || (v->n_gates > 0 && (!v->gate_type || !v->arg_ptr)))
        return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "bad dag view");
    const int32_t n_nodes = v->n_events + v->n_gates;
    if (v->top < 0 || v->top >= n_nodes) return fail(ctx, CANOPY_ERR_INVALID_MODEL, "top node id out of range");
    // validate + levels on the host (the DAG is small: it is the model, not the sample stream)
    std::vector<int32_t> level((size_t)v->n_gates, 0);
    int32_t n_levels = 0;
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
            if (!v->gate_min || n_args > kMaxAtleastArgs || v->gate_min[g] < 0 || v->gate_min[g] > 63)
                return fail(ctx, CANOPY_ERR_UNSUPPORTED, "ATLEAST gate " + std::to_string(g) + ": need 0 <= K <= 63 and <= 63 arguments");
        }
        int32_t lv = 0;
        for (int32_t k = b; k < e; ++k) {
            const int32_t raw = v->args[k];
            const int32_t id = raw >= 0 ? raw : ~raw;
            if (id < 0 || id >= v->n_events + g) // an event, or a gate strictly before this one
                return fail(ctx, CANOPY_ERR_INVALID_MODEL, "gate " + std::to_string(g) + ": argument is not an earlier node");
            if (id >= v->n_events) lv = std::max(lv, level[(size_t)(id - v->n_events)] + 1);
        }
        level[(size_t)g] = lv;
        n_levels = std::max(n_levels, lv + 1);
    }
    // level-ordered gate list (stable inside a level)
    std::vector<int32_t> order((size_t)v->n_gates);
    for (int32_t g = 0; g < v->n_gates; ++g) order[(size_t)g] = g;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return level[(size_t)x] < level[(size_t)y]; });
    std::vector<int32_t> l_type, l_min, l_node, l_ptr(1, 0), l_args, level_ptr((size_t)n_levels + 1, 0);
    for (int32_t g : order) {
        l_type.push_back(v->gate_type[g]);
        l_min.push_back(v->gate_min ? v->gate_min[g] : 0);
        l_node.push_back(v->n_events + g);
        for (int32_t k = v->arg_ptr[g]; k < v->arg_ptr[g + 1]; ++k) l_args.push_back(v->args[k]);
        l_ptr.push_back((int32_t)l_args.size());
        level_ptr[(size_t)level[(size_t)g] + 1] += 1;
    }
    for (int32_t l = 0; l < n_levels; ++l) level_ptr[(size_t)l + 1] += level_ptr[(size_t)l];

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    canopy_dag* d = new (std::nothrow) canopy_dag();
    if (!d) return CANOPY_ERR_OUT_OF_MEMORY;
    d->device = ctx->device;
    d->n_events = v->n_events; d->n_gates = v->n_gates; d->n_levels = n_levels; d->top = v->top;
    auto up = [&](int32_t** dst, const std::vector<int32_t>& src) -> hipError_t {
        hipError_t e = hipMalloc(dst, sizeof(int32_t) * std::max<size_t>(src.size(), 1));
        if (e == hipSuccess && !src.empty())
            e = hipMemcpy(*dst, src.data(), sizeof(int32_t) * src.size(), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(&d->d_gate_type, l_type);
    if (e == hipSuccess) e = up(&d->d_gate_min, l_min);
    if (e == hipSuccess) e = up(&d->d_gate_node, l_node);
    if (e == hipSuccess) e = up(&d->d_arg_ptr, l_ptr);
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
}

void canopy_dag_free(canopy_dag* dag) { free_dag(dag); }

canopy_status canopy_dag_set_probabilities(canopy_ctx* ctx, canopy_dag* d, const double* p) {
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
}

canopy_status canopy_mc_run_async(canopy_ctx* ctx, const canopy_dag* d, const canopy_mc_params* prm) {
    if (!ctx || !d || !prm) return CANOPY_ERR_INVALID_ARGUMENT;
    if (d->device != ctx->device) return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "dag lives on another device");
    if (!d->has_p) return fail(ctx, CANOPY_ERR_STATE, "canopy_dag_set_probabilities has not been called");
    if (prm->num_trials < 0 || prm->first_sample < 0 || (prm->first_sample & 63) != 0)
        return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "num_trials >= 0 and first_sample a non-negative multiple of 64");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_mc_hits) {
        HIP_TRY(ctx, hipMalloc(&ctx->d_mc_hits, sizeof(unsigned long long)));
        HIP_TRY(ctx, hipHostMalloc(&ctx->h_mc_hits, sizeof(unsigned long long)));
        HIP_TRY(ctx, hipEventCreate(&ctx->ev_mc[0]));
        HIP_TRY(ctx, hipEventCreate(&ctx->ev_mc[1]));
    }
    const size_t n_nodes = (size_t)d->n_events + (size_t)d->n_gates;
    const size_t lds = ((size_t)kMcWaves * n_nodes + 2 * (size_t)d->n_events) * sizeof(uint64_t);
    if (lds > 160 * 1024) return fail(ctx, CANOPY_ERR_UNSUPPORTED, "gate DAG too large for the LDS node arrays (events + gates <= 5120)");
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mc), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    McArgs a;
    a.gate_type = d->d_gate_type; a.gate_min = d->d_gate_min; a.gate_node = d->d_gate_node;
    a.arg_ptr = d->d_arg_ptr; a.args = d->d_args; a.level_ptr = d->d_level_ptr; a.thresh = d->d_thresh;
    a.n_events = d->n_events; a.n_gates = d->n_gates; a.n_levels = d->n_levels; a.top = d->top;
    a.key = stream_key(prm->seed, kStreamMc);
    a.first_word = prm->first_sample / 64;
    a.n_words = (prm->num_trials + 63) / 64;
    a.num_trials = prm->num_trials;
    a.hits = ctx->d_mc_hits;
    int64_t grid = (a.n_words + kMcWaves - 1) / kMcWaves;
    const int64_t blocks_per_cu = std::max<int64_t>(1, std::min<int64_t>(8, (160 * 1024) / std::max<size_t>(lds, 1)));
    grid = std::min<int64_t>(grid, (int64_t)ctx->n_cu * blocks_per_cu);
    if (grid < 1) grid = 1;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_mc_hits, 0, sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_mc[0], ctx->stream));
    hipLaunchKernelGGL(k_mc, dim3((unsigned)grid), dim3(kMcWaves * kWave), lds, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_mc[1], ctx->stream));
    ctx->mc_last_trials = prm->num_trials;
    return CANOPY_OK;
}

canopy_status canopy_mc_fetch(canopy_ctx* ctx, canopy_mc_result* r) {
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
}

canopy_status canopy_mc_run(canopy_ctx* ctx, const canopy_dag* d, const canopy_mc_params* prm, canopy_mc_result* r) {
    if (!r) return CANOPY_ERR_INVALID_ARGUMENT;
    canopy_status st = canopy_mc_run_async(ctx, d, prm);
    if (st != CANOPY_OK) return st;
    return canopy_mc_fetch(ctx, r);
}

} // extern "C"
