// kernels_fr.hip -- "filter then refine" (FR) variant of K1..K4 for large basic-event counts and a
// positive cut-off, gfx950 / wave64 only.
//
// With the reference's default cut_off = 1e-8 (settings.h:332) fewer than 1 % of the cut sets survive
// the filter s_i >= log(cut_off). Everything the path reports (sums, count, argmax, keep mask) depends
// on the survivors only. So:
//
//   pass A  k_fr_filter: an UNCONSTRAINED fast pass over all entries whose only job is to find the rows
//           that might survive. A workgroup owns a slab of 8192 (or 16384) consecutive rows whose running
//           sums live in LDS; the slab's entries arrive as a column-bucket-sorted stream of packed u32
//           (col << 14 | row_local) built once at prepare time (k_fr_build_sorted), any wave takes any
//           chunk: 64 gathers of neighbouring columns, then 64 unordered ds_add_f64. (k_fr_filter_q8, the
//           form AUTO runs for E <= 112 K, replaces the gathers by byte reads of a conservative 8-bit table
//           in LDS and the fp64 sums by u32 sums: see there.)
//           The order of additions is NOT the CSR order here, so s~_i differs from s_i by rounding only
//           (<= len * 2^-53 * sum|logp|, about 1e-13). Rows with s~_i >= log(cut_off) - delta
//           (delta = 1e-9 >> that error) go into the slab's candidate list (u16 local row ids, ascending,
//           positions from an in-workgroup prefix scan of the ballots: deterministic).
//   pass B  k_fr_refine: every candidate row is recomputed from the canonical CSR in exact CSR order
//           (the same plain v_add_f64 chain as the oracle) and goes through the exact filter, exp / log1p
//           sums, count, argmax, keep mask. No row that survives exactly can be missed (its s~ is within
//           1e-13 of s >= log(cut_off)), and no reported value comes from pass A.
//
// Per-row log-probability output needs every s_i exactly and therefore uses the exact stream kernels (kernels_bs.hip,
// kernels_ws.hip).
#include <stdlib.h>

#include "canopy_device.h"
#include "canopy_internal.h"

namespace canopy {

namespace {

constexpr int kFrChunk = 256;

int grid_for(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (int)g;
}

// configuration: rows per slab (the slot after them is the no-op sink), waves per workgroup
template <int SLAB_ROWS>
struct FrCfg {
    static constexpr int slab_rows = SLAB_ROWS;
    static constexpr int row_bits = SLAB_ROWS > 8192 ? 15 : 14;
    static constexpr uint32_t row_mask = (1u << row_bits) - 1;
    static constexpr int slots = SLAB_ROWS + 64;
    static constexpr int waves = SLAB_ROWS / 1024; // 8192 rows: 8 waves, two workgroups per CU; 16384: 16 waves
    static constexpr int threads = waves * kWave;
    static constexpr size_t lds = (size_t)slots * 8;
};

// 16384-row slabs (one 16-wave workgroup per CU) measured faster than 8192-row slabs (two 8-wave
// workgroups): 0.85 vs 0.98 ms at config 3. They need 15 row bits, i.e. E <= 2^17; beyond that 8192.
constexpr int kQ8MaxEvents = 112 * 1024; // 8-bit table + 8192 u32 row sums + 128 counts must fit the 160 KiB of LDS
constexpr int kQ8Levels = 224;           // threshold level; entries clamp at 255 (alone enough to kill a row)

// Beyond 112 K events the 8-bit table no longer fits LDS beside the row sums: the columns are cut into H = 2 or 3 PARTS
// (whole column buckets, <= 112 K events each). The stream of a slab keeps its parts apart (each padded to whole chunks,
// one chunk offset per (slab, part)), and the quantised filter makes H sweeps over its slabs with one part of the table
// resident per sweep, carrying every row's partial sum from sweep to sweep as a saturated u16 in the (not yet written)
// candidate-list buffer: +4 B per row and part boundary on a 24 B-per-cut-set stream instead of a gathered fp64 filter.
constexpr int kQ8MaxParts = 3;
constexpr int kFrBucketsC = 512; // column buckets of the builder's counting sort (= kFrBuckets below)

bool fr_q8_allowed() {
    static const bool off = CANOPY_TUNE_ENV("CANOPY_FR_NOQ8") != nullptr; // tuning only
    return !off;
}
bool fr_use_q8(int32_t n_events) { return fr_q8_allowed() && n_events <= kQ8MaxParts * kQ8MaxEvents && n_events <= (1 << 18); }
// Round 3: between 112 K and 254 K (260 096) events the table is kept at FOUR bits per event (levels 0..14, 15 = "alone enough to kill a
// row"): it fits LDS in one piece, so the model needs no column parts and no second sweep; the coarser steps list about twice as
// many rows for the exact pass (2 % instead of 0.9 % at the default cut-off), which costs less than the sweep it saves.
constexpr int kQ4MaxEvents = 260096; // = 2 x (160 KiB - the 8192 + 128 u32 row sums - 128 counts): the 4-bit table fills the LDS to the byte
constexpr int kQ4Levels = 14;
bool fr_q8_nibble(int32_t n_events) {
    static const bool off = CANOPY_TUNE_ENV("CANOPY_FR_NONIBBLE") != nullptr; // tuning only
    return !off && fr_use_q8(n_events) && n_events > kQ8MaxEvents && n_events <= kQ4MaxEvents;
}
int fr_parts_for(int32_t n_events) {
    if (!fr_use_q8(n_events) || n_events <= kQ8MaxEvents || fr_q8_nibble(n_events)) return 1;
    return (n_events + kQ8MaxEvents - 1) / kQ8MaxEvents;
}
__host__ __device__ inline unsigned fr_bucket_of(uint32_t c, uint64_t ne) { return (unsigned)(((uint64_t)c * (uint64_t)kFrBucketsC) / ne); }
__host__ __device__ inline int fr_part_of_bucket(unsigned b, int H) { return (int)((b * (unsigned)H) / (unsigned)kFrBucketsC); }
// first column of part h (h = H: n_events)
int32_t fr_part_first_col(int h, int H, int32_t n_events) {
    if (h <= 0) return 0;
    if (h >= H) return n_events;
    unsigned b = 0;
    while (b < (unsigned)kFrBucketsC && fr_part_of_bucket(b, H) < h) ++b;
    return (int32_t)(((uint64_t)b * (uint64_t)n_events + kFrBucketsC - 1) / kFrBucketsC); // smallest c with bucket(c) >= b
}

int fr_slab_rows_for(int32_t n_events) {
    static const int forced = CANOPY_TUNE_ENV("CANOPY_FR_SLAB") ? atoi(CANOPY_TUNE_ENV("CANOPY_FR_SLAB")) : 0; // tuning only
    if (forced == 8192 || forced == 16384) return forced;
    if (fr_use_q8(n_events)) return 8192; // quantised LDS-table filter: 8192-row slabs
    return n_events <= (1 << 17) ? 16384 : 8192;
}

// ---- builder: one workgroup per slab, counting sort of the slab's entries by column bucket -------------------
// The filters add in any order, so the stream needs no group structure: a slab's entries are bucket-sorted by
// column (<= 512 buckets over [0, E)) and cut into 256-entry chunks; only the slab's last chunk is padded
// (col 0 -> the sink row). Chunks are therefore narrow in column (what the 24-bit repacking and the gathered
// fp64 filter both want), the stream is as dense as it can be, and the build is parallel.
constexpr int kFrBuckets = 512;
constexpr int kFrColHist = 576; // >= the columns of one bucket + 1 for E <= 2^18 (513), a multiple of 64

// LDS traffic between the lanes of ONE wave (no workgroup barrier): the wave's LDS operations complete in order; this
// keeps the compiler from moving them across
__device__ __forceinline__ void fr_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void k_fr_count_chunks(const int64_t* __restrict__ row_ptr, int64_t n_rows, int slab_rows,
                                                         int64_t n_slabs, int64_t* __restrict__ chunks) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slabs) return;
    const int64_t r0 = s * slab_rows;
    const int64_t r1 = r0 + slab_rows < n_rows ? r0 + slab_rows : n_rows;
    chunks[s] = (row_ptr[r1] - row_ptr[r0] + kFrChunk - 1) / kFrChunk;
}

// Where the entry of sorted position `pos` of a slab goes: inside its 256-entry chunk the order is transposed, so that
// lane k's 16-byte load brings sorted entries k, 64 + k, 128 + k, 192 + k -- the 64 lanes of ONE table-gather
// instruction then cover a contiguous quarter of the chunk's column range (a handful of cache lines) instead of a
// stride-4 sample of all of it. The LDS-table filter does not care; the gathered passes issue 4x fewer line requests
// per instruction.
__device__ __forceinline__ unsigned fr_chunk_slot(unsigned pos) {
    const unsigned p = pos & (kFrChunk - 1);
    return (pos - p) + ((p & 63u) << 2) + (p >> 6);
}

// chunks per (slab, part) when the columns are cut into H > 1 parts: one workgroup per slab counts its entries per part
__global__ __launch_bounds__(256) void k_fr_count_chunks_parts(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                               int64_t n_rows, int32_t n_events, int slab_rows, int H,
                                                               int64_t* __restrict__ chunks /* [n_slabs * H] */) {
    __shared__ unsigned cnt[kQ8MaxParts];
    const int64_t s = blockIdx.x;
    const int64_t r0 = s * slab_rows;
    const int64_t r1 = r0 + slab_rows < n_rows ? r0 + slab_rows : n_rows;
    if (threadIdx.x < kQ8MaxParts) cnt[threadIdx.x] = 0u;
    __syncthreads();
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256)
        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k)
            atomicAdd(&cnt[fr_part_of_bucket(fr_bucket_of((uint32_t)col_idx[k], (uint64_t)n_events), H)], 1u);
    __syncthreads();
    if ((int)threadIdx.x < H) chunks[s * H + threadIdx.x] = (int64_t)((cnt[threadIdx.x] + kFrChunk - 1) / kFrChunk);
}

__global__ __launch_bounds__(256) void k_fr_build_sorted(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                         int64_t n_rows, int32_t n_events, int slab_rows, int row_bits, int H,
                                                         const int64_t* __restrict__ fr_off, uint32_t* __restrict__ tmp,
                                                         uint32_t* __restrict__ stream) {
    __shared__ unsigned hist[kFrBuckets + 1];
    __shared__ unsigned ps[kQ8MaxParts + 1]; // first (unpadded) sorted slot of each part; ps[H] = the slab's entries
    __shared__ unsigned chist[4][kFrColHist]; // phase 2: one wave's per-column counts of one bucket
    const int64_t s = blockIdx.x;
    const int64_t r0 = s * slab_rows;
    const int64_t r1 = r0 + slab_rows < n_rows ? r0 + slab_rows : n_rows;
    const int tid = threadIdx.x;
    for (int i = tid; i <= kFrBuckets; i += 256) hist[i] = 0u;
    __syncthreads();
    const uint64_t nb = (uint64_t)kFrBuckets, ne = (uint64_t)n_events;
    for (int64_t r = r0 + tid; r < r1; r += 256)
        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k)
            atomicAdd(&hist[(unsigned)(((uint64_t)(uint32_t)col_idx[k] * nb) / ne) + 1u], 1u);
    __syncthreads();
    if (tid < kWave) { // inclusive scan of the 512 counts (8 per lane) -> hist[b] = first slot of bucket b
        unsigned loc[8], tot = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { loc[i] = hist[1 + tid * 8 + i]; tot += loc[i]; }
        unsigned incl = tot;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const unsigned o = __shfl_up(incl, d, kWave);
            if (tid >= d) incl += o;
        }
        unsigned run = incl - tot;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const unsigned c = loc[i]; hist[1 + tid * 8 + i] = run; run += c; }
    }
    const int64_t nnz = row_ptr[r1] - row_ptr[r0];
    if (tid <= H) { // where each part starts in the sorted order (before the cursors move)
        unsigned b = 0;
        while (b < (unsigned)kFrBuckets && fr_part_of_bucket(b, H) < tid) ++b;
        ps[tid] = tid == H ? (unsigned)nnz : hist[1 + b];
    }
    __syncthreads();
    // hist[1 + b] is now the cursor of bucket b; part h of the slab starts at chunk fr_off[s * H + h]
    uint32_t* out = stream + fr_off[s * H] * kFrChunk;
    uint32_t* tq = tmp + fr_off[s * H] * kFrChunk; // phase 1: bucket order, linear positions
    for (int64_t r = r0 + tid; r < r1; r += 256)
        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {
            const uint32_t c = (uint32_t)col_idx[k];
            const unsigned b = (unsigned)(((uint64_t)c * nb) / ne);
            const unsigned pos = atomicAdd(&hist[b + 1u], 1u);
            tq[pos] = (c << row_bits) | (uint32_t)(r - r0);
        }
    __syncthreads(); // hist[b] .. hist[b + 1] is now bucket b's range of tq (hist[0] = 0); the slab's tq is complete
    // phase 2: every bucket is sorted by COLUMN (a counting sort over its <= E/512 + 1 columns, one wave per bucket), so
    // that the 64 lanes of one table-gather instruction of the gathered passes carry ascending addresses with
    // neighbouring lanes in the same cache line: the texture addresser takes such an instruction in ~29 clocks instead
    // of ~41 for the same lines in random lane order (benchmarks/microbench5.hip). Entries of one column keep an
    // arbitrary order (no pass depends on it: integer sums commute, the fp64 filter is conservative by 1e-9).
    {
        const int wv = tid >> 6, lane = tid & (kWave - 1);
        unsigned* ch = chist[wv];
        for (unsigned b = (unsigned)wv; b < (unsigned)kFrBuckets; b += 4u) {
            const unsigned lo = hist[b], hi = hist[b + 1u];
            if (hi == lo) continue; // wave-uniform
            const uint32_t c_lo = (uint32_t)(((uint64_t)b * ne + nb - 1) / nb), c_hi = (uint32_t)(((uint64_t)(b + 1u) * ne + nb - 1) / nb);
            const unsigned n_cols = c_hi - c_lo; // <= kFrColHist - 1 (E <= 2^18)
            for (unsigned i = (unsigned)lane; i < (unsigned)kFrColHist; i += kWave) ch[i] = 0u;
            fr_wave_lds_sync();
            for (unsigned p = lo + (unsigned)lane; p < hi; p += kWave) atomicAdd(&ch[(tq[p] >> row_bits) - c_lo], 1u);
            fr_wave_lds_sync();
            { // exclusive scan of the counts, kFrColHist / 64 per lane
                constexpr int PER = kFrColHist / kWave;
                unsigned loc[PER], tot = 0;
#pragma unroll
                for (int i = 0; i < PER; ++i) { loc[i] = ch[lane * PER + i]; tot += loc[i]; }
                unsigned incl = tot;
#pragma unroll
                for (int d = 1; d < kWave; d <<= 1) {
                    const unsigned o = __shfl_up(incl, d, kWave);
                    if (lane >= d) incl += o;
                }
                unsigned run = incl - tot;
#pragma unroll
                for (int i = 0; i < PER; ++i) { const unsigned c = loc[i]; ch[lane * PER + i] = run; run += c; }
            }
            fr_wave_lds_sync();
            const int h = fr_part_of_bucket(b, H);
            const unsigned base = (unsigned)((fr_off[s * H + h] - fr_off[s * H]) * kFrChunk) - ps[h];
            (void)n_cols;
            for (unsigned p = lo + (unsigned)lane; p < hi; p += kWave) {
                const uint32_t e = tq[p];
                const unsigned q = lo + atomicAdd(&ch[(e >> row_bits) - c_lo], 1u);
                out[fr_chunk_slot(base + q)] = e;
            }
            fr_wave_lds_sync();
        }
    }
    // padding of every part: col 0 -> one of the 64 sink rows, chosen so that the 64 lanes of one LDS add hit 64
    // DIFFERENT slots (64 atomics on one address serialise: a padded chunk then costs more than a real one)
    for (int h = 0; h < H; ++h) {
        const unsigned base = (unsigned)((fr_off[s * H + h] - fr_off[s * H]) * kFrChunk), n_h = ps[h + 1] - ps[h];
        const unsigned padded = (n_h + kFrChunk - 1) / kFrChunk * kFrChunk;
        for (unsigned i = n_h + tid; i < padded; i += 256) {
            const unsigned ph = fr_chunk_slot(base + i);
            out[ph] = (uint32_t)slab_rows + ((ph >> 2) & 63u);
        }
    }
}

// the no-op chunk: col 0 -> the 64 sink rows, lane k's four entries on sink slot k
__global__ __launch_bounds__(256) void k_fr_fill(uint32_t* p, int64_t n, uint32_t v) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v + (uint32_t)((i >> 2) & 63);
}

// ---- the 24-bit form of the slab stream ("P24": what the quantised filter streams when the model allows) --------
// The same chunks at the same chunk indices, three bytes per entry: (col - first column of the chunk) << 14 | row_local.
// A chunk of 256 column-sorted entries spans ~520 columns at config 3; a slab with a chunk that spans 1024 columns or more
// (a sparse slab: the last, partial one of a model) is an EXCEPTION slab: the filter reads nothing of it and hands all its
// rows to the exact pass (<= 8192 rows). More than 1 % of such slabs: the model keeps the 32-bit form only. Lane k's 12
// bytes hold its four entries (sorted positions k, 64 + k, ...), one u32 per chunk holds the chunk's first column.
// 18.1 B per cut set instead of 24.1 at no extra LDS work and ~1.5 more vector instructions per 64 entries. Measured:
// 1.25e7-row shard (the 8-GPU case) 54.7 -> 49.6 us per pipelined step; whole 1e8-row model 0.440 -> 0.422 ms per serial
// pass and no change pipelined (0.374: there the pass is level with its own instruction stream).
constexpr uint32_t kP24ColSpan = 1u << 10;

__global__ __launch_bounds__(256) void k_fr_pack24(const uint4* __restrict__ stream32, const int64_t* __restrict__ fr_off,
                                                   int64_t n_slabs, int64_t total_chunks, int row_bits, int slab_rows,
                                                   uint32_t* __restrict__ out24, uint32_t* __restrict__ base,
                                                   int32_t* __restrict__ exc, int* __restrict__ n_exc) {
    __shared__ int s_bad;
    const int64_t s = blockIdx.x; // slab n_slabs = the no-op chunk
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int64_t c0 = s < n_slabs ? fr_off[s] : total_chunks, c1 = s < n_slabs ? fr_off[s + 1] : total_chunks + 1;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    const uint32_t row_mask = (1u << row_bits) - 1u;
    bool bad = false;
    for (int64_t c = c0 + wave; c < c1; c += 4) {
        const uint4 q = stream32[c * kWave + lane];
        const uint32_t e[4] = {q.x, q.y, q.z, q.w};
        // sorted position 0 = lane 0's first entry; a padding entry there means the chunk is all padding
        const uint32_t first = __builtin_amdgcn_readfirstlane(q.x);
        const uint32_t col0 = (first & row_mask) >= (uint32_t)slab_rows ? 0u : (first >> row_bits);
        uint32_t pk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t rl = e[k] & row_mask, col = e[k] >> row_bits;
            const bool pad = rl >= (uint32_t)slab_rows;
            const uint32_t off = pad ? 0u : col - col0; // ascending columns: col >= col0
            bad = bad || off >= kP24ColSpan;
            // padding: sink row 8192 + 2 k (the overlapped filter packs two rows per LDS word: sink k = the low half of
            // word 4096 + k; the plain filter keeps 128 sink slots)
            const uint32_t rl24 = pad ? (uint32_t)slab_rows + 2u * ((rl - (uint32_t)slab_rows) & 63u) : rl;
            pk[k] = ((off & (kP24ColSpan - 1u)) << 14) | rl24;
        }
        uint32_t* o = out24 + (c * kWave + lane) * 3;
        o[0] = pk[0] | (pk[1] << 24);
        o[1] = (pk[1] >> 8) | (pk[2] << 16);
        o[2] = (pk[2] >> 16) | (pk[3] << 8);
        if (lane == 0) base[c] = col0;
    }
    if (bad) s_bad = 1;
    __syncthreads();
    if (threadIdx.x == 0 && s < n_slabs) {
        exc[s] = s_bad;
        if (s_bad) atomicAdd(n_exc, 1);
    }
}

__device__ __forceinline__ int64_t fr_uload(const int64_t* __restrict__ p) {
    const int64_t v = *p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | (uint64_t)lo);
}

// a wave-uniform pointer pinned to scalar registers: `p + lane_offset` then becomes a load with a scalar base and a 32-bit
// vector offset instead of a 64-bit vector address the compiler builds with multiply-adds per load
// (address space 1 = global: the rebuilt pointer must not decay to a generic one, whose loads are flat_load)
template <typename T>
__device__ __forceinline__ const T __attribute__((address_space(1))) * fr_uniform_ptr(const T* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const T __attribute__((address_space(1))) *)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ void fr_barrier() { // LDS drained, then the workgroup barrier (no vmcnt drain)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- candidate emission (shared by both filters) --------------------------------------------------------
// After the row sums of a slab are final, thread tid holds the ballots m[i] of rows i * THREADS + tid,
// i < NI. The candidates of the slab are written to its list segment in ascending row order = (i, wave,
// lane) order: every wave publishes popcount(m[i]) to s_cnt[i * NW + wave], one workgroup barrier, every
// wave scans all NI * NW counts and writes its own candidates. Positions never depend on timing.
// lo / cap: only list positions [lo, lo + cap) are written, at seg[position - lo] (the one-launch pass keeps a
// 1024-entry list in LDS and takes a slab with more candidates in several turns).
template <int NI, int NW>
__device__ __forceinline__ unsigned fr_emit_candidates(const unsigned long long (&m)[NI], const unsigned (&fl)[NI],
                                                       const unsigned* s_cnt, int lane, int wave, int tid,
                                                       uint16_t* __restrict__ seg, unsigned lo = 0u, unsigned cap = ~0u) {
    constexpr int NC = NI * NW, PER = NC / kWave;
    static_assert(NC % kWave == 0 && (PER == 2 || PER == 4), "counts per lane");
    unsigned c[PER], offs[PER], tot = 0;
#pragma unroll
    for (int p = 0; p < PER; ++p) { c[p] = s_cnt[lane * PER + p]; tot += c[p]; }
    unsigned incl = tot;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const unsigned o = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += o;
    }
    offs[0] = incl - tot;
#pragma unroll
    for (int p = 1; p < PER; ++p) offs[p] = offs[p - 1] + c[p - 1];
    const unsigned total = __shfl(incl, kWave - 1, kWave);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = i * NW + wave; // wave-uniform
        unsigned sel = offs[0];
#pragma unroll
        for (int p = 1; p < PER; ++p) sel = (idx % PER) == p ? offs[p] : sel;
        const unsigned off = __shfl(sel, idx / PER, kWave);
        if ((m[i] >> lane) & 1ULL) {
            const unsigned pos = off + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[i], 0u)) - lo;
            if (pos < cap) // (unsigned: positions below lo wrap around and fail the test too)
                seg[pos] = (uint16_t)(fl[i] | (unsigned)(i * (NW * kWave) + tid)); // fl: the fixed-point pass's flag bits (0 for the filters)
        }
    }
    return total;
}

// ---- pass A ------------------------------------------------------------------------------------------
template <int SLAB_ROWS, int kFrU>
__global__ __launch_bounds__(FrCfg<SLAB_ROWS>::threads) void k_fr_filter(const uint4* __restrict__ stream,
                                                             const int64_t* __restrict__ fr_off, int64_t total_chunks,
                                                             const double* __restrict__ logp, int64_t n_rows,
                                                             int64_t n_slabs, double cand_cut,
                                                             uint16_t* __restrict__ cand_list, unsigned* __restrict__ cand_cnt,
                                                             unsigned long long* __restrict__ cand_count) {
    using Cf = FrCfg<SLAB_ROWS>;
    constexpr int kFrRowBits = Cf::row_bits, kFrSlabRows = Cf::slab_rows, kFrSlots = Cf::slots, kFrWaves = Cf::waves,
                  kFrThreads = Cf::threads;
    constexpr uint32_t kFrRowMask = Cf::row_mask;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sums = reinterpret_cast<double*>(smem);
    unsigned* s_cnt = reinterpret_cast<unsigned*>(smem + Cf::lds); // [slab_rows / 64] candidate counts
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    for (int i = tid; i < kFrSlots; i += kFrThreads) sums[i] = 0.0;
    fr_barrier();

    unsigned long long n_cand = 0; // counted by lane 0 of each wave
    const int64_t G = (int64_t)gridDim.x;
    for (int64_t s = (int64_t)blockIdx.x; s < n_slabs; s += G) {
        const int64_t c0 = fr_uload(fr_off + s), c1 = fr_uload(fr_off + s + 1);
        // -------- phase 1: any wave, any chunk: gather, unordered LDS add
        uint4 q[kFrU];
#pragma unroll
        for (int u = 0; u < kFrU; ++u) {
            const int64_t c = c0 + wave + (int64_t)u * kFrWaves;
            q[u] = stream[(c < c1 ? c : total_chunks) * kWave + lane]; // past the end: the no-op chunk
        }
        for (int64_t cb = c0 + wave; cb < c1; cb += (int64_t)kFrU * kFrWaves) {
            double v[kFrU][4];
#pragma unroll
            for (int u = 0; u < kFrU; ++u) {
                const uint32_t e[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) v[u][k] = logp[e[k] >> kFrRowBits];
            }
            uint4 qn[kFrU]; // next step's chunks: issued behind the gathers, so the adds do not wait for them
#pragma unroll
            for (int u = 0; u < kFrU; ++u) {
                const int64_t c = cb + (int64_t)(kFrU + u) * kFrWaves;
                qn[u] = stream[(c < c1 ? c : total_chunks) * kWave + lane];
            }
#pragma unroll
            for (int u = 0; u < kFrU; ++u) {
                const uint32_t e[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    __hip_atomic_fetch_add(&sums[e[k] & kFrRowMask], v[u][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int u = 0; u < kFrU; ++u) q[u] = qn[u];
        }
        fr_barrier();
        // -------- phase 2: list the rows that might survive, clear the sums
        const int64_t r0 = s * kFrSlabRows;
        constexpr int NI = kFrSlabRows / kFrThreads;
        unsigned long long m[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int rl = i * kFrThreads + tid;
            const double sv = sums[rl];
            sums[rl] = 0.0;
            m[i] = __ballot(r0 + rl < n_rows && sv >= cand_cut);
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) s_cnt[i * kFrWaves + wave] = (unsigned)__builtin_popcountll(m[i]);
        }
        if (tid < 64) sums[kFrSlabRows + tid] = 0.0; // the sink
        fr_barrier();
        const unsigned no_flags[NI] = {};
        const unsigned total = fr_emit_candidates<NI, kFrWaves>(m, no_flags, s_cnt, lane, wave, tid, cand_list + s * kFrSlabRows);
        if (tid == 0) cand_cnt[s] = total;
        n_cand += total; // wave-uniform
    }
    // The count lets the host see when the filter stops paying off: one global atomic per workgroup
    // (one per wave -- thousands of same-address atomics -- cost ~50 us). n_cand is workgroup-uniform.
    if (tid == 0 && n_cand) atomicAdd(cand_count, n_cand);
}

// ---- pass A, quantised form ---------------------------------------------------------------------------
// The filter only has to be CONSERVATIVE, not precise: q_j = floor(-logp_j / step) (clamped to 255) is a
// lower bound of -logp_j in units of `step`, so step * sum(q) <= -s_i and
//     s_i >= log(cut_off)   ==>   sum(q) <= T = floor(-log(cut_off) / step).
// With step = -log(cut_off) / 224 one byte per basic event suffices and the whole table (E <= 112 K) sits
// in LDS beside 8192 u32 row sums: the filter pass issues NO texture-path gathers (the path that bounds
// the fp64 filter at ~33 clk per 64 lanes), only ds_read_u8 + ds_add_u32, and the TA just streams.
// The price is a looser bound (about 0.08 per entry in log units): more candidates for the exact pass B.
__global__ __launch_bounds__(256) void k_fr_q8_table(const double* __restrict__ logp, int32_t n_events, double inv_step,
                                                     uint8_t* __restrict__ q8) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_events) { // the row stream's two sentinel columns (kernels_ls.hip): a padding entry, a row past the model's end
        if (j == n_events) q8[j] = 0;
        else if (j == n_events + 1) q8[j] = 255;
        return;
    }
    const double neg = -logp[j]; // >= 0, +inf for p = 0
    // (1 - 2^-30): the bound must stay a lower bound although (a) the product may round up across an
    // integer and (b) the CSR-order fp64 sum s_i differs from the exact sum by up to len * 2^-53 relative
    // (2^-30 covers rows of up to 2^22 entries)
    const double lv = neg * inv_step * (1.0 - 0x1.0p-30);
    q8[j] = (uint8_t)(lv >= 255.0 ? 255 : (int)lv);
}

// the 4-bit form: two events per byte (event 2 b in the low half), levels clamp at 15
__global__ __launch_bounds__(256) void k_fr_q4_table(const double* __restrict__ logp, int32_t n_events, double inv_step,
                                                     uint8_t* __restrict__ q4) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * b >= n_events + 2) return;
    unsigned v = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = 2 * b + h;
        // behind the last event: the row stream's two sentinel columns (kernels_ls.hip) -- a padding entry (0), a row past the
        // model's end (15: alone enough to kill it); the chunk filter reads neither
        unsigned q = j == n_events ? 0u : 15u;
        if (j < n_events) {
            const double lv = -logp[j] * inv_step * (1.0 - 0x1.0p-30); // as in k_fr_q8_table: a lower bound
            q = lv >= 15.0 ? 15u : (unsigned)(int)lv;
        }
        v |= q << (4 * h);
    }
    q4[b] = (uint8_t)v;
}

constexpr int kQ8SlabRows = 8192, kQ8RowBits = 14, kQ8Waves = 16, kQ8Threads = kQ8Waves * kWave;
constexpr uint32_t kQ8RowMask = (1u << kQ8RowBits) - 1;
constexpr int kQ8Sinks = 128; // (the 24-bit stream's padding entries use the even ones: 8192 + 2 k)
constexpr size_t kQ8SumsBytes = (size_t)(kQ8SlabRows + kQ8Sinks) * 4;

struct Q8Chunk { // one lane's share of a chunk in flight: 32-bit form x y z w = four entries; 24-bit form x y z = 12 packed bytes, w = the chunk's first column
    uint32_t x, y, z, w;
};

template <bool P24>
__device__ __forceinline__ Q8Chunk q8_load_chunk(const uint32_t* __restrict__ stream, const uint32_t* __restrict__ cbase, int64_t c, int lane,
                                                 uint32_t zero = 0u, bool zero_given = false) {
    Q8Chunk r;
    if (P24) {
        // (wave-uniform chunk pointer + a 32-bit lane offset: the load takes the pointer from scalar registers; as
        // `stream + (c * 64 + lane) * 3` the whole index went through 64-bit vector multiplies, two per load)
        const uint32_t __attribute__((address_space(1))) * chunk = fr_uniform_ptr(stream + c * (int64_t)(kWave * 3));
        const uint32_t __attribute__((address_space(1))) * p = chunk + (uint32_t)lane * 3u;
        r.x = p[0]; r.y = p[1]; r.z = p[2];
        // The chunk's first column: a VECTOR load (all lanes the same address). Left to the compiler this wave-uniform load
        // becomes s_load_dword, whose completion shares lgkmcnt with the LDS operations and returns out of order: every
        // step then waits (lgkmcnt(0)) for the NEXT step's scalar loads -- a full memory latency -- before its first LDS
        // add. (That is what held the first 24-bit and 16-bit attempts at the 32-bit stream's time.)
        // (address space 1 = global: a generic pointer would become flat_load, which counts in lgkmcnt as well)
        // (scalar base + an opaque zero in a vector register: no address arithmetic on the vector ALU)
        // (a BYTE offset on a char pointer: as an element index the compiler scaled it with a 64-bit vector add into a register
        // pair it had just used as the destination of a chunk load in flight -- and waited for that load: `vmcnt(0)` every pass)
        // (`zero`: the caller's opaque zero, made ONCE per kernel and kept in a register of its own -- made here, per load, it
        // landed in a register that was the destination of a chunk load still in flight: the same wait)
        if (!zero_given) asm volatile("" : "+v"(zero));
        const char __attribute__((address_space(1))) * bp = reinterpret_cast<const char __attribute__((address_space(1))) *>(fr_uniform_ptr(cbase + c));
        r.w = *reinterpret_cast<const uint32_t __attribute__((address_space(1))) *>(bp + zero);
    } else {
        const uint32_t __attribute__((address_space(1))) * p = fr_uniform_ptr(stream + c * (int64_t)(kWave * 4)) + (uint32_t)lane * 4u;
        r.x = p[0]; r.y = p[1]; r.z = p[2]; r.w = p[3];
    }
    return r;
}

// column and row slot of a chunk's four entries
template <bool P24>
__device__ __forceinline__ void q8_decode(const Q8Chunk& q, uint32_t (&col)[4], uint32_t (&row)[4]) {
    if (P24) {
        const uint32_t e0 = q.x, e1 = __builtin_amdgcn_alignbit(q.y, q.x, 24), e2 = __builtin_amdgcn_alignbit(q.z, q.y, 16), e3 = q.z >> 8;
        col[0] = q.w + ((e0 >> 14) & (kP24ColSpan - 1u)); row[0] = e0 & kQ8RowMask;
        col[1] = q.w + ((e1 >> 14) & (kP24ColSpan - 1u)); row[1] = e1 & kQ8RowMask;
        col[2] = q.w + ((e2 >> 14) & (kP24ColSpan - 1u)); row[2] = e2 & kQ8RowMask;
        col[3] = q.w + (e3 >> 14);                         row[3] = e3 & kQ8RowMask;
    } else {
        const uint32_t e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { col[k] = e[k] >> kQ8RowBits; row[k] = e[k] & kQ8RowMask; }
    }
}

template <int U, bool P24, bool NIB = false> // NIB: the table holds four bits per event (112 K < E <= 260 096)
__global__ __launch_bounds__(kQ8Threads) void k_fr_filter_q8(const uint32_t* __restrict__ stream, const uint32_t* __restrict__ cbase,
                                                             const int32_t* __restrict__ exc,
                                                             const int64_t* __restrict__ fr_off, int64_t total_chunks,
                                                             const uint8_t* __restrict__ q8, int32_t n_events,
                                                             int64_t n_rows, int64_t n_slabs, int threshold,
                                                             uint16_t* __restrict__ cand_list, unsigned* __restrict__ cand_cnt,
                                                             unsigned long long* __restrict__ cand_count, int abl_arg) {
#ifdef CANOPY_ABLATIONS
    const int abl = abl_arg; // CANOPY_FR_DEBUG: timing ablations of benchmarks/ (wrong results)
#else
    constexpr int abl = 0;   // the default build has no ablation branch in the loop
    (void)abl_arg;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* sums = reinterpret_cast<unsigned*>(smem);
    uint8_t* tab = reinterpret_cast<uint8_t*>(smem + kQ8SumsBytes);
    const int tab_bytes = NIB ? (n_events + 1) / 2 : n_events;
    unsigned* s_cnt = reinterpret_cast<unsigned*>(smem + kQ8SumsBytes + (((size_t)tab_bytes + 15) / 16) * 16); // [128]
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int64_t G = (int64_t)gridDim.x;
    // the first chunks of a slab are requested BEFORE the workgroup-wide steps in front of it (table staging here,
    // the previous slab's end step in the loop): no slab starts with a cold stream
    Q8Chunk q[U];
#define FR_FIRST_CHUNKS(SLAB)                                                                       \
    if ((SLAB) < n_slabs) {                                                                         \
        const int64_t f0 = fr_uload(fr_off + (SLAB)), f1 = fr_uload(fr_off + (SLAB) + 1);           \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                             \
            const int64_t c = f0 + wave + (int64_t)u * kQ8Waves;                                    \
            q[u] = q8_load_chunk<P24>(stream, cbase, c < f1 ? c : total_chunks, lane);              \
        }                                                                                           \
    }
    FR_FIRST_CHUNKS((int64_t)blockIdx.x)
    for (int i = tid; i < kQ8SlabRows + kQ8Sinks; i += kQ8Threads) sums[i] = 0u;
    { // the table, 16 bytes per lane per step (its buffer is padded to a multiple of 16)
        const uint4* src = reinterpret_cast<const uint4*>(q8);
        uint4* dst = reinterpret_cast<uint4*>(tab);
        for (int i = tid; i < (tab_bytes + 15) / 16; i += kQ8Threads) dst[i] = src[i];
    }
    fr_barrier();

    unsigned long long n_cand = 0;
    for (int64_t s = (int64_t)blockIdx.x; s < n_slabs; s += G) {
        const int64_t c0 = fr_uload(fr_off + s);
        // an exception slab of the 24-bit form (a chunk too wide for the 10-bit column offset): no chunk is read, every row
        // goes to the exact pass
        const bool all_rows = P24 && __builtin_amdgcn_readfirstlane(exc[s]) != 0;
        const int64_t c1 = all_rows ? c0 : fr_uload(fr_off + s + 1);
        if (abl & 16) { FR_FIRST_CHUNKS(s) } // A/B switch (CANOPY_FR_DEBUG=16): the slab's first chunks requested only now
        for (int64_t cb = c0 + wave; cb < c1; cb += (int64_t)U * kQ8Waves) {
            Q8Chunk qn[U]; // next step's chunks first: nothing else is in the vector-memory queue
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t c = cb + (int64_t)(U + u) * kQ8Waves;
                // timing ablation (CANOPY_FR_DEBUG=1, wrong results): every slab re-reads its first 64 chunks, which
                // stay cache-resident -> the pass without its HBM stream
                const int64_t cc = (abl & 1) ? c0 + ((c - c0) & 63) : c;
                qn[u] = q8_load_chunk<P24>(stream, cbase, c < c1 ? cc : total_chunks, lane);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t col[4], row[4];
                q8_decode<P24>(q[u], col, row);
                unsigned v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = NIB ? ((unsigned)tab[col[k] >> 1] >> ((col[k] & 1u) << 2)) & 15u : (unsigned)tab[col[k]];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    __hip_atomic_fetch_add(&sums[row[k]], v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (abl & 2) { // timing ablation (CANOPY_FR_DEBUG=2, wrong results): the LDS work of every chunk twice
                    unsigned v2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v2[k] = tab[col[k] ^ 1u];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        __hip_atomic_fetch_add(&sums[row[k] ^ 1u], v2[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) q[u] = qn[u];
        }
        if (!(abl & 16)) { FR_FIRST_CHUNKS(s + G) } // the next slab's stream starts under this slab's end step
        if (abl & 8) continue;                    // timing ablation (wrong results): no end step, no barrier
        fr_barrier();
        if (abl & 4) continue;                    // timing ablation (wrong results): the barrier only
        const int64_t r0 = s * kQ8SlabRows;
        constexpr int NI = kQ8SlabRows / kQ8Threads;
        unsigned long long m[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int rl = i * kQ8Threads + tid;
            const int sv = (int)sums[rl];
            sums[rl] = 0u;
            m[i] = __ballot(r0 + rl < n_rows && (all_rows || sv <= threshold));
        }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) s_cnt[i * kQ8Waves + wave] = (unsigned)__builtin_popcountll(m[i]);
        }
        if (tid < kQ8Sinks) sums[kQ8SlabRows + tid] = 0u; // the sinks
        fr_barrier();
        const unsigned no_flags[NI] = {};
        const unsigned total = fr_emit_candidates<NI, kQ8Waves>(m, no_flags, s_cnt, lane, wave, tid, cand_list + s * kQ8SlabRows);
        if (tid == 0) cand_cnt[s] = total;
        n_cand += total; // workgroup-uniform
    }
#undef FR_FIRST_CHUNKS
    if (tid == 0 && n_cand) atomicAdd(cand_count, n_cand); // one global atomic per workgroup, as in k_fr_filter
}

// ---- pass A, quantised form, the slab's end step folded into the next slabs' work (24-bit stream, rows of <= 257 entries)
// k_fr_filter_q8 ends every slab with a workgroup-wide step (two barriers, the ballots over the 8192 sums, the ordered
// candidate list) during which nothing is added: with the 24-bit stream that step is what separates the pass from the HBM
// time of its stream (end step removed: 0.368 -> 0.295 ms at config 3). Here the row sums are PACKED two to a 32-bit word
// (u16 each: a row of L entries sums to <= 255 L, so L <= 257 cannot carry into its neighbour), which makes room for TWO
// slabs of sums beside the table, and the end step of slab k is spread over the two following iterations of all 16 waves:
//   iteration k    every wave adds its chunks of slab k into buffer k & 1;                               barrier
//   iteration k+1  every wave takes ITS 512 rows of slab k out of that buffer (threshold, ballots kept in registers,
//                  clear), publishes its count, then adds slab k+1 into the other buffer;                barrier
//   iteration k+2  every wave reads the 16 counts of slab k and writes its candidates at their place in the slab's list.
// One barrier per slab, no wave ever waits for the list of a slab, the lists are the same ascending lists as before.
// (Dedicated finishing waves, 14 + 2, were measured first: 0.343 ms -- the pass is bound by what ITS adding waves get
// through, and two fewer of them cost what the hidden end step gained; 15 + 1: 0.60, one wave cannot finish 8192 rows.)
constexpr int kQ8oBufWords = kQ8SlabRows / 2 + 64;
constexpr size_t kQ8oSumsBytes = (size_t)2 * kQ8oBufWords * 4;
constexpr int kQ8oMaxRowLen = 257;
constexpr int kQ8oWaveRows = kQ8SlabRows / kQ8Waves; // 512 rows = 256 words = 4 words per lane

template <int U>
__global__ __launch_bounds__(kQ8Threads) void k_fr_filter_q8o(const uint32_t* __restrict__ stream, const uint32_t* __restrict__ cbase,
                                                              const int32_t* __restrict__ exc, const int64_t* __restrict__ fr_off,
                                                              int64_t total_chunks, const uint8_t* __restrict__ q8, int32_t n_events,
                                                              int64_t n_rows, int64_t n_slabs, int threshold,
                                                              uint16_t* __restrict__ cand_list, unsigned* __restrict__ cand_cnt,
                                                              unsigned long long* __restrict__ cand_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* bufs = reinterpret_cast<unsigned*>(smem);                                  // [2][kQ8oBufWords]
    uint8_t* tab = reinterpret_cast<uint8_t*>(smem + kQ8oSumsBytes);
    unsigned* s_cnt = reinterpret_cast<unsigned*>(smem + kQ8oSumsBytes + (((size_t)n_events + 15) / 16) * 16); // [2][16]
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    uint32_t zero_v = 0u; // an opaque zero byte offset in a vector register (q8_load_chunk)
    asm volatile("" : "+v"(zero_v));
    const int64_t G = (int64_t)gridDim.x;
    // A CONTIGUOUS block of slabs per workgroup (the first n_slabs % G workgroups take one more), not slabs blockIdx, blockIdx + G,
    // ...: 1 % faster at config 3 (0.3276 against 0.3313 ms per step, three alternated runs each on one box; giving every XCD a
    // contiguous eighth of the model instead: 0.3322, no better than round-robin)
    const int64_t q_wg = n_slabs / G, r_wg = n_slabs % G, b_wg = (int64_t)blockIdx.x;
    const int64_t s_begin = b_wg * q_wg + (b_wg < r_wg ? b_wg : r_wg), s_end = s_begin + q_wg + (b_wg < r_wg ? 1 : 0);
    Q8Chunk q[U];
#define FRO_FIRST_CHUNKS(SLAB)                                                                      \
    if ((SLAB) < s_end) {                                                                         \
        const int64_t f0 = fr_uload(fr_off + (SLAB));                                               \
        const int64_t f1 = __builtin_amdgcn_readfirstlane(exc[(SLAB)]) != 0 ? f0 : fr_uload(fr_off + (SLAB) + 1); \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                             \
            const int64_t c = f0 + wave + (int64_t)u * kQ8Waves;                                    \
            q[u] = q8_load_chunk<true>(stream, cbase, c < f1 ? c : total_chunks, lane, zero_v, true); \
        }                                                                                           \
    }
    FRO_FIRST_CHUNKS(s_begin)
    for (int i = tid; i < 2 * kQ8oBufWords; i += kQ8Threads) bufs[i] = 0u;
    {
        const uint4* src = reinterpret_cast<const uint4*>(q8);
        uint4* dst = reinterpret_cast<uint4*>(tab);
        for (int i = tid; i < (n_events + 15) / 16; i += kQ8Threads) dst[i] = src[i];
    }
    fr_barrier();

    unsigned long long n_cand = 0;                    // wave 0: candidates of the slabs this workgroup listed
    unsigned long long mlo[4] = {0, 0, 0, 0}, mhi[4] = {0, 0, 0, 0}; // the ballots of the slab whose counts are published (wave-uniform)
    int64_t s_count = -1, s_emit = -1;                // slab to take out of its buffer / slab to list, this iteration
    bool all_count = false;
    const unsigned long long below = (1ULL << lane) - 1ULL;
    int it = 0;
    for (int64_t s = s_begin;; s += 1, ++it) {
        const bool have = s < s_end;
        // an exception slab of the 24-bit form: no chunk is read, every row goes to the exact pass
        const bool all_rows = have && __builtin_amdgcn_readfirstlane(exc[have ? s : 0]) != 0;
        if (s_emit >= 0) { // ---- the list of slab s_emit: this wave's candidates behind those of the waves before it
            const unsigned* cnt = s_cnt + ((it - 1) & 1) * kQ8Waves; // published during the previous iteration
            unsigned before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < kQ8Waves; ++w) {
                const unsigned c = cnt[w];
                before += w < wave ? c : 0u;
                total += c;
            }
            uint16_t* seg = cand_list + s_emit * (int64_t)kQ8SlabRows;
            unsigned n_list = before;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned long long a = mlo[u], b = mhi[u];
                if ((a | b) != 0ULL) { // wave-uniform
                    const int rl = wave * kQ8oWaveRows + 2 * (u * kWave + lane); // the word's low row, local to the slab
                    const bool k_lo = (a >> lane) & 1ULL, k_hi = (b >> lane) & 1ULL;
                    const unsigned pre = n_list + (unsigned)__builtin_popcountll(a & below) + (unsigned)__builtin_popcountll(b & below);
                    if (k_lo) seg[pre] = (uint16_t)rl;
                    if (k_hi) seg[pre + (k_lo ? 1u : 0u)] = (uint16_t)(rl + 1);
                    n_list += (unsigned)__builtin_popcountll(a) + (unsigned)__builtin_popcountll(b);
                }
            }
            if (tid == 0) cand_cnt[s_emit] = total;
            n_cand += total;
        }
        if (s_count >= 0) { // ---- this wave's 512 rows of slab s_count out of its buffer; the count is published
            unsigned* sums = bufs + (size_t)((it - 1) & 1) * kQ8oBufWords;
            const int64_t r0 = s_count * kQ8SlabRows + (int64_t)wave * kQ8oWaveRows;
            unsigned d[4], n_w = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int w = wave * (kQ8oWaveRows / 2) + u * kWave + lane;
                d[u] = sums[w];
                sums[w] = 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t r = r0 + 2 * (u * kWave + lane);
                const bool k_lo = r < n_rows && (all_count || (int)(d[u] & 0xffffu) <= threshold);
                const bool k_hi = r + 1 < n_rows && (all_count || (int)(d[u] >> 16) <= threshold);
                mlo[u] = __ballot(k_lo);
                mhi[u] = __ballot(k_hi);
                n_w += (unsigned)__builtin_popcountll(mlo[u]) + (unsigned)__builtin_popcountll(mhi[u]);
            }
            if (wave == 0) sums[kQ8SlabRows / 2 + lane] = 0u; // the sinks
            if (lane == 0) s_cnt[(it & 1) * kQ8Waves + wave] = n_w;
        }
        if (have) { // ---- add slab s into buffer it & 1
            unsigned* sums = bufs + (size_t)(it & 1) * kQ8oBufWords;
            const int64_t c0 = fr_uload(fr_off + s);
            const int64_t c1 = all_rows ? c0 : fr_uload(fr_off + s + 1);
            // two register sets in ping-pong (a copy "q = qn" at the end of the body makes the compiler wait for all of the
            // next chunks every iteration)
#define Q8O_HALF(QC, QN, CB)                                                                                           \
            {                                                                                                          \
                _Pragma("unroll") for (int u = 0; u < U; ++u) { /* next step's chunks first */                          \
                    const int rc = (CB) + (U + u) * kQ8Waves; /* relative to the slab's first chunk: 32-bit SCALAR compares */ \
                    QN[u] = q8_load_chunk<true>(stream, cbase, rc < n_c ? c0 + rc : total_chunks, lane, zero_v, true); \
                } /* (pinned in front of the step's LDS work with a scheduling barrier: 0.330 against 0.315 ms per step -- the        \
                     compiler's own placement of these requests is the better one) */                                        \
                _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                        \
                    uint32_t col[4], row[4];                                                                           \
                    q8_decode<true>(QC[u], col, row);                                                                  \
                    unsigned v[4];                                                                                     \
                    _Pragma("unroll") for (int k = 0; k < 4; ++k) v[k] = tab[col[k]];                                  \
                    _Pragma("unroll") for (int k = 0; k < 4; ++k) /* word row / 2, half row & 1 */                     \
                        __hip_atomic_fetch_add(&sums[row[k] >> 1], v[k] << ((row[k] & 1u) << 4), __ATOMIC_RELAXED,     \
                                               __HIP_MEMORY_SCOPE_WORKGROUP);                                          \
                }                                                                                                      \
            }
            {
                Q8Chunk q2[U];
                bool in_q = true; // which set holds the chunks requested last (after the loop: the no-op chunks past the end)
                // (a 64-bit `cb < c1` is a VECTOR compare whose operand the compiler moved into registers it had just used as the
                // destination of a chunk load still in flight -- and waited for that load first: `s_waitcnt vmcnt(0)` at the top
                // of this loop, every pass)
                const int n_c = (int)(c1 - c0);
                for (int cb = wave; cb < n_c;) {
                    Q8O_HALF(q, q2, cb)
                    cb += U * kQ8Waves;
                    in_q = false;
                    if (cb >= n_c) break;
                    Q8O_HALF(q2, q, cb)
                    cb += U * kQ8Waves;
                    in_q = true;
                }
                (void)in_q;
            }
#undef Q8O_HALF
            FRO_FIRST_CHUNKS(s + 1) // the next slab's stream starts before this slab's barrier (always into q)
        }
        s_emit = s_count;           // (its counts are published once this iteration's barrier is passed)
        s_count = have ? s : -1;
        all_count = all_rows;
        if (!have && s_emit < 0) break; // nothing left in flight (no barrier needed: nobody reads anything of ours any more)
        fr_barrier();
    }
#undef FRO_FIRST_CHUNKS
    if (tid == 0 && n_cand) atomicAdd(cand_count, n_cand); // one global atomic per workgroup
}

// ---- pass A, quantised form, wide event spaces (112 K < E <= 2^18): one part of the table resident per sweep ----
struct Q8Parts {
    int H;
    int32_t c_lo[kQ8MaxParts + 1]; // first column of part h; c_lo[H] = n_events
};

template <int U>
__global__ __launch_bounds__(kQ8Threads) void k_fr_filter_q8p(const uint4* __restrict__ stream, const int64_t* __restrict__ fr_off,
                                                              int64_t total_chunks, const uint8_t* __restrict__ q8, Q8Parts P,
                                                              int64_t n_rows, int64_t n_slabs, int threshold,
                                                              uint16_t* __restrict__ cand_list, unsigned* __restrict__ cand_cnt,
                                                              unsigned long long* __restrict__ cand_count, size_t tab_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* sums = reinterpret_cast<unsigned*>(smem);
    uint8_t* tab = reinterpret_cast<uint8_t*>(smem + kQ8SumsBytes);
    unsigned* s_cnt = reinterpret_cast<unsigned*>(smem + kQ8SumsBytes + tab_bytes); // [128]
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int64_t G = (int64_t)gridDim.x;
    const int H = P.H;
    constexpr int NI = kQ8SlabRows / kQ8Threads;
    uint4 q[U];
    for (int i = tid; i < kQ8SlabRows + kQ8Sinks; i += kQ8Threads) sums[i] = 0u;
    unsigned long long n_cand = 0;
    for (int h = 0; h < H; ++h) {
        const int a0 = P.c_lo[h] & ~15, a1 = P.c_lo[h + 1]; // the table bytes of this part (16-byte aligned start)
#define FRP_FIRST_CHUNKS(SLAB)                                                                      \
        if ((SLAB) < n_slabs) {                                                                     \
            const int64_t f0 = fr_uload(fr_off + (SLAB) * H + h), f1 = fr_uload(fr_off + (SLAB) * H + h + 1); \
            _Pragma("unroll") for (int u = 0; u < U; ++u) {                                         \
                const int64_t c = f0 + wave + (int64_t)u * kQ8Waves;                                \
                q[u] = stream[(c < f1 ? c : total_chunks) * kWave + lane];                          \
            }                                                                                       \
        }
        FRP_FIRST_CHUNKS((int64_t)blockIdx.x)
        fr_barrier(); // everybody is done with the previous part's table
        {
            const uint4* src = reinterpret_cast<const uint4*>(q8 + a0);
            uint4* dst = reinterpret_cast<uint4*>(tab);
            for (int i = tid; i < (a1 - a0 + 15) / 16; i += kQ8Threads) dst[i] = src[i];
        }
        fr_barrier();
        for (int64_t s = (int64_t)blockIdx.x; s < n_slabs; s += G) {
            const int64_t c0 = fr_uload(fr_off + s * H + h), c1 = fr_uload(fr_off + s * H + h + 1);
            uint16_t* part_sums = cand_list + s * kQ8SlabRows; // this slab's row sums between two sweeps (its list comes last)
            // resume: the sums of the earlier parts are requested now and ADDED after the chunk loop (adds commute; other
            // waves may already be adding to these rows): their latency hides behind the slab's own work
            unsigned resume[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) resume[i] = h > 0 ? (unsigned)part_sums[i * kQ8Threads + tid] : 0u;
            for (int64_t cb = c0 + wave; cb < c1; cb += (int64_t)U * kQ8Waves) {
                uint4 qn[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t c = cb + (int64_t)(U + u) * kQ8Waves;
                    qn[u] = stream[(c < c1 ? c : total_chunks) * kWave + lane];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t e[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                    unsigned v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int col = (int)(e[k] >> kQ8RowBits);
                        v[k] = tab[col >= a0 ? col - a0 : 0]; // (padding entries carry column 0 and a sink row)
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        __hip_atomic_fetch_add(&sums[e[k] & kQ8RowMask], v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) q[u] = qn[u];
            }
            FRP_FIRST_CHUNKS(s + G)
            if (h > 0) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    __hip_atomic_fetch_add(&sums[i * kQ8Threads + tid], resume[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            fr_barrier();
            const int64_t r0 = s * kQ8SlabRows;
            if (h + 1 < H) { // not the last part: park the sums (saturated: 65535 is above every threshold)
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int rl = i * kQ8Threads + tid;
                    const unsigned sv = sums[rl];
                    sums[rl] = 0u;
                    part_sums[rl] = (uint16_t)(sv < 65535u ? sv : 65535u);
                }
                if (tid < kQ8Sinks) sums[kQ8SlabRows + tid] = 0u; // the sinks
                fr_barrier();
                continue;
            }
            unsigned long long m[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int rl = i * kQ8Threads + tid;
                const int sv = (int)sums[rl];
                sums[rl] = 0u;
                m[i] = __ballot(r0 + rl < n_rows && sv <= threshold);
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < NI; ++i) s_cnt[i * kQ8Waves + wave] = (unsigned)__builtin_popcountll(m[i]);
            }
            if (tid < kQ8Sinks) sums[kQ8SlabRows + tid] = 0u; // the sinks
            fr_barrier();
            const unsigned no_flags[NI] = {};
            const unsigned total = fr_emit_candidates<NI, kQ8Waves>(m, no_flags, s_cnt, lane, wave, tid, cand_list + s * kQ8SlabRows);
            if (tid == 0) cand_cnt[s] = total;
            n_cand += total; // workgroup-uniform
        }
#undef FRP_FIRST_CHUNKS
    }
    if (tid == 0 && n_cand) atomicAdd(cand_count, n_cand);
}

// ---- pass A, quantised form, wide event spaces, the end step folded into the next iterations (rows of <= 257 entries) ----
// k_fr_filter_q8p with k_fr_filter_q8o's structure: packed u16 row sums in two buffers, ONE barrier per (slab, part); every
// wave takes its 512 rows of the previous (slab, part) out of the other buffer after the barrier -- parks them as they
// are (packed words, in the slab's list segment) when more parts follow, or thresholds them and publishes its count after
// the last part -- while its loads of the current (slab, part) are in flight, and lists its candidates one iteration later.
template <int U>
__global__ __launch_bounds__(kQ8Threads) void k_fr_filter_q8po(const uint4* __restrict__ stream, const int64_t* __restrict__ fr_off,
                                                               int64_t total_chunks, const uint8_t* __restrict__ q8, Q8Parts P,
                                                               int64_t n_rows, int64_t n_slabs, int threshold,
                                                               uint16_t* __restrict__ cand_list, unsigned* __restrict__ cand_cnt,
                                                               unsigned long long* __restrict__ cand_count, size_t tab_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* bufs = reinterpret_cast<unsigned*>(smem);                                  // [2][kQ8oBufWords]
    uint8_t* tab = reinterpret_cast<uint8_t*>(smem + kQ8oSumsBytes);
    unsigned* s_cnt = reinterpret_cast<unsigned*>(smem + kQ8oSumsBytes + tab_bytes);     // [2][16]
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int64_t G = (int64_t)gridDim.x;
    const int H = P.H;
    uint4 q[U];
    for (int i = tid; i < 2 * kQ8oBufWords; i += kQ8Threads) bufs[i] = 0u;
    unsigned long long n_cand = 0;
    unsigned long long mlo[4] = {0, 0, 0, 0}, mhi[4] = {0, 0, 0, 0};
    int64_t s_take = -1, s_emit = -1; // slab whose (previous) part is taken out of its buffer / slab to list, this iteration
    bool take_last = false;           // ... and whether that part was the slab's last
    const unsigned long long below = (1ULL << lane) - 1ULL;
    int it = 0;
    for (int h = 0; h <= H; ++h) { // h = H: the drain iterations (nothing to add any more)
        const int a0 = h < H ? (P.c_lo[h] & ~15) : 0, a1 = h < H ? P.c_lo[h + 1] : 0;
#define FRPO_FIRST_CHUNKS(SLAB)                                                                     \
        if (h < H && (SLAB) < n_slabs) {                                                            \
            const int64_t f0 = fr_uload(fr_off + (SLAB) * H + h), f1 = fr_uload(fr_off + (SLAB) * H + h + 1); \
            _Pragma("unroll") for (int u = 0; u < U; ++u) {                                         \
                const int64_t c = f0 + wave + (int64_t)u * kQ8Waves;                                \
                q[u] = stream[(c < f1 ? c : total_chunks) * kWave + lane];                          \
            }                                                                                       \
        }
        FRPO_FIRST_CHUNKS((int64_t)blockIdx.x)
        fr_barrier(); // everybody is done adding with the previous part's table (and the first buffers are cleared)
        if (h < H) {
            const uint4* src = reinterpret_cast<const uint4*>(q8 + a0);
            uint4* dst = reinterpret_cast<uint4*>(tab);
            for (int i = tid; i < (a1 - a0 + 15) / 16; i += kQ8Threads) dst[i] = src[i];
        }
        fr_barrier();
        for (int64_t s = (int64_t)blockIdx.x;; s += G, ++it) {
            const bool have = h < H && s < n_slabs;
            if (!have && s_take < 0 && s_emit < 0) break; // (workgroup-uniform)
            if (h == H && s_take < 0 && s_emit < 0) break;
            if (s_emit >= 0) { // ---- the list of slab s_emit
                const unsigned* cnt = s_cnt + ((it - 1) & 1) * kQ8Waves;
                unsigned before = 0, total = 0;
#pragma unroll
                for (int w = 0; w < kQ8Waves; ++w) {
                    const unsigned c = cnt[w];
                    before += w < wave ? c : 0u;
                    total += c;
                }
                uint16_t* seg = cand_list + s_emit * (int64_t)kQ8SlabRows;
                unsigned n_list = before;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned long long a = mlo[u], b = mhi[u];
                    if ((a | b) != 0ULL) {
                        const int rl = wave * kQ8oWaveRows + 2 * (u * kWave + lane);
                        const bool k_lo = (a >> lane) & 1ULL, k_hi = (b >> lane) & 1ULL;
                        const unsigned pre = n_list + (unsigned)__builtin_popcountll(a & below) + (unsigned)__builtin_popcountll(b & below);
                        if (k_lo) seg[pre] = (uint16_t)rl;
                        if (k_hi) seg[pre + (k_lo ? 1u : 0u)] = (uint16_t)(rl + 1);
                        n_list += (unsigned)__builtin_popcountll(a) + (unsigned)__builtin_popcountll(b);
                    }
                }
                if (tid == 0) cand_cnt[s_emit] = total;
                n_cand += total;
            }
            s_emit = -1;
            // the sums of the earlier parts of the CURRENT slab: requested now, added after its chunk loop (adds commute)
            unsigned resume[4] = {0u, 0u, 0u, 0u};
            if (have && h > 0) {
                const unsigned* parked = reinterpret_cast<const unsigned*>(cand_list + s * (int64_t)kQ8SlabRows);
#pragma unroll
                for (int u = 0; u < 4; ++u) resume[u] = parked[wave * (kQ8oWaveRows / 2) + u * kWave + lane];
            }
            if (s_take >= 0) { // ---- this wave's 512 rows of the previous (slab, part) out of its buffer
                unsigned* sums = bufs + (size_t)((it - 1) & 1) * kQ8oBufWords;
                unsigned d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int w = wave * (kQ8oWaveRows / 2) + u * kWave + lane;
                    d[u] = sums[w];
                    sums[w] = 0u;
                }
                if (wave == 0) sums[kQ8SlabRows / 2 + lane] = 0u; // the sinks
                if (!take_last) { // more parts follow: park the packed words in the slab's (not yet written) list segment
                    unsigned* parked = reinterpret_cast<unsigned*>(cand_list + s_take * (int64_t)kQ8SlabRows);
#pragma unroll
                    for (int u = 0; u < 4; ++u) parked[wave * (kQ8oWaveRows / 2) + u * kWave + lane] = d[u];
                } else {
                    const int64_t r0 = s_take * kQ8SlabRows + (int64_t)wave * kQ8oWaveRows;
                    unsigned n_w = 0;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int64_t r = r0 + 2 * (u * kWave + lane);
                        const bool k_lo = r < n_rows && (int)(d[u] & 0xffffu) <= threshold;
                        const bool k_hi = r + 1 < n_rows && (int)(d[u] >> 16) <= threshold;
                        mlo[u] = __ballot(k_lo);
                        mhi[u] = __ballot(k_hi);
                        n_w += (unsigned)__builtin_popcountll(mlo[u]) + (unsigned)__builtin_popcountll(mhi[u]);
                    }
                    if (lane == 0) s_cnt[(it & 1) * kQ8Waves + wave] = n_w;
                    s_emit = s_take; // listed next iteration, once every wave's count is published
                }
            }
            s_take = -1;
            if (have) { // ---- add (slab s, part h) into buffer it & 1
                unsigned* sums = bufs + (size_t)(it & 1) * kQ8oBufWords;
                const int64_t c0 = fr_uload(fr_off + s * H + h), c1 = fr_uload(fr_off + s * H + h + 1);
                for (int64_t cb = c0 + wave; cb < c1; cb += (int64_t)U * kQ8Waves) {
                    uint4 qn[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int64_t c = cb + (int64_t)(U + u) * kQ8Waves;
                        qn[u] = stream[(c < c1 ? c : total_chunks) * kWave + lane];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t e[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                        unsigned v[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int col = (int)(e[k] >> kQ8RowBits);
                            v[k] = tab[col >= a0 ? col - a0 : 0]; // (padding entries carry column 0 and a sink row)
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint32_t row = e[k] & kQ8RowMask;
                            __hip_atomic_fetch_add(&sums[row >> 1], v[k] << ((row & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) q[u] = qn[u];
                }
                FRPO_FIRST_CHUNKS(s + G)
                if (h > 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        __hip_atomic_fetch_add(&sums[wave * (kQ8oWaveRows / 2) + u * kWave + lane], resume[u], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                s_take = s;
                take_last = h + 1 == H;
            }
            fr_barrier();
        }
#undef FRPO_FIRST_CHUNKS
    }
    if (tid == 0 && n_cand) atomicAdd(cand_count, n_cand);
}

// ---- pass B ------------------------------------------------------------------------------------------
// Two waves per slab walk its candidate list, 64 candidates per trip each: a lane sums one candidate row
// from the canonical CSR in CSR order. Fixed wave <- (slab, half) and lane <- list position assignment =>
// reproducible. The exact keep bits are OR-ed into a pre-cleared mask when the caller wants one.
constexpr int kRefineWaves = 4;

constexpr int kOffBlockBits = 13; // d_off16: row starts relative to their 8192-row block

struct __attribute__((packed, aligned(4))) Cols4 { int32_t c[4]; }; // one 16-byte load at 4-byte alignment (col_idx has 64 B of slack)

// Bounds of row r. With the 2-byte offsets a candidate costs a 2-byte pair out of a 128-byte line that serves 64 rows
// (the int64 row_ptr line serves 16): what the exact pass pays per listed row at HBM is lines, not bytes.
__device__ __forceinline__ void refine_row_bounds(const RowKernelArgs& a, int64_t r, int64_t* b_out, int* len_out) {
    if (a.off16 != nullptr) {
        const int64_t blk = (r >> kOffBlockBits) << kOffBlockBits;
        const int64_t base = a.row_ptr[blk]; // the lanes of a list unit share their block: one address
        const int64_t nxt = r + 1;
        const int64_t b = base + (int64_t)a.off16[r];
        const int64_t e = ((nxt & ((1 << kOffBlockBits) - 1)) == 0 || nxt == a.n_rows) ? a.row_ptr[nxt] : base + (int64_t)a.off16[nxt];
        *b_out = b;
        *len_out = (int)(e - b);
    } else {
        const int64_t b = a.row_ptr[r], e = a.row_ptr[r + 1];
        *b_out = b;
        *len_out = (int)(e - b);
    }
}

// The exact CSR-order sum of one row per lane. Columns come four at a time in 16-byte loads, table entries as one gather per
// entry position the longest row of the WAVE still has (wave-uniform trip counts: no per-lane predicated loads).
__device__ __forceinline__ double refine_row_sum(const RowKernelArgs& a, int64_t b, int len, bool on) {
    int wave_max = len;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(wave_max, d, kWave); wave_max = o > wave_max ? o : wave_max; }
    wave_max = __builtin_amdgcn_readfirstlane(wave_max);
    double s = 0.0;
    for (int k0 = 0; k0 < wave_max; k0 += 8) {
        const int n_here = wave_max - k0 < 8 ? wave_max - k0 : 8; // wave-uniform
        Cols4 q0, q1;
        q0 = *reinterpret_cast<const Cols4*>(a.col_idx + (on ? b + k0 : 0));
        if (n_here > 4) q1 = *reinterpret_cast<const Cols4*>(a.col_idx + (on ? b + k0 + 4 : 0));
        else { q1.c[0] = q1.c[1] = q1.c[2] = q1.c[3] = 0; }
        int32_t c[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) { c[u] = (k0 + u < len) ? q0.c[u] : 0; c[4 + u] = (k0 + 4 + u < len) ? q1.c[u] : 0; }
        // The gathers of a trip go out TOGETHER: four unconditional ones (positions past the row gather entry 0: a cached line),
        // the other four under ONE wave-uniform test. One test per gather put every gather into a basic block of its own with a
        // `s_waitcnt vmcnt(0)` in front (the ISA showed): two or three memory round trips per trip instead of one.
        double v[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = a.logp[c[u]];
        if (n_here > 4) {
#pragma unroll
            for (int u = 4; u < 8; ++u) v[u] = a.logp[c[u]];
        } else {
#pragma unroll
            for (int u = 4; u < 8; ++u) v[u] = 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (u < n_here && k0 + u < len) s = s + v[u]; // plain adds, CSR order
    }
    return s;
}

// FLAGGED (lists of the fixed-point pass k_fx_sum): bit 15 of an entry = "pass A left this row out: account for it
// here if it is kept", bit 14 = "argmax contender: its exact s_i takes part in max / argmax"; the filters' lists carry
// no flags and mean both. Records go to block_partials[part_offset + blockIdx.x].
// PERM (lists of the row stream, kernels_ls.hip): an entry is a POSITION of its slab; posinfo[position] names the row AND, for
// nearly every row, its bounds (start relative to the slab's first entry, length = the item's length minus a 3-bit difference):
// the candidates of an item sit side by side in the list, so they share the 128-byte line of their posinfo words, where the
// row-bounds lookup costs a random line per candidate. The rows of a lane do not ascend, so an argmax tie is settled here
// (lower row) and not by arrival.
template <bool FLAGGED, bool PERM = false>
__global__ __launch_bounds__(kRefineWaves * kWave) void k_fr_refine(RowKernelArgs a, const uint16_t* __restrict__ cand_list,
                                                                   const unsigned* __restrict__ cand_cnt, int slab_rows,
                                                                   int64_t n_slabs, uint64_t* __restrict__ keep_mask,
                                                                   int part_offset, const uint32_t* __restrict__ posinfo = nullptr,
                                                                   const uint8_t* __restrict__ item_len8 = nullptr) {
    __shared__ double s_red[kRefineWaves * kAccDoubles];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    Acc acc;
    acc_init(acc);
    // a "unit" = slab_rows consecutive rows with their own list segment and count (a whole slab for the filters, the
    // share of one folding wave for k_fx2); two waves per unit
    const int64_t GW = (int64_t)gridDim.x * kRefineWaves;
    for (int64_t u = (int64_t)blockIdx.x * kRefineWaves + wave; u < 2 * n_slabs; u += GW) {
        const int64_t s = u >> 1;
        const int n = (int)__builtin_amdgcn_readfirstlane(cand_cnt[s]);
        const int64_t base = s * slab_rows;
        for (int j0 = (int)(u & 1) * kWave; j0 < n; j0 += 2 * kWave) {
            const bool on = j0 + lane < n;
            unsigned entry = on ? (unsigned)cand_list[base + j0 + lane] : 0u;
            uint32_t info = 0u;
            unsigned pos = entry;
            if (PERM) {
                info = on ? posinfo[base + entry] : 0u;
                entry = info & 8191u;
            }
            const bool account = !FLAGGED || (entry & 0x8000u) != 0, contend = !FLAGGED || (entry & 0x4000u) != 0;
            const int64_t r = base + (int64_t)(FLAGGED ? (entry & 0x3FFFu) : entry);
            int len = 0;
            int64_t rb = 0;
            if (on) {
                const unsigned code = (info >> 13) & 7u;
                if (PERM && code != 7u) {
                    len = (int)item_len8[s * 128 + (int64_t)(pos >> 6)] - (int)code;
                    rb = a.row_ptr[base] + (int64_t)(info >> 16); // (one address for the whole unit)
                } else {
                    refine_row_bounds(a, r, &rb, &len);
                }
            }
            const double sv = refine_row_sum(a, rb, len, on);
            const bool keep = on && (sv >= a.log_cut) && (a.limit_order == 0 || len <= a.limit_order);
            fold_kept_row(acc.rare, acc.l1p, acc.n_certain, sv, keep && account);
            if (keep) {
                if (account) acc.n_kept += 1.0;
                // rows arrive in ascending order per lane, ties resolve to the lower row in acc_merge
                if (contend && (sv > acc.max_logp || (PERM && sv == acc.max_logp && (double)(a.row_offset + r) < acc.argmax))) {
                    acc.max_logp = sv;
                    acc.argmax = (double)(a.row_offset + r);
                }
                if (account && keep_mask != nullptr)
                    atomicOr(reinterpret_cast<unsigned long long*>(&keep_mask[r >> 6]), 1ULL << (r & 63));
            }
        }
    }
    acc_wave_reduce(acc);
    if (lane == 0) acc_store(s_red + wave * kAccDoubles, acc);
    __syncthreads();
    if (tid == 0) {
        Acc t;
        acc_load(t, s_red);
        for (int k = 1; k < kRefineWaves; ++k) {
            Acc o;
            acc_load(o, s_red + k * kAccDoubles);
            acc_merge(t, o);
        }
        acc_store(a.block_partials + (size_t)(part_offset + (int)blockIdx.x) * kAccDoubles, t);
    }
}

// ---- the exact all-rows pass without ordering constraints: fixed-point sums ----------------------------------
// cut_off = 0, or a cut-off most rows survive: every row contributes, so there is nothing to filter, and the exact
// CSR-order kernels pay for their ordering (k_ws: one wave per 4096 rows, gather and LDS phases serialised).
// The contract only needs the ORDER for the bit-exact outputs (which rows are kept, which row is the argmax):
// rare-event and MCUB carry a 1e-12 tolerance. So this pass adds in ANY order, exactly:
//   fx[j]  = round(-logp[j] * 2^47)                (u64, clamped to 2^56; built when p changes)
//   F_i    = sum of fx over the row                (integer adds commute: ds_add_u64, all 16 waves share a slab and
//                                                   the column-sorted stream of the filter pass; deterministic)
//   s~_i   = -F_i * 2^-47,  |s~_i - s_i| <= L*2^-48 + L*|s_i|*2^-53      (L = row length; table rounding + the
//                                                   rounding of the canonical fp64 chain + one conversion)
// P_i = exp(s~_i) and log1p(-P_i) go into the sums for every row whose keep decision is beyond doubt at that error
// (F_i + e <= T or F_i - e > T in integers, e = L*(2 + F_i/2^52) covering twice the bound). Rows it cannot decide,
// rows whose |s~| is so large that the bound exceeds 5e-13, and the argmax contenders (F_i - e <= the smallest
// F_j + e seen) go into the slab's candidate list with flag bits; k_fr_refine<true> recomputes exactly those in CSR
// order: the integer outputs are the canonical ones, bit for bit.
constexpr uint64_t kFxClamp = 1ULL << 56;
constexpr double kFxScale = 0x1.0p47;

__global__ __launch_bounds__(256) void k_fx_table(const double* __restrict__ logp, int32_t n_events, uint64_t* __restrict__ fx) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_events) return;
    const double v = -logp[j] * kFxScale; // >= 0, +inf for p = 0
    fx[j] = v >= 0x1.0p56 ? kFxClamp : (uint64_t)__double2ll_rn(v);
}

struct FxAcc { CSum rare, l1p; double n_kept, n_certain; };

struct FxScalars {
    int64_t total_chunks, n_rows, n_slabs;
    long long t_keep;   // F + e <= t_keep  => s_i >= log(cut_off) beyond doubt
    long long t_drop;   // F - e >  t_drop  => s_i <  log(cut_off) beyond doubt
    long long f_smax;   // F > f_smax: the sum's own error bound exceeds the budget (or an entry was clamped) -> exact pass
    int32_t L;          // max_row_len (bounded by limit_order when that is smaller)
    int32_t limit_order, need_len;
    int32_t H;          // column parts of the stream: slab s holds chunks [fr_off[s * H], fr_off[(s + 1) * H])
};

template <int SLAB_ROWS, int WAVES, int U, bool TABLE_LDS>
__global__ __launch_bounds__(WAVES * kWave) void k_fx_sum(const uint4* __restrict__ stream, const int64_t* __restrict__ fr_off,
                                                          const uint64_t* __restrict__ fx, int32_t n_events,
                                                          const int64_t* __restrict__ row_ptr, FxScalars sc,
                                                          double* __restrict__ block_partials, uint16_t* __restrict__ cand_list,
                                                          unsigned* __restrict__ cand_cnt, unsigned long long* __restrict__ cand_count,
                                                          uint64_t* __restrict__ keep_mask) {
    constexpr int kRowBits = SLAB_ROWS > 8192 ? 15 : 14, kThreads = WAVES * kWave, NI = SLAB_ROWS / kThreads;
    constexpr uint32_t kRowMask = (1u << kRowBits) - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* sums = reinterpret_cast<unsigned long long*>(smem);                 // [SLAB_ROWS + 64]
    unsigned* s_cnt = reinterpret_cast<unsigned*>(smem + (size_t)(SLAB_ROWS + 64) * 8);     // [SLAB_ROWS / 64]
    unsigned long long* s_min = reinterpret_cast<unsigned long long*>(smem + (size_t)(SLAB_ROWS + 64) * 8 + (SLAB_ROWS / 64) * 4);
    const unsigned long long* tab = s_min + 2;                                              // [n_events] when TABLE_LDS
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int64_t G = (int64_t)gridDim.x;
    uint4 q[U];
#define FX_FIRST_CHUNKS(SLAB)                                                                       \
    if ((SLAB) < sc.n_slabs) {                                                                      \
        const int64_t f0 = fr_uload(fr_off + (SLAB) * sc.H), f1 = fr_uload(fr_off + ((SLAB) + 1) * sc.H); \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                             \
            const int64_t c = f0 + wave + (int64_t)u * WAVES;                                       \
            q[u] = stream[(c < f1 ? c : sc.total_chunks) * kWave + lane];                           \
        }                                                                                           \
    }
    FX_FIRST_CHUNKS((int64_t)blockIdx.x)
    for (int i = tid; i < SLAB_ROWS + 64; i += kThreads) sums[i] = 0ULL;
    if (tid == 0) s_min[0] = ~0ULL;
    if (TABLE_LDS) {
        unsigned long long* t = const_cast<unsigned long long*>(tab);
        for (int i = tid; i < n_events; i += kThreads) t[i] = fx[i];
    }
    fr_barrier();

    FxAcc acc = {{0.0, 0.0}, {0.0, 0.0}, 0.0, 0.0};
    unsigned long long n_cand = 0;
    for (int64_t s = (int64_t)blockIdx.x; s < sc.n_slabs; s += G) {
        const int64_t c0 = fr_uload(fr_off + s * sc.H), c1 = fr_uload(fr_off + (s + 1) * sc.H);
        // -------- phase 1: any wave, any chunk: table value, unordered exact LDS add
        for (int64_t cb = c0 + wave; cb < c1; cb += (int64_t)U * WAVES) {
            unsigned long long v[U][4];
            if (!TABLE_LDS) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t e[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[u][k] = fx[e[k] >> kRowBits];
                }
            }
            uint4 qn[U]; // next step's chunks: issued behind the gathers, so the adds do not wait for them
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t c = cb + (int64_t)(U + u) * WAVES;
                qn[u] = stream[(c < c1 ? c : sc.total_chunks) * kWave + lane];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t e[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
                if (TABLE_LDS) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[u][k] = tab[e[k] >> kRowBits];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    __hip_atomic_fetch_add(&sums[e[k] & kRowMask], v[u][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) q[u] = qn[u];
        }
        FX_FIRST_CHUNKS(s + G) // the next slab's stream starts under this slab's end step
        fr_barrier();
        // -------- phase 2: classify, fold the decided rows, list the others
        const int64_t r0 = s * SLAB_ROWS;
        const unsigned long long m_run = s_min[0]; // smallest F + e of a decided kept row in this workgroup's earlier slabs
        // (a) the best decided row of this wave: only the keep rule, no state kept (the sums are read again below)
        unsigned long long my_min = ~0ULL;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const unsigned long long Fi = sums[i * kThreads + tid];
            const int64_t r = r0 + i * kThreads + tid;
            bool ok = r < sc.n_rows;
            if (sc.need_len && ok) ok = (int)(row_ptr[r + 1] - row_ptr[r]) <= sc.limit_order;
            const unsigned long long e = (unsigned long long)sc.L * (2ULL + (Fi >> 52));
            ok = ok && (long long)(Fi + e) <= sc.t_keep && (long long)Fi <= sc.f_smax;
            if (ok) my_min = Fi + e < my_min ? Fi + e : my_min;
        }
        unsigned long long w_min = my_min;
#pragma unroll
        for (int d = kWave / 2; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_xor(w_min, d, kWave);
            w_min = o < w_min ? o : w_min;
        }
        // argmax contenders: anything that may still beat the best decided row seen so far (this wave's included)
        const unsigned long long bar = w_min < m_run ? w_min : m_run;
        if (lane == 0 && w_min != ~0ULL)
            __hip_atomic_fetch_min(&s_min[0], w_min, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        // (b) four rows at a time: classification, flag bits (two per row, packed), ballots, the decided rows' exp / log1p
        unsigned flpack = 0u;
        constexpr int FOLD = WAVES > 8 ? 2 : 4; // independent exp chains per lane: four waves per SIMD hide the rest
#pragma unroll 1 // ONE copy of the exp / log1p code (unrolled, the kernel is 9000 instructions and spills)
        for (int i0 = 0; i0 < NI; i0 += FOLD) {
            double sv[FOLD];
            bool on[FOLD];
#pragma unroll
            for (int u = 0; u < FOLD; ++u) {
                const int i = i0 + u;
                const int rl = i * kThreads + tid;
                const unsigned long long Fi = sums[rl];
                sums[rl] = 0ULL;
                const int64_t r = r0 + rl;
                const bool valid = r < sc.n_rows;
                bool len_ok = true;
                if (sc.need_len && valid) len_ok = (int)(row_ptr[r + 1] - row_ptr[r]) <= sc.limit_order;
                const unsigned long long e = (unsigned long long)sc.L * (2ULL + (Fi >> 52));
                const bool big = (long long)Fi > sc.f_smax;
                const bool kc = (long long)(Fi + e) <= sc.t_keep;
                const bool dc = Fi < kFxClamp && (long long)Fi - (long long)e > sc.t_drop;
                const bool kept = valid && len_ok && kc && !big;
                const bool listed = valid && len_ok && !dc && (big || !kc); // bit 15: left to the exact pass
                const bool contend = (kept || listed) && (big || bar == ~0ULL || Fi <= bar + e); // F - e <= bar
                flpack |= ((listed ? 2u : 0u) | (contend ? 1u : 0u)) << (2 * i);
                const unsigned long long mi = __ballot(listed || contend);
                if (lane == 0) s_cnt[i * WAVES + wave] = (unsigned)__builtin_popcountll(mi);
                if (keep_mask != nullptr) {
                    const unsigned long long km = __ballot(kept);
                    const int64_t rw = r0 + i * kThreads + wave * kWave;
                    if (lane == 0 && rw < sc.n_rows) keep_mask[rw >> 6] = km; // the exact pass ORs its rows in afterwards
                }
                on[u] = kept;
                sv[u] = -((double)Fi * 0x1.0p-47);
            }
            fold_kept_rows<FOLD>(acc.rare, acc.l1p, acc.n_certain, sv, on);
#pragma unroll
            for (int u = 0; u < FOLD; ++u) acc.n_kept += on[u] ? 1.0 : 0.0;
        }
        if (tid < 64) sums[SLAB_ROWS + tid] = 0ULL; // the sink
        unsigned long long m[NI];
        unsigned fl[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            fl[i] = ((flpack >> (2 * i)) & 3u) << 14;
            m[i] = __ballot(fl[i] != 0u);
        }
        fr_barrier();
        const unsigned total = fr_emit_candidates<NI, WAVES>(m, fl, s_cnt, lane, wave, tid, cand_list + s * SLAB_ROWS);
        if (tid == 0) cand_cnt[s] = total;
        n_cand += total; // workgroup-uniform
    }
#undef FX_FIRST_CHUNKS
    if (tid == 0 && n_cand) atomicAdd(cand_count, n_cand);

    __syncthreads(); // the sums are dead: the workgroup's record
    double* s_red = reinterpret_cast<double*>(smem);
    Acc full;
    acc_init(full); // no max / argmax from this pass: they come from the exact pass over the contenders
    full.rare = acc.rare; full.l1p = acc.l1p; full.n_kept = acc.n_kept; full.n_certain = acc.n_certain;
    acc_wave_reduce(full);
    if (lane == 0) acc_store(s_red + wave * kAccDoubles, full);
    __syncthreads();
    if (tid == 0) {
        Acc t;
        acc_load(t, s_red);
        for (int w = 1; w < WAVES; ++w) {
            Acc o;
            acc_load(o, s_red + w * kAccDoubles);
            acc_merge(t, o);
        }
        acc_store(block_partials + (size_t)blockIdx.x * kAccDoubles, t);
    }
}


// ---- the same pass with the two phases overlapped: adding waves and folding waves ------------------------------
// k_fx_sum runs its phases one after the other (barriers), so the texture path / LDS atomics of phase 1 and the
// vector ALU of phase 2 (an exp per row) never work at the same time. Here a 16-wave workgroup keeps TWO slabs of
// row sums in LDS: waves [0, ADD) add slab k into one buffer while waves [ADD, 16) fold slab k-1 out of the other
// (classification, exp / log1p, candidate lists, keep mask) and clear it; one workgroup barrier per slab.
// Each folding wave owns a fixed share of the slab's rows and writes its OWN list segment and count (no
// cross-wave scan): the lists are ordered and deterministic, and k_fr_refine walks them as "units".
// exp(s~): table-driven (2^(j/64) in LDS, degree-6 polynomial on |r| <= ln2/128), about a third of ocml's exp.
constexpr int kFx2Slab = 8192, kFx2Waves = 16;

__device__ __forceinline__ double fx_exp_neg(double x, const double* __restrict__ etab) { // x <= 0 (may be -inf)
    const double kf = __builtin_rint(x * 0x1.71547652b82fep+6); // 64 / ln 2
    double r = __builtin_fma(kf, -0x1.62e42fefa0000p-7, x);     // ln2/64, high part (trailing zeros: kf * hi is exact)
    r = __builtin_fma(kf, -0x1.cf79abc9e3b3ap-46, r);           // low part
    double q = __builtin_fma(r, 0x1.6c16c16c16c17p-10, 0x1.1111111111111p-7); // 1/720, 1/120
    q = __builtin_fma(q, r, 0x1.5555555555555p-5);  // 1/24
    q = __builtin_fma(q, r, 0x1.5555555555555p-3);  // 1/6
    q = __builtin_fma(q, r, 0.5);
    const double p = __builtin_fma(q * r, r, r);    // r + r^2 * q
    const int k = (int)kf;                          // |kf| <= 3.1e6 for any finite fixed-point sum
    const double t = etab[k & 63];
    return __builtin_ldexp(__builtin_fma(t, p, t), k >> 6);
}

template <int ADD, int U, bool TABLE_LDS, bool ALLKEPT>
__global__ __launch_bounds__(kFx2Waves * kWave) void k_fx2(const uint4* __restrict__ stream, const int64_t* __restrict__ fr_off,
                                                           const uint64_t* __restrict__ fx, int32_t n_events,
                                                           const int64_t* __restrict__ row_ptr, FxScalars sc,
                                                           double* __restrict__ block_partials, uint16_t* __restrict__ cand_list,
                                                           unsigned* __restrict__ cand_cnt, unsigned long long* __restrict__ cand_count,
                                                           uint64_t* __restrict__ keep_mask) {
    constexpr int kRowBits = 14, FW = kFx2Waves - ADD, kUnit = kFx2Slab / FW, NIF = kUnit / kWave, kThreads = kFx2Waves * kWave;
    constexpr uint32_t kRowMask = (1u << kRowBits) - 1;
    constexpr int kBuf = kFx2Slab + 64; // + the sink slots
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* bufs = reinterpret_cast<unsigned long long*>(smem);      // [2][kBuf]
    unsigned long long* s_min = bufs + 2 * kBuf;                                 // [0] running min, [1] candidate total
    double* etab = reinterpret_cast<double*>(s_min + 2);                         // [64] 2^(j/64)
    const unsigned long long* tab = reinterpret_cast<const unsigned long long*>(etab + 64); // [n_events] when TABLE_LDS
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int64_t G = (int64_t)gridDim.x; // (slabs blockIdx, blockIdx + G, ...: contiguous blocks per workgroup, which pay in k_fr_filter_q8o, cost 1 % here)
    const bool adder = wave < ADD;
    uint4 q[U], q2[U];
#define FX2_FIRST_CHUNKS(SLAB)                                                                      \
    if ((SLAB) < sc.n_slabs) {                                                                      \
        const int64_t f0 = fr_uload(fr_off + (SLAB) * sc.H), f1 = fr_uload(fr_off + ((SLAB) + 1) * sc.H); \
        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                             \
            const int64_t c = f0 + wave + (int64_t)u * ADD;                                         \
            q[u] = stream[(c < f1 ? c : sc.total_chunks) * kWave + lane];                           \
        }                                                                                           \
    }
    if (adder) { FX2_FIRST_CHUNKS((int64_t)blockIdx.x) }
    for (int i = tid; i < 2 * kBuf; i += kThreads) bufs[i] = 0ULL;
    if (tid == 0) { s_min[0] = ~0ULL; s_min[1] = 0ULL; }
    if (tid < 64) etab[tid] = exp2((double)tid * 0x1.0p-6);
    if (TABLE_LDS) {
        unsigned long long* t = const_cast<unsigned long long*>(tab);
        for (int i = tid; i < n_events; i += kThreads) t[i] = fx[i];
    }
    fr_barrier();

    FxAcc acc = {{0.0, 0.0}, {0.0, 0.0}, 0.0, 0.0};
    unsigned long long n_cand = 0; // folding waves: wave-uniform
    int64_t s_prev = -1;
    int it = 0;
    for (int64_t s = (int64_t)blockIdx.x;; s += G, ++it) {
        const bool have = s < sc.n_slabs;
        if (adder) {
            if (have) { // -------- add slab s into buffer it & 1: any adding wave, any chunk
                unsigned long long* sums = bufs + (size_t)(it & 1) * kBuf;
                const int64_t c0 = fr_uload(fr_off + s * sc.H), c1 = fr_uload(fr_off + (s + 1) * sc.H);
                // two register sets in ping-pong (a copy "q = qn" at the end of the body makes the compiler wait for all
                // of the next chunks every iteration)
#define FX2_HALF(QC, QN, CB)                                                                                        \
                {                                                                                                   \
                    unsigned long long v[U][4];                                                                     \
                    if (!TABLE_LDS) {                                                                               \
                        _Pragma("unroll") for (int u = 0; u < U; ++u) {                                             \
                            const uint32_t e[4] = {QC[u].x, QC[u].y, QC[u].z, QC[u].w};                             \
                            _Pragma("unroll") for (int k = 0; k < 4; ++k) v[u][k] = fx[e[k] >> kRowBits];           \
                        }                                                                                           \
                    }                                                                                               \
                    _Pragma("unroll") for (int u = 0; u < U; ++u) { /* next chunks: behind the gathers */           \
                        const int64_t c = (CB) + (int64_t)(U + u) * ADD;                                            \
                        QN[u] = stream[(c < c1 ? c : sc.total_chunks) * kWave + lane];                              \
                    }                                                                                               \
                    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                 \
                        const uint32_t e[4] = {QC[u].x, QC[u].y, QC[u].z, QC[u].w};                                 \
                        if (TABLE_LDS) {                                                                            \
                            _Pragma("unroll") for (int k = 0; k < 4; ++k) v[u][k] = tab[e[k] >> kRowBits];          \
                        }                                                                                           \
                        _Pragma("unroll") for (int k = 0; k < 4; ++k)                                               \
                            __hip_atomic_fetch_add(&sums[e[k] & kRowMask], v[u][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
                    }                                                                                               \
                }
                for (int64_t cb = c0 + wave; cb < c1;) {
                    FX2_HALF(q, q2, cb)
                    cb += (int64_t)U * ADD;
                    if (cb >= c1) break;
                    FX2_HALF(q2, q, cb)
                    cb += (int64_t)U * ADD;
                }
#undef FX2_HALF
                FX2_FIRST_CHUNKS(s + G)
            }
        } else if (s_prev >= 0) { // -------- fold slab s_prev out of the other buffer
            unsigned long long* sums = bufs + (size_t)((it - 1) & 1) * kBuf;
            const int fw = wave - ADD;
            const int64_t r0 = s_prev * kFx2Slab + (int64_t)fw * kUnit; // first row of this wave's unit
            const int l0 = fw * kUnit;
            unsigned long long bar = s_min[0];
            if (bar == ~0ULL) { // nothing decided yet in this workgroup: the wave's own best row sets the bar
                unsigned long long mm = ~0ULL;
                for (int i = 0; i < NIF; ++i) {
                    const unsigned long long Fi = sums[l0 + i * kWave + lane];
                    const int64_t r = r0 + i * kWave + lane;
                    bool ok = r < sc.n_rows && (long long)Fi <= sc.f_smax;
                    if (!ALLKEPT) {
                        if (sc.need_len && ok) ok = (int)(row_ptr[r + 1] - row_ptr[r]) <= sc.limit_order;
                        const unsigned long long e = (unsigned long long)sc.L * (2ULL + (Fi >> 52));
                        ok = ok && (long long)(Fi + e) <= sc.t_keep;
                    }
                    if (ok) mm = Fi < mm ? Fi : mm;
                }
#pragma unroll
                for (int d = kWave / 2; d >= 1; d >>= 1) {
                    const unsigned long long o = __shfl_xor(mm, d, kWave);
                    mm = o < mm ? o : mm;
                }
                bar = mm;
            }
            // argmax contenders: F - e <= bar + e'; one uniform slack covers both error terms (e grows with F by < L per 2^52)
            const unsigned long long thr = bar == ~0ULL ? ~0ULL : bar + 2ULL * (unsigned long long)sc.L * (3ULL + (bar >> 52));
            uint16_t* seg = cand_list + (s_prev * FW + fw) * (int64_t)kUnit;
            unsigned n_list = 0;     // wave-uniform
            unsigned n_kept_w = 0;   // wave-uniform: decided kept rows of this wave in this slab
            unsigned long long my_min = ~0ULL;
#pragma unroll 1 // one copy of the exp / log1p code
            for (int i0 = 0; i0 < NIF; i0 += 4) {
                unsigned long long Fv[4];
                bool on[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { // all four LDS reads first
                    const int rl = l0 + (i0 + u) * kWave + lane;
                    Fv[u] = sums[rl];
                    sums[rl] = 0ULL;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u;
                    const unsigned long long Fi = Fv[u];
                    const int64_t r = r0 + i * kWave + lane;
                    const bool valid = r < sc.n_rows;
                    const bool big = (long long)Fi > sc.f_smax;
                    bool kept, listed;
                    if (ALLKEPT) { // cut_off = 0, no length rule: every valid row is kept; only the accuracy budget lists rows
                        kept = valid && !big;
                        listed = valid && big;
                    } else {
                        bool len_ok = true;
                        if (sc.need_len && valid) len_ok = (int)(row_ptr[r + 1] - row_ptr[r]) <= sc.limit_order;
                        const unsigned long long e = (unsigned long long)sc.L * (2ULL + (Fi >> 52));
                        const bool kc = (long long)(Fi + e) <= sc.t_keep;
                        const bool dc = Fi < kFxClamp && (long long)Fi - (long long)e > sc.t_drop;
                        kept = valid && len_ok && kc && !big;
                        listed = valid && len_ok && !dc && (big || !kc);
                    }
                    const bool contend = (kept || listed) && (big || Fi <= thr);
                    const unsigned long long mi = __ballot(listed || contend);
                    if (mi != 0ULL) { // wave-uniform; rare
                        if (listed || contend)
                            seg[n_list + __builtin_amdgcn_mbcnt_hi((uint32_t)(mi >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mi, 0u))] =
                                (uint16_t)((listed ? 0x8000u : 0u) | (contend ? 0x4000u : 0u) | (unsigned)(i * kWave + lane));
                        n_list += (unsigned)__builtin_popcountll(mi);
                    }
                    const unsigned long long km = __ballot(kept);
                    n_kept_w += (unsigned)__builtin_popcountll(km);
                    if (keep_mask != nullptr) {
                        const int64_t rw = r0 + i * kWave;
                        if (lane == 0 && rw < sc.n_rows) keep_mask[rw >> 6] = km; // the exact pass ORs its rows in afterwards
                    }
                    if (kept) my_min = Fi < my_min ? Fi : my_min;
                    on[u] = kept;
                }
                // P = exp(s~) for all four rows unconditionally (four independent chains the scheduler interleaves; a
                // branch per row serialises them), masked afterwards
                double P[4], l[4];
                bool any_big = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) P[u] = fx_exp_neg(-((double)Fv[u] * 0x1.0p-47), etab);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    P[u] = on[u] ? P[u] : 0.0;
                    any_big = any_big || P[u] >= 0x1.0p-12;
                }
                if (!__any(any_big)) {
                    // every P < 2^-12: log1p(-P) = -P(1 + P/2 + P^2/3 + P^3/4 + P^4/5 + P^5/6) to < 1e-19 relative. The four
                    // terms are added plainly first (positive terms: <= 3 ulp relative), ONE compensated add per batch
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        double t = __builtin_fma(P[u], 1.0 / 6.0, 0.2);
                        t = __builtin_fma(t, P[u], 0.25);
                        t = __builtin_fma(t, P[u], 1.0 / 3.0);
                        t = __builtin_fma(t, P[u], 0.5);
                        t = __builtin_fma(t, P[u], 1.0);
                        l[u] = -P[u] * t;
                    }
                    csum_add(acc.rare, (P[0] + P[1]) + (P[2] + P[3]));
                    csum_add(acc.l1p, (l[0] + l[1]) + (l[2] + l[3]));
                } else { // some lane has a large P (or a certain row, P >= 1): the careful per-row path
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        double lu = log1p_neg_small(P[u]);
                        if (on[u] && P[u] >= 0x1.0p-12) lu = (P[u] < 1.0) ? log1p(-P[u]) : 0.0;
                        fold_terms(acc.rare, acc.l1p, acc.n_certain, P[u], lu, on[u]);
                    }
                }
            }
            if (lane == 0) acc.n_kept += (double)n_kept_w;
            if (fw == 0) sums[kFx2Slab + lane] = 0ULL; // the sink
#pragma unroll
            for (int d = kWave / 2; d >= 1; d >>= 1) {
                const unsigned long long o = __shfl_xor(my_min, d, kWave);
                my_min = o < my_min ? o : my_min;
            }
            if (lane == 0) {
                cand_cnt[s_prev * FW + fw] = n_list;
                if (my_min != ~0ULL) { // F + e of the best decided row
                    const unsigned long long up = my_min + (unsigned long long)sc.L * (2ULL + (my_min >> 52));
                    __hip_atomic_fetch_min(&s_min[0], up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            n_cand += n_list;
        }
        fr_barrier();
        if (!have) break;
        s_prev = s;
    }
#undef FX2_FIRST_CHUNKS
    if (!adder && lane == 0 && n_cand)
        __hip_atomic_fetch_add(&s_min[1], n_cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads(); // the sums are dead: the workgroup's record
    if (tid == 0 && s_min[1]) atomicAdd(cand_count, s_min[1]); // one global atomic per workgroup
    double* s_red = reinterpret_cast<double*>(smem);
    Acc full;
    acc_init(full); // no max / argmax from this pass: they come from the exact pass over the contenders
    full.rare = acc.rare; full.l1p = acc.l1p; full.n_kept = acc.n_kept; full.n_certain = acc.n_certain;
    acc_wave_reduce(full);
    __syncthreads();
    if (lane == 0) acc_store(s_red + wave * kAccDoubles, full);
    __syncthreads();
    if (tid == 0) {
        Acc t;
        acc_load(t, s_red);
        for (int w = 1; w < kFx2Waves; ++w) {
            Acc o;
            acc_load(o, s_red + w * kAccDoubles);
            acc_merge(t, o);
        }
        acc_store(block_partials + (size_t)blockIdx.x * kAccDoubles, t);
    }
}

} // namespace

// The 24-bit form of an existing 32-bit slab stream (one column part, 8192-row slabs). p->stream stays null (nothing
// allocated) when more than 1 % of the slabs are exception slabs: the model then keeps the 32-bit form.
bool fr_p24_applies(int32_t n_events) {
    static const bool off = CANOPY_TUNE_ENV("CANOPY_FR_NOP24") != nullptr; // tuning only
    // (not for the 4-bit table's range: beyond 112 K events a chunk spans more columns than the 10-bit offset holds)
    return !off && fr_use_q8(n_events) && !fr_q8_nibble(n_events) && fr_parts_for(n_events) == 1 && fr_slab_rows_for(n_events) == kQ8SlabRows;
}

hipError_t fr_build_pack24(const uint32_t* d_stream32, const int64_t* d_fr_off, int64_t n_slabs, int64_t total_chunks, int32_t n_events,
                           FrP24Owned* p, hipStream_t s) {
    p->stream = nullptr; p->base = nullptr; p->exc = nullptr; p->n_exception_slabs = 0;
    if (!fr_p24_applies(n_events) || n_slabs <= 0) return hipSuccess;
    const int64_t n = total_chunks + 1; // + the no-op chunk
    uint32_t* out = nullptr;
    uint32_t* base = nullptr;
    int32_t* exc = nullptr;
    int* d_n = nullptr;
    int n_exc = 0;
    hipError_t e = hipMalloc(&out, sizeof(uint32_t) * ((size_t)n * kWave * 3 + 4));
    if (e == hipSuccess) e = hipMalloc(&base, sizeof(uint32_t) * (size_t)n);
    if (e == hipSuccess) e = hipMalloc(&exc, sizeof(int32_t) * (size_t)(n_slabs + 1));
    if (e == hipSuccess) e = hipMalloc(&d_n, sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(d_n, 0, sizeof(int), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_fr_pack24, dim3((unsigned)(n_slabs + 1)), dim3(256), 0, s, reinterpret_cast<const uint4*>(d_stream32), d_fr_off,
                           n_slabs, total_chunks, kQ8RowBits, kQ8SlabRows, out, base, exc, d_n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&n_exc, d_n, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (d_n) (void)hipFree(d_n);
    if (e != hipSuccess || (int64_t)n_exc * 100 > n_slabs + 99) { // (a one-slab model may be its own exception)
        if (out) (void)hipFree(out);
        if (base) (void)hipFree(base);
        if (exc) (void)hipFree(exc);
        return e;
    }
    p->stream = out; p->base = base; p->exc = exc; p->n_exception_slabs = n_exc;
    return hipSuccess;
}

int fr_max_events() { return 1 << 18; }
bool fr_filter_is_quantised(int32_t n_events) { return fr_use_q8(n_events) && fr_slab_rows_for(n_events) == 8192; }
int fr_refine_grid(int n_cu) { return n_cu * 8; } // (16 per CU: the exact pass 49.3 us against 45.3, the final reduce 10.3 against 7.2)

namespace {
__global__ void k_fr_off16(const int64_t* __restrict__ row_ptr, int64_t n_rows, uint16_t* __restrict__ off16, int* __restrict__ too_wide) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n_rows; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t blk = (r >> kOffBlockBits) << kOffBlockBits;
        const int64_t d = row_ptr[r] - row_ptr[blk];
        if (d > 65535) atomicOr(too_wide, 1);
        off16[r] = (uint16_t)d;
    }
}
} // namespace

hipError_t fr_build_off16(const int64_t* d_row_ptr, int64_t n_rows, uint16_t** d_off16_out, hipStream_t s) {
    *d_off16_out = nullptr;
    uint16_t* d = nullptr;
    int* flag = nullptr;
    hipError_t e = hipMalloc(&d, sizeof(uint16_t) * (size_t)(n_rows + 1));
    if (e == hipSuccess) e = hipMalloc(&flag, sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(flag, 0, sizeof(int), s);
    int wide = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_fr_off16, dim3(grid_for(n_rows + 1, 256) > 65535 * 4 ? 65535 * 4 : grid_for(n_rows + 1, 256)), dim3(256), 0, s,
                           d_row_ptr, n_rows, d, flag);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&wide, flag, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (flag) (void)hipFree(flag);
    if (e != hipSuccess || wide) { // (a block with > 65535 entries: the exact pass keeps reading the int64 offsets)
        if (d) (void)hipFree(d);
        return e;
    }
    *d_off16_out = d;
    return hipSuccess;
}
int fr_slab_rows(int32_t n_events) { return fr_slab_rows_for(n_events); }

// Chunk offsets of the (slab, part) pairs: fr_off[s * H + h], H = fr_parts_for(n_events) (1 up to 112 K events).
hipError_t fr_offsets(const int64_t* d_row_ptr, const int32_t* d_col_idx, int64_t n_rows, int32_t n_events,
                      int64_t** d_fr_off_out, int64_t* total_chunks_out, int64_t* n_slabs_out, bool* ok, hipStream_t s) {
    *ok = false;
    if (n_events > fr_max_events() || n_events < 1) return hipSuccess;
    const int slab_rows = fr_slab_rows_for(n_events), H = fr_parts_for(n_events);
    const int64_t n_slabs = (n_rows + slab_rows - 1) / slab_rows, n_off = n_slabs * H;
    *n_slabs_out = n_slabs;
    int64_t* d_off = nullptr;
    hipError_t e = hipMalloc(&d_off, sizeof(int64_t) * (size_t)(n_off + 1));
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_off, 0, sizeof(int64_t), s);
    int64_t total = 0;
    if (e == hipSuccess && n_slabs > 0) {
        if (H == 1)
            hipLaunchKernelGGL(k_fr_count_chunks, dim3(grid_for(n_slabs, 256)), dim3(256), 0, s, d_row_ptr, n_rows, slab_rows, n_slabs,
                               d_off + 1);
        else
            hipLaunchKernelGGL(k_fr_count_chunks_parts, dim3((unsigned)n_slabs), dim3(256), 0, s, d_row_ptr, d_col_idx, n_rows, n_events,
                               slab_rows, H, d_off + 1);
        e = hipGetLastError();
        if (e == hipSuccess) e = scan_inclusive_i64(d_off + 1, n_off, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&total, d_off + n_off, sizeof(int64_t), hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { (void)hipFree(d_off); return e; }
    *d_fr_off_out = d_off;
    *total_chunks_out = total;
    *ok = true;
    return hipSuccess;
}

// The column-bucket-sorted 32-bit stream all slab passes read.
hipError_t fr_build_sorted32(const int64_t* d_row_ptr, const int32_t* d_col_idx, int64_t n_rows, int32_t n_events,
                             const int64_t* d_fr_off, int64_t total_chunks, int64_t n_slabs, uint32_t** d_stream_out,
                             hipStream_t s) {
    const int slab_rows = fr_slab_rows_for(n_events), row_bits = slab_rows > 8192 ? 15 : 14;
    if (n_events > (1 << 18)) return hipErrorInvalidValue; // 14 row bits + 18 column bits; kFrColHist
    uint32_t* d_stream = nullptr;
    uint32_t* d_tmp = nullptr; // the slabs in bucket order, before the in-bucket column sort: build time only
    hipError_t e = hipMalloc(&d_stream, sizeof(uint32_t) * (size_t)((total_chunks + 1) * kFrChunk));
    if (e != hipSuccess) return e;
    e = hipMalloc(&d_tmp, sizeof(uint32_t) * (size_t)((total_chunks + 1) * kFrChunk));
    if (e != hipSuccess) { (void)hipFree(d_stream); return e; }
    hipLaunchKernelGGL(k_fr_fill, dim3(1), dim3(256), 0, s, d_stream + total_chunks * kFrChunk, (int64_t)kFrChunk, (uint32_t)slab_rows); // the no-op chunk
    if (n_slabs > 0) {
        hipLaunchKernelGGL(k_fr_build_sorted, dim3((unsigned)n_slabs), dim3(256), 0, s, d_row_ptr, d_col_idx, n_rows, n_events,
                           slab_rows, row_bits, fr_parts_for(n_events), d_fr_off, d_tmp, d_stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_tmp);
    if (e != hipSuccess) { (void)hipFree(d_stream); return e; }
    *d_stream_out = d_stream;
    return hipSuccess;
}

int fr_parts(int32_t n_events) { return fr_parts_for(n_events); }

namespace {
template <int SLAB_ROWS, int U>
hipError_t launch_fr_filter_t(const uint32_t* stream, const int64_t* fr_off, int64_t total_chunks, const double* logp,
                              int64_t n_rows, int64_t n_slabs, double cand_cut, uint16_t* cand_list, unsigned* cand_cnt,
                              unsigned long long* cand_count, int n_cu, hipStream_t s) {
    using Cf = FrCfg<SLAB_ROWS>;
    // per device and cheap: set on every launch (no process-wide "done" flag: contexts on other devices / threads)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fr_filter<SLAB_ROWS, U>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    int64_t grid = (int64_t)n_cu * (SLAB_ROWS > 8192 ? 1 : 2);
    if (grid > n_slabs) grid = n_slabs;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((k_fr_filter<SLAB_ROWS, U>), dim3((unsigned)grid), dim3(Cf::threads), Cf::lds + (SLAB_ROWS / 64) * 4, s,
                       reinterpret_cast<const uint4*>(stream), fr_off, total_chunks, logp, n_rows, n_slabs, cand_cut,
                       cand_list, cand_cnt, cand_count);
    return hipGetLastError();
}
} // namespace

hipError_t launch_fr_filter(const uint32_t* stream, const int64_t* fr_off, int64_t total_chunks, const double* logp,
                            int64_t n_rows, int32_t n_events, int64_t n_slabs, double cand_cut, uint16_t* cand_list,
                            unsigned* cand_cnt, unsigned long long* cand_count, int n_cu, hipStream_t s) {
    static const int u = CANOPY_TUNE_ENV("CANOPY_FR_U") ? atoi(CANOPY_TUNE_ENV("CANOPY_FR_U")) : 4; // tuning only
    if (fr_slab_rows_for(n_events) == 16384) {
        if (u == 1) return launch_fr_filter_t<16384, 1>(stream, fr_off, total_chunks, logp, n_rows, n_slabs, cand_cut, cand_list, cand_cnt, cand_count, n_cu, s);
        if (u == 4) return launch_fr_filter_t<16384, 4>(stream, fr_off, total_chunks, logp, n_rows, n_slabs, cand_cut, cand_list, cand_cnt, cand_count, n_cu, s);
        if (u == 2) return launch_fr_filter_t<16384, 2>(stream, fr_off, total_chunks, logp, n_rows, n_slabs, cand_cut, cand_list, cand_cnt, cand_count, n_cu, s);
        return launch_fr_filter_t<16384, 4>(stream, fr_off, total_chunks, logp, n_rows, n_slabs, cand_cut, cand_list, cand_cnt, cand_count, n_cu, s);
    }
    if (u == 1) return launch_fr_filter_t<8192, 1>(stream, fr_off, total_chunks, logp, n_rows, n_slabs, cand_cut, cand_list, cand_cnt, cand_count, n_cu, s);
    if (u == 4) return launch_fr_filter_t<8192, 4>(stream, fr_off, total_chunks, logp, n_rows, n_slabs, cand_cut, cand_list, cand_cnt, cand_count, n_cu, s);
    if (u == 2) return launch_fr_filter_t<8192, 2>(stream, fr_off, total_chunks, logp, n_rows, n_slabs, cand_cut, cand_list, cand_cnt, cand_count, n_cu, s);
    return launch_fr_filter_t<8192, 4>(stream, fr_off, total_chunks, logp, n_rows, n_slabs, cand_cut, cand_list, cand_cnt, cand_count, n_cu, s);
}

// Quantised filter: builds the 8-bit table for (logp, cut-off) unless the caller says q8_buf already holds
// it (table_valid), and runs k_fr_filter_q8. q8_buf holds n_events bytes rounded up to 16.
// overlapped: run k_fr_filter_q8o (fr_q8_overlapped says when it applies)
bool fr_q8_overlapped(const FrP24* p24, int32_t n_events, int32_t max_row_len) {
    static const bool off = CANOPY_TUNE_ENV("CANOPY_FR_NOOVERLAP") != nullptr; // tuning only
    return !off && p24 != nullptr && p24->stream != nullptr && max_row_len <= kQ8oMaxRowLen && fr_parts_for(n_events) == 1 &&
           kQ8oSumsBytes + (((size_t)n_events + 15) / 16) * 16 + 2 * kQ8Waves * 4 <= (size_t)160 * 1024;
}
void fr_q8_overlapped_units(int* unit_rows, int* units_per_slab) { *unit_rows = kQ8SlabRows; *units_per_slab = 1; } // (slab lists, as the plain filter's)

// the 8-bit table of (logp, cut-off) by itself: q8_buf holds n_events bytes rounded up to 16
hipError_t launch_fr_q8_table(const double* logp, int32_t n_events, double log_cut, uint8_t* q8_buf, hipStream_t s) {
    const double mag = -log_cut > 1.0 ? -log_cut : 1.0;
    if (fr_q8_nibble(n_events)) {
        hipLaunchKernelGGL(k_fr_q4_table, dim3(grid_for((n_events + 3) / 2, 256)), dim3(256), 0, s, logp, n_events, (double)kQ4Levels / mag, q8_buf);
        return hipGetLastError();
    }
    const double inv_step = (double)kQ8Levels / mag;
    hipLaunchKernelGGL(k_fr_q8_table, dim3(grid_for(n_events + 2, 256)), dim3(256), 0, s, logp, n_events, inv_step, q8_buf);
    return hipGetLastError();
}

hipError_t launch_fr_filter_q8(const uint32_t* stream, const FrP24* p24, bool overlapped, int32_t max_row_len, const int64_t* fr_off,
                               int64_t total_chunks, const double* logp, uint8_t* q8_buf, int64_t n_rows, int32_t n_events, int64_t n_slabs,
                               double log_cut, uint16_t* cand_list, unsigned* cand_cnt, unsigned long long* cand_count,
                               int n_cu, bool table_valid, hipStream_t s, const LsView* ls) {
    const double mag = -log_cut > 1.0 ? -log_cut : 1.0;
    const bool nib = fr_q8_nibble(n_events);
    const double inv_step = (double)(nib ? kQ4Levels : kQ8Levels) / mag;
    // s >= log_cut  ==>  sum(q) <= -log_cut / step; (1 + 2^-28): a hair generous, never too strict
    const double t = -log_cut * inv_step * (1.0 + 0x1.0p-28);
    const int threshold = t < 0.0 ? -1 : (t > 2.0e9 ? 2000000000 : (int)t);
    hipError_t e = hipSuccess;
    if (!table_valid) {
        e = launch_fr_q8_table(logp, n_events, log_cut, q8_buf, s);
        if (e != hipSuccess) return e;
    }
    if (ls != nullptr) // the row stream (kernels_ls.hip): the same table and threshold, lists of positions
        return launch_ls_filter(*ls, q8_buf, n_events, nib, ls->alone, threshold, cand_list, cand_cnt, cand_count, n_cu, s);
    static const int u = CANOPY_TUNE_ENV("CANOPY_FR_QU") ? atoi(CANOPY_TUNE_ENV("CANOPY_FR_QU")) : 4; // tuning only (chunks in flight per wave)
    static const int fr_dbg = CANOPY_TUNE_ENV("CANOPY_FR_DEBUG") ? atoi(CANOPY_TUNE_ENV("CANOPY_FR_DEBUG")) : 0; // timing ablations only
    const int H = fr_parts_for(n_events);
    if (H > 1) { // wide event space: one part of the table resident per sweep
        Q8Parts P;
        P.H = H;
        size_t part_max = 0;
        for (int h = 0; h <= H; ++h) P.c_lo[h] = fr_part_first_col(h, H, n_events);
        for (int h = 0; h < H; ++h) {
            const size_t bytes = (size_t)(P.c_lo[h + 1] - (P.c_lo[h] & ~15) + 15) / 16 * 16;
            part_max = bytes > part_max ? bytes : part_max;
        }
        const size_t lds_p = kQ8SumsBytes + part_max + (kQ8SlabRows / 64) * 4;
        if (lds_p > (size_t)160 * 1024) return hipErrorInvalidValue;
        int64_t gp = n_cu;
        if (gp > n_slabs) gp = n_slabs;
        if (gp < 1) gp = 1;
#define Q8P_GO(UU)                                                                                                     \
        do {                                                                                                           \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fr_filter_q8p<UU>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e != hipSuccess) return e;                                                                             \
            hipLaunchKernelGGL(k_fr_filter_q8p<UU>, dim3((unsigned)gp), dim3(kQ8Threads), lds_p, s, reinterpret_cast<const uint4*>(stream), \
                               fr_off, total_chunks, q8_buf, P, n_rows, n_slabs, threshold, cand_list, cand_cnt, cand_count, part_max); \
            return hipGetLastError();                                                                                  \
        } while (0)
        static const bool no_fold = CANOPY_TUNE_ENV("CANOPY_FR_NOOVERLAP") != nullptr; // tuning only
        const size_t lds_po = kQ8oSumsBytes + part_max + 2 * kQ8Waves * 4;
        const bool folded = !no_fold && max_row_len <= kQ8oMaxRowLen && lds_po <= (size_t)160 * 1024;
        // chunks in flight per wave: a (slab, part) holds total_chunks / (n_slabs * H) chunks on average (97 at E = 2e5); with
        // U = 4 its loop is one full step of 64 chunks and a second, nearly empty one that costs the same latency; U = 6
        // covers it in one step (measured at E = 2e5: 0.6245 -> 0.598 ms; U = 8 0.616)
        static const bool u_forced = CANOPY_TUNE_ENV("CANOPY_FR_QU") != nullptr;
        int up = u;
        if (!u_forced) {
            const double avg = n_slabs > 0 ? (double)total_chunks / ((double)n_slabs * H) : 0.0;
            up = (avg * 1.1 > 4.0 * kQ8Waves && avg * 1.1 <= 6.0 * kQ8Waves) ? 6 : 4;
        }
#define Q8PO_GO(UU)                                                                                                    \
        do {                                                                                                           \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fr_filter_q8po<UU>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e != hipSuccess) return e;                                                                             \
            hipLaunchKernelGGL(k_fr_filter_q8po<UU>, dim3((unsigned)gp), dim3(kQ8Threads), lds_po, s, reinterpret_cast<const uint4*>(stream), \
                               fr_off, total_chunks, q8_buf, P, n_rows, n_slabs, threshold, cand_list, cand_cnt, cand_count, part_max); \
            return hipGetLastError();                                                                                  \
        } while (0)
        if (folded) {
            if (up == 6) Q8PO_GO(6);
            Q8PO_GO(4);
        }
#undef Q8PO_GO
        if (up == 2) Q8P_GO(2);
        if (up == 3) Q8P_GO(3);
        if (up == 6) Q8P_GO(6);
        Q8P_GO(4);
#undef Q8P_GO
    }
    const size_t lds = kQ8SumsBytes + (((size_t)(nib ? (n_events + 1) / 2 : n_events) + 15) / 16) * 16 + (kQ8SlabRows / 64) * 4;
    int64_t grid = n_cu;
    if (grid > n_slabs) grid = n_slabs;
    if (grid < 1) grid = 1;
#define Q8_GO(UU, PP)                                                                                                   \
    do {                                                                                                                \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fr_filter_q8<UU, PP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); /* per device: every launch */ \
        if (e != hipSuccess) return e;                                                                                  \
        hipLaunchKernelGGL((k_fr_filter_q8<UU, PP>), dim3((unsigned)grid), dim3(kQ8Threads), lds, s, PP ? p24->stream : stream, \
                           PP ? p24->base : nullptr, PP ? p24->exc : nullptr, fr_off, total_chunks, q8_buf, n_events, n_rows, \
                           n_slabs, threshold, cand_list, cand_cnt, cand_count, fr_dbg);                                \
        return hipGetLastError();                                                                                       \
    } while (0)
    if (nib) { // 112 K < E <= 260 096: the 4-bit table, the 32-bit stream (a chunk spans more columns than the 24-bit form's offset holds)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fr_filter_q8<4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_fr_filter_q8<4, false, true>), dim3((unsigned)grid), dim3(kQ8Threads), lds, s, stream, (const uint32_t*)nullptr,
                           (const int32_t*)nullptr, fr_off, total_chunks, q8_buf, n_events, n_rows, n_slabs, threshold, cand_list, cand_cnt,
                           cand_count, fr_dbg);
        return hipGetLastError();
    }
    if (overlapped) { // adding and finishing waves side by side (list units of 4096 rows)
        if (p24 == nullptr || p24->stream == nullptr) return hipErrorInvalidValue;
        const size_t lds_o = kQ8oSumsBytes + (((size_t)n_events + 15) / 16) * 16 + 2 * kQ8Waves * 4;
        if (lds_o > (size_t)160 * 1024) return hipErrorInvalidValue;
#define Q8O_GO(UU)                                                                                                      \
        do {                                                                                                            \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fr_filter_q8o<UU>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e != hipSuccess) return e;                                                                              \
            hipLaunchKernelGGL((k_fr_filter_q8o<UU>), dim3((unsigned)grid), dim3(kQ8Threads), lds_o, s, p24->stream, p24->base, p24->exc, \
                               fr_off, total_chunks, q8_buf, n_events, n_rows, n_slabs, threshold, cand_list, cand_cnt, cand_count); \
            return hipGetLastError();                                                                                   \
        } while (0)
        if (u == 2) Q8O_GO(2);
        if (u == 3) Q8O_GO(3);
        if (u == 6) Q8O_GO(6);
        Q8O_GO(4);
#undef Q8O_GO
    }
    if (p24 != nullptr && p24->stream != nullptr) { // the 24-bit form of the same chunks
        if (u == 2) Q8_GO(2, true);
        if (u == 3) Q8_GO(3, true);
        if (u == 6) Q8_GO(6, true);
        Q8_GO(4, true);
    }
    if (u == 1) Q8_GO(1, false);
    if (u == 2) Q8_GO(2, false);
    if (u == 3) Q8_GO(3, false);
    if (u == 6) Q8_GO(6, false);
    Q8_GO(4, false);
#undef Q8_GO
}

hipError_t launch_fr_refine(const RowKernelArgs& a, const uint16_t* cand_list, const unsigned* cand_cnt, int slab_rows,
                            int64_t n_slabs, uint64_t* keep_mask, int grid, bool flagged, int part_offset, hipStream_t s,
                            const uint32_t* posinfo, const uint8_t* item_len8) {
    if (posinfo != nullptr && !flagged) {
        hipLaunchKernelGGL((k_fr_refine<false, true>), dim3(grid), dim3(kRefineWaves * kWave), 0, s, a, cand_list, cand_cnt, slab_rows,
                           n_slabs, keep_mask, part_offset, posinfo, item_len8);
        return hipGetLastError();
    }
    if (flagged)
        hipLaunchKernelGGL((k_fr_refine<true, false>), dim3(grid), dim3(kRefineWaves * kWave), 0, s, a, cand_list, cand_cnt, slab_rows,
                           n_slabs, keep_mask, part_offset, (const uint32_t*)nullptr, (const uint8_t*)nullptr);
    else
        hipLaunchKernelGGL((k_fr_refine<false, false>), dim3(grid), dim3(kRefineWaves * kWave), 0, s, a, cand_list, cand_cnt, slab_rows,
                           n_slabs, keep_mask, part_offset, (const uint32_t*)nullptr, (const uint8_t*)nullptr);
    return hipGetLastError();
}

// ---- fixed-point sum pass -------------------------------------------------------------------------------------
int fx_max_row_len() { return 63; } // 63 entries of at most 2^56 stay below 2^62
size_t fx_table_lds_max_events(int slab_rows) { // the u64 table beside the row sums of one workgroup
    const size_t per_wg = (size_t)160 * 1024;
    const size_t fixed = (size_t)(slab_rows + 64) * 8 + (size_t)(slab_rows / 64) * 4 + 16 + 64;
    return per_wg > fixed ? (per_wg - fixed) / 8 : 0;
}

hipError_t launch_fx_table(const double* logp, int32_t n_events, uint64_t* fx, hipStream_t s) {
    if (n_events <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fx_table, dim3(grid_for(n_events, 256)), dim3(256), 0, s, logp, n_events, fx);
    return hipGetLastError();
}

namespace {
size_t fx_lds_bytes(int32_t n_events, int slab_rows, bool* tab_lds) {
    static const bool no_lds = CANOPY_TUNE_ENV("CANOPY_FX_NOLDS") != nullptr; // tuning only
    *tab_lds = !no_lds && (size_t)n_events <= fx_table_lds_max_events(slab_rows);
    return (size_t)(slab_rows + 64) * 8 + (size_t)(slab_rows / 64) * 4 + 16 + (*tab_lds ? (size_t)n_events * 8 : 0);
}
} // namespace

namespace {
bool fx_use_overlapped(int32_t n_events) { // k_fx2 needs two 8192-row buffers
    static const bool old_only = CANOPY_TUNE_ENV("CANOPY_FX_OLD") != nullptr; // tuning only
    return !old_only && fr_slab_rows_for(n_events) == 8192;
}
constexpr size_t kFx2FixedLds = (size_t)2 * (kFx2Slab + 64) * 8 + 16 + 64 * 8;
} // namespace

int fx_sum_grid(int n_cu, int32_t n_events, int64_t n_slabs) {
    int per_cu = 1;
    if (!fx_use_overlapped(n_events)) {
        bool tab_lds = false;
        const size_t lds = fx_lds_bytes(n_events, fr_slab_rows_for(n_events), &tab_lds);
        per_cu = lds <= (size_t)80 * 1024 ? 2 : 1; // two workgroups per CU while their LDS fits
    }
    int64_t g = (int64_t)n_cu * per_cu;
    if (g > n_slabs) g = n_slabs;
    return g < 1 ? 1 : (int)g;
}

// rows per candidate-list unit and units per slab of the pass launch_fx_sum will run for this model
void fx_list_units(int32_t n_events, int* unit_rows, int* units_per_slab) {
    if (fx_use_overlapped(n_events)) {
        static const int add_env = CANOPY_TUNE_ENV("CANOPY_FX_ADD") ? atoi(CANOPY_TUNE_ENV("CANOPY_FX_ADD")) : 0; // tuning only
        const int fw = add_env == 14 ? 2 : (add_env == 8 ? 8 : 4);
        *unit_rows = kFx2Slab / fw;
        *units_per_slab = fw;
    } else {
        *unit_rows = fr_slab_rows_for(n_events);
        *units_per_slab = 1;
    }
}

namespace {
template <int SLAB_ROWS, int WAVES, int U, bool TABLE_LDS>
hipError_t launch_fx_t(const uint32_t* stream, const int64_t* fr_off, const uint64_t* fx, int32_t n_events, const int64_t* row_ptr,
                       const FxScalars& sc, double* block_partials, uint16_t* cand_list, unsigned* cand_cnt,
                       unsigned long long* cand_count, uint64_t* keep_mask, int grid, hipStream_t s) {
    const void* fn = reinterpret_cast<const void*>(&k_fx_sum<SLAB_ROWS, WAVES, U, TABLE_LDS>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); // per device: every launch
    if (e != hipSuccess) return e;
    bool tl = false;
    const size_t lds = fx_lds_bytes(n_events, SLAB_ROWS, &tl);
    hipLaunchKernelGGL((k_fx_sum<SLAB_ROWS, WAVES, U, TABLE_LDS>), dim3((unsigned)grid), dim3(WAVES * kWave), lds, s,
                       reinterpret_cast<const uint4*>(stream), fr_off, fx, n_events, row_ptr, sc, block_partials, cand_list,
                       cand_cnt, cand_count, keep_mask);
    return hipGetLastError();
}
} // namespace

// One pass A launch. log_cut = -inf for cut_off = 0. The caller runs launch_fr_refine(flagged = true) behind it.
hipError_t launch_fx_sum(const uint32_t* stream, const int64_t* fr_off, int64_t total_chunks, const uint64_t* fx,
                         const int64_t* row_ptr, int64_t n_rows, int32_t n_events, int64_t n_slabs, double log_cut,
                         int32_t limit_order, int32_t max_row_len, double* block_partials, uint16_t* cand_list,
                         unsigned* cand_cnt, unsigned long long* cand_count, uint64_t* keep_mask, int grid, hipStream_t s) {
    FxScalars sc;
    sc.total_chunks = total_chunks; sc.n_rows = n_rows; sc.n_slabs = n_slabs;
    sc.H = fr_parts_for(n_events);
    sc.limit_order = limit_order;
    sc.need_len = (limit_order != 0 && limit_order < max_row_len) ? 1 : 0;
    int L = max_row_len;
    if (limit_order != 0 && limit_order < L) L = limit_order; // longer rows are dropped by the exact length rule
    if (L < 1) L = 1;
    sc.L = L;
    // |s~ - s| <= L*2^-48 + L*|s|*2^-53 must stay below 5e-13 for a row to be summed from s~
    const double smax = (5e-13 - (double)L * 0x1.0p-48) / ((double)L * 0x1.0p-53);
    const double fs = smax * kFxScale;
    sc.f_smax = fs >= (double)(kFxClamp - 1) ? (long long)(kFxClamp - 1) : (fs < 0.0 ? -1 : (long long)fs);
    const long long kAll = (long long)1 << 62; // cut_off = 0: every F (< 2^62) is kept beyond doubt
    if (log_cut == -INFINITY) { sc.t_keep = kAll; sc.t_drop = kAll; }
    else {
        const double t = -log_cut * kFxScale; // the cut-off in fixed point; may be negative (cut_off > 1)
        const double lo = floor(t) - 2.0, hi = ceil(t) + 2.0;
        sc.t_keep = lo < -0x1.0p62 ? -kAll : (lo > 0x1.0p62 ? kAll : (long long)lo);
        sc.t_drop = hi < -0x1.0p62 ? -kAll : (hi > 0x1.0p62 ? kAll : (long long)hi);
    }
    static const int u_env = CANOPY_TUNE_ENV("CANOPY_FX_U") ? atoi(CANOPY_TUNE_ENV("CANOPY_FX_U")) : 0;  // tuning only
    static const int w_env = CANOPY_TUNE_ENV("CANOPY_FX_W") ? atoi(CANOPY_TUNE_ENV("CANOPY_FX_W")) : 0;  // tuning only (waves per workgroup)
    const int slab_rows = fr_slab_rows_for(n_events);
    if (fx_use_overlapped(n_events)) { // adding and folding waves side by side
        static const int add_env = CANOPY_TUNE_ENV("CANOPY_FX_ADD") ? atoi(CANOPY_TUNE_ENV("CANOPY_FX_ADD")) : 0; // tuning only
        const bool tl = kFx2FixedLds + (size_t)n_events * 8 + 64 <= (size_t)160 * 1024 && !CANOPY_TUNE_ENV("CANOPY_FX_NOLDS");
        const bool allkept = log_cut == -INFINITY && !sc.need_len;
        const size_t lds = kFx2FixedLds + (tl ? (size_t)n_events * 8 : 0);
#define FX2_GO(AD, UU, TL, AK)                                                                                          \
        do {                                                                                                            \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fx2<AD, UU, TL, AK>),                   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                 \
            if (e_ != hipSuccess) return e_;                                                                            \
            hipLaunchKernelGGL((k_fx2<AD, UU, TL, AK>), dim3((unsigned)grid), dim3(kFx2Waves * kWave), lds, s,          \
                               reinterpret_cast<const uint4*>(stream), fr_off, fx, n_events, row_ptr, sc, block_partials, \
                               cand_list, cand_cnt, cand_count, keep_mask);                                             \
            return hipGetLastError();                                                                                   \
        } while (0)
        if (tl) { if (allkept) FX2_GO(12, 2, true, true); FX2_GO(12, 2, true, false); }
        if (add_env == 14) {
            if (u_env == 4) { if (allkept) FX2_GO(14, 4, false, true); FX2_GO(14, 4, false, false); }
            if (allkept) FX2_GO(14, 2, false, true); FX2_GO(14, 2, false, false);
        }
        if (add_env == 8) {
            if (u_env == 4) { if (allkept) FX2_GO(8, 4, false, true); FX2_GO(8, 4, false, false); }
            if (allkept) FX2_GO(8, 2, false, true); FX2_GO(8, 2, false, false);
        }
        if (u_env == 2) { if (allkept) FX2_GO(12, 2, false, true); FX2_GO(12, 2, false, false); }
        if (allkept) FX2_GO(12, 4, false, true); // measured at config 3, cut_off 0: 0.687 ms (U = 2: 0.81, 8 adders: 0.71, 14 adders: 0.94)
        FX2_GO(12, 4, false, false);
#undef FX2_GO
    }
    bool tab_lds = false;
    (void)fx_lds_bytes(n_events, slab_rows, &tab_lds);
#define FX_GO(SR, W, UU, TL) return launch_fx_t<SR, W, UU, TL>(stream, fr_off, fx, n_events, row_ptr, sc, block_partials, cand_list, cand_cnt, cand_count, keep_mask, grid, s)
    if (slab_rows > 8192) {
        if (tab_lds) FX_GO(16384, 16, 2, true);
        if (u_env == 4) FX_GO(16384, 16, 4, false);
        FX_GO(16384, 16, 2, false);
    }
    if (tab_lds) { if (w_env == 8) FX_GO(8192, 8, 2, true); FX_GO(8192, 16, 2, true); }
    if (w_env == 8) { if (u_env == 4) FX_GO(8192, 8, 4, false); FX_GO(8192, 8, 2, false); }
    if (u_env == 4) FX_GO(8192, 16, 4, false);
    if (u_env == 1) FX_GO(8192, 16, 1, false);
    FX_GO(8192, 16, 2, false);
#undef FX_GO
}

} // namespace canopy
