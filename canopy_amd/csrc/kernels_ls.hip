// kernels_ls.hip -- the ROW STREAM of the quantised filter (pass A of FILTER_REFINE, E <= 260 096), gfx950 / wave64 only.
//
// The quantised filter (kernels_fr.hip, k_fr_filter_q8o) reads a column-sorted slab stream of 3-byte entries (row 13 bits |
// column offset 11 bits) and scatters one LDS atomic per entry into the slab's row sums. Its sums are small integers: the
// order of a row's entries is free, and so is the order of the ROWS as long as the list can name them. This layout turns
// the slab around (benchmarks/proto_ls.hip is the stand-alone prototype it was measured with):
//   * the 8192 rows of a slab are sorted by length and cut into 128 ITEMS of 64 rows (one row per lane); an item of
//     length L is L "entry rows": the lane's k-th column, k < L (rows shorter than their item are padded with a column
//     whose table value is 0, the rows past the model's end carry a column whose value kills them) -- no row ids in the
//     stream at all;
//   * a lane's columns are a bit stream of BITS bits per entry -- 12 / 14 / 16 / 17 / 18 by the number of events (17 at config 3;
//     beyond 112 K events the table is the 4-bit one) --; 32 entry rows = BITS dwords per lane = one UNIT of BITS 256-byte lines:
//     2.125 bytes per entry instead of 3.03;
//   * the sorted items of a slab go to the 16 waves in a serpentine (every wave gets the same share of every length class:
//     ls_item_of); the 8 items a wave has in each of the consecutive slabs of a STEP (four; two or one in a model so small that
//     steps of four would leave most CUs idle) are contiguous: a wave streams one RANGE per step, in units, one workgroup
//     barrier per step;
//   * per entry: one bit-field extract -> one ds_read_u8 of the 8-bit table in LDS -> a register add. No LDS atomics, no
//     row sums in LDS. A bit mask per unit says where items end: there one ballot goes to LDS. After the step's barrier
//     every wave turns the step's 512 ballots into list offsets (8 per lane, a 16-lane DPP scan = one list per slab) and
//     writes the candidates of its own items;
//   * the lists hold POSITIONS p = j * 1024 + wave * 64 + lane (item j of wave `wave`, row `lane` of the item),
//     ascending; the exact pass translates them through posinfo[slab * 8192 + p]: the local row id, and where the row starts
//     and how long it is (so that it need not look its bounds up).
// The candidate SET is the one k_fr_filter_q8o lists (same table, same threshold, integer sums); only its order differs.
#include "canopy_device.h"
#include "canopy_internal.h"

namespace canopy {

namespace {

#define AS1 __attribute__((address_space(1)))
#ifndef LS_ABL
#define LS_ABL 0 // timing ablations (WRONG results, benchmarks/build_ls_variant.sh only): 1 = table reads at conflict-free addresses
                 // (other table entries: the filter then misses survivors), 2 = a stream whose columns are conflict-free by construction,
                 // 3 = no table read (the low byte of the column instead), 4 = no list written
#endif
#ifndef LS_SS
#define LS_SS 4 // slabs per step (one workgroup barrier per step): 4 or 8
#endif
#ifndef LS_DEPTH
#define LS_DEPTH 2 // register sets of the unit pipeline: 2 = one unit requested ahead of the adds, 3 = two (measured level: 0.2346
                   // against 0.2333 ms per step at config 3 -- the pass is not waiting for its loads)
#endif
typedef uint32_t ls_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kLsSlabRows = 8192, kLsWaves = 16, kLsIPW = 8, kLsStepSlabs = LS_SS, kLsUnitRows = 32, kLsThreads = kLsWaves * kWave;
constexpr int kLsItems = kLsIPW * kLsStepSlabs;          // items of a wave per step
constexpr int kLsSlots = kLsItems * kLsWaves;            // ballots of a step
static_assert(kLsStepSlabs == 4 || kLsStepSlabs == 8, "the list offsets are scanned four slabs (64 lanes x 8 slots) at a time");
constexpr int kLsMaxEvents8 = 112 * 1024;                // the 8-bit table's limit, as the chunk filter's (kernels_fr.hip kQ8MaxEvents)
constexpr int kLsMaxEvents4 = 260096;                    // the 4-bit table's (kQ4MaxEvents): two events per byte, levels 0 .. 14, 15 kills
constexpr int kLsKeyMax = 255;                           // rows of >= 255 entries sort as one class

struct LsDesc { // one per (step, wave): 16 bytes
    uint32_t line0;  // first 256-byte line of the range
    uint32_t n_rows; // entry rows of the range
    uint32_t bm0;    // index of its first unit in the boundary-mask array
    uint32_t pad;
};

// The sorted item that is item j of wave w: a serpentine over the 16 waves, so that every wave gets the same share of every
// length class (with w, w + 16, ... wave 15's eight items are each one class longer than wave 0's: 52 against 44 entry rows
// per slab at config 3, i.e. seven units against six per step and a barrier that waits for the long ones).
__host__ __device__ __forceinline__ int ls_item_of(int j, int w) { return j * kLsWaves + ((j & 1) ? kLsWaves - 1 - w : w); }

__device__ __forceinline__ int ls_eff_len(const int64_t* __restrict__ row_ptr, int64_t r, int64_t n_rows) {
    if (r >= n_rows) return 1; // a padding row: one killing entry
    const int64_t l = row_ptr[r + 1] - row_ptr[r];
    return l < 1 ? 1 : (l > 0x7fffffff ? 0x7fffffff : (int)l); // an empty row: one entry of value 0
}

// ---- builder 1: entry rows of every (slab, wave) from the slab's length histogram ----------------------------------------
// Item i holds the sorted rows 64 i .. 64 i + 63; its length is the length of the last of them (the slab's longest row for
// the class of 255 and more).
__global__ __launch_bounds__(256) void k_ls_sizes(const int64_t* __restrict__ row_ptr, int64_t n_rows, int64_t n_slabs, uint32_t* __restrict__ sizes) {
    __shared__ unsigned hist[kLsKeyMax + 1];
    __shared__ unsigned item_len[128];
    __shared__ int s_max;
    const int64_t slab = blockIdx.x;
    if (slab >= n_slabs) return;
    for (int i = threadIdx.x; i <= kLsKeyMax; i += 256) hist[i] = 0u;
    if (threadIdx.x == 0) s_max = 1;
    __syncthreads();
    int mx = 1;
    for (int i = threadIdx.x; i < kLsSlabRows; i += 256) {
        const int l = ls_eff_len(row_ptr, slab * kLsSlabRows + i, n_rows);
        mx = l > mx ? l : mx;
        atomicAdd(&hist[l > kLsKeyMax ? kLsKeyMax : l], 1u);
    }
    atomicMax(&s_max, mx);
    __syncthreads();
    if (threadIdx.x == 0) { // 256 bins, 128 items: a serial walk
        unsigned cum = 0;
        int item = 0;
        for (int b = 0; b <= kLsKeyMax && item < 128; ++b) {
            cum += hist[b];
            while (item < 128 && (unsigned)(64 * item + 64) <= cum) item_len[item++] = (unsigned)(b == kLsKeyMax ? s_max : b);
        }
    }
    __syncthreads();
    if (threadIdx.x < kLsWaves) {
        unsigned n = 0;
        for (int j = 0; j < kLsIPW; ++j) n += item_len[ls_item_of(j, (int)threadIdx.x)];
        sizes[slab * kLsWaves + threadIdx.x] = n;
    }
}

// ---- builder 2: lines and units of every range (step, wave) ------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ls_range_sizes(const uint32_t* __restrict__ sizes, int64_t n_slabs, int64_t n_ranges, int bits, int ss,
                                                        int64_t* __restrict__ lines, int64_t* __restrict__ units, uint32_t* __restrict__ rows) {
    const int64_t rg = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (rg >= n_ranges) return;
    const int64_t step = rg / kLsWaves;
    const int w = (int)(rg % kLsWaves);
    uint64_t n = 0;
    for (int ls = 0; ls < ss; ++ls) {
        const int64_t slab = step * ss + ls;
        if (slab < n_slabs) n += sizes[slab * kLsWaves + w];
    }
    rows[rg] = (uint32_t)(n > 0xffffffffULL ? 0xffffffffULL : n);
    lines[rg] = (int64_t)((n * (uint64_t)bits + 31) / 32);
    units[rg] = (int64_t)((n + kLsUnitRows - 1) / kLsUnitRows);
}
__global__ __launch_bounds__(256) void k_ls_desc(const int64_t* __restrict__ lines_incl, const int64_t* __restrict__ units_incl,
                                                 const uint32_t* __restrict__ rows, int64_t n_ranges, LsDesc* __restrict__ desc) {
    const int64_t rg = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (rg > n_ranges) return;
    LsDesc d;
    if (rg == n_ranges) { // the empty range behind the last: its "unit" is the zero lines behind the stream
        d.line0 = (uint32_t)lines_incl[n_ranges - 1]; d.n_rows = 0; d.bm0 = (uint32_t)units_incl[n_ranges - 1]; d.pad = 0;
    } else {
        d.line0 = (uint32_t)(rg > 0 ? lines_incl[rg - 1] : 0); d.n_rows = rows[rg]; d.bm0 = (uint32_t)(rg > 0 ? units_incl[rg - 1] : 0); d.pad = 0;
    }
    desc[rg] = d;
}

// ---- builder 3: one workgroup per step: sort every slab's rows by (length, row), write the position words (row id, start, length), the bit streams
// ---- of the 16 ranges and their boundary masks ---------------------------------------------------------------------------
template <int BITS>
__global__ __launch_bounds__(kLsThreads) void k_ls_fill(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx, int64_t n_rows,
                                                         int32_t n_events, int64_t n_slabs, int ss, const LsDesc* __restrict__ desc,
                                                         uint32_t* __restrict__ ent, uint32_t* __restrict__ bm, uint32_t* __restrict__ posinfo,
                                                         uint8_t* __restrict__ item_len8) {
    __shared__ uint32_t key[kLsSlabRows]; // clamped length << 13 | local row
    __shared__ int s_max;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int64_t step = blockIdx.x;
    const LsDesc d = desc[step * kLsWaves + wave];
    const uint32_t pad_col = (uint32_t)n_events, kill_col = (uint32_t)n_events + 1u;
    uint64_t acc = 0; // the lane's bit stream
    int nb = 0;
    uint32_t dw = 0;
    uint32_t t_row = 0, bm_word = 0; // entry rows written so far; the boundary mask of the unit being written (lane 0)
    for (int ls = 0; ls < ss; ++ls) {
        const int64_t slab = step * ss + ls;
        if (slab >= n_slabs) break; // (workgroup-uniform)
        if (tid == 0) s_max = 1;
        __syncthreads();
        int mx = 1;
        for (int i = tid; i < kLsSlabRows; i += kLsThreads) {
            const int l = ls_eff_len(row_ptr, slab * kLsSlabRows + i, n_rows);
            mx = l > mx ? l : mx;
            key[i] = ((uint32_t)(l > kLsKeyMax ? kLsKeyMax : l) << 13) | (uint32_t)i;
        }
        atomicMax(&s_max, mx);
        __syncthreads();
        for (int k = 2; k <= kLsSlabRows; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < kLsSlabRows / 2; t += kLsThreads) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int p = i | j;
                    const bool up = (i & k) == 0;
                    const uint32_t a = key[i], c = key[p];
                    if ((a > c) == up) { key[i] = c; key[p] = a; }
                }
                __syncthreads();
            }
        const int slab_max = s_max;
        // what the exact pass needs of a listed row, beside its id: where it starts (relative to the slab's first entry: 16 bits, as
        // off16) and how long it is (its item's length minus a small difference)
        const int64_t blk_b = row_ptr[slab * kLsSlabRows];
        const int64_t blk_e = row_ptr[(slab + 1) * kLsSlabRows < n_rows ? (slab + 1) * kLsSlabRows : n_rows];
        const bool narrow = blk_e - blk_b <= 65535;
        for (int j = 0; j < kLsIPW; ++j) {
            const int item = ls_item_of(j, wave);
            const uint32_t kk = key[item * kWave + lane];
            const uint32_t last = key[item * kWave + kWave - 1] >> 13;
            const int L = last == (uint32_t)kLsKeyMax ? slab_max : (int)last; // (wave-uniform)
            const uint32_t local = kk & 8191u;
            const int64_t r = slab * kLsSlabRows + (int64_t)local;
            const bool real = r < n_rows;
            const int64_t b = real ? row_ptr[r] : 0;
            const int len = real ? (int)(row_ptr[r + 1] - b) : 0;
            {
                uint32_t code = 7u, start = 0u; // 7: the exact pass looks the bounds up (off16 / row_ptr)
                if (real && narrow && L <= 255 && L - len <= 6) { code = (uint32_t)(L - len); start = (uint32_t)(b - blk_b); }
                posinfo[slab * kLsSlabRows + j * 1024 + tid] = local | (code << 13) | (start << 16);
                if (lane == 0) item_len8[slab * 128 + j * kLsWaves + wave] = (uint8_t)(L > 255 ? 255 : L);
            }
            for (int k = 0; k < L; ++k) {
                uint32_t col = real ? (k < len ? (uint32_t)col_idx[b + k] : pad_col) : kill_col;
                if (LS_ABL == 2 && real && k < len) { // timing ablation (WRONG results): every lane of a half on a bank of its own
                    const uint64_t h = mix64(((uint64_t)r << 8) | (uint64_t)k);
                    col = (uint32_t)(h % (uint64_t)(n_events / 128)) * 128u + (uint32_t)(lane & 31) * 4u + (uint32_t)((h >> 40) & 3u);
                }
                acc |= (uint64_t)col << nb;
                nb += BITS;
                if (nb >= 32) { ent[((size_t)d.line0 + dw) * kWave + lane] = (uint32_t)acc; acc >>= 32; nb -= 32; ++dw; }
            }
            if (lane == 0) { // the item ends with entry row t_row + L - 1
                const uint32_t e = t_row + (uint32_t)L - 1u;
                // (units between the previous boundary and this one carry no boundary: written as they are passed)
                while ((t_row / kLsUnitRows) < (e / kLsUnitRows)) { bm[d.bm0 + t_row / kLsUnitRows] = bm_word; bm_word = 0; t_row = (t_row / kLsUnitRows + 1) * kLsUnitRows; }
                bm_word |= 1u << (e % kLsUnitRows);
                t_row = e + 1;
                if (t_row % kLsUnitRows == 0) { bm[d.bm0 + e / kLsUnitRows] = bm_word; bm_word = 0; }
            }
        }
        __syncthreads(); // (key is rewritten for the next slab)
    }
    if (nb > 0) ent[((size_t)d.line0 + dw) * kWave + lane] = (uint32_t)acc;
    if (lane == 0 && t_row % kLsUnitRows != 0) bm[d.bm0 + t_row / kLsUnitRows] = bm_word;
}

// ---- the filter --------------------------------------------------------------------------------------------------------------
template <int BITS>
struct LsUnit {
    uint32_t d[BITS];
    uint32_t bm;
};

__device__ __forceinline__ void ls_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T>
__device__ __forceinline__ const T AS1* ls_uniform_ptr(const T* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const T AS1*)(((uint64_t)hi << 32) | lo);
}

// base + a wave-uniform byte offset made of values that already sit in scalar registers: the arithmetic stays on the scalar
// ALU. (Through ls_uniform_ptr's v_readfirstlane of the finished pointer the whole chain -- line index, select, shift, add --
// ran on the vector ALU, in temporaries that were the destinations of unit loads still in flight: a `s_waitcnt vmcnt(0..2)`
// in front of every request.)
template <typename T>
__device__ __forceinline__ const T AS1* ls_sgpr_ptr(const T* base, uint64_t byte_off) {
    return (const T AS1*)((uint64_t)base + byte_off);
}

// a unit: BITS lines of 256 bytes + its boundary mask (a vector load of ONE address: a scalar load would share lgkmcnt with
// the LDS reads and return out of order, kernels_fr.hip q8_load_chunk)
template <int BITS, bool NT = false>
__device__ __forceinline__ void ls_load_unit(LsUnit<BITS>& U, const uint32_t* __restrict__ ent, const uint32_t* __restrict__ bm, uint32_t line,
                                             uint32_t bmi, int lane, uint32_t zero) {
    const uint32_t AS1* p = ls_sgpr_ptr(ent, (uint64_t)line * (kWave * 4)) + (uint32_t)lane;
#pragma unroll
    for (int q = 0; q < BITS; ++q) U.d[q] = NT ? __builtin_nontemporal_load(p + q * kWave) : p[q * kWave];
    const char AS1* bp = reinterpret_cast<const char AS1*>(ls_sgpr_ptr(bm, (uint64_t)bmi * 4));
    U.bm = *reinterpret_cast<const uint32_t AS1*>(bp + zero);
}

__device__ __forceinline__ LsDesc ls_load_desc(const LsDesc* __restrict__ desc, int64_t idx, uint32_t zero) {
    const char AS1* bp = reinterpret_cast<const char AS1*>(ls_uniform_ptr(desc + idx));
    const ls_u32x4 v = *reinterpret_cast<const ls_u32x4 AS1*>(bp + zero);
    LsDesc d;
    d.line0 = v.x; d.n_rows = v.y; d.bm0 = v.z; d.pad = v.w;
    return d;
}
__device__ __forceinline__ void ls_desc_uniform(LsDesc& d) {
    d.line0 = __builtin_amdgcn_readfirstlane(d.line0);
    d.n_rows = __builtin_amdgcn_readfirstlane(d.n_rows);
    d.bm0 = __builtin_amdgcn_readfirstlane(d.bm0);
}

// x + (x of the lane `shift` below it in its row of 16, or 0)
template <int CTRL>
__device__ __forceinline__ unsigned ls_row_shr_add(unsigned x) {
    return x + (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
}

// NIB: the table holds four bits per event (112 K < E <= 260 096, as the chunk filter's: k_fr_q4_table)
// NT: the stream is read with non-temporal loads -- for a pass that runs ALONE. The lines of the stream then do not push the exact
// pass's working set (the logp table, the position words, the CSR lines it shares between candidates) out of the L2 / Infinity
// Cache: exact pass 45.0 -> 36.7 us, the filter itself level or 1 % faster, a serial pass 0.310 -> 0.290 ms. In a PIPELINED sequence
// the next pass's filter shares the memory system with this pass's exact pass, and the same loads cost 4 % (0.239 against 0.2295 ms
// per step): the launch picks the form by what the pass is.
template <int BITS, bool NIB, bool NT>
__global__ __launch_bounds__(kLsThreads) void k_fr_filter_ls(const uint32_t* __restrict__ ent, const uint32_t* __restrict__ bm,
                                                              const LsDesc* __restrict__ desc, const uint8_t* __restrict__ q8, int32_t n_events,
                                                              int64_t n_steps, int64_t n_slabs, int ss, int threshold, uint16_t* __restrict__ cand_list,
                                                              unsigned* __restrict__ cand_cnt, unsigned long long* __restrict__ cand_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* s_m = reinterpret_cast<unsigned long long*>(smem); // [2][kLsSlots] ballots, slot (ls * 8 + j) * 16 + wave
    uint8_t* tab = reinterpret_cast<uint8_t*>(smem + 2 * kLsSlots * 8);
    // The table is read at ABSOLUTE LDS addresses (this kernel has no static LDS: its dynamic LDS starts at 0). Through `tab[col]`
    // every read paid a `v_add_u32 v, 0, v` -- the link-time base of `smem` -- i.e. 32 of the ~110 vector instructions of a unit.
    typedef const uint8_t __attribute__((address_space(3))) * lds_bytes;
    if (__builtin_expect((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem != 0u, 0)) __builtin_trap();
    constexpr uint32_t kTabAt = 2 * kLsSlots * 8;
    constexpr uint32_t kColMask = (1u << BITS) - 1u;
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    uint32_t zero_v = 0u; // an opaque zero byte offset in a vector register (see q8_load_chunk)
    asm volatile("" : "+v"(zero_v));
    const int64_t G = (int64_t)gridDim.x;
    const int64_t q_wg = n_steps / G, r_wg = n_steps % G, b_wg = (int64_t)blockIdx.x;
    const int64_t s_begin = b_wg * q_wg + (b_wg < r_wg ? b_wg : r_wg), s_end = s_begin + q_wg + (b_wg < r_wg ? 1 : 0);
    const int64_t last = n_steps * kLsWaves; // desc[last]: the empty range (its lines: zeros behind the stream)
    auto didx = [&](int64_t s) { return s < s_end ? s * kLsWaves + wave : last; };
    // descriptors run TWO steps ahead of the adds: their loads are then older than a whole step's unit loads on every path,
    // and the wait the compiler puts in front of their use is one the unit loads have long satisfied
    LsDesc d_cur = ls_load_desc(desc, didx(s_begin), zero_v);
    LsDesc d_nxt = ls_load_desc(desc, didx(s_begin + 1), zero_v);
    ls_desc_uniform(d_cur);
    LsUnit<BITS> A, B;
#if LS_DEPTH == 3
    LsUnit<BITS> C;
#endif
    ls_load_unit<BITS, NT>(A, ent, bm, d_cur.line0, d_cur.bm0, lane, zero_v);
#if LS_DEPTH == 3
    {
        const bool two = d_cur.n_rows > (uint32_t)kLsUnitRows;
        // (zero_line is not known yet: a range of one unit asks for its own lines again)
        ls_load_unit<BITS, NT>(B, ent, bm, two ? d_cur.line0 + (uint32_t)BITS : d_cur.line0, two ? d_cur.bm0 + 1u : d_cur.bm0, lane, zero_v);
    }
#endif
    {
        const uint4* src = reinterpret_cast<const uint4*>(q8);
        uint4* dst = reinterpret_cast<uint4*>(tab);
        const int tab_bytes = NIB ? (n_events + 3) / 2 : n_events + 2; // (+ the two sentinels)
        for (int i = tid; i < (tab_bytes + 15) / 16; i += kLsThreads) dst[i] = src[i];
    }
    for (int i = tid; i < 2 * kLsSlots; i += kLsThreads) s_m[i] = 0ULL;
    // the LDS table's bytes behind the two sentinels are never addressed: every column of the stream is < n_events + 2
    ls_barrier();
    uint32_t zero_line, zero_bm; // the unit of zero lines behind the stream
    {
        LsDesc e = ls_load_desc(desc, last, zero_v);
        ls_desc_uniform(e);
        zero_line = e.line0;
        zero_bm = e.bm0;
    }
    unsigned long long n_cand = 0;
    uint64_t nz_prev = 0u; // the items of the previous step this wave found candidates in
    int it = 0;
    int64_t s_emit = -1;
    for (int64_t s = s_begin;; ++s, ++it) {
        const bool have = s < s_end;
        LsDesc d_nn = ls_load_desc(desc, didx(s + 2), zero_v);
        if (s_emit >= 0) // ---- the lists of the previous step's slabs, four slabs at a time
#pragma unroll 1
        for (int h4 = 0; h4 < kLsStepSlabs / 4; ++h4) {
            const unsigned long long* mall = s_m + ((it - 1) & 1) * kLsSlots + h4 * 512;
            if (h4 * 4 >= ss) break;
            const int64_t slab0 = s_emit * ss + h4 * 4;
            // lane l: slots 8 l .. 8 l + 7; a row of 16 lanes = the 128 slots of one slab
            const uint4* q = reinterpret_cast<const uint4*>(mall + 8 * lane);
            const uint4 x0 = q[0], x1 = q[1], x2 = q[2], x3 = q[3];
            unsigned c[8];
            c[0] = __builtin_popcount(x0.x) + __builtin_popcount(x0.y); c[1] = __builtin_popcount(x0.z) + __builtin_popcount(x0.w);
            c[2] = __builtin_popcount(x1.x) + __builtin_popcount(x1.y); c[3] = __builtin_popcount(x1.z) + __builtin_popcount(x1.w);
            c[4] = __builtin_popcount(x2.x) + __builtin_popcount(x2.y); c[5] = __builtin_popcount(x2.z) + __builtin_popcount(x2.w);
            c[6] = __builtin_popcount(x3.x) + __builtin_popcount(x3.y); c[7] = __builtin_popcount(x3.z) + __builtin_popcount(x3.w);
            unsigned ex[8], run = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { ex[k] = run; run += c[k]; }
            unsigned inc = run; // inclusive scan over the row of 16 lanes
            inc = ls_row_shr_add<0x111>(inc);
            inc = ls_row_shr_add<0x112>(inc);
            inc = ls_row_shr_add<0x114>(inc);
            inc = ls_row_shr_add<0x118>(inc);
            const int wk = wave & 7;
            unsigned sel = ex[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) sel = wk == k ? ex[k] : sel;
            const unsigned myoff = inc - run + sel; // list offset of the slot of THIS wave that lane l covers
            if (wave == 0 && (lane & 15) == 15) {
                const int64_t slab = slab0 + (lane >> 4);
                if (h4 * 4 + (lane >> 4) < ss && slab < n_slabs) cand_cnt[slab] = inc;
            }
#pragma unroll
            for (int ls = 0; ls < 4; ++ls)
                if (h4 * 4 + ls < ss && slab0 + ls < n_slabs) n_cand += (unsigned)__builtin_amdgcn_readlane((int)inc, 16 * ls + 15);
            uint32_t nz = (uint32_t)(nz_prev >> (32 * h4));
            while (nz != 0u) { // (wave-uniform)
                const int jj = __builtin_amdgcn_readfirstlane(__builtin_ctz(nz));
                nz &= nz - 1u;
                const unsigned long long mv = mall[jj * kLsWaves + wave]; // (one address: a broadcast read)
                const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(mv >> 32)) << 32) |
                                             (uint32_t)__builtin_amdgcn_readfirstlane((int)mv);
                const unsigned off = (unsigned)__builtin_amdgcn_readlane((int)myoff, jj * 2 + (wave >> 3));
                if ((m >> lane) & 1ULL) {
                    const unsigned pos = off + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    cand_list[(slab0 + (jj >> 3)) * (int64_t)kLsSlabRows + pos] = (uint16_t)((jj & 7) * 1024 + tid);
                }
            }
        }
        uint64_t nz_now = 0u;
        if (have) {
            unsigned long long* mw = s_m + (it & 1) * kLsSlots + wave;
            const int n_units = (int)((d_cur.n_rows + kLsUnitRows - 1) / kLsUnitRows);
            const uint32_t line0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d_cur.line0), bm0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d_cur.bm0);
            int jj = 0;
            unsigned sum = 0;
            // two register sets in ping-pong, the next unit requested before this one's LDS work. A range's last half asks for the
            // zero lines behind the stream (a cached unit): the same number of loads in flight on every path. (Asking for the lines
            // that FOLLOW the range instead -- no select on the way to the address -- costs 10 % of the step: 0.259 against 0.236
            // ms, the next wave's first unit fetched twice.)
#define LS_HALF(UC, UN, UI)                                                                                                  \
            {                                                                                                                \
                const bool more = (UI) + (LS_DEPTH - 1) < n_units;                                                           \
                uint32_t ln = more ? line0 + (uint32_t)((UI) + (LS_DEPTH - 1)) * BITS : zero_line;                           \
                uint32_t bn = more ? bm0 + (uint32_t)(UI) + (uint32_t)(LS_DEPTH - 1) : zero_bm;                              \
                ls_load_unit<BITS, NT>(UN, ent, bm, ln, bn, lane, zero_v);                                                       \
                unsigned v[kLsUnitRows];                                                                                     \
                _Pragma("unroll") for (int t = 0; t < kLsUnitRows; ++t) {                                                    \
                    const int bo = t * BITS, w = bo >> 5, sh = bo & 31;                                                      \
                    uint32_t col;                                                                                            \
                    if (sh + BITS <= 32) col = __builtin_amdgcn_ubfe(UC.d[w], (uint32_t)sh, (uint32_t)BITS);                 \
                    else col = __builtin_amdgcn_alignbit(UC.d[w + 1 < BITS ? w + 1 : w], UC.d[w], (uint32_t)sh) & kColMask;  \
                    if (LS_ABL == 1) col = (col & ~127u) | ((uint32_t)lane << 2); /* timing ablation: no two lanes of a half on one bank */ \
                    if (LS_ABL == 3) v[t] = col & 255u; /* timing ablation: no table read at all */                                  \
                    else if (NIB) v[t] = ((unsigned)*(lds_bytes)(uintptr_t)((col >> 1) + kTabAt) >> ((col & 1u) << 2)) & 15u;            \
                    else v[t] = *(lds_bytes)(uintptr_t)(col + kTabAt);                                                       \
                }                                                                                                            \
                const uint32_t bmv = (uint32_t)__builtin_amdgcn_readfirstlane((int)UC.bm);                                   \
                _Pragma("unroll") for (int t = 0; t < kLsUnitRows; ++t) {                                                    \
                    sum += v[t];                                                                                             \
                    if (__builtin_expect((bmv >> t) & 1u, 0)) { /* an item ends here */                                      \
                        const unsigned long long m = __ballot((int)sum <= threshold);                                        \
                        if (lane == 0) mw[jj * kLsWaves] = m;                                                                \
                        nz_now |= (uint64_t)(m != 0ULL ? 1u : 0u) << jj;                                                     \
                        sum = 0;                                                                                             \
                        jj = __builtin_amdgcn_readfirstlane(jj + 1);                                                         \
                    }                                                                                                        \
                }                                                                                                            \
            }
#if LS_DEPTH == 3
            // (unit 1 of this step is in B since the end of the previous step)
            for (int u = 0; u < n_units;) {
                LS_HALF(A, C, u)
                ++u;
                if (u >= n_units) break;
                LS_HALF(B, A, u)
                ++u;
                if (u >= n_units) break;
                LS_HALF(C, B, u)
                ++u;
            }
#else
            for (int u = 0; u < n_units;) {
                LS_HALF(A, B, u)
                ++u;
                if (u >= n_units) break;
                LS_HALF(B, A, u)
                ++u;
            }
#endif
#undef LS_HALF
        }
        // the next step's first unit(s) go out before this step's barrier (always into A, B)
        ls_desc_uniform(d_nxt);
        ls_load_unit<BITS, NT>(A, ent, bm, d_nxt.line0, d_nxt.bm0, lane, zero_v);
#if LS_DEPTH == 3
        {
            const bool two = d_nxt.n_rows > (uint32_t)kLsUnitRows;
            ls_load_unit<BITS, NT>(B, ent, bm, two ? d_nxt.line0 + (uint32_t)BITS : zero_line, two ? d_nxt.bm0 + 1u : zero_bm, lane, zero_v);
        }
#endif
        d_cur = d_nxt;
        d_nxt = d_nn;
        nz_prev = LS_ABL == 4 ? 0ULL // (ablation 4: no list is written)
                              : ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(nz_now >> 32)) << 32) |
                                    (uint32_t)__builtin_amdgcn_readfirstlane((int)nz_now);
        s_emit = have ? s : -1;
        if (!have) break; // nothing left in flight: nobody reads anything of ours any more
        ls_barrier();
    }
    if (tid == 0 && n_cand) atomicAdd(cand_count, n_cand); // one global atomic per workgroup
}

} // namespace

// ---- host side -----------------------------------------------------------------------------------------------------------------
bool ls_applies(int32_t n_events) {
    static const bool off = CANOPY_TUNE_ENV("CANOPY_FR_NOLS") != nullptr; // tuning only
    return !off && n_events >= 1 && n_events <= kLsMaxEvents4;
}
// bits per entry: the columns 0 .. n_events + 1 (two sentinels) must fit
int ls_bits_for(int32_t n_events) {
    const int64_t c = (int64_t)n_events + 2;
    return c <= 4096 ? 12 : (c <= 16384 ? 14 : (c <= 65536 ? 16 : (c <= 131072 ? 17 : 18)));
}

void ls_free(LsOwned* L) {
    if (L->ent) (void)hipFree(L->ent);
    if (L->bm) (void)hipFree(L->bm);
    if (L->desc) (void)hipFree(L->desc);
    if (L->posinfo) (void)hipFree(L->posinfo);
    if (L->item_len8) (void)hipFree(L->item_len8);
    *L = LsOwned();
}

// Builds the row stream of a model (waits for the stream). *ok = false: the model has no such layout (no rows, too many
// events, a range too long for its 32-bit counters) -- not an error.
hipError_t ls_build(const int64_t* d_row_ptr, const int32_t* d_col_idx, int64_t n_rows, int64_t nnz, int32_t n_events, int n_cu, LsOwned* out,
                    bool* ok, hipStream_t s) {
    *ok = false;
    *out = LsOwned();
    if (!ls_applies(n_events) || n_rows < 1) return hipSuccess;
    const int64_t n_slabs = (n_rows + kLsSlabRows - 1) / kLsSlabRows;
    // Slabs per step (= per workgroup barrier): as many as the filter's buffers hold, unless that leaves most CUs without work -- a
    // model of 2e6 rows has 245 slabs: 62 steps of four occupy a quarter of the CUs, 245 steps of one nearly all of them. (Not at
    // the first sign of imbalance: the 1.25e7-row shard of an 8-GPU run has 1526 slabs = 382 steps of four = two rounds on 256 CUs,
    // the second half empty. 763 steps of two take the pass alone from 53.2 to 49.6 us, but a PIPELINED sequence fills the idle CUs
    // with the next pass's filter anyway and only pays for the extra barriers: 34.5 against 33.8 us per step.)
    int ss = kLsStepSlabs;
    {
        static const int forced = CANOPY_TUNE_ENV("CANOPY_LS_SS") ? atoi(CANOPY_TUNE_ENV("CANOPY_LS_SS")) : 0; // tuning only
        const int64_t G = n_cu > 0 ? n_cu : 256;
        double best = -1.0;
        for (int c = kLsStepSlabs; c >= 1; c >>= 1) {
            const int64_t steps = (n_slabs + c - 1) / c, rounds = (steps + G - 1) / G;
            const double eff = (double)steps / (double)(rounds * G);
            if (eff >= 0.7) { ss = c; break; }          // the largest step that keeps most CUs busy
            if (eff > best + 1e-9) { best = eff; ss = c; } // else the best there is (a small model: one slab per step)
        }
        if (forced == 1 || forced == 2 || forced == 4 || forced == kLsStepSlabs) ss = forced;
    }
    const int64_t n_steps = (n_slabs + ss - 1) / ss;
    const int64_t n_ranges = n_steps * kLsWaves;
    const int bits = ls_bits_for(n_events);
    uint32_t *d_sizes = nullptr, *d_rows = nullptr;
    int64_t *d_lines = nullptr, *d_units = nullptr;
    LsOwned L;
    auto bail = [&](hipError_t err) {
        (void)hipGetLastError();
        if (d_sizes) (void)hipFree(d_sizes);
        if (d_rows) (void)hipFree(d_rows);
        if (d_lines) (void)hipFree(d_lines);
        if (d_units) (void)hipFree(d_units);
        ls_free(&L);
        return err;
    };
    hipError_t e = hipMalloc(&d_sizes, sizeof(uint32_t) * (size_t)(n_slabs * kLsWaves));
    if (e == hipSuccess) e = hipMalloc(&d_rows, sizeof(uint32_t) * (size_t)n_ranges);
    if (e == hipSuccess) e = hipMalloc(&d_lines, sizeof(int64_t) * (size_t)n_ranges);
    if (e == hipSuccess) e = hipMalloc(&d_units, sizeof(int64_t) * (size_t)n_ranges);
    if (e != hipSuccess) return bail(e);
    hipLaunchKernelGGL(k_ls_sizes, dim3((unsigned)n_slabs), dim3(256), 0, s, d_row_ptr, n_rows, n_slabs, d_sizes);
    e = hipGetLastError();
    if (e != hipSuccess) return bail(e);
    hipLaunchKernelGGL(k_ls_range_sizes, dim3((unsigned)((n_ranges + 255) / 256)), dim3(256), 0, s, (const uint32_t*)d_sizes, n_slabs, n_ranges, bits, ss,
                       d_lines, d_units, d_rows);
    e = hipGetLastError();
    if (e == hipSuccess) e = scan_inclusive_i64(d_lines, n_ranges, s);
    if (e == hipSuccess) e = scan_inclusive_i64(d_units, n_ranges, s);
    int64_t total_lines = 0, total_units = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&total_lines, d_lines + (n_ranges - 1), sizeof(int64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&total_units, d_units + (n_ranges - 1), sizeof(int64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bail(e);
    // 32-bit line and unit indices; a range of >= 2^32 entry rows cannot be described (a model of absurdly long rows)
    if (total_lines + 2 * bits >= (int64_t)0xffffffffLL || total_units + 2 >= (int64_t)0xffffffffLL) { (void)bail(hipSuccess); return hipSuccess; }
    // An item is as long as its longest row: a few rows of thousands of entries among short ones pad their items (64 rows each).
    // A model whose stream would be more than twice its entries (+ a line per range and the padding rows of the last slab) keeps
    // the chunk stream: the filter must not read more than it saves.
    {
        const int64_t ideal_lines = (int64_t)(((__int128)(nnz + n_slabs * kLsSlabRows - n_rows) * bits + 2047) / 2048);
        if (total_lines > 2 * ideal_lines + 2 * n_ranges) { (void)bail(hipSuccess); return hipSuccess; }
    }
    L.n_lines = total_lines + 2 * bits; // (a unit of zero lines behind the stream: what a range's last half and an idle wave request)
    L.n_units = total_units + 1;
    L.n_steps = n_steps;
    L.ss = ss;
    L.n_slabs = n_slabs;
    L.bits = bits;
    e = hipMalloc(&L.ent, (size_t)L.n_lines * 256);
    if (e == hipSuccess) e = hipMalloc(&L.bm, sizeof(uint32_t) * (size_t)L.n_units);
    if (e == hipSuccess) e = hipMalloc(&L.desc, sizeof(LsDesc) * (size_t)(n_ranges + 1));
    if (e == hipSuccess) e = hipMalloc(&L.posinfo, sizeof(uint32_t) * (size_t)(n_slabs * kLsSlabRows));
    if (e == hipSuccess) e = hipMalloc(&L.item_len8, (size_t)(n_slabs * 128));
    // (every line of a range is written whole except the bits behind its last entry row; the zero unit must be zero)
    if (e == hipSuccess) e = hipMemsetAsync(L.ent, 0, (size_t)L.n_lines * 256, s);
    if (e == hipSuccess) e = hipMemsetAsync(L.bm, 0, sizeof(uint32_t) * (size_t)L.n_units, s);
    if (e != hipSuccess) return bail(e);
    hipLaunchKernelGGL(k_ls_desc, dim3((unsigned)((n_ranges + 1 + 255) / 256)), dim3(256), 0, s, (const int64_t*)d_lines, (const int64_t*)d_units,
                       (const uint32_t*)d_rows, n_ranges, reinterpret_cast<LsDesc*>(L.desc));
    e = hipGetLastError();
    if (e != hipSuccess) return bail(e);
#define LS_FILL(B)                                                                                                         \
    hipLaunchKernelGGL(k_ls_fill<B>, dim3((unsigned)n_steps), dim3(kLsThreads), 0, s, d_row_ptr, d_col_idx, n_rows, n_events, n_slabs, ss, \
                       reinterpret_cast<const LsDesc*>(L.desc), L.ent, L.bm, L.posinfo, L.item_len8)
    switch (bits) {
        case 12: LS_FILL(12); break;
        case 14: LS_FILL(14); break;
        case 16: LS_FILL(16); break;
        case 17: LS_FILL(17); break;
        default: LS_FILL(18); break;
    }
#undef LS_FILL
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bail(e);
    (void)hipFree(d_sizes); (void)hipFree(d_rows); (void)hipFree(d_lines); (void)hipFree(d_units);
    L.stream_bytes = L.n_lines * 256 + L.n_units * 4 + (n_ranges + 1) * 16;
    L.device_bytes = L.stream_bytes + n_slabs * kLsSlabRows * 4 + n_slabs * 128;
    *out = L;
    *ok = true;
    return hipSuccess;
}

size_t ls_filter_lds(int32_t n_events, bool nib) {
    const size_t tab = nib ? ((size_t)n_events + 3) / 2 : (size_t)n_events + 2;
    return (size_t)2 * kLsSlots * 8 + ((tab + 15) / 16) * 16;
}

// q8: the table of the pass (8 bits per event, or 4: nib) with its two sentinels behind it (launch_fr_q8_table)
hipError_t launch_ls_filter(const LsView& L, const uint8_t* q8, int32_t n_events, bool nib, bool alone, int threshold, uint16_t* cand_list,
                            unsigned* cand_cnt, unsigned long long* cand_count, int n_cu, hipStream_t s) {
    const size_t lds = ls_filter_lds(n_events, nib);
    if (lds > (size_t)160 * 1024 || (!nib && n_events > kLsMaxEvents8)) return hipErrorInvalidValue;
    static const int forced = CANOPY_TUNE_ENV("CANOPY_LS_GRID") ? atoi(CANOPY_TUNE_ENV("CANOPY_LS_GRID")) : 0; // tuning only
    int64_t g = forced > 0 ? forced : n_cu;
    if (g > L.n_steps) g = L.n_steps;
    if (g < 1) g = 1;
    const LsDesc* desc = reinterpret_cast<const LsDesc*>(L.desc);
#define LS_GO1(B, N, T)                                                                                                         \
    do {                                                                                                                        \
        /* (per device, so at every launch: one process may drive several devices) */                                          \
        const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_fr_filter_ls<B, N, T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        if (attr != hipSuccess) return attr;                                                                                    \
        hipLaunchKernelGGL((k_fr_filter_ls<B, N, T>), dim3((unsigned)g), dim3(kLsThreads), lds, s, L.ent, L.bm, desc, q8, n_events, L.n_steps, L.n_slabs, L.ss, \
                           threshold, cand_list, cand_cnt, cand_count);                                                         \
        return hipGetLastError();                                                                                               \
    } while (0)
#define LS_GO(B, N)                                                                                                             \
    do {                                                                                                                        \
        if (alone) LS_GO1(B, N, true);                                                                                          \
        LS_GO1(B, N, false);                                                                                                    \
    } while (0)
    if (nib) {
        if (L.bits == 17) LS_GO(17, true);
        if (L.bits == 18) LS_GO(18, true);
        return hipErrorInvalidValue; // (the 4-bit table is for more than 112 K events)
    }
    switch (L.bits) {
        case 12: LS_GO(12, false);
        case 14: LS_GO(14, false);
        case 16: LS_GO(16, false);
        case 17: LS_GO(17, false);
        default: return hipErrorInvalidValue;
    }
#undef LS_GO
#undef LS_GO1
}

} // namespace canopy
