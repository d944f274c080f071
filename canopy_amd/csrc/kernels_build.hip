// kernels_build.hip -- one-off model construction on the device: CSR validation at upload, and
// the synthetic cut-set generator of the benchmark driver (the reference's benchmarks/ is a
// 0-byte CMakeLists.txt, reference benchmarks/CMakeLists.txt). Not on the timed path.
#include "canopy_device.h"
#include "canopy_internal.h"

namespace canopy {

namespace {

// ---- validation: row_ptr monotone, columns in range and strictly ascending within a row -------
__global__ __launch_bounds__(256) void k_validate(const int64_t* __restrict__ row_ptr,
                                                  const int32_t* __restrict__ col_idx, int64_t n_rows,
                                                  int64_t nnz, int32_t n_events, unsigned long long* status) {
    // status[0] = lowest bad row (init ~0), status[1] = max row length
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long bad = ~0ULL, maxlen = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
        const int64_t b = row_ptr[r], e = row_ptr[r + 1];
        bool ok = b >= 0 && e >= b && e <= nnz; // never index col_idx outside [0, nnz)
        if (ok) {
            int32_t prev = -1;
            for (int64_t k = b; k < e; ++k) {
                const int32_t c = col_idx[k];
                ok = ok && (c > prev) && (c < n_events);
                prev = c;
            }
            const unsigned long long len = (unsigned long long)(e - b);
            maxlen = len > maxlen ? len : maxlen;
        }
        if (!ok && (unsigned long long)r < bad) bad = (unsigned long long)r;
    }
    if (bad != ~0ULL) atomicMin(&status[0], bad);
    if (maxlen) atomicMax(&status[1], maxlen);
}

__global__ __launch_bounds__(256) void k_rebase(int64_t* row_ptr, int64_t n, int64_t base) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) row_ptr[i] -= base;
}

// ---- synthetic generator -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_synth_len(uint64_t key_len, int64_t first_row, int64_t n_rows,
                                                   uint32_t span, int64_t* row_ptr /* [n_rows+1] */) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += stride) {
        const int64_t len = 2 + (int64_t)bounded(draw_keyed(key_len, (uint64_t)(first_row + i), 0), span);
        row_ptr[i + 1] = len;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) row_ptr[0] = 0;
}

// Inclusive scan of int64, three-level: per-block scan (2048 items) + recursive scan of block sums.
constexpr int kScanBlock = 256, kScanItems = 8, kScanTile = kScanBlock * kScanItems;

__global__ __launch_bounds__(kScanBlock) void k_scan_tile(int64_t* data, int64_t n, int64_t* tile_sums) {
    __shared__ int64_t s_wave[kScanBlock / kWave];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    int64_t v[kScanItems];
    int64_t run = 0;
#pragma unroll
    for (int u = 0; u < kScanItems; ++u) {
        const int64_t x = (base + u < n) ? data[base + u] : 0;
        run += x;
        v[u] = run;
    }
    // wave inclusive scan of the per-thread totals
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t incl = run;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int64_t o = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += o;
    }
    if (lane == kWave - 1) s_wave[wave] = incl;
    __syncthreads();
    int64_t wave_prefix = 0;
    for (int w = 0; w < wave; ++w) wave_prefix += s_wave[w];
    const int64_t excl = wave_prefix + incl - run;
#pragma unroll
    for (int u = 0; u < kScanItems; ++u)
        if (base + u < n) data[base + u] = v[u] + excl;
    if (threadIdx.x == kScanBlock - 1 && tile_sums) tile_sums[blockIdx.x] = excl + run;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_add(int64_t* data, int64_t n, const int64_t* tile_prefix) {
    if (blockIdx.x == 0) return;
    const int64_t add = tile_prefix[blockIdx.x - 1];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
    for (int u = 0; u < kScanItems; ++u)
        if (base + u < n) data[base + u] += add;
}

hipError_t scan_inclusive(int64_t* data, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int64_t tiles = (n + kScanTile - 1) / kScanTile;
    if (tiles == 1) {
        hipLaunchKernelGGL(k_scan_tile, dim3(1), dim3(kScanBlock), 0, s, data, n, (int64_t*)nullptr);
        return hipGetLastError();
    }
    int64_t* sums = nullptr;
    hipError_t e = hipMalloc(&sums, sizeof(int64_t) * (size_t)tiles);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_scan_tile, dim3((unsigned)tiles), dim3(kScanBlock), 0, s, data, n, sums);
    e = scan_inclusive(sums, tiles, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)tiles), dim3(kScanBlock), 0, s, data, n, sums);
        e = hipGetLastError();
    }
    hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(sums);
    return e != hipSuccess ? e : e2;
}

__global__ __launch_bounds__(256) void k_synth_cols(uint64_t key_col, int64_t first_row, int64_t n_rows,
                                                    uint32_t n_events, const int64_t* __restrict__ row_ptr,
                                                    int32_t* col_idx) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += stride) {
        const int64_t b = row_ptr[i];
        const int len = (int)(row_ptr[i + 1] - b);
        int32_t* out = col_idx + b; // the row is built in place (insertion sort on <= 32 entries)
        int got = 0;
        for (uint64_t t = 0; got < len; ++t) {
            const int32_t c = (int32_t)bounded(draw_keyed(key_col, (uint64_t)(first_row + i), t), n_events);
            bool dup = false;
            for (int k = 0; k < got; ++k) dup |= (out[k] == c);
            if (dup) continue;
            int k = got++;
            while (k > 0 && out[k - 1] > c) { out[k] = out[k - 1]; --k; }
            out[k] = c;
        }
    }
}

int grid_for(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

} // namespace

hipError_t scan_inclusive_i64(int64_t* data, int64_t n, hipStream_t s) { return scan_inclusive(data, n, s); }

namespace {
// Importance contributions: a thread walks the kept rows of one mask word; P_i = exp(s_i) with s_i the same
// CSR-order sum as the pass itself, added to every event of the row (fp64 global atomics, no fixed order).
__global__ __launch_bounds__(256) void k_importance(const uint64_t* __restrict__ mask, int64_t n_words, int64_t n_rows,
                                                    const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                    const double* __restrict__ logp, double* __restrict__ contrib) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t m = mask[w];
    const int64_t left = n_rows - w * 64;
    if (left < 64) m &= (1ULL << left) - 1ULL;
    for (; m != 0ULL; m &= m - 1) {
        const int64_t r = w * 64 + __builtin_ctzll(m);
        const int64_t b = row_ptr[r], e = row_ptr[r + 1];
        double s = 0.0;
        for (int64_t k = b; k < e; ++k) s = s + logp[col_idx[k]]; // plain adds, CSR order
        const double P = exp(s);
        for (int64_t k = b; k < e; ++k) unsafeAtomicAdd(&contrib[col_idx[k]], P);
    }
}
} // namespace

hipError_t launch_importance(const uint64_t* keep_mask, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx,
                             const double* logp, double* contrib, hipStream_t s) {
    const int64_t n_words = (n_rows + 63) / 64;
    if (n_words == 0) return hipSuccess;
    hipLaunchKernelGGL(k_importance, dim3(grid_for(n_words, 256)), dim3(256), 0, s, keep_mask, n_words, n_rows, row_ptr,
                       col_idx, logp, contrib);
    return hipGetLastError();
}

hipError_t launch_validate_csr(const int64_t* row_ptr, const int32_t* col_idx, int64_t n_rows, int64_t nnz,
                               int32_t n_events, int64_t* d_status, hipStream_t s) {
    hipLaunchKernelGGL(k_validate, dim3(grid_for(n_rows, 256)), dim3(256), 0, s, row_ptr, col_idx, n_rows, nnz,
                       n_events, reinterpret_cast<unsigned long long*>(d_status));
    return hipGetLastError();
}

hipError_t launch_rebase_row_ptr(int64_t* row_ptr, int64_t n_plus_1, int64_t base, hipStream_t s) {
    hipLaunchKernelGGL(k_rebase, dim3(grid_for(n_plus_1, 256)), dim3(256), 0, s, row_ptr, n_plus_1, base);
    return hipGetLastError();
}

// row_ptr (rebased to 0) of rows [first_row, first_row + n_rows) of the synthetic family: lengths + scan.
hipError_t synth_row_ptr(uint64_t seed, int64_t first_row, int64_t n_rows, int32_t avg_nnz, int64_t** d_row_ptr_out,
                         hipStream_t s) {
    int64_t* d_row_ptr = nullptr;
    hipError_t e = hipMalloc(&d_row_ptr, sizeof(int64_t) * (size_t)(n_rows + 1));
    if (e != hipSuccess) return e;
    const uint32_t span = (uint32_t)(2 * avg_nnz - 3);
    hipLaunchKernelGGL(k_synth_len, dim3(grid_for(n_rows, 256)), dim3(256), 0, s,
                       stream_key(seed, kStreamLen), first_row, n_rows, span, d_row_ptr);
    e = hipGetLastError();
    if (e == hipSuccess) e = scan_inclusive(d_row_ptr + 1, n_rows, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { (void)hipFree(d_row_ptr); return e; }
    *d_row_ptr_out = d_row_ptr;
    return hipSuccess;
}

hipError_t synth_build_csr(uint64_t seed, int64_t first_row, int64_t n_rows, int32_t n_events,
                           int32_t avg_nnz, int64_t** d_row_ptr_out, int32_t** d_col_idx_out,
                           int64_t* nnz_out, hipStream_t s) {
    int64_t* d_row_ptr = nullptr;
    int32_t* d_col = nullptr;
    hipError_t e = synth_row_ptr(seed, first_row, n_rows, avg_nnz, &d_row_ptr, s);
    if (e != hipSuccess) return e;
    int64_t nnz = 0;
    if (e == hipSuccess)
        e = hipMemcpyAsync(&nnz, d_row_ptr + n_rows, sizeof(int64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipMalloc(&d_col, sizeof(int32_t) * (size_t)nnz + 64);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_synth_cols, dim3(grid_for(n_rows, 256)), dim3(256), 0, s,
                           stream_key(seed, kStreamCol), first_row, n_rows, (uint32_t)n_events, d_row_ptr, d_col);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        (void)hipFree(d_row_ptr);
        if (d_col) (void)hipFree(d_col);
        return e;
    }
    *d_row_ptr_out = d_row_ptr;
    *d_col_idx_out = d_col;
    *nnz_out = nnz;
    return hipSuccess;
}

} // namespace canopy
