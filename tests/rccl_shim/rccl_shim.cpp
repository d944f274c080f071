// rccl_shim.cpp -- TEST DOUBLE of the RCCL entry points libcanopy.so binds (comm_rccl.hip), so that canopy_comm can run at
// world size > 1 on ONE GPU: real RCCL refuses two ranks on one device, and the GPU box of this project has one device.
// Test infrastructure only (tests/test_comm_world2.py builds it as librccl.so.1 and LD_PRELOADs it into rank processes that
// import neither torch nor anything else carrying an RCCL; the product never sees it).
//
// What it keeps of the real thing -- because that is what the library's ordering logic has to get right:
//   * a collective is ASYNCHRONOUS work on the stream it was issued to (copy out, a host function, copy back);
//   * the ranks must issue the collectives of a communicator in the SAME ORDER (the k-th call of every rank meets the k-th
//     call of the others; count and type are checked), and a rank's k-th collective cannot complete before its peers' k-th
//     has been reached -- while a process has ONE thread for its host functions, so two collectives of one communicator that
//     become runnable out of issue order (two lanes, no edge between them) dead-lock here, where RCCL would be left to its
//     internal serialisation. A run under this shim therefore proves the CHAINED / OWN_STREAM ordering of comm_rccl.hip.
// Exchange: a POSIX shared-memory segment named by the unique id; 64 slots in rotation; every rank sums all ranks' words in
// rank order (identical results everywhere). A rank that waits more than 60 s for its peers flags an error and returns.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <new>

namespace {
constexpr int kSlots = 64, kMaxRanks = 8, kMaxWords = 128;
struct Slot {
    std::atomic<uint32_t> gen, arrived, departed;
    uint32_t count[kMaxRanks], dtype[kMaxRanks];
    uint64_t data[kMaxRanks][kMaxWords];
};
struct Shm {
    std::atomic<uint32_t> attached, error;
    Slot slots[kSlots];
};
struct Call {
    struct ShimComm* c;
    uint64_t seq;
    uint32_t count, dtype;
    uint64_t* h_in;
    uint64_t* h_out;
};
struct ShimComm {
    Shm* shm = nullptr;
    char name[64] = {0};
    int n = 1, rank = 0;
    uint64_t seq = 0;
    uint64_t* h_ring = nullptr; // pinned [kSlots][2][kMaxWords]
    Call calls[kSlots];
};
double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

void exchange(void* p) { // the host function of one collective: runs on the process's callback thread, in stream order
    Call* k = static_cast<Call*>(p);
    ShimComm* c = k->c;
    Slot& s = c->shm->slots[k->seq % kSlots];
    const uint32_t want_gen = (uint32_t)(k->seq / kSlots);
    const double t0 = now_s();
    auto late = [&] { return now_s() - t0 > 60.0 || c->shm->error.load() != 0; };
    while (s.gen.load(std::memory_order_acquire) != want_gen) if (late()) { c->shm->error.store(1); return; }
    s.count[c->rank] = k->count; s.dtype[c->rank] = k->dtype;
    memcpy(s.data[c->rank], k->h_in, sizeof(uint64_t) * k->count);
    s.arrived.fetch_add(1, std::memory_order_acq_rel);
    while (s.arrived.load(std::memory_order_acquire) < (uint32_t)c->n) if (late()) { c->shm->error.store(2); return; }
    for (int r = 0; r < c->n; ++r)
        if (s.count[r] != k->count || s.dtype[r] != k->dtype) c->shm->error.store(3); // the ranks' k-th collectives differ: order broken
    for (uint32_t i = 0; i < k->count; ++i) {
        if (k->dtype == (uint32_t)ncclDouble) {
            double acc = 0.0;
            for (int r = 0; r < c->n; ++r) { double v; memcpy(&v, &s.data[r][i], 8); acc += v; }
            memcpy(&k->h_out[i], &acc, 8);
        } else {
            uint64_t acc = 0;
            for (int r = 0; r < c->n; ++r) acc += s.data[r][i];
            k->h_out[i] = acc;
        }
    }
    if (s.departed.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->n) { // the last one out hands the slot on
        s.arrived.store(0); s.departed.store(0);
        s.gen.fetch_add(1, std::memory_order_release);
    }
}
} // namespace

extern "C" {
ncclResult_t ncclGetVersion(int* v) { if (v) *v = 99999; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "rccl_shim error"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/canopy_rccl_shim_%d_%ld", (int)getpid(), (long)time(nullptr));
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int n, ncclUniqueId id, int rank) {
    if (n < 1 || n > kMaxRanks || rank < 0 || rank >= n) return ncclInvalidArgument;
    ShimComm* c = new (std::nothrow) ShimComm();
    if (!c) return ncclSystemError;
    c->n = n; c->rank = rank;
    strncpy(c->name, id.internal, sizeof(c->name) - 1);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shm)) != 0) { delete c; return ncclSystemError; }
    void* m = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); // (a fresh segment is zero-filled: gen 0, nobody arrived)
    close(fd);
    if (m == MAP_FAILED) { delete c; return ncclSystemError; }
    c->shm = static_cast<Shm*>(m);
    if (hipHostMalloc(&c->h_ring, sizeof(uint64_t) * kSlots * 2 * kMaxWords) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
    c->shm->attached.fetch_add(1);
    const double t0 = now_s();
    while (c->shm->attached.load() < (uint32_t)n) if (now_s() - t0 > 60.0) return ncclSystemError; // (like the real one: init is collective)
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t*, int, const int*) { return ncclInvalidUsage; } // one process, N devices: not on a one-GPU box
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    ShimComm* c = reinterpret_cast<ShimComm*>(comm);
    if (!c) return ncclSuccess;
    if (c->shm) { if (c->rank == 0) shm_unlink(c->name); munmap(c->shm, sizeof(Shm)); }
    if (c->h_ring) (void)hipHostFree(c->h_ring);
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }
ncclResult_t ncclAllGather(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) { return ncclInvalidUsage; }
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t st) {
    ShimComm* c = reinterpret_cast<ShimComm*>(comm);
    if (!c || op != ncclSum || count > (size_t)kMaxWords || (dt != ncclDouble && dt != ncclUint64 && dt != ncclInt64)) return ncclInvalidArgument;
    if (c->shm->error.load() != 0) return ncclSystemError;
    const uint64_t k = c->seq++;
    Call* call = &c->calls[k % kSlots]; // (64 collectives in flight at most: the library keeps at most four)
    call->c = c; call->seq = k; call->count = (uint32_t)count; call->dtype = (uint32_t)dt;
    call->h_in = c->h_ring + (k % kSlots) * 2 * kMaxWords;
    call->h_out = call->h_in + kMaxWords;
    if (hipMemcpyAsync(call->h_in, send, count * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(st, exchange, call) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpyAsync(recv, call->h_out, count * 8, hipMemcpyHostToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}
// (what a test reads after the run: 0 = every collective met its peers in order)
int canopy_rccl_shim_error(ncclComm_t comm) { ShimComm* c = reinterpret_cast<ShimComm*>(comm); return c && c->shm ? (int)c->shm->error.load() : -1; }
}
