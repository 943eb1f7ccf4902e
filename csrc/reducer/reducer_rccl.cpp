// RCCL transport for the gradient-bucket reducer.
//
// One communicator per process (one process per GPU).  All collectives are issued on a
// dedicated non-blocking HIP stream owned by this file; ordering against the stream the
// training step runs on is expressed only with HIP events, so backward kernels and bucket
// collectives overlap:
//
//   step stream:  ... wgrad(bucket k complete) --record(ready[k])--> ... more backward ...
//   comm stream:  wait(ready[k]) -> [pack] -> collective(bucket k) -> [unpack] -> record(done[k])
//   step stream:  (end of backward) wait(done[0..n)) -> optimizer kernel
//
// The step stream is an argument of every launch / wait call (the stream current on the calling
// thread), never a stored handle: inside a hipGraph capture the event record lands on the capturing
// stream, the wait pulls the communication stream into the capture, and the collective becomes a
// node of the graph that every replay executes.
//
// Per bucket the collective is one of
//   * ncclAllReduce in place (ncclAvg), or
//   * ncclReduceScatter + ncclAllGather in place (each rank owns 1/world of the bucket between the
//     two calls) - the two-phase direct form a fully connected xGMI node favours,
// on the bucket's own element type or, for fp32 buckets, on a bf16 wire copy: a pack kernel casts
// and divides by the world size, the collective sums in bf16, an unpack kernel writes the fp32
// gradients back.  Half the bytes cross the links; accumulation inside a rank stays fp32.
//
// The unique id is created on rank 0 (pdt_rccl_unique_id) and distributed by the caller
// through whatever key-value store it already has (torch's TCPStore in this repository).
//
// Role parity: c10d ProcessGroupNCCL + Reducer as reached from
// /root/reference/src/main.py:39-41,53,78.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "reducer_core.h"

namespace {

// Last error text, process-wide: bucket launches run on autograd's device thread while the error is
// usually read from the main thread.
std::mutex g_err_mu;
std::string g_last_error;
std::atomic<int> g_live_streams{0};  // HIP streams created by this library and not yet destroyed

void set_error(const std::string& s) {
    std::lock_guard<std::mutex> g(g_err_mu);
    g_last_error = s;
}

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}

bool nccl_ok(ncclResult_t e, const char* what) {
    if (e == ncclSuccess) return true;
    set_error(std::string(what) + ": " + ncclGetErrorString(e));
    return false;
}

ncclDataType_t to_nccl_dtype(int dtype) {
    switch (dtype) {
        case 0: return ncclFloat32;
        case 1: return ncclBfloat16;
        case 2: return ncclFloat16;
        case 3: return ncclFloat64;
        default: return ncclFloat32;
    }
}

size_t dtype_bytes(int dtype) { return dtype == 0 ? 4 : (dtype == 3 ? 8 : 2); }

// fp32 -> bf16 with a scale (1 / world size); 8 elements per thread, n is a multiple of 8
__global__ void pack_bf16_kernel(const float4* __restrict__ src, uint4* __restrict__ dst, int64_t n8, float scale) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = src[2 * i], b = src[2 * i + 1];
        __hip_bfloat162 p0 = __float22bfloat162_rn(float2{a.x * scale, a.y * scale});
        __hip_bfloat162 p1 = __float22bfloat162_rn(float2{a.z * scale, a.w * scale});
        __hip_bfloat162 p2 = __float22bfloat162_rn(float2{b.x * scale, b.y * scale});
        __hip_bfloat162 p3 = __float22bfloat162_rn(float2{b.z * scale, b.w * scale});
        uint4 o;
        o.x = *reinterpret_cast<uint32_t*>(&p0);
        o.y = *reinterpret_cast<uint32_t*>(&p1);
        o.z = *reinterpret_cast<uint32_t*>(&p2);
        o.w = *reinterpret_cast<uint32_t*>(&p3);
        dst[i] = o;
    }
}

__global__ void unpack_bf16_kernel(const uint4* __restrict__ src, float4* __restrict__ dst, int64_t n8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        // a bf16 is the upper half of the fp32 with the same value
        float4 a, b;
        a.x = __uint_as_float(v.x << 16);
        a.y = __uint_as_float(v.x & 0xffff0000u);
        a.z = __uint_as_float(v.y << 16);
        a.w = __uint_as_float(v.y & 0xffff0000u);
        b.x = __uint_as_float(v.z << 16);
        b.y = __uint_as_float(v.z & 0xffff0000u);
        b.z = __uint_as_float(v.w << 16);
        b.w = __uint_as_float(v.w & 0xffff0000u);
        dst[2 * i] = a;
        dst[2 * i + 1] = b;
    }
}

// One increment per executed bucket collective.  Part of the stream work, so a replayed hipGraph
// counts too: the counter is the evidence that a recorded step really contains its collectives.
__global__ void count_kernel(unsigned long long* ctr) { atomicAdd(ctr, 1ull); }

// Stand-in for the footprint of a multi-rank ring collective on ONE GPU (contention rehearsal, see
// rehearsal_channels below): `channels` workgroups of 256 threads stay resident for as long as the bucket would take at
// a given link rate.  They do not compute; what matters to the persistent one-workgroup-per-CU GEMM kernels is that a
// CU with a resident wave of another kernel cannot take a workgroup that needs the whole register file.  The wait is
// bounded by the constant-rate counter, so every wave leaves.
__global__ __launch_bounds__(256) void resident_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

// The same with the footprint of a real collective's workgroup (PDT_REHEARSE_HEAVY=1): 512 threads that hold at
// least 128 vector registers per lane and PDT_REHEARSE_LDS_KB (default 48) KB of LDS.  Two such waves per SIMD take half
// of the register file, so a 256-register ring workgroup (two waves per SIMD, the whole file) cannot share the CU
// with it - the situation the light variant (a handful of registers, no LDS) does not create.
__global__ __launch_bounds__(512) void resident_heavy_kernel(long long ticks) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    asm volatile("" ::: "v127");  // the register allocation of the kernel reaches v127
    lds[threadIdx.x * 16] = static_cast<char>(ticks);
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (lds[(threadIdx.x ^ 1) * 16] == 77 && ticks < 0) lds[0] = 1;  // keeps the LDS traffic above alive
}

// PDT_REHEARSE_CHANNELS=c (with a one-rank collective kept by PDT_SELF_ALLREDUCE=1): c resident workgroups per bucket;
// PDT_REHEARSE_GBPS: the rate the bucket's duration is derived from (default 150, about one xGMI link).
int rehearsal_channels() {
    static int v = -1;
    if (v < 0) {
        const char* e = std::getenv("PDT_REHEARSE_CHANNELS");
        v = e ? std::atoi(e) : 0;
    }
    return v;
}
double rehearsal_gbps() {
    static double v = -1.0;
    if (v < 0) {
        const char* e = std::getenv("PDT_REHEARSE_GBPS");
        v = e ? std::atof(e) : 150.0;
        if (v <= 0) v = 150.0;
    }
    return v;
}

bool rehearsal_heavy() {
    static int v = -1;
    if (v < 0) {
        const char* e = std::getenv("PDT_REHEARSE_HEAVY");
        v = (e && std::atoi(e) != 0) ? 1 : 0;
    }
    return v == 1;
}
int rehearsal_lds_bytes() {
    static int v = -1;
    if (v < 0) {
        const char* e = std::getenv("PDT_REHEARSE_LDS_KB");
        v = (e ? std::atoi(e) : 48) * 1024;
        if (v < 8192) v = 8192;
        if (v > 64 * 1024) v = 64 * 1024;
    }
    return v;
}

int pack_grid(int64_t n8) {
    const int64_t blocks = (n8 + 255) / 256;
    return static_cast<int>(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks));
}

}  // namespace

struct PdtComm {
    ncclComm_t comm = nullptr;
    int rank = 0;
    int world = 1;
    int device = 0;
    hipStream_t stream = nullptr;  // dedicated communication stream
    hipEvent_t fence_in = nullptr, fence_out = nullptr;  // ordering of one-off collectives against a caller's stream
    std::atomic<int> aborted{0};
};

enum : int {
    PDT_RCCL_AVERAGE = 1,          // mean instead of sum
    PDT_RCCL_SCATTER_GATHER = 2,   // reduce-scatter + all-gather instead of all-reduce
    PDT_RCCL_KEEP_ONE_RANK = 4,    // issue the collective even when the group has one rank
    PDT_RCCL_COUNT_ON_DEVICE = 8,  // device-side counter of executed bucket collectives
};

struct PdtRcclBuckets {
    PdtComm* comm = nullptr;
    std::vector<void*> ptr;
    std::vector<void*> wire;  // bf16 staging buffer of an fp32 bucket, or nullptr: the bucket itself goes on the wire
    std::vector<int64_t> count;
    std::vector<int> dtype;
    std::vector<hipEvent_t> ready;
    std::vector<hipEvent_t> done;
    int flags = PDT_RCCL_AVERAGE;
    unsigned long long* device_count = nullptr;
    std::atomic<int> errors{0};
    std::atomic<int64_t> bytes_on_wire{0};
};

static bool bucket_collective(PdtRcclBuckets* s, int b) {
    PdtComm* c = s->comm;
    const bool average = (s->flags & PDT_RCCL_AVERAGE) != 0;
    const bool on_wire = s->wire[b] != nullptr;
    const int64_t n = s->count[b];
    void* buf = on_wire ? s->wire[b] : s->ptr[b];
    const int dt = on_wire ? 1 : s->dtype[b];
    if (on_wire) {
        pack_bf16_kernel<<<pack_grid(n / 8), 256, 0, c->stream>>>(static_cast<const float4*>(s->ptr[b]),
                                                                 static_cast<uint4*>(buf), n / 8,
                                                                 average ? 1.0f / static_cast<float>(c->world) : 1.0f);
        if (!hip_ok(hipGetLastError(), "pack_bf16_kernel")) return false;
    }
    // the pack kernel has already divided by the world size
    const ncclRedOp_t op = (average && !on_wire) ? ncclAvg : ncclSum;
    if (c->world > 1 || (s->flags & PDT_RCCL_KEEP_ONE_RANK)) {
        if (s->flags & PDT_RCCL_SCATTER_GATHER) {
            const size_t shard = static_cast<size_t>(n / c->world);
            char* mine = static_cast<char*>(buf) + static_cast<size_t>(c->rank) * shard * dtype_bytes(dt);
            if (!nccl_ok(ncclReduceScatter(buf, mine, shard, to_nccl_dtype(dt), op, c->comm, c->stream), "ncclReduceScatter") ||
                !nccl_ok(ncclAllGather(mine, buf, shard, to_nccl_dtype(dt), c->comm, c->stream), "ncclAllGather"))
                return false;
        } else if (!nccl_ok(ncclAllReduce(buf, buf, static_cast<size_t>(n), to_nccl_dtype(dt), op, c->comm, c->stream),
                            "ncclAllReduce")) {
            return false;
        }
        s->bytes_on_wire += n * static_cast<int64_t>(dtype_bytes(dt));
        if (c->world == 1 && rehearsal_channels() > 0) {
            // ring all-reduce: every byte crosses a link twice; wall_clock64 ticks at 100 MHz
            const double seconds = 2.0 * static_cast<double>(n) * dtype_bytes(dt) / (rehearsal_gbps() * 1e9);
            const long long ticks = static_cast<long long>(seconds * 1e8) + 1;
            if (rehearsal_heavy())
                resident_heavy_kernel<<<rehearsal_channels(), 512, rehearsal_lds_bytes(), c->stream>>>(ticks);
            else
                resident_kernel<<<rehearsal_channels(), 256, 0, c->stream>>>(ticks);
            if (!hip_ok(hipGetLastError(), "resident_kernel")) return false;
        }
    }
    if (on_wire) {
        unpack_bf16_kernel<<<pack_grid(n / 8), 256, 0, c->stream>>>(static_cast<const uint4*>(buf),
                                                                   static_cast<float4*>(s->ptr[b]), n / 8);
        if (!hip_ok(hipGetLastError(), "unpack_bf16_kernel")) return false;
    }
    if (s->device_count) {
        count_kernel<<<1, 1, 0, c->stream>>>(s->device_count);
        if (!hip_ok(hipGetLastError(), "count_kernel")) return false;
    }
    return true;
}

static void rccl_launch(void* user, int b, void* stream) {
    auto* s = static_cast<PdtRcclBuckets*>(user);
    auto step = static_cast<hipStream_t>(stream);
    // A one-rank group has nothing to reduce; the event chain is kept so the step is ordered as with N ranks.
    bool ok = hip_ok(hipEventRecord(s->ready[b], step), "hipEventRecord(ready)") &&
              hip_ok(hipStreamWaitEvent(s->comm->stream, s->ready[b], 0), "hipStreamWaitEvent(comm)") &&
              bucket_collective(s, b) && hip_ok(hipEventRecord(s->done[b], s->comm->stream), "hipEventRecord(done)");
    if (!ok) s->errors++;
}

static void rccl_wait(void* user, int b, void* stream) {
    auto* s = static_cast<PdtRcclBuckets*>(user);
    if (!hip_ok(hipStreamWaitEvent(static_cast<hipStream_t>(stream), s->done[b], 0), "hipStreamWaitEvent(step)")) s->errors++;
}

extern "C" {

// The returned pointer stays valid until the calling thread asks again.
const char* pdt_rccl_last_error() {
    thread_local std::string snapshot;
    std::lock_guard<std::mutex> g(g_err_mu);
    snapshot = g_last_error;
    return snapshot.c_str();
}

int pdt_rccl_unique_id_bytes() { return static_cast<int>(sizeof(ncclUniqueId)); }

int pdt_rccl_unique_id(char* out) {
    ncclUniqueId id;
    if (!nccl_ok(ncclGetUniqueId(&id), "ncclGetUniqueId")) return -1;
    std::memcpy(out, &id, sizeof(id));
    return 0;
}

// `priority`: 0 = the priority of ordinary streams (default), 1 = the device's highest, -1 = its lowest.
// The communication stream spends most of its life waiting (a barrier packet on a "bucket ready" event of the step's
// stream).  On a raised-priority hardware queue that wait is a priority inversion once the process has more streams
// than the packet processor serves at a time (step stream + copy stream + torch's own communicator stream + this one):
// the waiting queue keeps being served ahead of the queues that would produce the event, the device stops making
// progress and the host blocks in its next kernel launch.  Observed on MI355X with a one-rank group, ResNet-50 at batch
// 256 with a prefetching copy stream: hung in the second step at priority 1, fine at 0 and -1.
PdtComm* pdt_rccl_comm_create(const char* id_bytes, int rank, int world, int device, int priority) {
    if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return nullptr;
    auto* c = new PdtComm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    int lo = 0, hi = 0;  // numerically lowest value = highest priority
    if (!hip_ok(hipDeviceGetStreamPriorityRange(&lo, &hi), "hipDeviceGetStreamPriorityRange")) {
        delete c;
        return nullptr;
    }
    // (lo is the numerically greatest value = lowest priority; ordinary streams have priority 0)
    const int prio = priority > 0 ? hi : (priority < 0 ? lo : (lo < 0 ? lo : (hi > 0 ? hi : 0)));
    if (!hip_ok(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio),
                "hipStreamCreateWithPriority") ||
        !hip_ok(hipEventCreateWithFlags(&c->fence_in, hipEventDisableTiming), "hipEventCreate") ||
        !hip_ok(hipEventCreateWithFlags(&c->fence_out, hipEventDisableTiming), "hipEventCreate")) {
        delete c;
        return nullptr;
    }
    g_live_streams++;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    if (!nccl_ok(ncclCommInitRank(&c->comm, world, id, rank), "ncclCommInitRank")) {
        (void)hipStreamDestroy(c->stream);
        g_live_streams--;
        delete c;
        return nullptr;
    }
    return c;
}

void pdt_rccl_comm_destroy(PdtComm* c) {
    if (!c) return;
    if (!c->aborted.load()) {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (c->comm) (void)ncclCommDestroy(c->comm);
    }
    if (c->fence_in) (void)hipEventDestroy(c->fence_in);
    if (c->fence_out) (void)hipEventDestroy(c->fence_out);
    if (c->stream) {
        (void)hipStreamDestroy(c->stream);
        g_live_streams--;
    }
    delete c;
}

// HIP streams this library currently owns (one per live communicator): part of the process's stream census.
int pdt_rccl_live_streams() { return g_live_streams.load(); }

void* pdt_rccl_comm_stream(PdtComm* c) { return c->stream; }

// Number of ranks RCCL itself reports for the communicator (-1 on error).
int pdt_rccl_comm_count(PdtComm* c) {
    int n = -1;
    return nccl_ok(ncclCommCount(c->comm, &n), "ncclCommCount") ? n : -1;
}

// Asynchronous error state of the communicator: 0 = healthy, > 0 = the ncclResult_t RCCL reports (a peer died, a
// link failed, ...), -1 = the query itself failed.  Cheap enough to poll from a monitoring thread.
int pdt_rccl_comm_async_error(PdtComm* c) {
    if (c->aborted.load()) return static_cast<int>(ncclInternalError);
    ncclResult_t state = ncclSuccess;
    if (!nccl_ok(ncclCommGetAsyncError(c->comm, &state), "ncclCommGetAsyncError")) return -1;
    if (state == ncclSuccess || state == ncclInProgress) return 0;
    set_error(std::string("asynchronous RCCL error: ") + ncclGetErrorString(state));
    return static_cast<int>(state);
}

// Tear the communicator down without waiting for outstanding collectives (they can never finish once a peer is
// gone).  Kernels blocked inside a collective return, so the process can exit instead of hanging.
int pdt_rccl_comm_abort(PdtComm* c) {
    if (c->aborted.exchange(1)) return 0;
    return nccl_ok(ncclCommAbort(c->comm), "ncclCommAbort") ? 0 : -1;
}

// Collectives on an explicit stream (nullptr = the communicator's own stream).
int pdt_rccl_allreduce(PdtComm* c, void* buf, int64_t count, int dtype, int average, void* stream) {
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    return nccl_ok(ncclAllReduce(buf, buf, static_cast<size_t>(count), to_nccl_dtype(dtype),
                                 average ? ncclAvg : ncclSum, c->comm, s),
                   "ncclAllReduce")
               ? 0
               : -1;
}

int pdt_rccl_broadcast(PdtComm* c, void* buf, int64_t bytes, int root, void* stream) {
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    return nccl_ok(ncclBroadcast(buf, buf, static_cast<size_t>(bytes), ncclInt8, root, c->comm, s), "ncclBroadcast")
               ? 0
               : -1;
}

// Collectives on the communication stream, ordered after everything already submitted to `step_stream` and before
// everything submitted to it afterwards (two event edges, no host synchronisation; recordable into a hipGraph).
// kind: 0 broadcast of `count` bytes from `root`, 1 all-reduce sum, 2 all-reduce mean, 3 all-reduce max, 4 min.
int pdt_rccl_ordered(PdtComm* c, int kind, void* buf, int64_t count, int dtype, int root, void* step_stream) {
    auto step = static_cast<hipStream_t>(step_stream);
    if (!hip_ok(hipEventRecord(c->fence_in, step), "hipEventRecord(fence_in)") ||
        !hip_ok(hipStreamWaitEvent(c->stream, c->fence_in, 0), "hipStreamWaitEvent(comm)"))
        return -1;
    bool ok;
    if (kind == 0) {
        ok = nccl_ok(ncclBroadcast(buf, buf, static_cast<size_t>(count), ncclInt8, root, c->comm, c->stream), "ncclBroadcast");
    } else {
        const ncclRedOp_t op = kind == 1 ? ncclSum : (kind == 2 ? ncclAvg : (kind == 4 ? ncclMin : ncclMax));
        ok = nccl_ok(ncclAllReduce(buf, buf, static_cast<size_t>(count), to_nccl_dtype(dtype), op, c->comm, c->stream),
                     "ncclAllReduce");
    }
    if (!ok) return -1;
    if (!hip_ok(hipEventRecord(c->fence_out, c->stream), "hipEventRecord(fence_out)") ||
        !hip_ok(hipStreamWaitEvent(step, c->fence_out, 0), "hipStreamWaitEvent(step)"))
        return -1;
    return 0;
}

int pdt_rccl_stream_sync(PdtComm* c) { return hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize") ? 0 : -1; }

// Mean time in microseconds of one in-place all-reduce of `count` elements, measured with
// HIP events on the communication stream.  Used by the bucket-size sweep.
double pdt_rccl_time_allreduce(PdtComm* c, void* buf, int64_t count, int dtype, int warmup, int iters) {
    hipEvent_t t0, t1;
    if (!hip_ok(hipEventCreate(&t0), "hipEventCreate") || !hip_ok(hipEventCreate(&t1), "hipEventCreate")) return -1.0;
    for (int i = 0; i < warmup; ++i)
        if (pdt_rccl_allreduce(c, buf, count, dtype, 0, nullptr) != 0) return -1.0;
    (void)hipEventRecord(t0, c->stream);
    for (int i = 0; i < iters; ++i)
        if (pdt_rccl_allreduce(c, buf, count, dtype, 0, nullptr) != 0) return -1.0;
    (void)hipEventRecord(t1, c->stream);
    if (!hip_ok(hipEventSynchronize(t1), "hipEventSynchronize")) return -1.0;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, t0, t1);
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return iters > 0 ? 1e3 * ms / iters : 0.0;
}

// `wire[i]`: bf16 staging buffer of `count[i]` elements for an fp32 bucket that should cross the links as bf16, or
// nullptr.  `flags`: PDT_RCCL_* bits.  With PDT_RCCL_SCATTER_GATHER every count must be a multiple of the world
// size; with a wire buffer, a multiple of 8.
PdtRcclBuckets* pdt_rccl_buckets_create(PdtComm* c, int n, void** ptr, void** wire, const int64_t* count,
                                        const int32_t* dtype, int flags) {
    auto* s = new PdtRcclBuckets();
    s->comm = c;
    s->ptr.assign(ptr, ptr + n);
    if (wire) s->wire.assign(wire, wire + n); else s->wire.assign(n, nullptr);
    s->count.assign(count, count + n);
    s->dtype.assign(dtype, dtype + n);
    s->flags = flags;
    for (int i = 0; i < n; ++i) {
        const bool bad_wire = s->wire[i] && (s->dtype[i] != 0 || s->count[i] % 8 != 0);
        const bool bad_shard = (flags & PDT_RCCL_SCATTER_GATHER) && s->count[i] % c->world != 0;
        if (bad_wire || bad_shard) {
            set_error(bad_wire ? "a bf16 wire buffer needs an fp32 bucket whose length is a multiple of 8"
                               : "reduce-scatter needs bucket lengths that are multiples of the world size");
            delete s;
            return nullptr;
        }
    }
    if ((flags & PDT_RCCL_COUNT_ON_DEVICE) &&
        (!hip_ok(hipMalloc(&s->device_count, sizeof(unsigned long long)), "hipMalloc") ||
         !hip_ok(hipMemset(s->device_count, 0, sizeof(unsigned long long)), "hipMemset"))) {
        delete s;
        return nullptr;
    }
    s->ready.assign(n, nullptr);
    s->done.assign(n, nullptr);
    for (int i = 0; i < n; ++i) {
        if (!hip_ok(hipEventCreateWithFlags(&s->ready[i], hipEventDisableTiming), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&s->done[i], hipEventDisableTiming), "hipEventCreate")) {
            delete s;
            return nullptr;
        }
    }
    return s;
}

void pdt_rccl_buckets_destroy(PdtRcclBuckets* s) {
    if (!s) return;
    for (auto e : s->ready)
        if (e) (void)hipEventDestroy(e);
    for (auto e : s->done)
        if (e) (void)hipEventDestroy(e);
    if (s->device_count) (void)hipFree(s->device_count);
    delete s;
}

int pdt_rccl_buckets_errors(const PdtRcclBuckets* s) { return s->errors.load(); }

// Bytes handed to RCCL by bucket collectives so far (host-side count: a replayed hipGraph does not add to it).
int64_t pdt_rccl_buckets_wire_bytes(const PdtRcclBuckets* s) { return s->bytes_on_wire.load(); }

// Bucket collectives executed on the device so far (needs PDT_RCCL_COUNT_ON_DEVICE; waits for the communication
// stream).  -1 when the counter is off or the copy failed.
int64_t pdt_rccl_buckets_device_count(PdtRcclBuckets* s) {
    if (!s->device_count) return -1;
    unsigned long long v = 0;
    if (!hip_ok(hipStreamSynchronize(s->comm->stream), "hipStreamSynchronize") ||
        !hip_ok(hipMemcpy(&v, s->device_count, sizeof(v), hipMemcpyDeviceToHost), "hipMemcpy"))
        return -1;
    return static_cast<int64_t>(v);
}

pdt_bucket_fn pdt_rccl_launch_fn() { return &rccl_launch; }
pdt_bucket_fn pdt_rccl_wait_fn() { return &rccl_wait; }

}  // extern "C"
