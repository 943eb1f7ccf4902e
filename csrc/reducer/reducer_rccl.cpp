// RCCL transport for the gradient-bucket reducer.
//
// One communicator per process (one process per GPU).  All collectives are issued on a
// dedicated, high-priority HIP stream owned by this file; ordering against the compute
// stream is expressed only with HIP events, so backward kernels and bucket all-reduces
// overlap:
//
//   compute stream:  ... wgrad(bucket k complete) --record(ready[k])--> ... more backward ...
//   comm stream:     wait(ready[k]) -> ncclAllReduce(bucket k) -> record(done[k])
//   compute stream:  (end of backward) wait(done[0..n)) -> optimizer kernel
//
// The unique id is created on rank 0 (pdt_rccl_unique_id) and distributed by the caller
// through whatever key-value store it already has (torch's TCPStore in this repository).
//
// Role parity: c10d ProcessGroupNCCL + Reducer as reached from
// /root/reference/src/main.py:39-41,53,78.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "reducer_core.h"

namespace {

thread_local std::string g_last_error;

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

bool nccl_ok(ncclResult_t e, const char* what) {
    if (e == ncclSuccess) return true;
    g_last_error = std::string(what) + ": " + ncclGetErrorString(e);
    return false;
}

ncclDataType_t to_nccl_dtype(int dtype) {
    switch (dtype) {
        case 0: return ncclFloat32;
        case 1: return ncclBfloat16;
        case 2: return ncclFloat16;
        default: return ncclFloat32;
    }
}

}  // namespace

struct PdtComm {
    ncclComm_t comm = nullptr;
    int rank = 0;
    int world = 1;
    int device = 0;
    hipStream_t stream = nullptr;  // dedicated communication stream
};

struct PdtRcclBuckets {
    PdtComm* comm = nullptr;
    std::vector<void*> ptr;
    std::vector<int64_t> count;
    std::vector<int> dtype;
    std::vector<hipEvent_t> ready;
    std::vector<hipEvent_t> done;
    hipStream_t compute = nullptr;
    ncclRedOp_t op = ncclAvg;
    int errors = 0;
};

static void rccl_launch(void* user, int b) {
    auto* s = static_cast<PdtRcclBuckets*>(user);
    // A one-rank group has nothing to reduce (sum and mean over one rank are the identity); RCCL would still run
    // a copy kernel over the whole bucket.  The event chain is kept so the step is ordered as with N ranks.
    bool ok = hip_ok(hipEventRecord(s->ready[b], s->compute), "hipEventRecord(ready)") &&
              hip_ok(hipStreamWaitEvent(s->comm->stream, s->ready[b], 0), "hipStreamWaitEvent(comm)") &&
              (s->comm->world == 1 ||
               nccl_ok(ncclAllReduce(s->ptr[b], s->ptr[b], static_cast<size_t>(s->count[b]),
                                     to_nccl_dtype(s->dtype[b]), s->op, s->comm->comm, s->comm->stream),
                       "ncclAllReduce")) &&
              hip_ok(hipEventRecord(s->done[b], s->comm->stream), "hipEventRecord(done)");
    if (!ok) s->errors++;
}

static void rccl_wait(void* user, int b) {
    auto* s = static_cast<PdtRcclBuckets*>(user);
    if (!hip_ok(hipStreamWaitEvent(s->compute, s->done[b], 0), "hipStreamWaitEvent(compute)")) s->errors++;
}

extern "C" {

const char* pdt_rccl_last_error() { return g_last_error.c_str(); }

int pdt_rccl_unique_id_bytes() { return static_cast<int>(sizeof(ncclUniqueId)); }

int pdt_rccl_unique_id(char* out) {
    ncclUniqueId id;
    if (!nccl_ok(ncclGetUniqueId(&id), "ncclGetUniqueId")) return -1;
    std::memcpy(out, &id, sizeof(id));
    return 0;
}

PdtComm* pdt_rccl_comm_create(const char* id_bytes, int rank, int world, int device, int high_priority) {
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
    if (!hip_ok(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, high_priority ? hi : lo),
                "hipStreamCreateWithPriority")) {
        delete c;
        return nullptr;
    }
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    if (!nccl_ok(ncclCommInitRank(&c->comm, world, id, rank), "ncclCommInitRank")) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return nullptr;
    }
    return c;
}

void pdt_rccl_comm_destroy(PdtComm* c) {
    if (!c) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void* pdt_rccl_comm_stream(PdtComm* c) { return c->stream; }

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

PdtRcclBuckets* pdt_rccl_buckets_create(PdtComm* c, int n, void** ptr, const int64_t* count, const int32_t* dtype,
                                        int average, void* compute_stream) {
    auto* s = new PdtRcclBuckets();
    s->comm = c;
    s->ptr.assign(ptr, ptr + n);
    s->count.assign(count, count + n);
    s->dtype.assign(dtype, dtype + n);
    s->op = average ? ncclAvg : ncclSum;
    s->compute = static_cast<hipStream_t>(compute_stream);
    s->ready.resize(n);
    s->done.resize(n);
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
    for (auto e : s->ready) (void)hipEventDestroy(e);
    for (auto e : s->done) (void)hipEventDestroy(e);
    delete s;
}

void pdt_rccl_buckets_set_compute_stream(PdtRcclBuckets* s, void* stream) { s->compute = static_cast<hipStream_t>(stream); }
int pdt_rccl_buckets_errors(const PdtRcclBuckets* s) { return s->errors; }

pdt_bucket_fn pdt_rccl_launch_fn() { return &rccl_launch; }
pdt_bucket_fn pdt_rccl_wait_fn() { return &rccl_wait; }

}  // extern "C"
