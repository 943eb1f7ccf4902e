// Gradient-bucket reducer core: bucket planning, grad-ready bookkeeping, in-order
// launch of bucket collectives and end-of-backward finalisation.
//
// The core is transport-agnostic.  A transport supplies two C callbacks:
//   launch(user, bucket, stream)  -- start an asynchronous all-reduce of one bucket whose gradients
//                                    were produced on `stream`
//   wait(user, bucket, stream)    -- make work submitted to `stream` afterwards wait for it
// `stream` is the HIP stream that is current on the calling thread at the moment of the call (nullptr
// on the CPU).  It is passed with every call instead of being stored, so a step that runs on a side
// stream - a hipGraph capture, a `torch.cuda.stream(...)` block - orders its collectives against the
// stream its kernels really run on.
// Transports in this repository: RCCL on a dedicated HIP stream
// (reducer_rccl.cpp), an in-process loopback fabric whose ranks are threads
// (loopback.cpp, used by CPU tests) and a Python callback that forwards to
// torch.distributed (gloo).
//
// Role parity: c10d::Reducer as used by DistributedDataParallel at
// /root/reference/src/main.py:53,78 (bucketed all-reduce overlapped with
// backward).  This is an independent design: gradients live permanently inside
// the flat bucket buffers, so there is no copy-in/copy-out stage.
#pragma once
#include <cstdint>

extern "C" {

typedef void (*pdt_bucket_fn)(void* user, int bucket, void* stream);

struct PdtReducer;

// Assign parameters to buckets.
//   numel[i], elem_size[i]: size of parameter i (registration order)
//   order[i]: launch-priority order of parameters (order[0] becomes ready first);
//             pass nullptr for "reverse registration order".
//   first_cap_bytes: capacity of bucket 0 (small, so communication starts early)
//   cap_bytes: capacity of every later bucket
//   last_cap_bytes: when > 0, the parameters that become ready last are split off into a final
//             bucket of at most this size: the final collective is the only one no backward kernel
//             is left to hide, so it should be short
//   align_bytes: every parameter slot starts at a multiple of this
// Outputs (arrays of n): bucket_of[i], offset_bytes[i] (offset inside the bucket).
// bucket_bytes must have room for n entries.  A bucket never mixes element sizes.
// Returns the number of buckets.
int pdt_plan_buckets(const int64_t* numel, const int32_t* elem_size, int n, const int32_t* order,
                     int64_t first_cap_bytes, int64_t cap_bytes, int64_t last_cap_bytes, int64_t align_bytes,
                     int32_t* bucket_of, int64_t* offset_bytes, int64_t* bucket_bytes);

PdtReducer* pdt_reducer_create(int n_params, const int32_t* bucket_of, int n_buckets,
                               pdt_bucket_fn launch, pdt_bucket_fn wait, void* user);
void pdt_reducer_destroy(PdtReducer* r);

// Called at the start of every backward pass that will synchronise.
void pdt_reducer_prepare(PdtReducer* r);

// Parameter `param` has its final gradient for this iteration.  Launches every
// bucket that became launchable (buckets are always launched in index order so
// all ranks issue the same collective sequence).  Returns the number of buckets
// launched by this call, or -1 if the parameter was already marked.
int pdt_reducer_mark_ready(PdtReducer* r, int param, void* stream);

// End of backward: parameters that produced no gradient are treated as ready
// (their slots hold zeros), remaining buckets are launched, then wait() is called
// for every bucket in order.  Returns the number of parameters that were never
// marked ready in this iteration.
int pdt_reducer_finalize(PdtReducer* r, void* stream);

// Order in which parameters became ready during the last finalised iteration
// (n_params entries; unmarked parameters follow in reverse registration order).
void pdt_reducer_ready_order(const PdtReducer* r, int32_t* out);

// Counters for tests/diagnostics.
int64_t pdt_reducer_launch_count(const PdtReducer* r);
int pdt_reducer_pending_buckets(const PdtReducer* r);

}  // extern "C"
