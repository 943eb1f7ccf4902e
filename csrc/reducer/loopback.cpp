// In-process loopback fabric: the "ranks" are threads of one process that share
// host memory.  It exists so the reducer core's bucketing / ordering / finalise
// logic can be exercised with world_size > 1 on a machine without GPUs or any
// network backend.  The all-reduce uses the same two-phase direct algorithm a
// fully connected xGMI node favours (every rank reduces one shard read from all
// peers, then every rank gathers all shards), so shard arithmetic is covered too.
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

#include "reducer_core.h"

namespace {

class Barrier {
   public:
    explicit Barrier(int n) : n_(n) {}
    void arrive_and_wait() {
        std::unique_lock<std::mutex> lk(mu_);
        const int64_t gen = gen_;
        if (++count_ == n_) {
            count_ = 0;
            ++gen_;
            cv_.notify_all();
        } else {
            cv_.wait(lk, [&] { return gen_ != gen; });
        }
    }

   private:
    int n_;
    int count_ = 0;
    int64_t gen_ = 0;
    std::mutex mu_;
    std::condition_variable cv_;
};

// Round to the nearest bf16 (ties to even) and return it as a float: what a value looks like after crossing the
// wire in bf16.
inline float bf16_round(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) != 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    std::memcpy(&x, &u, 4);
    return x;
}

}  // namespace

struct PdtLoopback {
    int world;
    Barrier barrier;
    std::vector<float*> ptr;  // buffer published by each rank for the collective in flight
    std::vector<float> scratch;
    std::mutex scratch_mu;
    explicit PdtLoopback(int w) : world(w), barrier(w), ptr(w, nullptr) {}
};

struct PdtLoopbackRank {
    PdtLoopback* fabric;
    int rank;
    int flags;
    std::vector<float*> bucket_ptr;
    std::vector<int64_t> bucket_numel;
    int64_t collectives = 0;
};

// flags: bit 0 = mean instead of sum; bit 1 = bf16 on the wire (every rank's contribution is divided by the world
// size and rounded to bf16 first, the sum is rounded to bf16 again - the arithmetic of the RCCL transport's
// pack / bf16 collective / unpack sequence)
static void loopback_allreduce(PdtLoopback* f, int rank, float* data, int64_t n, int flags) {
    const int w = f->world;
    const int average = flags & 1;
    const bool wire16 = (flags & 2) != 0;
    f->ptr[rank] = data;
    f->barrier.arrive_and_wait();  // all pointers published
    const int64_t lo = n * rank / w, hi = n * (rank + 1) / w;
    std::vector<float> acc(static_cast<size_t>(hi - lo));
    for (int64_t i = lo; i < hi; ++i) {
        float s = 0.f;
        if (wire16) {
            const float pre = average ? 1.0f / static_cast<float>(w) : 1.0f;
            for (int r = 0; r < w; ++r) s += bf16_round(f->ptr[r][i] * pre);
            acc[i - lo] = bf16_round(s);
            continue;
        }
        for (int r = 0; r < w; ++r) s += f->ptr[r][i];  // fixed rank order: identical result on every rank
        acc[i - lo] = average ? s / static_cast<float>(w) : s;
    }
    f->barrier.arrive_and_wait();  // every shard reduced before anyone overwrites inputs
    for (int r = 0; r < w; ++r) std::copy(acc.begin(), acc.end(), f->ptr[r] + lo);
    f->barrier.arrive_and_wait();  // all shards scattered
}

static void loopback_launch(void* user, int bucket, void*) {
    auto* c = static_cast<PdtLoopbackRank*>(user);
    loopback_allreduce(c->fabric, c->rank, c->bucket_ptr[bucket], c->bucket_numel[bucket], c->flags);
    c->collectives++;
}

static void loopback_wait(void*, int, void*) {}  // the collective is synchronous

extern "C" {

PdtLoopback* pdt_loopback_create(int world) { return world > 0 ? new PdtLoopback(world) : nullptr; }
void pdt_loopback_destroy(PdtLoopback* f) { delete f; }

PdtLoopbackRank* pdt_loopback_rank_create(PdtLoopback* f, int rank, int n_buckets, float** bucket_ptr,
                                          const int64_t* bucket_numel, int flags) {
    if (!f || rank < 0 || rank >= f->world) return nullptr;
    auto* c = new PdtLoopbackRank();
    c->fabric = f;
    c->rank = rank;
    c->flags = flags;
    c->bucket_ptr.assign(bucket_ptr, bucket_ptr + n_buckets);
    c->bucket_numel.assign(bucket_numel, bucket_numel + n_buckets);
    return c;
}
void pdt_loopback_rank_destroy(PdtLoopbackRank* c) { delete c; }
int64_t pdt_loopback_rank_collectives(const PdtLoopbackRank* c) { return c->collectives; }

void pdt_loopback_allreduce(PdtLoopback* f, int rank, float* data, int64_t n, int flags) {
    loopback_allreduce(f, rank, data, n, flags);
}

pdt_bucket_fn pdt_loopback_launch_fn() { return &loopback_launch; }
pdt_bucket_fn pdt_loopback_wait_fn() { return &loopback_wait; }

}  // extern "C"
