// Stand-alone stress driver for the reducer core and the loopback fabric, meant to be built with
// -fsanitize=address / thread / undefined (host code only; `python csrc/build.py --sanitize thread`).
//
// W ranks are threads of this process.  Every rank plans buckets for the same random parameter list, owns a
// reducer over the loopback transport and runs `iters` "backward passes": the parameters are marked ready in a
// shuffled order from two helper threads per rank (autograd may fire gradient hooks from more than one thread),
// the pass is finalised, and every gradient slot is compared with the mean over ranks.  Some passes leave a few
// parameters unmarked (unused parameters are reduced as whatever their slots hold).  After the first pass the
// buckets are re-planned from rank 0's observed ready order, as the Python wrapper does.
//
// Exit status 0 = every check passed; sanitizer reports go to stderr and make the run fail through their own
// exit codes (-fno-sanitize-recover for UBSan).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <thread>
#include <vector>

#include "reducer_core.h"

extern "C" {
struct PdtLoopback;
struct PdtLoopbackRank;
PdtLoopback* pdt_loopback_create(int world);
void pdt_loopback_destroy(PdtLoopback* f);
PdtLoopbackRank* pdt_loopback_rank_create(PdtLoopback* f, int rank, int n_buckets, float** bucket_ptr,
                                          const int64_t* bucket_numel, int flags);
void pdt_loopback_rank_destroy(PdtLoopbackRank* c);
int64_t pdt_loopback_rank_collectives(const PdtLoopbackRank* c);
void pdt_loopback_allreduce(PdtLoopback* f, int rank, float* data, int64_t n, int flags);
pdt_bucket_fn pdt_loopback_launch_fn();
pdt_bucket_fn pdt_loopback_wait_fn();
}

namespace {

struct Plan {
    std::vector<int32_t> bucket_of;
    std::vector<int64_t> offset;
    std::vector<int64_t> bytes;
};

Plan make_plan(const std::vector<int64_t>& numel, const std::vector<int32_t>* order, int64_t last_cap) {
    const int n = static_cast<int>(numel.size());
    std::vector<int32_t> esize(n, 4);
    Plan p;
    p.bucket_of.resize(n);
    p.offset.resize(n);
    p.bytes.resize(n);
    const int nb = pdt_plan_buckets(numel.data(), esize.data(), n, order ? order->data() : nullptr, 512, 4096, last_cap, 64,
                                    p.bucket_of.data(), p.offset.data(), p.bytes.data());
    p.bytes.resize(nb);
    return p;
}

float grad_value(int rank, int iter, int param, int64_t i) {
    return static_cast<float>((rank + 1) * 0.25 + iter * 0.5 + param * 0.125 + static_cast<double>(i % 7));
}

std::atomic<int> g_failures{0};
std::vector<int32_t> g_rank0_order;  // written by rank 0 between two fabric-wide collectives

void rank_main(PdtLoopback* fabric, int rank, int world, const std::vector<int64_t>& numel, int iters, int wire16) {
    const int n = static_cast<int>(numel.size());
    std::mt19937 rng(1234);  // same sequence on every rank: the launch order must agree across ranks
    Plan plan = make_plan(numel, nullptr, 256);
    std::vector<int32_t> order;
    for (int it = 0; it < iters; ++it) {
        if (it == 1) plan = make_plan(numel, &order, 256);  // re-plan from the observed ready order
        const int nb = static_cast<int>(plan.bytes.size());
        std::vector<std::vector<float>> buckets(nb);
        std::vector<float*> ptrs(nb);
        std::vector<int64_t> lens(nb);
        for (int b = 0; b < nb; ++b) {
            buckets[b].assign(static_cast<size_t>(plan.bytes[b] / 4), 0.f);
            ptrs[b] = buckets[b].data();
            lens[b] = plan.bytes[b] / 4;
        }
        PdtLoopbackRank* ctx = pdt_loopback_rank_create(fabric, rank, nb, ptrs.data(), lens.data(), 1 | (wire16 ? 2 : 0));
        PdtReducer* red = pdt_reducer_create(n, plan.bucket_of.data(), nb, pdt_loopback_launch_fn(), pdt_loopback_wait_fn(), ctx);
        if (!ctx || !red) {
            std::fprintf(stderr, "rank %d: create failed\n", rank);
            g_failures++;
            return;
        }
        // ready order of this pass and the parameters left unmarked (same draw on every rank)
        std::vector<int> ready(n);
        std::iota(ready.begin(), ready.end(), 0);
        std::shuffle(ready.begin(), ready.end(), rng);
        const int unused = (it % 3 == 2) ? 2 : 0;
        const int marked = n - unused;
        for (int p = 0; p < n; ++p) {
            float* slot = buckets[plan.bucket_of[p]].data() + plan.offset[p] / 4;
            for (int64_t i = 0; i < numel[p]; ++i) slot[i] = grad_value(rank, it, p, i);
        }
        pdt_reducer_prepare(red);
        // two hook threads per rank mark interleaved halves of the order concurrently
        auto hook_thread = [&](int lane) {
            for (int k = lane; k < marked; k += 2)
                if (pdt_reducer_mark_ready(red, ready[k], nullptr) < 0) g_failures++;
        };
        std::thread h0(hook_thread, 0), h1(hook_thread, 1);
        h0.join();
        h1.join();
        const int got_unused = pdt_reducer_finalize(red, nullptr);
        if (got_unused != unused) {
            std::fprintf(stderr, "rank %d iter %d: %d unused, expected %d\n", rank, it, got_unused, unused);
            g_failures++;
        }
        if (pdt_loopback_rank_collectives(ctx) != nb || pdt_reducer_launch_count(red) != nb || pdt_reducer_pending_buckets(red) != 0) {
            std::fprintf(stderr, "rank %d iter %d: not every bucket was launched exactly once\n", rank, it);
            g_failures++;
        }
        for (int p = 0; p < n; ++p) {
            const float* slot = buckets[plan.bucket_of[p]].data() + plan.offset[p] / 4;
            for (int64_t i = 0; i < numel[p]; ++i) {
                double mean = 0;
                for (int r = 0; r < world; ++r) mean += grad_value(r, it, p, i);
                mean /= world;
                const double tol = wire16 ? 2e-2 * std::fabs(mean) + 1e-3 : 1e-5 * std::fabs(mean) + 1e-6;
                if (std::fabs(slot[i] - mean) > tol) {
                    if (g_failures++ < 5)
                        std::fprintf(stderr, "rank %d iter %d param %d[%lld]: %g != %g\n", rank, it, p,
                                     static_cast<long long>(i), slot[i], mean);
                    break;
                }
            }
        }
        // every rank re-plans from rank 0's observed order (the ranks' own orders differ: their hook threads
        // interleave differently), exchanged between two collectives that act as barriers
        order.resize(n);
        pdt_reducer_ready_order(red, order.data());
        std::vector<int32_t> sorted(order);
        std::sort(sorted.begin(), sorted.end());
        for (int p = 0; p < n; ++p)
            if (sorted[p] != p) g_failures++;  // the observed order must be a permutation
        float token = 0.f;
        pdt_loopback_allreduce(fabric, rank, &token, 1, 0);
        if (rank == 0) g_rank0_order = order;
        pdt_loopback_allreduce(fabric, rank, &token, 1, 0);
        order = g_rank0_order;
        pdt_reducer_destroy(red);
        pdt_loopback_rank_destroy(ctx);
    }
}

}  // namespace

int main(int argc, char** argv) {
    const int world = argc > 1 ? std::atoi(argv[1]) : 4;
    const int iters = argc > 2 ? std::atoi(argv[2]) : 12;
    std::mt19937 rng(7);
    std::vector<int64_t> numel(37);
    for (auto& v : numel) v = 1 + static_cast<int64_t>(rng() % 700);
    for (int wire16 = 0; wire16 < 2; ++wire16) {
        PdtLoopback* fabric = pdt_loopback_create(world);
        std::vector<std::thread> ranks;
        for (int r = 0; r < world; ++r) ranks.emplace_back(rank_main, fabric, r, world, std::cref(numel), iters, wire16);
        for (auto& t : ranks) t.join();
        pdt_loopback_destroy(fabric);
    }
    if (g_failures.load()) {
        std::fprintf(stderr, "reducer_stress: %d failure(s)\n", g_failures.load());
        return 1;
    }
    std::printf("reducer_stress: ok (world %d, %d passes x 2 wire types)\n", world, iters);
    return 0;
}
