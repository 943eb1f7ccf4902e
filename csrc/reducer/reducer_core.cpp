// Implementation of the transport-agnostic reducer core (see reducer_core.h).
#include "reducer_core.h"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

inline int64_t round_up(int64_t x, int64_t a) { return a > 1 ? (x + a - 1) / a * a : x; }

}  // namespace

struct PdtReducer {
    int n_params = 0;
    int n_buckets = 0;
    std::vector<int32_t> bucket_of;
    std::vector<int32_t> bucket_size;     // parameters per bucket
    std::vector<int32_t> bucket_pending;  // parameters not yet ready, this iteration
    std::vector<uint8_t> marked;
    std::vector<uint8_t> launched;
    std::vector<int32_t> ready_order;  // order of arrival in the current iteration
    std::vector<int32_t> last_order;   // order of arrival in the last finalised iteration
    int next_launch = 0;               // lowest bucket index not launched yet
    int64_t launches = 0;
    pdt_bucket_fn launch = nullptr;
    pdt_bucket_fn wait = nullptr;
    void* user = nullptr;
    std::mutex mu;  // autograd may call hooks from several threads

    // Launch every consecutive ready bucket starting at next_launch.
    int drain(void* stream) {
        int n = 0;
        while (next_launch < n_buckets && bucket_pending[next_launch] == 0) {
            launch(user, next_launch, stream);
            launched[next_launch] = 1;
            ++next_launch;
            ++launches;
            ++n;
        }
        return n;
    }

    void reset() {
        bucket_pending = bucket_size;
        std::fill(marked.begin(), marked.end(), 0);
        std::fill(launched.begin(), launched.end(), 0);
        ready_order.clear();
        next_launch = 0;
    }
};

extern "C" {

int pdt_plan_buckets(const int64_t* numel, const int32_t* elem_size, int n, const int32_t* order,
                     int64_t first_cap_bytes, int64_t cap_bytes, int64_t last_cap_bytes, int64_t align_bytes,
                     int32_t* bucket_of, int64_t* offset_bytes, int64_t* bucket_bytes) {
    if (n <= 0) return 0;
    if (cap_bytes <= 0) cap_bytes = INT64_MAX;
    if (first_cap_bytes <= 0) first_cap_bytes = cap_bytes;
    auto param_at = [&](int k) { return order ? order[k] : n - 1 - k; };
    // Parameters from position `tail_from` of the visiting order on form the final bucket: the longest suffix of
    // one element size that fits the final capacity (never the whole list, never an empty suffix).
    int tail_from = n;
    if (last_cap_bytes > 0 && n > 1) {
        int64_t fill = 0;
        const int32_t es = elem_size[param_at(n - 1)];
        for (int k = n - 1; k >= 1; --k) {
            const int p = param_at(k);
            const int64_t bytes = round_up(numel[p] * elem_size[p], align_bytes);
            if (elem_size[p] != es || (fill > 0 && fill + bytes > last_cap_bytes)) break;
            if (fill == 0 && bytes > last_cap_bytes) break;  // a single large tensor: leave it where it falls
            fill += bytes;
            tail_from = k;
        }
    }
    int bucket = -1;
    int64_t fill = 0;
    int32_t cur_elem = 0;
    for (int k = 0; k < n; ++k) {
        const int p = param_at(k);
        const int64_t bytes = numel[p] * elem_size[p];
        const int64_t cap = (bucket <= 0) ? first_cap_bytes : cap_bytes;
        const int64_t start = round_up(fill, align_bytes);
        const bool fits = bucket >= 0 && k != tail_from && cur_elem == elem_size[p] &&
                          (fill == 0 || k > tail_from || start + bytes <= cap);
        if (!fits) {
            if (bucket >= 0) bucket_bytes[bucket] = round_up(fill, align_bytes);
            ++bucket;
            cur_elem = elem_size[p];
            bucket_of[p] = bucket;
            offset_bytes[p] = 0;
            fill = bytes;
        } else {
            bucket_of[p] = bucket;
            offset_bytes[p] = start;
            fill = start + bytes;
        }
    }
    bucket_bytes[bucket] = round_up(fill, align_bytes);
    return bucket + 1;
}

PdtReducer* pdt_reducer_create(int n_params, const int32_t* bucket_of, int n_buckets,
                               pdt_bucket_fn launch, pdt_bucket_fn wait, void* user) {
    if (n_params < 0 || n_buckets < 0 || !launch || !wait) return nullptr;
    for (int i = 0; i < n_params; ++i)
        if (bucket_of[i] < 0 || bucket_of[i] >= n_buckets) return nullptr;
    auto* r = new PdtReducer();
    r->n_params = n_params;
    r->n_buckets = n_buckets;
    r->bucket_of.assign(bucket_of, bucket_of + n_params);
    r->bucket_size.assign(n_buckets, 0);
    for (int i = 0; i < n_params; ++i) r->bucket_size[bucket_of[i]]++;
    r->marked.assign(n_params, 0);
    r->launched.assign(n_buckets, 0);
    r->launch = launch;
    r->wait = wait;
    r->user = user;
    r->last_order.resize(n_params);
    for (int i = 0; i < n_params; ++i) r->last_order[i] = n_params - 1 - i;
    r->reset();
    return r;
}

void pdt_reducer_destroy(PdtReducer* r) { delete r; }

void pdt_reducer_prepare(PdtReducer* r) {
    std::lock_guard<std::mutex> g(r->mu);
    r->reset();
}

int pdt_reducer_mark_ready(PdtReducer* r, int param, void* stream) {
    std::lock_guard<std::mutex> g(r->mu);
    if (param < 0 || param >= r->n_params || r->marked[param]) return -1;
    r->marked[param] = 1;
    r->ready_order.push_back(param);
    r->bucket_pending[r->bucket_of[param]]--;
    return r->drain(stream);
}

int pdt_reducer_finalize(PdtReducer* r, void* stream) {
    std::lock_guard<std::mutex> g(r->mu);
    int unused = 0;
    for (int p = r->n_params - 1; p >= 0; --p) {
        if (!r->marked[p]) {
            r->marked[p] = 1;
            r->ready_order.push_back(p);
            r->bucket_pending[r->bucket_of[p]]--;
            ++unused;
        }
    }
    r->drain(stream);
    for (int b = 0; b < r->n_buckets; ++b) r->wait(r->user, b, stream);
    r->last_order = r->ready_order;
    return unused;
}

void pdt_reducer_ready_order(const PdtReducer* r, int32_t* out) {
    std::memcpy(out, r->last_order.data(), sizeof(int32_t) * r->n_params);
}

int64_t pdt_reducer_launch_count(const PdtReducer* r) { return r->launches; }

int pdt_reducer_pending_buckets(const PdtReducer* r) { return r->n_buckets - r->next_launch; }

}  // extern "C"
