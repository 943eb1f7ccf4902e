// Fused softmax cross-entropy (mean over rows) for bf16 or fp32 logits.
//
// Forward: one pass per row computes max, log-sum-exp and the row loss (online softmax, so the
//          row is read once even when it does not fit in registers); the row's lse is kept.
// Backward: dlogits = (exp(x - lse) - onehot) * (g / rows), written in the logits' dtype in one
//          pass; probabilities are never materialised.
// Rows of up to 1024 classes are handled by one wave each (4 rows per workgroup); longer rows
// (language-model vocabularies) by one 256-thread workgroup each.
//
// Role parity: nn.CrossEntropyLoss (log_softmax + nll_loss, mean reduction) at
// /root/reference/src/main.py:62,76,78.
#include "common.h"

template <typename T>
__device__ __forceinline__ float ld(const T* p, long long i);
template <>
__device__ __forceinline__ float ld<bf16>(const bf16* p, long long i) { return bf16_to_f32(p[i]); }
template <>
__device__ __forceinline__ float ld<float>(const float* p, long long i) { return p[i]; }

template <typename T>
__device__ __forceinline__ void st(T* p, long long i, float v);
template <>
__device__ __forceinline__ void st<bf16>(bf16* p, long long i, float v) { p[i] = f32_to_bf16(v); }
template <>
__device__ __forceinline__ void st<float>(float* p, long long i, float v) { p[i] = v; }

// Running-max sentinel: finite, so exp(m - m) is 1 (not NaN) when two empty partials are merged and
// the code stays correct under -ffast-math (which assumes no infinities).
#define CE_NEG (-3.0e38f)

__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
}

// ---- wave-per-row forward
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_wave_kernel(const T* __restrict__ logits, const long long* __restrict__ labels,
                                                         float* __restrict__ row_loss, float* __restrict__ lse, int rows,
                                                         int classes, long long ld_row) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* x = logits + row * ld_row;
    float m = CE_NEG, s = 0.f;
    for (int i = lane; i < classes; i += 64) {
        const float v = ld<T>(x, i);
        const float mn = fmaxf(m, v);
        s = s * __expf(m - mn) + __expf(v - mn);
        m = mn;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        online_merge(m, s, m2, s2);
    }
    if (lane == 0) {
        const float l = m + __logf(s);
        const long long y = labels[row];
        lse[row] = l;
        row_loss[row] = (y >= 0 && y < classes) ? l - ld<T>(x, y) : 0.f;
    }
}

// ---- workgroup-per-row forward (long rows)
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_block_kernel(const T* __restrict__ logits, const long long* __restrict__ labels,
                                                          float* __restrict__ row_loss, float* __restrict__ lse,
                                                          int classes, long long ld_row) {
    __shared__ float sm[4], ss[4];
    const int row = blockIdx.x;
    const T* x = logits + row * ld_row;
    float m = CE_NEG, s = 0.f;
    for (int i = threadIdx.x; i < classes; i += 256) {
        const float v = ld<T>(x, i);
        const float mn = fmaxf(m, v);
        s = s * __expf(m - mn) + __expf(v - mn);
        m = mn;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        online_merge(m, s, m2, s2);
    }
    if ((threadIdx.x & 63) == 0) {
        sm[threadIdx.x >> 6] = m;
        ss[threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float M = sm[0], S = ss[0];
        for (int w = 1; w < 4; ++w)
            online_merge(M, S, sm[w], ss[w]);
        const float l = M + __logf(S);
        const long long y = labels[row];
        lse[row] = l;
        row_loss[row] = (y >= 0 && y < classes) ? l - ld<T>(x, y) : 0.f;
    }
}

// ---- backward: grid-stride over all elements of the [rows, classes] matrix
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, const long long* __restrict__ labels,
                                                    const float* __restrict__ lse, const float* __restrict__ gout,
                                                    T* __restrict__ dlogits, int rows, int classes, long long ld_row,
                                                    long long ld_drow, int pad_to) {
    const float g = gout[0] / rows;
    // one workgroup walks whole rows so lse/label are loaded once per row
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const float l = lse[row];
        const long long y = labels[row];
        const T* x = logits + row * ld_row;
        T* d = dlogits + row * ld_drow;
        const bool valid = y >= 0 && y < classes;
        for (int i = threadIdx.x; i < pad_to; i += 256) {
            float v = 0.f;
            if (i < classes && valid) v = (__expf(ld<T>(x, i) - l) - (i == y ? 1.f : 0.f)) * g;
            st<T>(d, i, v);
        }
    }
}

// deterministic mean of a float vector -> out[0]  (single workgroup)
__global__ __launch_bounds__(1024) void mean_kernel(const float* __restrict__ x, float* __restrict__ out, long long n,
                                                   float scale) {
    __shared__ float red[16];
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) s += x[i];
    const float t = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = t * scale;
}

// dtype: 0 = fp32 logits, 1 = bf16 logits.  row_loss/lse: [rows] fp32.  loss: [1] fp32.
PDT_API int pdt_softmax_ce_fwd(const void* logits, const long long* labels, float* row_loss, float* lse, float* loss,
                               int rows, int classes, long long ld_row, int dtype, void* stream) {
    if (rows <= 0 || classes <= 0 || ld_row < classes) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (classes <= 1024) {
        const dim3 g(pdt_cdiv(rows, 4)), b(256);
        if (dtype == 1) hipLaunchKernelGGL(ce_fwd_wave_kernel<bf16>, g, b, 0, s, static_cast<const bf16*>(logits), labels, row_loss, lse, rows, classes, ld_row);
        else hipLaunchKernelGGL(ce_fwd_wave_kernel<float>, g, b, 0, s, static_cast<const float*>(logits), labels, row_loss, lse, rows, classes, ld_row);
    } else {
        const dim3 g(rows), b(256);
        if (dtype == 1) hipLaunchKernelGGL(ce_fwd_block_kernel<bf16>, g, b, 0, s, static_cast<const bf16*>(logits), labels, row_loss, lse, classes, ld_row);
        else hipLaunchKernelGGL(ce_fwd_block_kernel<float>, g, b, 0, s, static_cast<const float*>(logits), labels, row_loss, lse, classes, ld_row);
    }
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, s, row_loss, loss, static_cast<long long>(rows), 1.f / rows);
    return pdt_launch_status();
}

// dlogits has row stride ld_drow; columns [classes, pad_to) are written as zeros (pad_to <= ld_drow).
PDT_API int pdt_softmax_ce_bwd(const void* logits, const long long* labels, const float* lse, const float* gout,
                               void* dlogits, int rows, int classes, long long ld_row, long long ld_drow, int pad_to,
                               int dtype, void* stream) {
    if (rows <= 0 || classes <= 0 || ld_row < classes || pad_to < classes || ld_drow < pad_to) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 g(rows < 8192 ? rows : 8192), b(256);
    if (dtype == 1) hipLaunchKernelGGL(ce_bwd_kernel<bf16>, g, b, 0, s, static_cast<const bf16*>(logits), labels, lse, gout, static_cast<bf16*>(dlogits), rows, classes, ld_row, ld_drow, pad_to);
    else hipLaunchKernelGGL(ce_bwd_kernel<float>, g, b, 0, s, static_cast<const float*>(logits), labels, lse, gout, static_cast<float*>(dlogits), rows, classes, ld_row, ld_drow, pad_to);
    return pdt_launch_status();
}
