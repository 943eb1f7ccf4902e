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

// ---- long bf16 rows, forward and gradient in one pass: a 512-thread workgroup holds its row in registers
// (NV 16-byte vectors per thread), so the logits are read from memory once and the gradient
// (softmax - onehot) / rows is written straight away; the backward pass only has to scale it by the
// incoming gradient, which for a loss that is the root of the graph is 1.
template <int NV>
__global__ __launch_bounds__(512) void ce_fused_row_kernel(const bf16* __restrict__ logits, const long long* __restrict__ labels,
                                                          float* __restrict__ row_loss, bf16* __restrict__ dlogits,
                                                          int rows, int classes, long long ld_row, long long ld_drow,
                                                          int pad_to, float inv_rows) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    const bf16* x = logits + row * ld_row;
    const int nvec = (classes + 7) >> 3;  // vectors that hold at least one class (ld_row covers them)
    bf16x8 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = threadIdx.x + i * 512;
        if (vi < nvec) v[i] = *reinterpret_cast<const bf16x8*>(x + vi * 8);
    }
    const long long y = labels[row];
    const bool valid = y >= 0 && y < classes;
    float m = CE_NEG, xy = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = threadIdx.x + i * 512;
        if (vi < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = vi * 8 + e;
                const float f = bf16_to_f32(v[i][e]);
                if (c < classes) m = fmaxf(m, f);
                if (c == y) xy = f;
            }
        }
    }
    m = block_max(m, red);
    const float ml2 = m * 1.4426950408889634f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = threadIdx.x + i * 512;
        if (vi < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (vi * 8 + e < classes) s += __builtin_amdgcn_exp2f(bf16_to_f32(v[i][e]) * 1.4426950408889634f - ml2);
        }
    }
    s = block_sum(s, red);
    xy = block_sum(xy, red);  // one thread holds the label's logit, the others 0
    const float l = m + __logf(s);
    if (threadIdx.x == 0) row_loss[row] = valid ? l - xy : 0.f;
    if (dlogits == nullptr) return;
    // (dlogits may be the logits buffer itself: every thread has its part of the row in registers by now, and the
    // block reductions above are barriers between the last load and the first store)
    const float g = valid ? inv_rows : 0.f;
    const float ll2 = l * 1.4426950408889634f;
    bf16* d = dlogits + row * ld_drow;
    const int pvec = pad_to >> 3;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = threadIdx.x + i * 512;
        if (vi < pvec) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = vi * 8 + e;
                float r = 0.f;
                if (vi < nvec && c < classes)
                    r = (__builtin_amdgcn_exp2f(bf16_to_f32(v[i][e]) * 1.4426950408889634f - ll2) - (c == y ? 1.f : 0.f)) * g;
                o[e] = f32_to_bf16(r);
            }
            *reinterpret_cast<bf16x8*>(d + vi * 8) = o;
        }
    }
}

// x *= g[0] unless g[0] == 1 (the gradient of a loss that is the root of the graph)
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(bf16* __restrict__ x, const float* __restrict__ g, long long nvec) {
    const float gv = g[0];
    if (gv == 1.f) return;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nvec; i += gridDim.x * 256ll) {
        bf16x8 v = *reinterpret_cast<bf16x8*>(x + i * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f32_to_bf16(bf16_to_f32(v[e]) * gv);
        *reinterpret_cast<bf16x8*>(x + i * 8) = v;
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

// Forward and gradient in one pass for bf16 logits with long rows (see ce_fused_row_kernel).  dlogits may be
// nullptr (no gradient wanted); it has row stride ld_drow and columns [classes, pad_to) are written as zeros.
// Needs classes > 1024, ld_row, ld_drow and pad_to multiples of 8, ld_row >= classes rounded up to 8, and at
// most 65536 classes.  The gradient is for an incoming gradient of 1: scale with pdt_scale_by_scalar.
PDT_API int pdt_softmax_ce_fused(const void* logits, const long long* labels, float* row_loss, float* loss, void* dlogits,
                                 int rows, int classes, long long ld_row, long long ld_drow, int pad_to, void* stream) {
    const int c8 = (classes + 7) / 8 * 8;
    if (rows <= 0 || classes <= 1024 || classes > 65536 || (ld_row % 8) != 0 || ld_row < c8) return -1;
    if (dlogits != nullptr && ((ld_drow % 8) != 0 || (pad_to % 8) != 0 || pad_to < c8 || ld_drow < pad_to || pad_to > 65536)) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int need = (((dlogits != nullptr ? pad_to : c8) >> 3) + 511) / 512;
    const bf16* x = static_cast<const bf16*>(logits);
    bf16* d = static_cast<bf16*>(dlogits);
#define PDT_CE(NV) hipLaunchKernelGGL(ce_fused_row_kernel<NV>, dim3(rows), dim3(512), 0, s, x, labels, row_loss, d, rows, classes, ld_row, ld_drow, pad_to, 1.f / rows)
    if (need <= 4) PDT_CE(4);
    else if (need <= 8) PDT_CE(8);
    else if (need <= 13) PDT_CE(13);
    else PDT_CE(16);
#undef PDT_CE
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, s, row_loss, loss, static_cast<long long>(rows), 1.f / rows);
    return pdt_launch_status();
}

// The same pass over one chunk of `rows` rows of a batch of `total_rows` rows: the gradient is scaled by
// 1 / total_rows and no mean is taken (pdt_mean_f32 over the row losses of all chunks does that).  `dlogits` may be
// the logits buffer itself (in-place): a language-model head then needs no second [rows, vocabulary] buffer.
PDT_API int pdt_softmax_ce_fused_chunk(const void* logits, const long long* labels, float* row_loss, void* dlogits,
                                       int rows, int classes, long long ld_row, long long ld_drow, int pad_to,
                                       long long total_rows, void* stream) {
    const int c8 = (classes + 7) / 8 * 8;
    if (rows <= 0 || total_rows < rows || classes <= 1024 || classes > 65536 || (ld_row % 8) != 0 || ld_row < c8) return -1;
    if (dlogits != nullptr && ((ld_drow % 8) != 0 || (pad_to % 8) != 0 || pad_to < c8 || ld_drow < pad_to || pad_to > 65536)) return -1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int need = (((dlogits != nullptr ? pad_to : c8) >> 3) + 511) / 512;
    const bf16* x = static_cast<const bf16*>(logits);
    bf16* d = static_cast<bf16*>(dlogits);
    const float inv = 1.f / static_cast<float>(total_rows);
#define PDT_CE(NV) hipLaunchKernelGGL(ce_fused_row_kernel<NV>, dim3(rows), dim3(512), 0, s, x, labels, row_loss, d, rows, classes, ld_row, ld_drow, pad_to, inv)
    if (need <= 4) PDT_CE(4);
    else if (need <= 8) PDT_CE(8);
    else if (need <= 13) PDT_CE(13);
    else PDT_CE(16);
#undef PDT_CE
    return pdt_launch_status();
}

// out[0] = scale * sum(x[0..n)) (deterministic, one workgroup)
PDT_API int pdt_mean_f32(const float* x, float* out, long long n, float scale, void* stream) {
    if (n <= 0) return -1;
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), x, out, n, scale);
    return pdt_launch_status();
}

// dst += g[0] * src over fp32 vectors (n a multiple of 4): a gradient computed for an incoming gradient of 1 is
// added to its destination once the incoming gradient is known
__global__ __launch_bounds__(256) void axpy_dev_scalar_kernel(float4* __restrict__ dst, const float4* __restrict__ src,
                                                              const float* __restrict__ g, long long n4) {
    const float gv = g[0];
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += gridDim.x * 256ll) {
        float4 d = dst[i];
        const float4 a = src[i];
        d.x += gv * a.x;
        d.y += gv * a.y;
        d.z += gv * a.z;
        d.w += gv * a.w;
        dst[i] = d;
    }
}

PDT_API int pdt_axpy_dev_scalar_f32(float* dst, const float* src, const float* g, long long n, void* stream) {
    if (n <= 0 || (n % 4) != 0) return -1;
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(axpy_dev_scalar_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<float4*>(dst), reinterpret_cast<const float4*>(src), g, n / 4);
    return pdt_launch_status();
}

PDT_API int pdt_scale_by_scalar(void* x, const float* g, long long numel, void* stream) {
    if (numel <= 0 || (numel % 8) != 0) return -1;
    long long blocks = (numel / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scale_by_scalar_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<bf16*>(x), g, numel / 8);
    return pdt_launch_status();
}
