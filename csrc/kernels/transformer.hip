// Transformer support kernels (bf16 activations, fp32 parameters): LayerNorm forward/backward,
// tanh-GELU forward/backward, bias + residual add, embedding gather and its scatter-add gradient,
// column sums (bias gradients).  All row-wise passes use one wave per row with 16-byte accesses.
//
// Not exercised by the reference (its model is a CNN, /root/reference/src/main.py:49); they serve the
// ViT-B/16 and GPT-2 data-parallel workloads.
#include "common.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------------
// LayerNorm
// ------------------------------------------------------------------------------------------------
// Activations beyond 160 MB are streamed past the caches (NT): nothing that large survives until its consumer runs,
// and ordinary stores of it evict what the next kernel could have reused.
template <bool NT>
__device__ __forceinline__ bf16x8 ln_ld8(const bf16* q) {
    return NT ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(q)) : *reinterpret_cast<const bf16x8*>(q);
}
template <bool NT>
__device__ __forceinline__ void ln_st8(bf16* q, const bf16x8& v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(q));
    else *reinterpret_cast<bf16x8*>(q) = v;
}
static inline bool ln_stream(int rows, int D) { return static_cast<long long>(rows) * D * 2 > (160ll << 20); }

// Persistent waves walk rows row0, row0 + stride, ...: a row (up to 64 * 8 * MAXV columns) is read once and stays in
// registers between the statistics and the output pass, the next row's loads are issued before the current row is
// reduced (two rows in flight per wave), and - up to 1024 columns - a lane keeps its gamma / beta in registers.
template <int MAXV, bool NT>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, bf16* __restrict__ y,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           int rows, int D, float eps) {
    constexpr bool KEEP = MAXV <= 2;
    constexpr int NK = KEEP ? MAXV : 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D >> 3;
    float gm[NK][8], bt[NK][8];
    if (KEEP) {
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int v = lane + 64 * i;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                gm[i][e] = v < nv ? gamma[v * 8 + e] : 0.f;
                bt[i][e] = v < nv ? beta[v * 8 + e] : 0.f;
            }
        }
    }
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto load_row = [&](bf16x8(&r)[MAXV], int row) {
        const bf16* xr = x + static_cast<size_t>(row) * D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int v = lane + 64 * i;
            r[i] = v < nv ? ln_ld8<NT>(xr + v * 8) : zero8;
        }
    };
    const int stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    bf16x8 cur[MAXV], nxt[MAXV];
    load_row(cur, row);
    const float inv_d = 1.f / D;
    for (; row < rows; row += stride) {
        const bool more = row + stride < rows;
        if (more) load_row(nxt, row + stride);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = bf16_to_f32(cur[i][e]);  // lanes beyond the row hold zeros
                s1 += f;
                s2 += f * f;
            }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const float mean = s1 * inv_d;
        const float rstd = rsqrtf(fmaxf(s2 * inv_d - mean * mean, 0.f) + eps);
        if (lane == 0) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
        bf16* yr = y + static_cast<size_t>(row) * D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int v = lane + 64 * i;
            if (v < nv) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float g = KEEP ? gm[i][e] : gamma[v * 8 + e], bb = KEEP ? bt[i][e] : beta[v * 8 + e];
                    o[e] = f32_to_bf16((bf16_to_f32(cur[i][e]) - mean) * rstd * g + bb);
                }
                ln_st8<NT>(yr + v * 8, o);
            }
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) cur[i] = nxt[i];
        }
    }
}

PDT_API int pdt_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                              int rows, int D, float eps, void* stream) {
    if (rows <= 0 || D <= 0 || (D % 8) != 0 || D > 8 * 64 * 8) return -1;
    const int maxv = (D / 8 + 63) / 64;
    // Two workgroups of four persistent waves per CU: a wave's preload of gamma / beta (8 KB) is then small against the
    // rows it walks; measured best for streamed and cache-resident tensors alike (bench/ln_bench.py).
    const bool nt = ln_stream(rows, D);
    const int want = pdt_cdiv(rows, 4), cap = 256 * 2;
    const dim3 grid(want < cap ? want : cap);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define PDT_LNF(V)                                                                                                                  \
    do {                                                                                                                            \
        if (nt) hipLaunchKernelGGL((layernorm_fwd_kernel<V, true>), grid, dim3(256), 0, s, static_cast<const bf16*>(x), gamma, beta, \
                                   static_cast<bf16*>(y), mean, rstd, rows, D, eps);                                                \
        else hipLaunchKernelGGL((layernorm_fwd_kernel<V, false>), grid, dim3(256), 0, s, static_cast<const bf16*>(x), gamma, beta,   \
                                static_cast<bf16*>(y), mean, rstd, rows, D, eps);                                                   \
    } while (0)
    if (maxv <= 1) PDT_LNF(1);
    else if (maxv <= 2) PDT_LNF(2);
    else if (maxv <= 4) PDT_LNF(4);
    else PDT_LNF(8);
#undef PDT_LNF
    return pdt_launch_status();
}

// dx = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)) (+ addend);  dgamma += dy*xhat;  dbeta += dy
// Each wave walks rows row0, row0 + stride, ...; a lane always owns the same columns, so it keeps gamma and its
// dgamma/dbeta partial sums in registers (MAXV vectors of 8 columns), combined per workgroup in LDS and
// written to the workgroup's row of `partial` ([workgroups][2][D]); layernorm_param_grad_kernel folds those
// rows into dgamma / dbeta.  (Global atomics on the 2 D result addresses from every workgroup cost more
// than the whole streaming pass.)  A row's dy and x stay in registers between
// the reduction and the output pass (rows of up to 2048 columns), so each is read from memory once.
// `addend` (optional) is a second gradient of the same input - the skip connection around the normalised
// branch - added here instead of by a separate element-wise pass.
template <int MAXV, int LN_BWD_WAVES, bool NT>
__global__ __launch_bounds__(LN_BWD_WAVES * 64) void layernorm_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                           const float* __restrict__ gamma, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const bf16* __restrict__ addend,
                                                           bf16* __restrict__ dx, float* __restrict__ partial, int rows, int D) {
    constexpr bool KEEP = MAXV <= 4;
    constexpr int NK = KEEP ? MAXV : 1;
    __shared__ float sred[2 * MAXV * 8 * 64];  // [dgamma | dbeta][vector i][element e][lane]: conflict-free per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D >> 3;
    float ag[MAXV][8], ab[MAXV][8], gm[NK][8];
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) ag[i][e] = ab[i][e] = 0.f;
    if (KEEP) {
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int v = lane + 64 * i;
#pragma unroll
            for (int e = 0; e < 8; ++e) gm[i][e] = v < nv ? gamma[v * 8 + e] : 0.f;
        }
    }
    // KEEP: a row's dy, x and addend vectors stay in registers, and the next row's loads are issued before the current
    // row is reduced, so two rows are in flight per wave.
    bf16x8 dk[NK], xk[NK], ak[NK], dn[NK], xn[NK], an[NK];
    float mu_n = 0.f, rs_n = 0.f;
    auto load_row = [&](bf16x8(&d)[NK], bf16x8(&xx)[NK], bf16x8(&aa)[NK], float& mu_o, float& rs_o, int row) {
        const bf16* dyr = dy + static_cast<size_t>(row) * D;
        const bf16* xr = x + static_cast<size_t>(row) * D;
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int v = lane + 64 * i;
            if (v < nv) {
                d[i] = ln_ld8<NT>(dyr + v * 8);
                xx[i] = ln_ld8<NT>(xr + v * 8);
                if (addend != nullptr) aa[i] = ln_ld8<NT>(addend + static_cast<size_t>(row) * D + v * 8);
            }
        }
        mu_o = mean[row];
        rs_o = rstd[row];
    };
    const int row_first = blockIdx.x * LN_BWD_WAVES + wave, row_stride = gridDim.x * LN_BWD_WAVES;
    if (KEEP && row_first < rows) load_row(dn, xn, an, mu_n, rs_n, row_first);
    for (int row = row_first; row < rows; row += row_stride) {
        const bf16* dyr = dy + static_cast<size_t>(row) * D;
        const bf16* xr = x + static_cast<size_t>(row) * D;
        const bf16* ar = addend != nullptr ? addend + static_cast<size_t>(row) * D : nullptr;
        float mu, rs;
        if (KEEP) {
#pragma unroll
            for (int i = 0; i < NK; ++i) {
                dk[i] = dn[i];
                xk[i] = xn[i];
                ak[i] = an[i];
            }
            mu = mu_n;
            rs = rs_n;
            if (row + row_stride < rows) load_row(dn, xn, an, mu_n, rs_n, row + row_stride);
        } else {
            mu = mean[row];
            rs = rstd[row];
        }
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int v = lane + 64 * i;
            if (v < nv) {
                const bf16x8 dv = KEEP ? dk[i] : *reinterpret_cast<const bf16x8*>(dyr + v * 8);
                const bf16x8 xv = KEEP ? xk[i] : *reinterpret_cast<const bf16x8*>(xr + v * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float g = KEEP ? gm[i][e] : gamma[v * 8 + e];
                    const float d = bf16_to_f32(dv[e]), xh = (bf16_to_f32(xv[e]) - mu) * rs;
                    c1 += d * g * xh;
                    c2 += d * g;
                    ag[i][e] += d * xh;
                    ab[i][e] += d;
                }
            }
        }
        c1 = wave_sum(c1) / D;
        c2 = wave_sum(c2) / D;
        bf16* dxr = dx + static_cast<size_t>(row) * D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int v = lane + 64 * i;
            if (v < nv) {
                const bf16x8 dv = KEEP ? dk[i] : *reinterpret_cast<const bf16x8*>(dyr + v * 8);
                const bf16x8 xv = KEEP ? xk[i] : *reinterpret_cast<const bf16x8*>(xr + v * 8);
                bf16x8 av;
                if (ar != nullptr) av = KEEP ? ak[i] : *reinterpret_cast<const bf16x8*>(ar + v * 8);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float g = KEEP ? gm[i][e] : gamma[v * 8 + e];
                    const float xh = (bf16_to_f32(xv[e]) - mu) * rs;
                    float r = rs * (bf16_to_f32(dv[e]) * g - c2 - xh * c1);
                    if (ar != nullptr) r += bf16_to_f32(av[e]);
                    o[e] = f32_to_bf16(r);
                }
                ln_st8<NT>(dxr + v * 8, o);
            }
        }
    }
    // The waves add their register sums into LDS one after the other (LDS float atomics are far slower than
    // this: they dominated the kernel's run time).
    for (int w = 0; w < LN_BWD_WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int a = (i * 8 + e) * 64 + lane, b = (MAXV * 8 + i * 8 + e) * 64 + lane;
                    sred[a] = w == 0 ? ag[i][e] : sred[a] + ag[i][e];
                    sred[b] = w == 0 ? ab[i][e] : sred[b] + ab[i][e];
                }
        }
        __syncthreads();
    }
    float* out = partial + static_cast<size_t>(blockIdx.x) * 2 * D;
    for (int idx = threadIdx.x; idx < 2 * MAXV * 8 * 64; idx += LN_BWD_WAVES * 64) {
        const int ln = idx & 63, j = idx >> 6;
        const int which = j / (MAXV * 8), r = j - which * (MAXV * 8);
        const int v = ln + 64 * (r >> 3);
        if (v < nv) out[which * D + v * 8 + (r & 7)] = sred[idx];
    }
}

// dgamma[c] += sum_b partial[b][c], dbeta[c] += sum_b partial[b][D + c]: one workgroup per 64 of the 2 D columns,
// its four waves take every fourth row of `partial`.
__global__ __launch_bounds__(256) void layernorm_param_grad_kernel(const float* __restrict__ partial, int nparts, int D,
                                                                  float* dgamma, float* dbeta) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;  // column of the [2 D] row
    float t = 0.f;
    if (c < 2 * D) {
#pragma unroll 8
        for (int b = wave; b < nparts; b += 4) t += partial[static_cast<size_t>(b) * 2 * D + c];
    }
    red[wave][lane] = t;
    __syncthreads();
    if (wave == 0 && c < 2 * D) {
        t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        if (c < D) dgamma[c] += t;
        else dbeta[c - D] += t;
    }
}

// Rows of up to 1024 columns: 8 waves per workgroup, one workgroup per CU (2048 waves; more waves or more
// workgroups only add reduction work).  Longer rows need more registers per lane: 4 waves per workgroup.
static void ln_bwd_config(int D, int* waves, int* cap) {
    const bool small = D <= 1024;
    *waves = small ? 8 : 4;
    *cap = small ? 256 : 512;
}

// Workgroups the backward kernel is launched with = rows of `partial` the caller must provide.
PDT_API int pdt_layernorm_bwd_parts(int rows, int D) {
    int waves, cap;
    ln_bwd_config(D, &waves, &cap);
    const int g = pdt_cdiv(rows, waves);
    return g < cap ? g : cap;
}

// dgamma / dbeta are added to (zero-initialise them, or pass a running sum); they must not be written by anything
// else while this runs.  addend: optional bf16 [rows, D] added to dx.  partial: fp32 [pdt_layernorm_bwd_parts(rows, D)][2 D].
PDT_API int pdt_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                              void* dx, float* dgamma, float* dbeta, int rows, int D, const void* addend, float* partial,
                              void* stream) {
    if (rows <= 0 || D <= 0 || (D % 8) != 0 || D > 8 * 64 * 8 || partial == nullptr) return -1;
    const int nv = D / 8, maxv = (nv + 63) / 64;
    const int grid = pdt_layernorm_bwd_parts(rows, D);
    int waves, cap;
    ln_bwd_config(D, &waves, &cap);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bf16 *pdy = static_cast<const bf16*>(dy), *px = static_cast<const bf16*>(x), *pa = static_cast<const bf16*>(addend);
    bf16* pdx = static_cast<bf16*>(dx);
    const bool nt = ln_stream(rows, D);
#define PDT_LN(V, W)                                                                                                                     \
    do {                                                                                                                                 \
        if (nt) hipLaunchKernelGGL((layernorm_bwd_kernel<V, W, true>), dim3(grid), dim3(W * 64), 0, s, pdy, px, gamma, mean, rstd, pa, pdx, \
                                   partial, rows, D);                                                                                    \
        else hipLaunchKernelGGL((layernorm_bwd_kernel<V, W, false>), dim3(grid), dim3(W * 64), 0, s, pdy, px, gamma, mean, rstd, pa, pdx,  \
                                partial, rows, D);                                                                                       \
    } while (0)
    if (maxv <= 1) PDT_LN(1, 8);
    else if (maxv <= 2) PDT_LN(2, 8);
    else if (maxv <= 4) PDT_LN(4, 4);
    else PDT_LN(8, 4);
#undef PDT_LN
    hipLaunchKernelGGL(layernorm_param_grad_kernel, dim3(pdt_cdiv(2 * D, 64)), dim3(256), 0, s, partial, grid, D, dgamma, dbeta);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// GELU (tanh form)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, long long nvec) {
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < nvec; v += gridDim.x * 256ll) {
        const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + v * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(pdt_gelu(bf16_to_f32(xv[e])));
        *reinterpret_cast<bf16x8*>(y + v * 8) = o;
    }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16* __restrict__ g, const bf16* __restrict__ x,
                                                      bf16* __restrict__ dx, long long nvec) {
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < nvec; v += gridDim.x * 256ll) {
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + v * 8);
        const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + v * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(bf16_to_f32(gv[e]) * pdt_gelu_grad(bf16_to_f32(xv[e])));
        *reinterpret_cast<bf16x8*>(dx + v * 8) = o;
    }
}

static inline int ew_grid8(long long nvec) {
    long long g = (nvec + 255) / 256;
    return static_cast<int>(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

PDT_API int pdt_gelu_fwd(const void* x, void* y, long long numel, void* stream) {
    if (numel <= 0 || (numel % 8) != 0) return -1;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ew_grid8(numel / 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16*>(x), static_cast<bf16*>(y), numel / 8);
    return pdt_launch_status();
}

PDT_API int pdt_gelu_bwd(const void* g, const void* x, void* dx, long long numel, void* stream) {
    if (numel <= 0 || (numel % 8) != 0) return -1;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_grid8(numel / 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16*>(g), static_cast<const bf16*>(x), static_cast<bf16*>(dx), numel / 8);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// out = x (+ bias[col]) (+ residual)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_residual_kernel(const bf16* __restrict__ x, const float* __restrict__ bias,
                                                           const bf16* __restrict__ res, bf16* __restrict__ out,
                                                           long long nvec, int D) {
    const int dvec = D >> 3;
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < nvec; v += gridDim.x * 256ll) {
        const int c = static_cast<int>(v % dvec) << 3;
        const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + v * 8);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = bf16_to_f32(xv[e]);
        if (bias != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += bias[c + e];
        }
        if (res != nullptr) {
            const bf16x8 rv = *reinterpret_cast<const bf16x8*>(res + v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += bf16_to_f32(rv[e]);
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(f[e]);
        *reinterpret_cast<bf16x8*>(out + v * 8) = o;
    }
}

PDT_API int pdt_bias_residual(const void* x, const float* bias, const void* res, void* out, long long numel, int D,
                              void* stream) {
    if (numel <= 0 || D <= 0 || (D % 8) != 0 || (numel % D) != 0) return -1;
    hipLaunchKernelGGL(bias_residual_kernel, dim3(ew_grid8(numel / 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16*>(x), bias, static_cast<const bf16*>(res), static_cast<bf16*>(out), numel / 8, D);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Column sums of a bf16 [rows, ld] matrix -> fp32 out[cols] (zero-initialised); used for bias gradients
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const bf16* __restrict__ x, float* out, int rows, int cols, int ld,
                                                    int rows_per_block) {
    __shared__ float red[32][64];
    const int chunk = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.y * 64 + chunk * 8;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    if (c < cols) {
        for (int r = r0 + rl; r < r1; r += 32) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + static_cast<size_t>(r) * ld + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += bf16_to_f32(v[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rl][chunk * 8 + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
        const int cc = blockIdx.y * 64 + threadIdx.x;
        if (cc < cols) __hip_atomic_fetch_add(out + cc, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// reads whole 8-column chunks: ld must be >= cols rounded up to 8
PDT_API int pdt_colsum(const void* x, float* out, int rows, int cols, int ld, void* stream) {
    if (rows <= 0 || cols <= 0 || (ld % 8) != 0 || ld < (cols + 7) / 8 * 8) return -1;
    const int groups = pdt_cdiv(cols, 64);
    int bx = 1024 / groups;
    if (bx < 1 || pdt_deterministic_flag) bx = 1;  // deterministic mode: one row block, so one adder per column
    int rpb = pdt_cdiv(rows, bx);
    rpb = pdt_cdiv(rpb, 32) * 32;
    hipLaunchKernelGGL(colsum_kernel, dim3(pdt_cdiv(rows, rpb), groups), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16*>(x), out, rows, cols, ld, rpb);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Embedding: out[i] = bf16(W[ids[i]]);  dW[ids[i]] += g[i]  (fp32 atomics, dW zero-initialised)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const long long* __restrict__ ids, const float* __restrict__ w,
                                                           bf16* __restrict__ out, long long n, int D, long long vocab) {
    const int dvec = D >> 3;
    const long long nvec = n * dvec;
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < nvec; v += gridDim.x * 256ll) {
        const long long i = v / dvec;
        const int c = static_cast<int>(v - i * dvec) << 3;
        long long id = ids[i];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const f32x4 a = *reinterpret_cast<const f32x4*>(w + id * D + c), b = *reinterpret_cast<const f32x4*>(w + id * D + c + 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = f32_to_bf16(a[e]);
            o[e + 4] = f32_to_bf16(b[e]);
        }
        *reinterpret_cast<bf16x8*>(out + v * 8) = o;
    }
}

__global__ __launch_bounds__(256) void embedding_bwd_kernel(const long long* __restrict__ ids, const bf16* __restrict__ g,
                                                           float* dw, long long n, int D, long long vocab) {
    const int dvec = D >> 3;
    const long long nvec = n * dvec;
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < nvec; v += gridDim.x * 256ll) {
        const long long i = v / dvec;
        const int c = static_cast<int>(v - i * dvec) << 3;
        const long long id = ids[i];
        if (id < 0 || id >= vocab) continue;
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>(g + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            __hip_atomic_fetch_add(dw + id * D + c + e, bf16_to_f32(gv[e]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

PDT_API int pdt_embedding_fwd(const long long* ids, const float* w, void* out, long long n, int D, long long vocab,
                              void* stream) {
    if (n <= 0 || D <= 0 || (D % 8) != 0 || vocab <= 0) return -1;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(ew_grid8(n * (D / 8))), dim3(256), 0, static_cast<hipStream_t>(stream), ids, w,
                       static_cast<bf16*>(out), n, D, vocab);
    return pdt_launch_status();
}

// Row form of the same scatter-add: a wave owns a token, its lanes walk the row with stride 64 elements, so every atomic
// instruction covers 256 contiguous bytes of one table row (the element-vector form above spreads the 64 lanes of an
// instruction over 64 different 32-byte segments).
__global__ __launch_bounds__(256) void embedding_bwd_rows_kernel(const long long* __restrict__ ids, const bf16* __restrict__ g,
                                                                float* dw, long long n, int D, long long vocab) {
    const int lane = threadIdx.x & 63;
    const long long waves = static_cast<long long>(gridDim.x) * 4;
    for (long long i = blockIdx.x * 4ll + (threadIdx.x >> 6); i < n; i += waves) {
        const long long id = ids[i];
        if (id < 0 || id >= vocab) continue;
        const bf16* src = g + i * D;
        float* dst = dw + id * D;
        for (int c = lane * 2; c < D; c += 128) {
            const bf16x2 v = *reinterpret_cast<const bf16x2*>(src + c);
            __hip_atomic_fetch_add(dst + c, bf16_to_f32(v[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(dst + c + 1, bf16_to_f32(v[1]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

PDT_API int pdt_embedding_bwd(const long long* ids, const void* g, float* dw, long long n, int D, long long vocab,
                              void* stream) {
    if (n <= 0 || D <= 0 || (D % 8) != 0 || vocab <= 0) return -1;
    static const bool rows = std::getenv("PDT_EMB_BWD_ROWS") == nullptr || std::atoi(std::getenv("PDT_EMB_BWD_ROWS")) != 0;
    if (rows) {
        const long long blocks = (n + 3) / 4;
        hipLaunchKernelGGL(embedding_bwd_rows_kernel, dim3(static_cast<unsigned>(blocks < 16384 ? blocks : 16384)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), ids, static_cast<const bf16*>(g), dw, n, D, vocab);
        return pdt_launch_status();
    }
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(ew_grid8(n * (D / 8))), dim3(256), 0, static_cast<hipStream_t>(stream), ids,
                       static_cast<const bf16*>(g), dw, n, D, vocab);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Token + position embedding in one pass (language models):  out[b][t] = bf16(W[ids[b][t]] + P[t]).
// Gradient of the position table: dP[t] += sum_b g[b][t] - one thread owns 8 channels of one position and walks
// the batch, so no atomics and no [B, S, D] -> [S, D] reduction pass; the token table uses embedding_bwd.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embedding_pos_fwd_kernel(const long long* __restrict__ ids, const float* __restrict__ w,
                                                               const float* __restrict__ pos, bf16* __restrict__ out, long long n,
                                                               int D, long long vocab, int S) {
    const int dvec = D >> 3;
    const long long nvec = n * dvec;
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < nvec; v += gridDim.x * 256ll) {
        const long long i = v / dvec;
        const int c = static_cast<int>(v - i * dvec) << 3;
        long long id = ids[i];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const int t = static_cast<int>(i % S);
        const f32x4 a = *reinterpret_cast<const f32x4*>(w + id * D + c), b = *reinterpret_cast<const f32x4*>(w + id * D + c + 4);
        const f32x4 pa = *reinterpret_cast<const f32x4*>(pos + static_cast<long long>(t) * D + c),
                    pb = *reinterpret_cast<const f32x4*>(pos + static_cast<long long>(t) * D + c + 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = f32_to_bf16(a[e] + pa[e]);
            o[e + 4] = f32_to_bf16(b[e] + pb[e]);
        }
        *reinterpret_cast<bf16x8*>(out + v * 8) = o;
    }
}

// dst[r][c] += sum over b of g[b][row0 + r][c]  for r in [0, R): g is [B][S][D] bf16, dst is [R][D] fp32.
// gridDim.y slices of the batch, each adding its partial sum with atomics (one slice: a plain read-modify-write) - a
// position table has only R D / 8 chunks, far fewer threads than the GPU holds, and the batch is the long axis.
__global__ __launch_bounds__(256) void batch_rowsum_kernel(const bf16* __restrict__ g, float* __restrict__ dst, int B, int S, int D,
                                                          int row0, int R) {
    const int dvec = D >> 3;
    const int nvec = R * dvec;
    const int per = (B + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
    const long long bstride = static_cast<long long>(S) * D;
    for (int v = blockIdx.x * 256 + threadIdx.x; v < nvec; v += gridDim.x * 256) {
        const int r = v / dvec;
        const int c = (v - r * dvec) << 3;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        const bf16* src = g + (static_cast<long long>(row0 + r)) * D + c;
        int bb = b0;
        for (; bb + 4 <= b1; bb += 4) {  // four independent loads in flight per lane
            bf16x8 gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) gv[u] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(src + (bb + u) * bstride));
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += bf16_to_f32(gv[u][e]);
        }
        for (; bb < b1; ++bb) {
            const bf16x8 gv = *reinterpret_cast<const bf16x8*>(src + bb * bstride);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf16_to_f32(gv[e]);
        }
        float* d = dst + static_cast<long long>(r) * D + c;
        if (gridDim.y == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) d[e] += acc[e];
        } else if (b0 < b1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) __hip_atomic_fetch_add(d + e, acc[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

PDT_API int pdt_embedding_pos_fwd(const long long* ids, const float* w, const float* pos, void* out, long long n, int D,
                                  long long vocab, int S, void* stream) {
    if (n <= 0 || D <= 0 || (D % 8) != 0 || vocab <= 0 || S <= 0) return -1;
    hipLaunchKernelGGL(embedding_pos_fwd_kernel, dim3(ew_grid8(n * (D / 8))), dim3(256), 0, static_cast<hipStream_t>(stream), ids, w,
                       pos, static_cast<bf16*>(out), n, D, vocab, S);
    return pdt_launch_status();
}

PDT_API int pdt_batch_rowsum(const void* g, float* dst, int B, int S, int D, int row0, int R, void* stream) {
    if (B <= 0 || S <= 0 || D <= 0 || (D % 8) != 0 || row0 < 0 || R <= 0 || row0 + R > S) return -1;
    const long long nvec = static_cast<long long>(R) * (D / 8);
    if (nvec >= (1ll << 31)) return -1;
    const int gx = ew_grid8(nvec);
    // enough workgroups for four per CU; deterministic mode: one adder per element, in batch order
    int gy = pdt_deterministic_flag ? 1 : (1024 + gx - 1) / gx;
    if (gy > B / 8) gy = B / 8;
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(batch_rowsum_kernel, dim3(gx, gy), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16*>(g), dst, B, S, D, row0, R);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Vision-transformer token assembly:  h[b][0] = cls + P[0],  h[b][1 + t] = tok[b][t] + P[1 + t]
// (class token and position table are fp32 parameters, tok the bf16 patch embedding), and the part of its gradient
// that is a copy: dtok[b][t] = g[b][1 + t].  The parameter gradients come from batch_rowsum_kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vit_assemble_fwd_kernel(const bf16* __restrict__ tok, const float* __restrict__ cls,
                                                              const float* __restrict__ pos, bf16* __restrict__ out, int B, int T,
                                                              int D) {
    const int dvec = D >> 3;
    const long long nvec = static_cast<long long>(B) * (T + 1) * dvec;
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < nvec; v += gridDim.x * 256ll) {
        const long long row = v / dvec;
        const int c = static_cast<int>(v - row * dvec) << 3;
        const int b = static_cast<int>(row / (T + 1)), t = static_cast<int>(row - static_cast<long long>(b) * (T + 1));
        const f32x4 pa = *reinterpret_cast<const f32x4*>(pos + static_cast<long long>(t) * D + c),
                    pb = *reinterpret_cast<const f32x4*>(pos + static_cast<long long>(t) * D + c + 4);
        float x[8];
        if (t == 0) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(cls + c), bb = *reinterpret_cast<const f32x4*>(cls + c + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = a[e];
                x[e + 4] = bb[e];
            }
        } else {
            const bf16x8 tv = *reinterpret_cast<const bf16x8*>(tok + (static_cast<long long>(b) * T + (t - 1)) * D + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = bf16_to_f32(tv[e]);
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = f32_to_bf16(x[e] + pa[e]);
            o[e + 4] = f32_to_bf16(x[e + 4] + pb[e]);
        }
        *reinterpret_cast<bf16x8*>(out + v * 8) = o;
    }
}

__global__ __launch_bounds__(256) void vit_assemble_bwd_kernel(const bf16* __restrict__ g, bf16* __restrict__ dtok, int B, int T, int D) {
    const int dvec = D >> 3;
    const long long nvec = static_cast<long long>(B) * T * dvec;
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < nvec; v += gridDim.x * 256ll) {
        const long long row = v / dvec;
        const int c = static_cast<int>(v - row * dvec) << 3;
        const long long b = row / T, t = row - b * T;
        *reinterpret_cast<bf16x8*>(dtok + v * 8) = *reinterpret_cast<const bf16x8*>(g + (b * (T + 1) + t + 1) * D + c);
    }
}

PDT_API int pdt_vit_assemble_fwd(const void* tok, const float* cls, const float* pos, void* out, int B, int T, int D, void* stream) {
    if (B <= 0 || T <= 0 || D <= 0 || (D % 8) != 0) return -1;
    hipLaunchKernelGGL(vit_assemble_fwd_kernel, dim3(ew_grid8(static_cast<long long>(B) * (T + 1) * (D / 8))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const bf16*>(tok), cls, pos, static_cast<bf16*>(out), B, T, D);
    return pdt_launch_status();
}

PDT_API int pdt_vit_assemble_bwd(const void* g, void* dtok, int B, int T, int D, void* stream) {
    if (B <= 0 || T <= 0 || D <= 0 || (D % 8) != 0) return -1;
    hipLaunchKernelGGL(vit_assemble_bwd_kernel, dim3(ew_grid8(static_cast<long long>(B) * T * (D / 8))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const bf16*>(g), static_cast<bf16*>(dtok), B, T, D);
    return pdt_launch_status();
}
