// Fused BatchNorm kernels for NHWC bf16 activations (training and inference).
//
// Forward:  the convolution epilogue has already accumulated per-channel (sum, sumsq) of its output
//           into PDT_STAT_SLOTS slots.  bn_finalize folds the slots into mean / rstd / scale / shift
//           and updates the running statistics; bn_act / bn_add_relu / bn_relu_maxpool then apply
//           y*scale + shift (+ReLU, +residual, +3x3/2 max-pool) in one pass.
// Backward: bn_bwd_reduce forms dz = dout * [out > 0] on the fly and accumulates sum(dz) and
//           sum(dz * xhat) per channel (optionally for a second BatchNorm sharing dz: the projected
//           shortcut); bn_bwd_finalize turns them into dgamma, dbeta and three per-channel
//           coefficients; bn_bwd_apply writes dy = A*dz + B*y + C (and dz itself for the identity
//           branch) in one pass.
// All element-wise passes move 16 bytes (8 channels) per lane.
//
// Role parity: the 20 BatchNorm2d + 17 ReLU + 8 residual adds + max-pool of torchvision's ResNet-18
// as exercised by /root/reference/src/main.py:49,74,78.
#include "common.h"

int pdt_deterministic_flag = 0;
PDT_API void pdt_set_deterministic(int on) { pdt_deterministic_flag = on ? 1 : 0; }
PDT_API int pdt_get_deterministic() { return pdt_deterministic_flag; }

// Deterministic mode, reductions over rows: every row block stores its partial sums to a scratch array (one writer per
// element) and a second kernel folds them into the accumulator slots in block order - as many row blocks as the fast
// path uses, so the pass still runs at memory speed.  The scratch array belongs to the stream the kernels run on (the
// two launches are adjacent in it); it is allocated at the first use, which therefore must not lie inside a capture.
static float* g_det_partials = nullptr;
static size_t g_det_partials_floats = 0;
static float* det_partials(size_t floats) {
    if (floats > g_det_partials_floats) {
        if (g_det_partials != nullptr) (void)hipFree(g_det_partials);
        g_det_partials = nullptr;
        g_det_partials_floats = 0;
        const size_t want = floats < (size_t(1) << 20) ? (size_t(1) << 20) : floats;
        if (hipMalloc(&g_det_partials, want * sizeof(float)) != hipSuccess) return nullptr;
        g_det_partials_floats = want;
    }
    return g_det_partials;
}

// slots[s][w][c] = sum over blocks b = s, s + SLOTS, s + 2 SLOTS, ... (in that order) of partials[b][w][c]
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ partials, int nblk, int rows, int used_rows,
                                                           int C, float* __restrict__ slots) {
    const int c = blockIdx.x * 256 + threadIdx.x, w = blockIdx.y, sl = blockIdx.z;
    if (c >= C || w >= used_rows) return;
    float t = 0.f;
    for (int b = sl; b < nblk; b += PDT_STAT_SLOTS) t += partials[(static_cast<size_t>(b) * rows + w) * C + c];
    slots[(static_cast<size_t>(sl) * rows + w) * C + c] = t;
}

// dst[i] += slabs[0][i] + slabs[1][i] + ... (in that order): the per-split weight-gradient slabs of deterministic mode
__global__ __launch_bounds__(256) void sum_slabs_kernel(float* __restrict__ dst, const float* __restrict__ slabs, int nsplit,
                                                       long long n, long long stride) {
    const long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    float t = slabs[i];
    for (int k = 1; k < nsplit; ++k) t += slabs[k * stride + i];
    dst[i] += t;
}

PDT_API int pdt_sum_slabs(float* dst, const float* slabs, int nsplit, long long n, long long stride, void* stream) {
    if (nsplit <= 0 || n <= 0 || stride < n) return -1;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dst, slabs, nsplit, n, stride);
    return pdt_launch_status();
}

static void launch_fold(const float* partials, int nblk, int rows, int used_rows, int C, float* slots, hipStream_t s) {
    hipLaunchKernelGGL(fold_partials_kernel, dim3(pdt_cdiv(C, 256), used_rows, PDT_STAT_SLOTS), dim3(256), 0, s, partials, nblk,
                       rows, used_rows, C, slots);
}

// (bn_finalize - slots -> per-channel constants - lives in bn_finalize.hip, which is compiled with IEEE NaN / Inf semantics)

// ------------------------------------------------------------------------------------------------
// element-wise forward:  out = act(y*scale + shift [+ ident | + ident*iscale + ishift])
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

template <bool RELU, int IDENT>  // IDENT: 0 none, 1 plain tensor, 2 tensor with its own scale/shift
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const bf16* __restrict__ y, const float* __restrict__ consts,
                                                          const bf16* __restrict__ ident,
                                                          const float* __restrict__ iconsts, bf16* __restrict__ out,
                                                          uint8_t* __restrict__ bits, long long nvec, int C) {
    const int cvec = C >> 3;
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < nvec; v += gridDim.x * 256ll) {
        const int c = static_cast<int>(v % cvec) << 3;
        float sc[8], sh[8];
        load8f(consts + 2 * C + c, sc);
        load8f(consts + 3 * C + c, sh);
        const bf16x8 yv = *reinterpret_cast<const bf16x8*>(y + v * 8);
        float z[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = bf16_to_f32(yv[e]) * sc[e] + sh[e];
        if (IDENT == 1) {
            const bf16x8 iv = *reinterpret_cast<const bf16x8*>(ident + v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] += bf16_to_f32(iv[e]);
        } else if (IDENT == 2) {
            float isc[8], ish[8];
            load8f(iconsts + 2 * C + c, isc);
            load8f(iconsts + 3 * C + c, ish);
            const bf16x8 iv = *reinterpret_cast<const bf16x8*>(ident + v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] += bf16_to_f32(iv[e]) * isc[e] + ish[e];
        }
        bf16x8 o;
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = f32_to_bf16(RELU ? fmaxf(z[e], 0.f) : z[e]);
            m |= (bf16_to_f32(o[e]) > 0.f ? 1u : 0u) << e;
        }
        *reinterpret_cast<bf16x8*>(out + v * 8) = o;
        if (bits != nullptr) bits[v] = static_cast<uint8_t>(m);  // ReLU mask, one bit per element, for backward
    }
}

// Streaming form of the element-wise passes, used when C/8 is a power of two <= 256 (every ResNet width): a
// workgroup walks tiles of 256*U vectors, so the channel chunk of a thread never changes and the per-channel
// constants stay in registers; the U loads of a tile are issued before the first use.  Tensors far larger than
// the caches are read and written with non-temporal accesses.  Measured against the grid-stride form on 2048-image
// ResNet-50 activations: 4.2-4.9 -> 5.9-6.1 TB/s (bench/stream_rate.hip).
constexpr int EW_U = 4;
constexpr long long EW_NT_BYTES = 64ll << 20;

template <bool NT>
__device__ __forceinline__ bf16x8 ew_load(const bf16* p, long long v) {
    const bf16x8* q = reinterpret_cast<const bf16x8*>(p) + v;
    return NT ? __builtin_nontemporal_load(q) : *q;
}
template <bool NT>
__device__ __forceinline__ void ew_store(bf16* p, long long v, bf16x8 x) {
    bf16x8* q = reinterpret_cast<bf16x8*>(p) + v;
    if (NT) __builtin_nontemporal_store(x, q);
    else *q = x;
}

static inline bool ew_tile_ok(int C) {
    const int cvec = C >> 3;
    return cvec >= 1 && cvec <= 256 && (cvec & (cvec - 1)) == 0;
}
static inline int ew_tile_grid(long long nvec, int u) {
    const long long g = (nvec + 256ll * u - 1) / (256ll * u);
    return static_cast<int>(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

template <bool RELU, int IDENT, bool NT>
__global__ __launch_bounds__(256) void bn_apply_fwd_tile_kernel(const bf16* __restrict__ y, const float* __restrict__ consts,
                                                               const bf16* __restrict__ ident,
                                                               const float* __restrict__ iconsts, bf16* __restrict__ out,
                                                               uint8_t* __restrict__ bits, long long nvec, int C) {
    const int c = (threadIdx.x & ((C >> 3) - 1)) << 3;
    float sc[8], sh[8], isc[8], ish[8];
    load8f(consts + 2 * C + c, sc);
    load8f(consts + 3 * C + c, sh);
    if (IDENT == 2) {
        load8f(iconsts + 2 * C + c, isc);
        load8f(iconsts + 3 * C + c, ish);
    }
    constexpr long long TILE = 256ll * EW_U;
    for (long long base = blockIdx.x * TILE + threadIdx.x; base < nvec; base += gridDim.x * TILE) {
        bf16x8 yv[EW_U], iv[EW_U];
#pragma unroll
        for (int u = 0; u < EW_U; ++u) {
            const long long v = base + u * 256;
            if (v < nvec) {
                yv[u] = ew_load<NT>(y, v);
                if (IDENT != 0) iv[u] = ew_load<NT>(ident, v);
            }
        }
#pragma unroll
        for (int u = 0; u < EW_U; ++u) {
            const long long v = base + u * 256;
            if (v < nvec) {
                bf16x8 o;
                unsigned m = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float z = bf16_to_f32(yv[u][e]) * sc[e] + sh[e];
                    if (IDENT == 1) z += bf16_to_f32(iv[u][e]);
                    if (IDENT == 2) z += bf16_to_f32(iv[u][e]) * isc[e] + ish[e];
                    o[e] = f32_to_bf16(RELU ? fmaxf(z, 0.f) : z);
                    m |= (bf16_to_f32(o[e]) > 0.f ? 1u : 0u) << e;
                }
                ew_store<NT>(out, v, o);
                if (bits != nullptr) bits[v] = static_cast<uint8_t>(m);
            }
        }
    }
}

static inline int ew_grid(long long nvec) {
    long long g = (nvec + 255) / 256;
    return static_cast<int>(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// bits (optional): [numel / 8] bytes receiving the ReLU mask of the output (bit e of byte v = out[8v + e] > 0).
PDT_API int pdt_bn_apply_fwd(const void* y, const float* consts, const void* ident, const float* iconsts, void* out,
                             void* bits, long long numel, int C, int relu, void* stream) {
    if (C <= 0 || (C % 8) != 0 || numel <= 0 || (numel % C) != 0) return -1;
    const long long nvec = numel / 8;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bf16* yy = static_cast<const bf16*>(y);
    const bf16* id = static_cast<const bf16*>(ident);
    bf16* o = static_cast<bf16*>(out);
    uint8_t* bt = static_cast<uint8_t*>(bits);
    if (ew_tile_ok(C)) {
        const dim3 tg(ew_tile_grid(nvec, EW_U)), tb(256);
        const int id_mode = ident == nullptr ? 0 : (iconsts == nullptr ? 1 : 2);
        const bool nt = numel * 2 >= EW_NT_BYTES;
#define PDT_FWD_T(RL, IDM, NTV) \
    hipLaunchKernelGGL((bn_apply_fwd_tile_kernel<RL, IDM, NTV>), tg, tb, 0, s, yy, consts, id, iconsts, o, bt, nvec, C)
#define PDT_FWD_T2(RL, IDM)             \
    do {                                \
        if (nt) PDT_FWD_T(RL, IDM, true); \
        else PDT_FWD_T(RL, IDM, false);  \
    } while (0)
        if (relu) {
            if (id_mode == 0) PDT_FWD_T2(true, 0);
            else if (id_mode == 1) PDT_FWD_T2(true, 1);
            else PDT_FWD_T2(true, 2);
        } else {
            if (id_mode == 0) PDT_FWD_T2(false, 0);
            else if (id_mode == 1) PDT_FWD_T2(false, 1);
            else PDT_FWD_T2(false, 2);
        }
#undef PDT_FWD_T2
#undef PDT_FWD_T
        return pdt_launch_status();
    }
    const dim3 g(ew_grid(nvec)), b(256);
    if (ident == nullptr) {
        if (relu) hipLaunchKernelGGL((bn_apply_fwd_kernel<true, 0>), g, b, 0, s, yy, consts, id, iconsts, o, bt, nvec, C);
        else hipLaunchKernelGGL((bn_apply_fwd_kernel<false, 0>), g, b, 0, s, yy, consts, id, iconsts, o, bt, nvec, C);
    } else if (iconsts == nullptr) {
        if (relu) hipLaunchKernelGGL((bn_apply_fwd_kernel<true, 1>), g, b, 0, s, yy, consts, id, iconsts, o, bt, nvec, C);
        else hipLaunchKernelGGL((bn_apply_fwd_kernel<false, 1>), g, b, 0, s, yy, consts, id, iconsts, o, bt, nvec, C);
    } else {
        if (relu) hipLaunchKernelGGL((bn_apply_fwd_kernel<true, 2>), g, b, 0, s, yy, consts, id, iconsts, o, bt, nvec, C);
        else hipLaunchKernelGGL((bn_apply_fwd_kernel<false, 2>), g, b, 0, s, yy, consts, id, iconsts, o, bt, nvec, C);
    }
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// backward reduction:  slots[S][3][C] += (sum dz, sum dz*xhat, sum dz*xhat2),  dz = dout * [out > 0]
//   workgroup = 32 row-lanes x 8 chunks (64 channels); grid.x = row blocks, grid.y = channel groups
// ------------------------------------------------------------------------------------------------
// MASK: 0 = dz is dout, 1 = mask from the stored output (out > 0), 2 = mask recomputed as y*scale + shift > 0,
//       3 = mask from the bit array written by the forward kernel (passed through the `out` pointer)
template <int MASK, bool SECOND>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ out,
                                                           const bf16* __restrict__ y, const float* __restrict__ consts,
                                                           const bf16* __restrict__ y2,
                                                           const float* __restrict__ consts2, float* slots,
                                                           int M, int C, int rows_per_block, float* partials) {
    __shared__ float red[32][3][64];
    const int chunk = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.y * 64 + chunk * 8;
    const bool cok = c < C;
    float mu[8], rs[8], mu2[8], rs2[8], sc[8], sh[8];
    if (cok) {
        load8f(consts + c, mu);
        load8f(consts + C + c, rs);
        if (MASK == 2) {
            load8f(consts + 2 * C + c, sc);
            load8f(consts + 3 * C + c, sh);
        }
        if (SECOND) {
            load8f(consts2 + c, mu2);
            load8f(consts2 + C + c, rs2);
        }
    }
    float s1[8], s2[8], s3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = s3[e] = 0.f;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    if (cok) {
        for (int r = r0 + rl; r < r1; r += 32) {
            const size_t off = static_cast<size_t>(r) * C + c;
            // every operand is read exactly once by this kernel: non-temporal loads keep them out of the caches
            const bf16x8 dv = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(dout + off));
            const bf16x8 yv = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(y + off));
            bf16x8 ov, y2v;
            unsigned mbits = 0;
            if (MASK == 1) ov = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(out + off));
            if (MASK == 3) mbits = reinterpret_cast<const uint8_t*>(out)[off >> 3];
            if (SECOND) y2v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(y2 + off));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float dz = bf16_to_f32(dv[e]);
                const float yf = bf16_to_f32(yv[e]);
                if (MASK == 1) dz = bf16_to_f32(ov[e]) > 0.f ? dz : 0.f;
                if (MASK == 2) dz = (yf * sc[e] + sh[e]) > 0.f ? dz : 0.f;
                if (MASK == 3) dz = ((mbits >> e) & 1u) ? dz : 0.f;
                s1[e] += dz;
                s2[e] += dz * (yf - mu[e]) * rs[e];
                if (SECOND) s3[e] += dz * (bf16_to_f32(y2v[e]) - mu2[e]) * rs2[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[rl][0][chunk * 8 + e] = s1[e];
        red[rl][1][chunk * 8 + e] = s2[e];
        red[rl][2][chunk * 8 + e] = s3[e];
    }
    __syncthreads();
    constexpr int NS = SECOND ? 3 : 2;
    if (threadIdx.x < NS * 64) {
        const int which = threadIdx.x >> 6, col = threadIdx.x & 63;
        float t = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) t += red[r][which][col];
        const int cc = blockIdx.y * 64 + col;
        if (cc < C) {
            if (partials != nullptr) {  // deterministic mode: one writer per element, folded by fold_partials_kernel
                partials[(static_cast<size_t>(blockIdx.x) * 3 + which) * C + cc] = t;
            } else {
                float* slot = slots + static_cast<size_t>(blockIdx.x % PDT_STAT_SLOTS) * 3 * C;
                __hip_atomic_fetch_add(slot + which * C + cc, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

static inline int reduce_rows_per_block(int M, int C) {
    const long long groups = (C + 63) / 64;
    long long blocks_x = 2048 / groups;
    if (blocks_x < 1) blocks_x = 1;
    long long rpb = (M + blocks_x - 1) / blocks_x;
    rpb = (rpb + 31) / 32 * 32;
    return static_cast<int>(rpb < 32 ? 32 : rpb);
}

// slots: zero-initialised [PDT_STAT_SLOTS][3][C].  mask_mode: 0 none, 1 from `out`, 2 recomputed from y.
PDT_API int pdt_bn_bwd_reduce(const void* dout, const void* out, const void* y, const float* consts, const void* y2,
                              const float* consts2, float* slots, int M, int C, int mask_mode, void* stream) {
    if (M <= 0 || C <= 0 || (C % 8) != 0 || mask_mode < 0 || mask_mode > 3) return -1;
    if ((mask_mode == 1 || mask_mode == 3) && out == nullptr) return -1;
    const int rpb = reduce_rows_per_block(M, C);
    const dim3 g(pdt_cdiv(M, rpb), pdt_cdiv(C, 64)), b(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bf16 *d = static_cast<const bf16*>(dout), *o = static_cast<const bf16*>(out), *yy = static_cast<const bf16*>(y),
               *yy2 = static_cast<const bf16*>(y2);
    const bool second = y2 != nullptr;
    float* partials = nullptr;
    if (pdt_deterministic_flag) {
        partials = det_partials(static_cast<size_t>(g.x) * 3 * C);
        if (partials == nullptr) return static_cast<int>(hipErrorOutOfMemory);
    }
#define PDT_RED(MK, SC) hipLaunchKernelGGL((bn_bwd_reduce_kernel<MK, SC>), g, b, 0, s, d, o, yy, consts, yy2, consts2, slots, M, C, rpb, partials)
    if (mask_mode == 1) { if (second) PDT_RED(1, true); else PDT_RED(1, false); }
    else if (mask_mode == 2) { if (second) PDT_RED(2, true); else PDT_RED(2, false); }
    else if (mask_mode == 3) { if (second) PDT_RED(3, true); else PDT_RED(3, false); }
    else { if (second) PDT_RED(0, true); else PDT_RED(0, false); }
#undef PDT_RED
    if (partials != nullptr) launch_fold(partials, static_cast<int>(g.x), 3, second ? 3 : 2, C, slots, s);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Forward statistics by a pass of their own (deterministic mode, where the convolution epilogues do not accumulate
// them with atomics): row block b stores sum(y - P), sum((y - P)^2) over its rows to partials[b][2][C], and
// fold_partials_kernel adds them into the slots of [PDT_STAT_SLOTS + 1][2][C] in block order; the pivot P (or zero)
// goes to the first row of the last slot, as the epilogues do it.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_stats_rows_kernel(const bf16* __restrict__ y, const float* __restrict__ pivot,
                                                           float* __restrict__ slots, int M, int C, int rows_per_block,
                                                           float* __restrict__ partials) {
    __shared__ float red[32][2][64];
    const int chunk = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.y * 64 + chunk * 8;
    const bool cok = c < C;
    float pv[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        pv[e] = (cok && pivot != nullptr) ? pivot[c + e] : 0.f;
        s1[e] = s2[e] = 0.f;
    }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    if (cok) {
        for (int r = r0 + rl; r < r1; r += 32) {
            const bf16x8 v = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(y + static_cast<size_t>(r) * C + c));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = bf16_to_f32(v[e]) - pv[e];
                s1[e] += d;
                s2[e] += d * d;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[rl][0][chunk * 8 + e] = s1[e];
        red[rl][1][chunk * 8 + e] = s2[e];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, col = threadIdx.x & 63;
        float t = 0.f;
        for (int r = 0; r < 32; ++r) t += red[r][which][col];
        const int cc = blockIdx.y * 64 + col;
        if (cc < C) {
            partials[(static_cast<size_t>(blockIdx.x) * 2 + which) * C + cc] = t;
            if (blockIdx.x == 0 && which == 0) slots[static_cast<size_t>(PDT_STAT_SLOTS) * 2 * C + cc] = pivot != nullptr ? pivot[cc] : 0.f;
        }
    }
}

PDT_API int pdt_bn_stats_rows(const void* y, const float* pivot, float* slots, int M, int C, void* stream) {
    if (M <= 0 || C <= 0 || (C % 8) != 0) return -1;
    const int rpb = reduce_rows_per_block(M, C);
    const int nblk = pdt_cdiv(M, rpb);
    float* partials = det_partials(static_cast<size_t>(nblk) * 2 * C);
    if (partials == nullptr) return static_cast<int>(hipErrorOutOfMemory);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bn_stats_rows_kernel, dim3(nblk, pdt_cdiv(C, 64)), dim3(256), 0, s, static_cast<const bf16*>(y), pivot,
                       slots, M, C, rpb, partials);
    launch_fold(partials, nblk, 2, 2, C, slots, s);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// bn_bwd_finalize: slots -> dgamma, dbeta, coef[3][C] with dy = coef0*dz + coef1*y + coef2
//   `which` selects the dz*xhat sum (1 = first BatchNorm, 2 = second)
// ------------------------------------------------------------------------------------------------
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ slots, int C, int which, float count,
                                       const float* __restrict__ gamma, const float* __restrict__ consts,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef,
                                       int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sdz = 0.f, sdzx = 0.f;
    for (int s = 0; s < PDT_STAT_SLOTS; ++s) {
        sdz += slots[(s * 3 + 0) * C + c];
        sdzx += slots[(s * 3 + which) * C + c];
    }
    const float mean = consts[c], rstd = consts[C + c];
    const float g = gamma != nullptr ? gamma[c] : 1.f;
    if (dgamma != nullptr) dgamma[c] = accumulate ? dgamma[c] + sdzx : sdzx;
    if (dbeta != nullptr) dbeta[c] = accumulate ? dbeta[c] + sdz : sdz;
    const float k1 = g * rstd;
    const float b = -k1 * rstd * sdzx / count;
    coef[c] = k1;
    coef[C + c] = b;
    coef[2 * C + c] = -k1 * sdz / count - b * mean;
}

// accumulate != 0: dgamma / dbeta are added to (gradient buffers that already hold a value).
PDT_API int pdt_bn_bwd_finalize(const float* slots, int C, int which, float count, const float* gamma,
                                const float* consts, float* dgamma, float* dbeta, float* coef, int accumulate,
                                void* stream) {
    if (C <= 0 || (which != 1 && which != 2)) return -1;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(pdt_cdiv(C, 64)), dim3(64), 0, static_cast<hipStream_t>(stream), slots, C,
                       which, count, gamma, consts, dgamma, dbeta, coef, accumulate);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// bn_bwd_apply:  dz = dout*[out>0];  dy = c0*dz + c1*y + c2;  optional dz store; optional second dy2
// ------------------------------------------------------------------------------------------------
template <int MASK, bool WRITE_DZ, bool SECOND>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ out,
                                                          const bf16* __restrict__ y, const float* __restrict__ coef,
                                                          const float* __restrict__ consts,
                                                          bf16* __restrict__ dy, bf16* __restrict__ dz_out,
                                                          const bf16* __restrict__ y2, const float* __restrict__ coef2,
                                                          bf16* __restrict__ dy2, long long nvec, int C) {
    const int cvec = C >> 3;
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < nvec; v += gridDim.x * 256ll) {
        const int c = static_cast<int>(v % cvec) << 3;
        float k0[8], k1[8], k2[8];
        load8f(coef + c, k0);
        load8f(coef + C + c, k1);
        load8f(coef + 2 * C + c, k2);
        const bf16x8 dv = *reinterpret_cast<const bf16x8*>(dout + v * 8);
        const bf16x8 yv = *reinterpret_cast<const bf16x8*>(y + v * 8);
        float dz[8];
        if (MASK == 1) {
            const bf16x8 ov = *reinterpret_cast<const bf16x8*>(out + v * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) dz[e] = bf16_to_f32(ov[e]) > 0.f ? bf16_to_f32(dv[e]) : 0.f;
        } else if (MASK == 2) {
            float sc[8], sh[8];
            load8f(consts + 2 * C + c, sc);
            load8f(consts + 3 * C + c, sh);
#pragma unroll
            for (int e = 0; e < 8; ++e) dz[e] = (bf16_to_f32(yv[e]) * sc[e] + sh[e]) > 0.f ? bf16_to_f32(dv[e]) : 0.f;
        } else if (MASK == 3) {
            const unsigned mbits = reinterpret_cast<const uint8_t*>(out)[v];
#pragma unroll
            for (int e = 0; e < 8; ++e) dz[e] = ((mbits >> e) & 1u) ? bf16_to_f32(dv[e]) : 0.f;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) dz[e] = bf16_to_f32(dv[e]);
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(k0[e] * dz[e] + k1[e] * bf16_to_f32(yv[e]) + k2[e]);
        *reinterpret_cast<bf16x8*>(dy + v * 8) = o;
        if (WRITE_DZ) {
            bf16x8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = f32_to_bf16(dz[e]);
            *reinterpret_cast<bf16x8*>(dz_out + v * 8) = z;
        }
        if (SECOND) {
            float q0[8], q1[8], q2[8];
            load8f(coef2 + c, q0);
            load8f(coef2 + C + c, q1);
            load8f(coef2 + 2 * C + c, q2);
            const bf16x8 y2v = *reinterpret_cast<const bf16x8*>(y2 + v * 8);
            bf16x8 o2;
#pragma unroll
            for (int e = 0; e < 8; ++e) o2[e] = f32_to_bf16(q0[e] * dz[e] + q1[e] * bf16_to_f32(y2v[e]) + q2[e]);
            *reinterpret_cast<bf16x8*>(dy2 + v * 8) = o2;
        }
    }
}

template <int MASK, bool WRITE_DZ, bool SECOND, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_apply_tile_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ out,
                                                               const bf16* __restrict__ y, const float* __restrict__ coef,
                                                               const float* __restrict__ consts, bf16* __restrict__ dy,
                                                               bf16* __restrict__ dz_out, const bf16* __restrict__ y2,
                                                               const float* __restrict__ coef2, bf16* __restrict__ dy2,
                                                               long long nvec, int C) {
    constexpr int U = SECOND ? 2 : EW_U;
    const int c = (threadIdx.x & ((C >> 3) - 1)) << 3;
    float k0[8], k1[8], k2[8], sc[8], sh[8], q0[8], q1[8], q2[8];
    load8f(coef + c, k0);
    load8f(coef + C + c, k1);
    load8f(coef + 2 * C + c, k2);
    if (MASK == 2) {
        load8f(consts + 2 * C + c, sc);
        load8f(consts + 3 * C + c, sh);
    }
    if (SECOND) {
        load8f(coef2 + c, q0);
        load8f(coef2 + C + c, q1);
        load8f(coef2 + 2 * C + c, q2);
    }
    constexpr long long TILE = 256ll * U;
    for (long long base = blockIdx.x * TILE + threadIdx.x; base < nvec; base += gridDim.x * TILE) {
        bf16x8 dv[U], yv[U], ov[U], y2v[U];
        unsigned mbits[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long v = base + u * 256;
            if (v < nvec) {
                dv[u] = ew_load<NT>(dout, v);
                yv[u] = ew_load<NT>(y, v);
                if (MASK == 1) ov[u] = ew_load<NT>(out, v);
                if (MASK == 3) mbits[u] = reinterpret_cast<const uint8_t*>(out)[v];
                if (SECOND) y2v[u] = ew_load<NT>(y2, v);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long v = base + u * 256;
            if (v < nvec) {
                float dz[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = bf16_to_f32(dv[u][e]);
                    if (MASK == 1) dz[e] = bf16_to_f32(ov[u][e]) > 0.f ? d : 0.f;
                    else if (MASK == 2) dz[e] = (bf16_to_f32(yv[u][e]) * sc[e] + sh[e]) > 0.f ? d : 0.f;
                    else if (MASK == 3) dz[e] = ((mbits[u] >> e) & 1u) ? d : 0.f;
                    else dz[e] = d;
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(k0[e] * dz[e] + k1[e] * bf16_to_f32(yv[u][e]) + k2[e]);
                ew_store<NT>(dy, v, o);
                if (WRITE_DZ) {
                    bf16x8 z;
#pragma unroll
                    for (int e = 0; e < 8; ++e) z[e] = f32_to_bf16(dz[e]);
                    ew_store<NT>(dz_out, v, z);
                }
                if (SECOND) {
                    bf16x8 o2;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o2[e] = f32_to_bf16(q0[e] * dz[e] + q1[e] * bf16_to_f32(y2v[u][e]) + q2[e]);
                    ew_store<NT>(dy2, v, o2);
                }
            }
        }
    }
}

// mask_mode as in pdt_bn_bwd_reduce; `consts` ([4][C] from bn_finalize) is needed for mask_mode 2.
PDT_API int pdt_bn_bwd_apply(const void* dout, const void* out, const void* y, const float* coef, const float* consts,
                             void* dy, void* dz_out, const void* y2, const float* coef2, void* dy2, long long numel,
                             int C, int mask_mode, void* stream) {
    if (C <= 0 || (C % 8) != 0 || numel <= 0 || (numel % C) != 0) return -1;
    if (mask_mode < 0 || mask_mode > 3 || ((mask_mode == 1 || mask_mode == 3) && out == nullptr) ||
        (mask_mode == 2 && consts == nullptr))
        return -1;
    if (dz_out != nullptr && y2 != nullptr) return -1;  // identity is either plain (needs dz) or projected (needs dy2)
    const long long nvec = numel / 8;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 g(ew_grid(nvec)), b(256);
    const bf16 *d = static_cast<const bf16*>(dout), *o = static_cast<const bf16*>(out), *yy = static_cast<const bf16*>(y),
               *yy2 = static_cast<const bf16*>(y2);
    bf16 *pdy = static_cast<bf16*>(dy), *pdz = static_cast<bf16*>(dz_out), *pdy2 = static_cast<bf16*>(dy2);
    const bool tile = ew_tile_ok(C), nt = numel * 2 >= EW_NT_BYTES;
    const dim3 tg(ew_tile_grid(nvec, y2 != nullptr ? 2 : EW_U));
#define PDT_APPLY(MK, WZ, SC)                                                                                              \
    do {                                                                                                                   \
        if (!tile)                                                                                                         \
            hipLaunchKernelGGL((bn_bwd_apply_kernel<MK, WZ, SC>), g, b, 0, s, d, o, yy, coef, consts, pdy, pdz, yy2, coef2, \
                               pdy2, nvec, C);                                                                             \
        else if (nt)                                                                                                       \
            hipLaunchKernelGGL((bn_bwd_apply_tile_kernel<MK, WZ, SC, true>), tg, b, 0, s, d, o, yy, coef, consts, pdy, pdz, \
                               yy2, coef2, pdy2, nvec, C);                                                                 \
        else                                                                                                               \
            hipLaunchKernelGGL((bn_bwd_apply_tile_kernel<MK, WZ, SC, false>), tg, b, 0, s, d, o, yy, coef, consts, pdy,     \
                               pdz, yy2, coef2, pdy2, nvec, C);                                                            \
    } while (0)
#define PDT_APPLY_M(MK)                                   \
    do {                                                  \
        if (dz_out != nullptr) PDT_APPLY(MK, true, false); \
        else if (y2 != nullptr) PDT_APPLY(MK, false, true); \
        else PDT_APPLY(MK, false, false);                 \
    } while (0)
    if (mask_mode == 1) PDT_APPLY_M(1);
    else if (mask_mode == 2) PDT_APPLY_M(2);
    else if (mask_mode == 3) PDT_APPLY_M(3);
    else PDT_APPLY_M(0);
#undef PDT_APPLY_M
#undef PDT_APPLY
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Stem tail: out = maxpool3x3/s2/p1(relu(y*scale + shift)), idx = winning tap (0..8) per channel
// ------------------------------------------------------------------------------------------------
// HOIST: C/8 divides 256, so a thread's channel chunk is the same in every iteration and the constants stay in registers.
template <bool HOIST>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const bf16* __restrict__ y, const float* __restrict__ consts,
                                                                 bf16* __restrict__ out, uint8_t* __restrict__ idx,
                                                                 bf16* __restrict__ ywin, int NB, int H, int W, int C,
                                                                 int Ho, int Wo, FastDiv div_wo, FastDiv div_ho, FastDiv div_cvec) {
    const uint32_t cvec = C >> 3;
    const uint32_t nvec = static_cast<uint32_t>(NB) * Ho * Wo * cvec;
    float sc[8], sh[8];
    auto load_consts = [&](int c) {
        load8f(consts + 2 * C + c, sc);
        load8f(consts + 3 * C + c, sh);
    };
    if (HOIST) load_consts(static_cast<int>(threadIdx.x & (cvec - 1)) << 3);
    for (uint32_t v = blockIdx.x * 256u + threadIdx.x; v < nvec; v += gridDim.x * 256u) {
        const uint32_t pix = fastdiv(v, div_cvec);
        const int c = static_cast<int>(v - pix * cvec) << 3;
        const uint32_t t = fastdiv(pix, div_wo);
        const int wo = static_cast<int>(pix - t * Wo);
        const uint32_t n = fastdiv(t, div_ho);
        const int ho = static_cast<int>(t - n * Ho);
        if (!HOIST) load_consts(c);
        float best[8];
        int bi[8];
        bf16x8 yw;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            best[e] = -1.f;  // relu output is >= 0, so any in-bounds tap wins over this
            bi[e] = 0;
            yw[e] = f32_to_bf16(0.f);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int h = ho * 2 - 1 + r;
            if (h < 0 || h >= H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int w = wo * 2 - 1 + s;
                if (w < 0 || w >= W) continue;
                const bf16x8 yv = *reinterpret_cast<const bf16x8*>(y + ((static_cast<size_t>(n) * H + h) * W + w) * C + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float z = fmaxf(bf16_to_f32(yv[e]) * sc[e] + sh[e], 0.f);
                    if (z > best[e]) {
                        best[e] = z;
                        bi[e] = r * 3 + s;
                        yw[e] = yv[e];
                    }
                }
            }
        }
        bf16x8 o;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = f32_to_bf16(best[e]);
            if (e < 4) lo |= static_cast<uint32_t>(bi[e]) << (8 * e);
            else hi |= static_cast<uint32_t>(bi[e]) << (8 * (e - 4));
        }
        const size_t ov = static_cast<size_t>(v) * 8;
        *reinterpret_cast<bf16x8*>(out + ov) = o;
        *reinterpret_cast<uint2*>(idx + ov) = make_uint2(lo, hi);
        if (ywin != nullptr) *reinterpret_cast<bf16x8*>(ywin + ov) = yw;
    }
}

// ywin (optional, shape of out): the pre-normalisation value of each window's winner.  BatchNorm's backward sums over
// the un-pooled gradient equal sums over the pooled one taken at the winners, so they need only (dout, ywin).
PDT_API int pdt_bn_relu_maxpool_fwd(const void* y, const float* consts, void* out, void* idx, void* ywin, int NB, int H,
                                    int W, int C, void* stream) {
    if (NB <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) != 0) return -1;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int cvec = C / 8;
    const long long nvec = static_cast<long long>(NB) * Ho * Wo * cvec;
    if (nvec >= (1ll << 31)) return -1;
    const bool hoist = cvec <= 256 && (cvec & (cvec - 1)) == 0;
    auto kernel = hoist ? bn_relu_maxpool_fwd_kernel<true> : bn_relu_maxpool_fwd_kernel<false>;
    hipLaunchKernelGGL(kernel, dim3(ew_grid(nvec)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16*>(y), consts, static_cast<bf16*>(out), static_cast<uint8_t*>(idx),
                       static_cast<bf16*>(ywin), NB, H, W, C, Ho, Wo, make_fastdiv(Wo), make_fastdiv(Ho), make_fastdiv(cvec));
    return pdt_launch_status();
}

// Backward of the stem tail.  The pooled gradient is never materialised at the input resolution:
//   dz[n,h,w,c] = [z > 0] * sum over pooling windows whose winning tap is (h,w) of dout
//   pass 1 (sums):  sum dz and sum dz*xhat over input pixels equal the sums of dout*[z_win > 0] and
//                   dout*[z_win > 0]*xhat(y_win) over pooled outputs: bn_bwd_reduce on (dout, ywin), both pooled-size
//   pass 2 (apply): dy = c0*dz + c1*y + c2 with dz gathered from (dout, idx), a quarter of y's size
// Second pass: dy = c0*dz + c1*y + c2 with dz gathered from (dout, idx).  A thread owns a 2x2 quad of input pixels
// (rows 2a, 2a+1; columns 2b, 2b+1) and 8 channels: the quad is covered by the four windows (a + di, b + dj), each
// loaded once; window (0,0) can have its winner on any of the four pixels (taps 4, 5, 7, 8), window (0,1) on the right
// column (taps 3, 6), window (1,0) on the lower row (taps 1, 2), window (1,1) on the lower right pixel (tap 0).
// HOIST: C/8 divides 256, so a thread's channel chunk is the same in every iteration and the constants stay in
// registers.
template <bool HOIST>
__global__ __launch_bounds__(256) void maxpool_bn_bwd_apply_kernel(const bf16* __restrict__ dout,
                                                                  const uint8_t* __restrict__ idx,
                                                                  const bf16* __restrict__ y,
                                                                  const float* __restrict__ consts,
                                                                  const float* __restrict__ coef, bf16* __restrict__ dy,
                                                                  int NB, int H, int W, int C, int Ho, int Wo,
                                                                  FastDiv div_wa, FastDiv div_ha, FastDiv div_cvec) {
    const uint32_t cvec = C >> 3;
    const uint32_t Ha = (H + 1) >> 1, Wa = (W + 1) >> 1;
    const uint32_t nvec = static_cast<uint32_t>(NB) * Ha * Wa * cvec;
    float sc[8], sh[8], k0[8], k1[8], k2[8];
    auto load_consts = [&](int c) {
        load8f(consts + 2 * C + c, sc);
        load8f(consts + 3 * C + c, sh);
        load8f(coef + c, k0);
        load8f(coef + C + c, k1);
        load8f(coef + 2 * C + c, k2);
    };
    if (HOIST) load_consts(static_cast<int>(threadIdx.x & (cvec - 1)) << 3);
    for (uint32_t v = blockIdx.x * 256u + threadIdx.x; v < nvec; v += gridDim.x * 256u) {
        const uint32_t quad = fastdiv(v, div_cvec);
        const int c = static_cast<int>(v - quad * cvec) << 3;
        const uint32_t t = fastdiv(quad, div_wa);
        const int b = static_cast<int>(quad - t * Wa);
        const uint32_t n = fastdiv(t, div_ha);
        const int a = static_cast<int>(t - n * Ha);
        if (!HOIST) load_consts(c);
        // the four windows (clamped addresses; a window outside the pooled image contributes nothing)
        const bool v1 = a + 1 < Ho, w1 = b + 1 < Wo;
        uint2 iv[4];
        bf16x8 dv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oh = min(a + (q >> 1), Ho - 1), ow = min(b + (q & 1), Wo - 1);
            const size_t o = ((static_cast<size_t>(n) * Ho + oh) * Wo + ow) * C + c;
            iv[q] = *reinterpret_cast<const uint2*>(idx + o);
            dv[q] = *reinterpret_cast<const bf16x8*>(dout + o);
        }
        // the quad's activations (pixels beyond an odd-sized image are skipped)
        const bool r1 = 2 * a + 1 < H, c1 = 2 * b + 1 < W;
        const size_t p00 = ((static_cast<size_t>(n) * H + 2 * a) * W + 2 * b) * C + c;
        const size_t poff[4] = {p00, p00 + C, p00 + static_cast<size_t>(W) * C, p00 + static_cast<size_t>(W) * C + C};
        const bool pok[4] = {true, c1, r1, r1 && c1};
        bf16x8 yv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (pok[q]) yv[q] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(y + poff[q]));
        float g[4][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            auto tap_of = [&](int q) {
                const uint32_t word = e < 4 ? iv[q].x : iv[q].y;
                return static_cast<int>((word >> (8 * (e & 3))) & 0xff);
            };
            const int t00 = tap_of(0), t01 = w1 ? tap_of(1) : -1, t10 = v1 ? tap_of(2) : -1, t11 = (v1 && w1) ? tap_of(3) : -1;
            const float d00 = bf16_to_f32(dv[0][e]), d01 = bf16_to_f32(dv[1][e]), d10 = bf16_to_f32(dv[2][e]),
                        d11 = bf16_to_f32(dv[3][e]);
            g[0][e] = (t00 == 4 ? d00 : 0.f);
            g[1][e] = (t00 == 5 ? d00 : 0.f) + (t01 == 3 ? d01 : 0.f);
            g[2][e] = (t00 == 7 ? d00 : 0.f) + (t10 == 1 ? d10 : 0.f);
            g[3][e] = (t00 == 8 ? d00 : 0.f) + (t01 == 6 ? d01 : 0.f) + (t10 == 2 ? d10 : 0.f) + (t11 == 0 ? d11 : 0.f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!pok[q]) continue;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float yf = bf16_to_f32(yv[q][e]);
                const float d = (yf * sc[e] + sh[e]) > 0.f ? g[q][e] : 0.f;
                o[e] = f32_to_bf16(k0[e] * d + k1[e] * yf + k2[e]);
            }
            __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(dy + poff[q]));
        }
    }
}

PDT_API int pdt_maxpool_bn_bwd_apply(const void* dout, const void* idx, const void* y, const float* consts,
                                     const float* coef, void* dy, int NB, int H, int W, int C, void* stream) {
    if (NB <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 8) != 0) return -1;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int Ha = (H + 1) / 2, Wa = (W + 1) / 2, cvec = C / 8;
    if (static_cast<long long>(NB) * H * W * cvec >= (1ll << 31)) return -1;
    const long long nvec = static_cast<long long>(NB) * Ha * Wa * cvec;
    long long grid = (nvec + 255) / 256;
    if (grid > 16384) grid = 16384;
    const bool hoist = cvec <= 256 && (cvec & (cvec - 1)) == 0;
    auto kernel = hoist ? maxpool_bn_bwd_apply_kernel<true> : maxpool_bn_bwd_apply_kernel<false>;
    hipLaunchKernelGGL(kernel, dim3(static_cast<int>(grid)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16*>(dout), static_cast<const uint8_t*>(idx), static_cast<const bf16*>(y), consts,
                       coef, static_cast<bf16*>(dy), NB, H, W, C, Ho, Wo, make_fastdiv(Wa), make_fastdiv(Ha),
                       make_fastdiv(cvec));
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Global average pool over H*W (NHWC):  out[n][c] = mean_hw a[n][hw][c];  backward broadcasts g/HW
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const bf16* __restrict__ a, bf16* __restrict__ out, int NB, int HW,
                                                         int C) {
    const int cvec = C >> 3;
    const long long nvec = static_cast<long long>(NB) * cvec;
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < nvec; v += gridDim.x * 256ll) {
        const int n = static_cast<int>(v / cvec), c = static_cast<int>(v % cvec) << 3;
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
        for (int i = 0; i < HW; ++i) {
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(a + (static_cast<size_t>(n) * HW + i) * C + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += bf16_to_f32(x[e]);
        }
        bf16x8 o;
        const float inv = 1.f / HW;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(s[e] * inv);
        *reinterpret_cast<bf16x8*>(out + static_cast<size_t>(n) * C + c) = o;
    }
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const bf16* __restrict__ g, bf16* __restrict__ da, int NB, int HW,
                                                         int C) {
    const int cvec = C >> 3;
    const long long nvec = static_cast<long long>(NB) * HW * cvec;
    const float inv = 1.f / HW;
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < nvec; v += gridDim.x * 256ll) {
        const int c = static_cast<int>(v % cvec) << 3;
        const int n = static_cast<int>(v / (static_cast<long long>(HW) * cvec));
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(g + static_cast<size_t>(n) * C + c);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(bf16_to_f32(x[e]) * inv);
        *reinterpret_cast<bf16x8*>(da + v * 8) = o;
    }
}

PDT_API int pdt_avgpool_fwd(const void* a, void* out, int NB, int HW, int C, void* stream) {
    if (NB <= 0 || HW <= 0 || C <= 0 || (C % 8) != 0) return -1;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(ew_grid(static_cast<long long>(NB) * (C / 8))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const bf16*>(a), static_cast<bf16*>(out), NB, HW, C);
    return pdt_launch_status();
}

PDT_API int pdt_avgpool_bwd(const void* g, void* da, int NB, int HW, int C, void* stream) {
    if (NB <= 0 || HW <= 0 || C <= 0 || (C % 8) != 0) return -1;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ew_grid(static_cast<long long>(NB) * HW * (C / 8))), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const bf16*>(g), static_cast<bf16*>(da), NB, HW, C);
    return pdt_launch_status();
}
