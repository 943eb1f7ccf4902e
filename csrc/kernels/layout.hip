// Layout / precision preparation kernels.
//   * NCHW float image batch -> NHWC bf16 (what every GPU kernel in this library reads)
//   * stem im2col: NHWC bf16 image with few channels -> [pixels][K] patch matrix (K padded to 64)
//     so the 7x7/2 stem convolution runs on the same MFMA GEMM as every other layer
//   * fp32 -> bf16 cast (initial bf16 copies of the weights; later refreshed by the optimizer kernel)
//   * weight transpose [Cout][taps][Cin] fp32 -> [Cin][taps][Cout] bf16 (operand of the data gradient)
//
// Role parity: the synchronous `.to(device)` upload and cuDNN's internal layout handling reached
// from /root/reference/src/main.py:69-70,74.
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const T* __restrict__ x, bf16* __restrict__ out, int NB, int C,
                                                          long long HW) {
    const long long npix = static_cast<long long>(NB) * HW;
    for (long long pix = blockIdx.x * 256ll + threadIdx.x; pix < npix; pix += gridDim.x * 256ll) {
        const long long n = pix / HW, hw = pix - n * HW;
        const T* src = x + n * C * HW + hw;
        bf16* dst = out + pix * C;
        for (int c = 0; c < C; ++c) dst[c] = f32_to_bf16(static_cast<float>(src[c * HW]));
    }
}

// dtype: 0 fp32, 1 bf16 source
PDT_API int pdt_nchw_to_nhwc_bf16(const void* x, void* out, int NB, int C, int H, int W, int dtype, void* stream) {
    if (NB <= 0 || C <= 0 || H <= 0 || W <= 0) return -1;
    const long long HW = static_cast<long long>(H) * W, npix = NB * HW;
    long long g = (npix + 255) / 256;
    g = g > 8192 ? 8192 : g;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == 0) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(g), dim3(256), 0, s, static_cast<const float*>(x), static_cast<bf16*>(out), NB, C, HW);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, dim3(g), dim3(256), 0, s, static_cast<const bf16*>(x), static_cast<bf16*>(out), NB, C, HW);
    return pdt_launch_status();
}

// uint8 NHWC image batch (what decoders and image files hold) -> normalised bf16 NHWC:
// out[i] = x[i] * scale[c] + shift[c], c = i mod 3.  A thread converts 8 pixels (24 bytes in, 48 bytes out), so the
// channel of every element is known at compile time.  The host therefore ships one byte per element over PCIe
// instead of four (fp32 NCHW), and ToTensor + Normalize + the layout change are this one pass on the GPU.
struct U8Norm {
    float scale[3];
    float shift[3];
};

__global__ __launch_bounds__(256) void u8_rgb_to_bf16_kernel(const uint8_t* __restrict__ x, bf16* __restrict__ out, long long n24,
                                                            long long tail_start, long long total, U8Norm nm) {
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < n24; v += gridDim.x * 256ll) {
        const uint2* src = reinterpret_cast<const uint2*>(x + v * 24);
        uint32_t w[6];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint2 t = src[i];
            w[2 * i] = t.x;
            w[2 * i + 1] = t.y;
        }
        bf16x8 o[3];
#pragma unroll
        for (int e = 0; e < 24; ++e) {
            const float f = static_cast<float>((w[e >> 2] >> ((e & 3) * 8)) & 0xffu);
            o[e >> 3][e & 7] = f32_to_bf16(f * nm.scale[e % 3] + nm.shift[e % 3]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<bf16x8*>(out + v * 24 + i * 8) = o[i];
    }
    // the last (total mod 24) elements, if any
    for (long long i = tail_start + blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll)
        out[i] = f32_to_bf16(static_cast<float>(x[i]) * nm.scale[i % 3] + nm.shift[i % 3]);
}

// x: [pixels][3] uint8 (8-byte aligned), out: [pixels][3] bf16 (16-byte aligned)
PDT_API int pdt_u8_rgb_to_bf16(const void* x, void* out, long long pixels, const float* scale, const float* shift, void* stream) {
    if (pixels <= 0 || (reinterpret_cast<uintptr_t>(x) & 7) || (reinterpret_cast<uintptr_t>(out) & 15)) return -1;
    U8Norm nm;
    for (int c = 0; c < 3; ++c) {
        nm.scale[c] = scale[c];
        nm.shift[c] = shift[c];
    }
    const long long total = pixels * 3, n24 = total / 24;
    long long g = (n24 + 255) / 256;
    g = g < 1 ? 1 : (g > 16384 ? 16384 : g);
    hipLaunchKernelGGL(u8_rgb_to_bf16_kernel, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const uint8_t*>(x),
                       static_cast<bf16*>(out), n24, n24 * 24, total, nm);
    return pdt_launch_status();
}

// Few-channel images for the packed-row convolution: out[n][hp][wp][0..CP) = x[n][hp - pad][wp - pad][0..C)
// inside the image, zero on the border and in the padding channels (CP in {4, 8}; one 8- or 16-byte store
// per pixel).  The convolution then reads whole kernel rows as contiguous, bounds-check-free vectors.
template <int CP>
__global__ __launch_bounds__(256) void pad_pixels_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, int NB, int H, int W,
                                                        int C, int Hp, int Wp, int pad) {
    typedef __attribute__((ext_vector_type(CP))) __bf16 vec_t;
    const long long npix = static_cast<long long>(NB) * Hp * Wp;
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < npix; v += gridDim.x * 256ll) {
        const int wp = static_cast<int>(v % Wp);
        long long t = v / Wp;
        const int hp = static_cast<int>(t % Hp);
        const int n = static_cast<int>(t / Hp);
        const int ih = hp - pad, iw = wp - pad;
        vec_t o;
#pragma unroll
        for (int e = 0; e < CP; ++e) o[e] = f32_to_bf16(0.f);
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
            const bf16* src = x + ((static_cast<size_t>(n) * H + ih) * W + iw) * C;
#pragma unroll
            for (int e = 0; e < CP; ++e)
                if (e < C) o[e] = src[e];
        }
        *reinterpret_cast<vec_t*>(out + v * CP) = o;
    }
}

PDT_API int pdt_pad_pixels(const void* x, void* out, int NB, int H, int W, int C, int CP, int Hp, int Wp, int pad, void* stream) {
    if (NB <= 0 || C <= 0 || C > CP || (CP != 4 && CP != 8) || Hp <= 0 || Wp <= 0 || pad < 0) return -1;
    long long g = (static_cast<long long>(NB) * Hp * Wp + 255) / 256;
    g = g > 32768 ? 32768 : g;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (CP == 4)
        hipLaunchKernelGGL(pad_pixels_kernel<4>, dim3(g), dim3(256), 0, s, static_cast<const bf16*>(x), static_cast<bf16*>(out), NB, H, W, C, Hp, Wp, pad);
    else
        hipLaunchKernelGGL(pad_pixels_kernel<8>, dim3(g), dim3(256), 0, s, static_cast<const bf16*>(x), static_cast<bf16*>(out), NB, H, W, C, Hp, Wp, pad);
    return pdt_launch_status();
}

// dst[row][0..cols_dst) = bf16(src[row][0..cols_src)), zero padded (cols_dst >= cols_src)
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long long rows,
                                                      int cols_src, int cols_dst) {
    const long long n = rows * cols_dst;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
        const long long r = i / cols_dst;
        const int c = static_cast<int>(i - r * cols_dst);
        dst[i] = f32_to_bf16(c < cols_src ? src[r * cols_src + c] : 0.f);
    }
}

PDT_API int pdt_cast_pad_f32_bf16(const float* src, void* dst, long long rows, int cols_src, int cols_dst, void* stream) {
    if (rows <= 0 || cols_src <= 0 || cols_dst < cols_src) return -1;
    long long g = (rows * cols_dst + 255) / 256;
    g = g > 4096 ? 4096 : g;
    hipLaunchKernelGGL(cast_pad_kernel, dim3(g), dim3(256), 0, static_cast<hipStream_t>(stream), src, static_cast<bf16*>(dst),
                       rows, cols_src, cols_dst);
    return pdt_launch_status();
}

// src fp32 [N][T][C]  ->  dst bf16 [C][T][N]   (32x32 tiles through LDS, coalesced on both sides)
__global__ __launch_bounds__(256) void transpose_w_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int N, int T,
                                                         int C) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z, c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (n < N && c < C) ? src[(static_cast<size_t>(n) * T + t) * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, n = n0 + tx;
        if (c < C && n < N) dst[(static_cast<size_t>(c) * T + t) * N + n] = f32_to_bf16(tile[tx][ty + 8 * i]);
    }
}

PDT_API int pdt_transpose_w(const float* src, void* dst, int N, int T, int C, void* stream) {
    if (N <= 0 || T <= 0 || C <= 0 || T > 65535) return -1;
    hipLaunchKernelGGL(transpose_w_kernel, dim3(pdt_cdiv(C, 32), pdt_cdiv(N, 32), T), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, static_cast<bf16*>(dst), N, T, C);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// The same transposition for many weights in one launch (all data-gradient copies of a parameter group right
// after the optimizer step): 64 x 64 tiles, fp32 rows read as 16-byte vectors, bf16 rows written as 16-byte
// vectors.  desc[i].tile_start is the first tile of tensor i in the launch-wide tile numbering (ascending).
// ------------------------------------------------------------------------------------------------
struct TransposeDesc {
    const float* src;  // [N][T][C]
    bf16* dst;         // [C][T][N]
    int N, T, C;
    int tile_start;
};

__global__ __launch_bounds__(256) void transpose_w_multi_kernel(const TransposeDesc* __restrict__ descs, int ntensors) {
    __shared__ float tile[64][65];
    const int tid = threadIdx.x;
    int lo = 0, hi = ntensors - 1;  // last tensor whose tile_start <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].tile_start <= static_cast<int>(blockIdx.x)) lo = mid;
        else hi = mid - 1;
    }
    const TransposeDesc d = descs[lo];
    int local = blockIdx.x - d.tile_start;
    const int tc = (d.C + 63) >> 6, tn = (d.N + 63) >> 6;
    const int ci = local % tc;
    local /= tc;
    const int ni = local % tn, t = local / tn;
    const int c0 = ci * 64, n0 = ni * 64;
    const bool vec_in = (d.C & 3) == 0, vec_out = (d.N & 7) == 0;
    // load: 64 rows (n) x 16 float4
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx >> 4, q = (idx & 15) << 2;
        const int n = n0 + r, c = c0 + q;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (n < d.N) {
            const float* sp = d.src + (static_cast<size_t>(n) * d.T + t) * d.C + c;
            if (vec_in && c + 3 < d.C) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(sp);
                v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < d.C) v[e] = sp[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[r][q + e] = v[e];
    }
    __syncthreads();
    // store: 64 rows (c) x 8 vectors of 8 bf16 along n
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx >> 3, q = (idx & 7) << 3;
        const int c = c0 + r, n = n0 + q;
        if (c >= d.C || n >= d.N) continue;
        bf16* dp = d.dst + (static_cast<size_t>(c) * d.T + t) * d.N + n;
        if (vec_out && n + 7 < d.N) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(tile[q + e][r]);
            *reinterpret_cast<bf16x8*>(dp) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < d.N) dp[e] = f32_to_bf16(tile[q + e][r]);
        }
    }
}

PDT_API int pdt_transpose_desc_bytes() { return static_cast<int>(sizeof(TransposeDesc)); }

PDT_API int pdt_transpose_w_multi(const void* descs, int ntensors, int total_tiles, void* stream) {
    if (ntensors <= 0 || total_tiles <= 0) return -1;
    hipLaunchKernelGGL(transpose_w_multi_kernel, dim3(total_tiles), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const TransposeDesc*>(descs), ntensors);
    return pdt_launch_status();
}
