#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// Streaming rate of BatchNorm-apply-shaped element-wise kernels (bf16 NHWC, per-channel fp32 constants).
//   hipcc --offload-arch=gfx950 -O3 -o stream_rate bench/stream_rate.hip && ./stream_rate
// variant "base": grid-stride loop, channel index by 64-bit modulo and constants re-read per 16-byte vector
// variant "tile": a workgroup walks tiles of 256*U vectors, the channel chunk of a thread is loop-invariant
//                 (256 % (C/8) == 0), constants live in registers, U independent loads are issued before use
// READS = number of bf16 input streams (1: y -> out; 2: dout, y -> dy)
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

template <int READS>
__global__ __launch_bounds__(256) void base_kernel(const bf16* __restrict__ y, const bf16* __restrict__ d,
                                                  const float* __restrict__ consts, bf16* __restrict__ out,
                                                  long long nvec, int C) {
    const int cvec = C >> 3;
    for (long long v = blockIdx.x * 256ll + threadIdx.x; v < nvec; v += gridDim.x * 256ll) {
        const int c = static_cast<int>(v % cvec) << 3;
        float sc[8], sh[8];
        load8f(consts + 2 * C + c, sc);
        load8f(consts + 3 * C + c, sh);
        const bf16x8 yv = *reinterpret_cast<const bf16x8*>(y + v * 8);
        bf16x8 dv;
        if (READS == 2) dv = *reinterpret_cast<const bf16x8*>(d + v * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float z = static_cast<float>(yv[e]) * sc[e] + sh[e];
            if (READS == 2) z = z > 0.f ? static_cast<float>(dv[e]) * sc[e] + z : 0.f;
            o[e] = static_cast<bf16>(fmaxf(z, 0.f));
        }
        *reinterpret_cast<bf16x8*>(out + v * 8) = o;
    }
}

template <int READS, int U, bool NT>
__global__ __launch_bounds__(256) void tile_kernel(const bf16* __restrict__ y, const bf16* __restrict__ d,
                                                  const float* __restrict__ consts, bf16* __restrict__ out,
                                                  long long nvec, int C) {
    const int cvec = C >> 3;
    const int c = (threadIdx.x & (cvec - 1)) << 3;  // cvec is a power of two <= 256
    float sc[8], sh[8];
    load8f(consts + 2 * C + c, sc);
    load8f(consts + 3 * C + c, sh);
    const long long tile = 256ll * U;
    for (long long base = blockIdx.x * tile; base < nvec; base += gridDim.x * tile) {
        bf16x8 yv[U], dv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long v = base + u * 256 + threadIdx.x;
            if (v < nvec) {
                const bf16x8* py = reinterpret_cast<const bf16x8*>(y) + v;
                yv[u] = NT ? __builtin_nontemporal_load(py) : *py;
                if (READS == 2) {
                    const bf16x8* pd = reinterpret_cast<const bf16x8*>(d) + v;
                    dv[u] = NT ? __builtin_nontemporal_load(pd) : *pd;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long v = base + u * 256 + threadIdx.x;
            if (v < nvec) {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float z = static_cast<float>(yv[u][e]) * sc[e] + sh[e];
                    if (READS == 2) z = z > 0.f ? static_cast<float>(dv[u][e]) * sc[e] + z : 0.f;
                    o[e] = static_cast<bf16>(fmaxf(z, 0.f));
                }
                bf16x8* po = reinterpret_cast<bf16x8*>(out) + v;
                if (NT) __builtin_nontemporal_store(o, po);
                else *po = o;
            }
        }
    }
}

template <typename F>
static float best_ms(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return best;
}

template <int READS, int U, bool NT>
static void run_tile(const bf16* y, const bf16* d, const float* consts, bf16* out, long long nvec, int C, int grid) {
    const float ms = best_ms([&] { hipLaunchKernelGGL((tile_kernel<READS, U, NT>), dim3(grid), dim3(256), 0, 0, y, d, consts, out, nvec, C); });
    printf("  tile reads=%d U=%d nt=%d grid=%5d: %8.1f us %6.0f GB/s\n", READS, U, (int)NT, grid, ms * 1e3,
           (READS + 1) * 16.0 * nvec / ms / 1e6);
}

int main() {
    const int shapes[][2] = {{56, 256}, {14, 1024}, {7, 2048}};
    for (auto& s : shapes) {
        const int hw = s[0], C = s[1];
        const long long n = 2048ll * hw * hw * C, nvec = n / 8;
        bf16 *y, *d, *out;
        float* consts;
        if (hipMalloc(&y, n * 2) != hipSuccess || hipMalloc(&d, n * 2) != hipSuccess || hipMalloc(&out, n * 2) != hipSuccess ||
            hipMalloc(&consts, 4 * C * 4) != hipSuccess)
            return 1;
        (void)hipMemset(y, 0x3c, n * 2);
        (void)hipMemset(d, 0x3c, n * 2);
        (void)hipMemset(consts, 0, 4 * C * 4);
        printf("%dx%dx%d (%.2f GB per stream)\n", hw, hw, C, n * 2 / 1e9);
        for (int reads = 1; reads <= 2; ++reads) {
            const float ms = best_ms([&] {
                if (reads == 1) hipLaunchKernelGGL(base_kernel<1>, dim3(4096), dim3(256), 0, 0, y, d, consts, out, nvec, C);
                else hipLaunchKernelGGL(base_kernel<2>, dim3(4096), dim3(256), 0, 0, y, d, consts, out, nvec, C);
            });
            printf("  base reads=%d grid= 4096:           %8.1f us %6.0f GB/s\n", reads, ms * 1e3, (reads + 1) * 16.0 * nvec / ms / 1e6);
        }
        for (int grid : {1024, 2048, 4096, 16384}) {
            run_tile<1, 1, false>(y, d, consts, out, nvec, C, grid);
            run_tile<1, 2, false>(y, d, consts, out, nvec, C, grid);
            run_tile<1, 4, false>(y, d, consts, out, nvec, C, grid);
            run_tile<1, 4, true>(y, d, consts, out, nvec, C, grid);
            run_tile<2, 2, false>(y, d, consts, out, nvec, C, grid);
            run_tile<2, 4, false>(y, d, consts, out, nvec, C, grid);
            run_tile<2, 4, true>(y, d, consts, out, nvec, C, grid);
        }
        (void)hipFree(y);
        (void)hipFree(d);
        (void)hipFree(out);
        (void)hipFree(consts);
    }
    return 0;
}
