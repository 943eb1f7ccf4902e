// LDS read throughput of one CU's worth of waves: ds_read_b128 against ds_read_b64_tr_b16 (the transposing read the
// weight-gradient kernels need for both operands).  512 threads per workgroup, one workgroup per CU, conflict-free
// addresses; prints bytes per clock and CU.      hipcc --offload-arch=gfx950 -O2 -o lds_read_rate bench/lds_read_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned* out, int iters, long long* clocks) {
    extern __shared__ char smem[];
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    const unsigned base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem)) + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    unsigned acc = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            u32x4 a, b, c, d;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:8192\n\tds_read_b128 %2, %4 offset:16384\n\tds_read_b128 %3, %4 offset:24576\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(base) : "memory");
            acc += a.x + b.y + c.z + d.w;
        } else {
            u32x2 a, b, c, d, e, f, g, h;
            asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:4096\n\tds_read_b64_tr_b16 %2, %8 offset:8192\n\tds_read_b64_tr_b16 %3, %8 offset:12288\n\t"
                         "ds_read_b64_tr_b16 %4, %8 offset:16384\n\tds_read_b64_tr_b16 %5, %8 offset:20480\n\tds_read_b64_tr_b16 %6, %8 offset:24576\n\tds_read_b64_tr_b16 %7, %8 offset:28672\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(h) : "v"(base - (threadIdx.x & 63) * 8) : "memory");
            acc += a.x + b.y + c.x + d.y + e.x + f.y + g.x + h.y;
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 512 + threadIdx.x] = acc;
    if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned* out; long long* clk; long long h[256];
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&clk, 256 * 8);
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 65536, 0, out, iters, clk);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 65536, 0, out, iters, clk);
            if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return 1; }
        }
        (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        const double bytes = 512.0 * 64 * iters;  // 64 bytes per thread and iteration in both modes
        printf("%s: %.1f bytes per clock64 tick and CU (%.0f ticks)\n", mode == 0 ? "ds_read_b128      " : "ds_read_b64_tr_b16", bytes / avg, avg);
    }
    return 0;
}
