#include <hip/hip_runtime.h>
#include <cstdio>
// Rate of fp32 global atomic adds as issued by the weight-gradient epilogue, on a [4096][1024] fp32 matrix.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_rate bench/atomic_rate.hip && ./atomic_rate
// pattern 0: each wave instruction adds to 4 rows x 16 consecutive floats (the MFMA accumulator layout)
// pattern 1: each wave instruction adds to 64 consecutive floats of one row
// pattern 2: plain read-modify-write with pattern 1 (no atomics; only meaningful with splits == 1)
// Measured on MI355X: 0.26 T elements/s (64 workgroups) to 0.31 T/s (256 workgroups) for patterns 0 and 1 alike.
template <int PAT>
__global__ __launch_bounds__(256) void k(float* w, int ld, int rows_per_block, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // block covers a tile of 256 rows x 256 cols like the wgrad kernel; wave (wr, wc) = (wave>>1... ) simplified:
    const int tile = blockIdx.x % ((ld / 256) * (rows_per_block));
    const int tr = tile / (ld / 256), tc = tile % (ld / 256);
    float* base = w + (size_t)tr * 256 * ld + tc * 256;
    for (int r = 0; r < reps; ++r) {
        // each wave handles a 64-row x 256-col strip: rows wave*64..
        for (int i = 0; i < 4; ++i)       // 16-row fragments
            for (int j = 0; j < 16; ++j)  // 16-col fragments
                for (int e = 0; e < 4; ++e) {
                    if (PAT == 0) {
                        const int row = wave * 64 + i * 16 + (lane >> 4) * 4 + e, col = j * 16 + (lane & 15);
                        __hip_atomic_fetch_add(base + (size_t)row * ld + col, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else if (PAT == 1) {
                        // same number of instructions; instr (i, j, e): row = i*16 + (j&3)*4 + e, cols (j>>2)*64 + lane
                        const int row = wave * 64 + i * 16 + (j & 3) * 4 + e, col = (j >> 2) * 64 + lane;
                        __hip_atomic_fetch_add(base + (size_t)row * ld + col, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else if (PAT == 2) {
                        const int row = wave * 64 + i * 16 + (j & 3) * 4 + e, col = (j >> 2) * 64 + lane;
                        float* p = base + (size_t)row * ld + col;
                        *p = *p + 1.0f;
                    }
                }
    }
}
int main() {
    const int N = 4096, ld = 1024;  // 16 MB fp32 like a [4096][1024] weight gradient
    float* w;
    if (hipMalloc(&w, (size_t)N * ld * 4) != hipSuccess || hipMemset(w, 0, (size_t)N * ld * 4) != hipSuccess) return 1;
    const int tiles = (N / 256) * (ld / 256);  // 64 tiles
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    for (int splits : {1, 4}) {
        for (int pat = 0; pat < 3; ++pat) {
            float best = 1e9;
            for (int it = 0; it < 5; ++it) {
                (void)hipEventRecord(e0);
                if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(tiles * splits), dim3(256), 0, 0, w, ld, N / 256, 1);
                if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(tiles * splits), dim3(256), 0, 0, w, ld, N / 256, 1);
                if (pat == 2) hipLaunchKernelGGL(k<2>, dim3(tiles * splits), dim3(256), 0, 0, w, ld, N / 256, 1);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double elems = (double)N * ld * splits;
            printf("splits %d pattern %d: %.1f us  %.1f G elem/s  (%.2f TB/s of fp32)\n", splits, pat, best * 1e3, elems / best / 1e6, elems * 4 / best / 1e9);
        }
    }
    return 0;
}
