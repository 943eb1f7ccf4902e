// Does v_mfma_f32_16x16x32_bf16 tolerate a destination that overlaps its B (or A) operand when srcC is a constant?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const bf16x8* a, const bf16x8* b, f32x4* ref, f32x4* ovb, f32x4* ova) {
    const int l = threadIdx.x;
    bf16x8 fa = a[l], fb = b[l];
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    ref[l] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, z, 0, 0, 0);
    f32x4 x = __builtin_bit_cast(f32x4, fb);
    asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %0, 0\n\ts_nop 15\n\ts_nop 15" : "+v"(x) : "v"(fa));
    ovb[l] = x;
    f32x4 y = __builtin_bit_cast(f32x4, fa);
    asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %0, %1, 0\n\ts_nop 15\n\ts_nop 15" : "+v"(y) : "v"(fb));
    ova[l] = y;
}
int main() {
    bf16x8 ha[64], hb[64]; f32x4 hr[64], h1[64], h2[64];
    bf16x8 *a, *b; f32x4 *r, *o1, *o2;
    hipMalloc(&a, 64 * 16); hipMalloc(&b, 64 * 16); hipMalloc(&r, 64 * 16); hipMalloc(&o1, 64 * 16); hipMalloc(&o2, 64 * 16);
    srand(1);
    for (int i = 0; i < 64; ++i) for (int e = 0; e < 8; ++e) { ha[i][e] = (__bf16)((rand() % 17 - 8) / 4.0f); hb[i][e] = (__bf16)((rand() % 13 - 6) / 2.0f); }
    hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, r, o1, o2);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    hipMemcpy(hr, r, sizeof(hr), hipMemcpyDeviceToHost); hipMemcpy(h1, o1, sizeof(h1), hipMemcpyDeviceToHost); hipMemcpy(h2, o2, sizeof(h2), hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0;
    for (int i = 0; i < 64; ++i) for (int e = 0; e < 4; ++e) { bad1 += hr[i][e] != h1[i][e]; bad2 += hr[i][e] != h2[i][e]; }
    printf("dst overlaps B: %d of 256 values differ; dst overlaps A: %d of 256 differ\n", bad1, bad2);
    return 0;
}
