// buffer_load ... lds with lanes whose offset lies beyond num_records: does the LDS receive zeros for them?
// (the gather loaders could then drop their page of zeros and their per-lane 64-bit address selects)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void_t;
__global__ void k(const unsigned* a, unsigned* out, int soff) {
    extern __shared__ unsigned smem[];
    for (int i = threadIdx.x; i < 256; i += 64) smem[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, 0x7fffffff, 0x00020000);
    // lanes 0-31 in range (16 bytes each, from byte offset soff), lanes 32-63 out of range
    const unsigned voff = threadIdx.x < 32 ? threadIdx.x * 16u : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)smem, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
}
int main() {
    unsigned h[1024], o[256]; unsigned *a, *out;
    for (int i = 0; i < 1024; ++i) h[i] = 1000 + i;
    (void)hipMalloc(&a, sizeof(h)); (void)hipMalloc(&out, sizeof(o));
    (void)hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, a, out, 64);
    if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); return 1; }
    (void)hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost);
    int ok_in = 0, zero_out = 0, stale_out = 0;
    for (int i = 0; i < 128; ++i) ok_in += o[i] == 1000u + 16 + i;   // soff = 64 bytes = 16 words
    for (int i = 128; i < 256; ++i) { zero_out += o[i] == 0; stale_out += o[i] == 0xdeadbeefu; }
    printf("in-range words correct: %d of 128; out-of-range words: %d zero, %d untouched\n", ok_in, zero_out, stale_out);
    return 0;
}
