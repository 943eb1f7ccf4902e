// Shared device helpers for the gfx950 kernels (wave64, MFMA bf16, LDS tiling).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define PDT_API extern "C" __attribute__((visibility("default")))
#define PDT_WAVE 64
// Per-channel statistics are accumulated with atomics into this many independent slots.
#define PDT_STAT_SLOTS 32

// Deterministic mode (pdt_set_deterministic, defined in bn.hip): reductions whose result depends on the order of
// floating-point atomic adds are launched so that every destination has exactly one adder, or not at all (the caller
// takes a different path); results are then bit-identical from run to run, at a price in speed.
extern int pdt_deterministic_flag;

// Launch-status convention of every exported launcher: 0 = launched, otherwise a hipError_t value.
static inline int pdt_launch_status() { return static_cast<int>(hipGetLastError()); }

static inline int pdt_cdiv(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

// Unsigned division by a run-time constant via multiply-high.  Valid for n < 2^31.
struct FastDiv {
    uint32_t mul;
    uint32_t shr;
    uint32_t d;
};

static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d ? d : 1;
    uint32_t l = 0;
    while ((1ull << l) < f.d) ++l;
    f.shr = l;
    f.mul = static_cast<uint32_t>(((1ull << 32) * ((1ull << l) - f.d)) / f.d + 1);
    return f;
}

__device__ __forceinline__ uint32_t fastdiv(uint32_t n, const FastDiv& f) { return (__umulhi(n, f.mul) + n) >> f.shr; }

__device__ __forceinline__ float bf16_to_f32(bf16 v) { return static_cast<float>(v); }
__device__ __forceinline__ bf16 f32_to_bf16(float v) { return static_cast<bf16>(v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// tanh-form GELU through the logistic function: 0.5 (1 + tanh u) = 1 / (1 + exp(-2u)),
// u = sqrt(2/pi) (x + 0.044715 x^3).  One exp2 and one reciprocal per element.
__device__ __forceinline__ float pdt_gelu_sigmoid(float x, float x2) {
    const float u2 = -2.f * 1.4426950408889634f * 0.7978845608028654f * x * (1.f + 0.044715f * x2);  // -2u log2(e)
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(u2));
}
__device__ __forceinline__ float pdt_gelu(float x) { return x * pdt_gelu_sigmoid(x, x * x); }
// gelu(x) and gelu'(x) from one evaluation of the logistic function (the fused feed-forward block stores both in
// forward, so that its backward multiplies by a stored number instead of evaluating two transcendentals per element)
__device__ __forceinline__ void pdt_gelu_both(float x, float& g, float& d) {
    const float x2 = x * x;
    const float s = pdt_gelu_sigmoid(x, x2);
    const float du2 = 2.f * 0.7978845608028654f * (1.f + 3.f * 0.044715f * x2);
    g = x * s;
    d = s + g * (1.f - s) * du2;
}
__device__ __forceinline__ float pdt_gelu_grad(float x) {
    const float x2 = x * x;
    const float s = pdt_gelu_sigmoid(x, x2);
    const float du2 = 2.f * 0.7978845608028654f * (1.f + 3.f * 0.044715f * x2);
    return s + x * s * (1.f - s) * du2;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` holds >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// Tiles that share an operand panel should run on one XCD (private L2).  Blocks b and b+8 share
// an XCD under round-robin dispatch; this bijection gives each XCD a contiguous range of tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, x = bid & 7, w = bid >> 3;
    const int base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + w;
}
