// Multi-tensor fused optimizer step: Adam with L2 decay, AdamW (decoupled decay), SGD with momentum.
//
// One launch updates every parameter tensor of a group.  A device-resident descriptor table holds,
// per tensor, the fp32 parameter / gradient / state pointers and the optional bf16 compute copy;
// a chunk table maps each workgroup to (tensor, offset).  Each lane moves 16 bytes per array per
// iteration.  The bf16 copy that the convolution / GEMM kernels read is refreshed in the same pass,
// so the weights are never re-read for a separate cast.
//
// Role parity: torch.optim.Adam(lr, weight_decay) stepping 62 tensors (fused/foreach Adam) at
// /root/reference/src/main.py:63,79; AdamW and SGD-momentum are the additional modes.
#include "common.h"

struct PdtTensorDesc {
    float* p;
    const float* g;
    float* s1;     // exp_avg | momentum buffer
    float* s2;     // exp_avg_sq | unused
    bf16* shadow;  // bf16 copy of p in the same element order, or nullptr
    long long n;
};

struct PdtChunk {
    int tensor;
    int pad;
    long long start;
};

#define PDT_OPT_CHUNK 32768  // elements per workgroup

struct OptHyper {
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, momentum, grad_scale;
    int mode;  // 0 adam (L2), 1 adamw, 2 sgd
    int first_step;
};

// hp != nullptr: hyper-parameters are read from device memory (written by set_hyper_kernel before every
// launch), so a launch recorded in a hipGraph picks up the current learning rate and bias corrections.
__global__ __launch_bounds__(256) void multi_tensor_step_kernel(const PdtTensorDesc* __restrict__ descs,
                                                               const PdtChunk* __restrict__ chunks, const OptHyper hv,
                                                               const OptHyper* __restrict__ hp) {
    const OptHyper h = hp != nullptr ? *hp : hv;
    const PdtChunk ck = chunks[blockIdx.x];
    const PdtTensorDesc d = descs[ck.tensor];
    const long long end = min(d.n, ck.start + PDT_OPT_CHUNK);
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(d.p) | reinterpret_cast<uintptr_t>(d.g) |
                          reinterpret_cast<uintptr_t>(d.s1) | reinterpret_cast<uintptr_t>(d.s2)) & 15) == 0 &&
                        (d.shadow == nullptr || (reinterpret_cast<uintptr_t>(d.shadow) & 7) == 0);
    auto update = [&](float& p, float g, float& s1, float& s2) {
        g *= h.grad_scale;
        if (h.mode == 2) {
            g += h.wd * p;
            if (h.momentum != 0.f) {
                s1 = h.first_step ? g : h.momentum * s1 + g;
                g = s1;
            }
            p -= h.lr * g;
        } else {
            if (h.mode == 0) g += h.wd * p;
            else p *= 1.f - h.lr * h.wd;
            s1 = s1 + (1.f - h.beta1) * (g - s1);
            s2 = h.beta2 * s2 + (1.f - h.beta2) * g * g;
            const float denom = sqrtf(s2) / h.bc2_sqrt + h.eps;
            p -= (h.lr / h.bc1) * (s1 / denom);
        }
    };
    long long i = ck.start + threadIdx.x * 4ll;
    if (vec_ok) {
        for (; i + 3 < end; i += 1024) {
            f32x4 p = *reinterpret_cast<f32x4*>(d.p + i);
            const f32x4 g = *reinterpret_cast<const f32x4*>(d.g + i);
            f32x4 s1 = *reinterpret_cast<f32x4*>(d.s1 + i);
            f32x4 s2 = {0.f, 0.f, 0.f, 0.f};
            if (h.mode != 2) s2 = *reinterpret_cast<f32x4*>(d.s2 + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = p[e], a = s1[e], b = s2[e];
                update(pe, g[e], a, b);
                p[e] = pe; s1[e] = a; s2[e] = b;
            }
            *reinterpret_cast<f32x4*>(d.p + i) = p;
            *reinterpret_cast<f32x4*>(d.s1 + i) = s1;
            if (h.mode != 2) *reinterpret_cast<f32x4*>(d.s2 + i) = s2;
            if (d.shadow != nullptr) {
                bf16x4 sh;
#pragma unroll
                for (int e = 0; e < 4; ++e) sh[e] = f32_to_bf16(p[e]);
                *reinterpret_cast<bf16x4*>(d.shadow + i) = sh;
            }
        }
    }
    // tail of the chunk (or the whole chunk when pointers are not 16-byte aligned)
    const long long tail0 = vec_ok ? ck.start + ((end - ck.start) / 4) * 4 : ck.start;
    for (long long j = tail0 + threadIdx.x; j < end; j += 256) {
        float p = d.p[j], a = d.s1[j], b = (h.mode != 2) ? d.s2[j] : 0.f;
        update(p, d.g[j], a, b);
        d.p[j] = p;
        d.s1[j] = a;
        if (h.mode != 2) d.s2[j] = b;
        if (d.shadow != nullptr) d.shadow[j] = f32_to_bf16(p);
    }
}

PDT_API int pdt_opt_chunk_elems() { return PDT_OPT_CHUNK; }
PDT_API int pdt_opt_desc_bytes() { return static_cast<int>(sizeof(PdtTensorDesc)); }
PDT_API int pdt_opt_chunk_bytes() { return static_cast<int>(sizeof(PdtChunk)); }

PDT_API int pdt_multi_tensor_step(const void* descs, const void* chunks, int n_chunks, int mode, float lr, float beta1,
                                  float beta2, float eps, float wd, float bc1, float bc2_sqrt, float momentum,
                                  float grad_scale, int first_step, void* stream) {
    if (n_chunks <= 0 || mode < 0 || mode > 2) return -1;
    OptHyper h;
    h.lr = lr; h.beta1 = beta1; h.beta2 = beta2; h.eps = eps; h.wd = wd; h.bc1 = bc1; h.bc2_sqrt = bc2_sqrt;
    h.momentum = momentum; h.grad_scale = grad_scale; h.mode = mode; h.first_step = first_step;
    hipLaunchKernelGGL(multi_tensor_step_kernel, dim3(n_chunks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const PdtTensorDesc*>(descs), static_cast<const PdtChunk*>(chunks), h,
                       static_cast<const OptHyper*>(nullptr));
    return pdt_launch_status();
}

__global__ void set_hyper_kernel(OptHyper* dst, const OptHyper h) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        // field-wise vector stores from one lane
        dst->lr = h.lr; dst->beta1 = h.beta1; dst->beta2 = h.beta2; dst->eps = h.eps; dst->wd = h.wd;
        dst->bc1 = h.bc1; dst->bc2_sqrt = h.bc2_sqrt; dst->momentum = h.momentum; dst->grad_scale = h.grad_scale;
        dst->mode = h.mode; dst->first_step = h.first_step;
    }
}

PDT_API int pdt_opt_hyper_bytes() { return static_cast<int>(sizeof(OptHyper)); }

// Stream-ordered update of a device-resident hyper-parameter block (the values travel as kernel arguments,
// so there is no host buffer whose lifetime the caller would have to manage).
PDT_API int pdt_opt_set_hyper(void* dst, int mode, float lr, float beta1, float beta2, float eps, float wd, float bc1,
                              float bc2_sqrt, float momentum, float grad_scale, int first_step, void* stream) {
    if (dst == nullptr || mode < 0 || mode > 2) return -1;
    OptHyper h;
    h.lr = lr; h.beta1 = beta1; h.beta2 = beta2; h.eps = eps; h.wd = wd; h.bc1 = bc1; h.bc2_sqrt = bc2_sqrt;
    h.momentum = momentum; h.grad_scale = grad_scale; h.mode = mode; h.first_step = first_step;
    hipLaunchKernelGGL(set_hyper_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<OptHyper*>(dst), h);
    return pdt_launch_status();
}

// The step with hyper-parameters taken from `hyper` (device memory, see pdt_opt_set_hyper).
PDT_API int pdt_multi_tensor_step_dev(const void* descs, const void* chunks, int n_chunks, const void* hyper, void* stream) {
    if (n_chunks <= 0 || hyper == nullptr) return -1;
    OptHyper unused = {};
    hipLaunchKernelGGL(multi_tensor_step_kernel, dim3(n_chunks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const PdtTensorDesc*>(descs), static_cast<const PdtChunk*>(chunks), unused,
                       static_cast<const OptHyper*>(hyper));
    return pdt_launch_status();
}
