// BatchNorm forward finalize: statistic slots -> per-channel constants, running statistics, the pivot of the next
// step's sums and the non-finite flag.  A file of its own because it must see NaN and Inf as what they are: it is
// compiled with IEEE semantics (csrc/build.py: PRECISE_MATH), while the element-wise passes in bn.hip keep the fast
// floating-point model.
//
// Role parity: the statistics half of the BatchNorm2d layers exercised by /root/reference/src/main.py:49,74.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// bn_finalize: slots -> per-channel constants
//   out consts layout: [4][C] = mean, rstd, scale, shift
// The slots hold sums about a per-channel pivot P (sum(y - P), sum((y - P)^2)); P itself sits in the first row of
// slot PDT_STAT_SLOTS (zero when the producer used none), so a copy of the slots stays self-contained:
// mean = P + s1 / n and var = s2 / n - (s1 / n)^2, which loses nothing to cancellation while P is near the mean.  The
// pivot buffer (optional) then receives this batch's mean for the next step's convolution to use (when the old pivot
// has drifted more than a quarter of the spread away from it).  A channel whose statistics
// are not finite (an Inf or NaN reached the convolution) raises `nonfinite` (a device word the trainer's
// --check-finite reads): BatchNorm followed by ReLU would otherwise swallow the NaN (max(NaN, 0) = 0).
// ------------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* __restrict__ slots, int C, float count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var,
                                   long long* num_batches, float momentum, float eps, int training,
                                   float* __restrict__ consts, float* pivot, int* nonfinite) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean, var;
    if (training) {
        double s1 = 0.0, s2 = 0.0;
        unsigned exp_all_ones = 0;  // any accumulator that holds an Inf or a NaN (tested on the loaded bits)
        for (int s = 0; s < PDT_STAT_SLOTS; ++s) {
            const float a = slots[(s * 2 + 0) * C + c], b = slots[(s * 2 + 1) * C + c];
            exp_all_ones |= static_cast<unsigned>((__float_as_uint(a) & 0x7f800000u) == 0x7f800000u) |
                            static_cast<unsigned>((__float_as_uint(b) & 0x7f800000u) == 0x7f800000u);
            s1 += a;
            s2 += b;
        }
        const double d1 = s1 / count;
        const float P = slots[PDT_STAT_SLOTS * 2 * C + c];
        mean = static_cast<float>(static_cast<double>(P) + d1);
        var = fmaxf(static_cast<float>(s2 / count - d1 * d1), 0.f);
        // (this file is compiled without finite-math assumptions: under them the compiler may assume that no float it
        // sees is an Inf or a NaN and fold this very test away, bit test or not)
        const bool finite = exp_all_ones == 0;
        // The pivot only has to be near the mean (within a quarter of the spread loses nothing): it is left alone while it
        // is, so a model that sees the same batch twice computes bit-identical statistics both times.
        if (pivot != nullptr && finite && d1 * d1 * 16.0 > static_cast<double>(var)) pivot[c] = mean;
        if (nonfinite != nullptr && !finite) atomicOr(nonfinite, 1);
        if (running_mean != nullptr) {
            const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
            if (c == 0 && num_batches != nullptr) *num_batches += 1;
        }
    } else {
        mean = running_mean[c];
        var = running_var[c];
    }
    const float rstd = rsqrtf(var + eps);
    const float g = gamma != nullptr ? gamma[c] : 1.f, b = beta != nullptr ? beta[c] : 0.f;
    consts[0 * C + c] = mean;
    consts[1 * C + c] = rstd;
    consts[2 * C + c] = g * rstd;
    consts[3 * C + c] = b - mean * g * rstd;
}

PDT_API int pdt_bn_finalize(const float* slots, int C, float count, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, long long* num_batches, float momentum, float eps,
                            int training, float* consts, float* pivot, int* nonfinite, void* stream) {
    if (C <= 0 || (!training && (running_mean == nullptr || running_var == nullptr))) return -1;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(pdt_cdiv(C, 64)), dim3(64), 0, static_cast<hipStream_t>(stream), slots, C,
                       count, gamma, beta, running_mean, running_var, num_batches, momentum, eps, training, consts, pivot,
                       nonfinite);
    return pdt_launch_status();
}

