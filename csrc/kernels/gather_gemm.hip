// Implicit-GEMM convolution / linear layer on MFMA (gfx950, bf16 in, fp32 accumulate).
//
//   Out[m][n] = sum_t sum_c A(m, t, c) * B[n][woff_t + c]        (+ bias[n])
//
// A row m is an output pixel (img, a, b) of this launch; for tap t its operand is the C-vector
// of input pixel (a*istr + dh_t, b*istr + dw_t) of image img, or zeros when that pixel is outside
// the input.  The tap table makes one kernel serve
//   * forward convolution of any stride (taps = kernel window),
//   * data gradient (taps = the window positions that reach an output-parity class, B = weights
//     transposed to [Cin][R][S][Cout]), written with an output stride for stride-2 layers,
//   * plain GEMM / nn.Linear (one tap, 1x1 "image").
// Activations are NHWC, so a tap's operand is contiguous and is fetched with 16-byte loads.
//
// Tiling: BM x BN output tile per 256-thread workgroup (4 waves), K-step 64, operands staged
// through LDS in an XOR-swizzled [row][64] layout that makes the ds_read_b128 fragment reads
// conflict-free, global->register prefetch of step k+1 issued before the MFMAs of step k,
// one barrier per K-step (two LDS buffers).  The MFMA is issued with the weight fragment as the
// first operand, so each lane ends up holding four consecutive output channels of one pixel:
// results are packed to bf16x4, staged through LDS and written as whole 16-byte row segments.
// The epilogue also produces per-channel partial sums / sums of squares of the stored values
// (BatchNorm statistics) so no extra pass over the output is needed.
//
// Role parity: the cuDNN/cuBLAS kernels behind `net(imgs)` / `losses.backward()` at
// /root/reference/src/main.py:74,78.
#include "common.h"
#include <cstdlib>
#include <type_traits>


#define GG_MAX_TAPS 49
#define GG_BK 64

struct GatherGemmArgs {
    const bf16* A;
    const bf16* B;
    bf16* Out;
    float* stats;       // [PDT_STAT_SLOTS + 1][2][N] zero-initialised (sum, sumsq) accumulators, or nullptr; the last slot's
                        // first row receives the pivot the sums were taken about (see `pivot`)
    const float* bias;  // [N] or nullptr
    const bf16* addend; // tensor with the layout of Out added to the result (residual gradient), or nullptr
    const uint8_t* addend_bits;  // optional ReLU mask for the addend: bit e of byte v gates element 8v + e
    const bf16* zeros;  // >= 16 bytes of zeros (ring variant: source of rows that are padding); may be nullptr otherwise
    // ring variant, data-gradient launches: the result is the gradient of a BatchNorm(+ReLU) output, and the
    // per-channel sums BatchNorm's backward needs (sum dz, sum dz * xhat, dz = result * [activation > 0]) are
    // accumulated here into stats ([slots][3][N]) instead of by a separate pass over the gradient
    const bf16* bn_y;          // pre-normalisation activations, layout of Out
    const float* bn_consts;    // [4][N]: mean, rstd, scale, shift
    const uint8_t* bn_bits;    // ReLU mask bits of the activation (block tails); nullptr: mask = y * scale + shift > 0
    const bf16* bn_y2;         // tails with a projected shortcut: the shortcut BatchNorm's input (layout of Out) ...
    const float* bn_consts2;   // ... and constants; the third row of a slot then receives sum dz * xhat2
    int stat_rows;             // rows per slot of `stats`: 2 (forward statistics) or 3 (backward sums)
    // activation fused into the epilogue: 1 = store gelu(result) to out2 and gelu'(result) to Out (what the backward
    // pass needs); 2 = multiply the result by act_src, the derivative stored by such a launch (layout of Out)
    // ring variant, single-tap launches: the A operand is relu(A * scale[k] + shift[k]) per input channel k, i.e. the
    // BatchNorm + ReLU of the layer before this convolution applied to its raw output on the way in (xf = that
    // BatchNorm's [4][C] constants: mean, rstd, scale, shift).  The normalised activation tensor is never written.
    const float* xf;
    // forward statistics are taken about a per-channel pivot: the slots receive sum(y - pivot) and sum((y - pivot)^2).
    // With the pivot near the channel mean (the caller keeps the previous step's batch mean there) the variance is not
    // formed as a difference of two large numbers.  nullptr: pivot 0.
    const float* pivot;
    int act;
    int out_nt;  // ring variant: store the result rows non-temporally (outputs far larger than the caches; +2-3 % on
                 // the bandwidth-bound 1x1 layers of a 2048-image ResNet-50 step)
    bf16* out2;
    const bf16* act_src;
    int M, N, ldb, ldo, C, T;
    int Hin, Win, istr;
    int Hout, Wout, ostr, oph, opw;
    FastDiv div_ab;  // divide by OA*OB
    FastDiv div_b;   // divide by OB
    int tw;          // packed-row mode: elements per kernel row (32 or 64)
    int cs;          // elements between consecutive input pixels (C, except in the ring kernels' packed-row mode)
    int OA, OB;
    int tap_dh[GG_MAX_TAPS];
    int tap_dw[GG_MAX_TAPS];
    int tap_woff[GG_MAX_TAPS];
};

// SMALLC: few input channels (the 3-channel stem), "packed rows".  The input has been copied into a
// zero-bordered image with the channels padded to C (4 or 8) elements per pixel, so one kernel row of an
// output pixel is `tw` (32 or 64) contiguous elements starting at pixel (a*istr + kh, b*istr): a K-step of
// 64 covers 64 / tw kernel rows, every operand chunk is one bounds-check-free 16-byte load, and no patch
// matrix is materialised.  B rows hold the weights as [kh][tw] with zeros at the positions that correspond
// to padding channels, to pixels beyond the kernel width and to kernel rows >= T.
template <int BM, int BN, int WM, int WN, bool SMALLC>
__global__ __launch_bounds__(WM * WN * 64) void gather_gemm_kernel(const GatherGemmArgs p) {
    constexpr int NT = WM * WN * 64;           // threads per workgroup (4 or 8 waves)
    constexpr int TM = BM / WM, TN = BN / WN;  // per-wave tile
    constexpr int FM = TM / 16, FN = TN / 16;  // 16x16 fragments per wave
    constexpr int LR = NT / 8;                 // rows covered by one loader pass (8 chunks per row)
    constexpr int AP = BM / LR, BP = BN / LR;  // loader passes
    static_assert(WM * WN == 4 || WM * WN == 8, "four or eight waves per workgroup");
    static_assert(BM % LR == 0 && BN % LR == 0, "tile rows must be a multiple of the loader pass");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // staging buffers: [stage][A tile BM x 64 | B tile BN x 64]; a single-step launch touches stage 0 only
    constexpr int STAGE = (BM + BN) * GG_BK;
    bf16* sA = reinterpret_cast<bf16*>(smem);
    bf16* sB = sA + BM * GG_BK;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread loader state
    const int chunk = tid & 7, lrow = tid >> 3;
    int a_pix[AP], a_h[AP], a_w[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + lrow + LR * i;
        if (m < p.M) {
            const uint32_t img = fastdiv(m, p.div_ab);
            const uint32_t rem = m - img * (p.OA * p.OB);
            const uint32_t a = fastdiv(rem, p.div_b);
            const uint32_t b = rem - a * p.OB;
            a_pix[i] = img * (p.Hin * p.Win);
            a_h[i] = a * p.istr;
            a_w[i] = b * p.istr;
        } else {
            a_pix[i] = 0;
            a_h[i] = -(1 << 24);
            a_w[i] = 0;
        }
    }
    uint32_t b_off[BP];
    bool b_ok[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int n = n0 + lrow + LR * i;
        b_ok[i] = n < p.N;
        b_off[i] = (b_ok[i] ? n : 0) * p.ldb + chunk * 8;
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int csteps = p.C / GG_BK;
    const int nsteps = SMALLC ? (p.T * p.tw + GG_BK - 1) / GG_BK : p.T * csteps;
    bf16x8 ra[AP], rb[BP];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    auto load_step = [&](int t, int c0) {
        if (SMALLC) {
            // here t is the K-step index
            const int cpr = p.tw >> 3;  // 16-byte chunks per kernel row (4 or 8)
            const int kh = min(t * (8 / cpr) + chunk / cpr, p.T - 1);  // rows >= T have zero weights: any valid row will do
            const int piece = chunk % cpr;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const uint32_t off = static_cast<uint32_t>(a_pix[i] + (a_h[i] + kh) * p.Win + a_w[i]) * p.C + piece * 8;
                ra[i] = a_h[i] >= 0 ? *reinterpret_cast<const bf16x8*>(p.A + off) : zero8;
            }
#pragma unroll
            for (int i = 0; i < BP; ++i)
                rb[i] = b_ok[i] ? *reinterpret_cast<const bf16x8*>(p.B + b_off[i] + t * GG_BK) : zero8;
            return;
        }
        const int dh = p.tap_dh[t], dw = p.tap_dw[t], woff = p.tap_woff[t] + c0;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int ih = a_h[i] + dh, iw = a_w[i] + dw;
            const bool ok = static_cast<unsigned>(ih) < static_cast<unsigned>(p.Hin) &&
                            static_cast<unsigned>(iw) < static_cast<unsigned>(p.Win);
            const uint32_t off = static_cast<uint32_t>(a_pix[i] + ih * p.Win + iw) * p.C + c0 + chunk * 8;
            ra[i] = ok ? *reinterpret_cast<const bf16x8*>(p.A + off) : zero8;
        }
#pragma unroll
        for (int i = 0; i < BP; ++i)
            rb[i] = b_ok[i] ? *reinterpret_cast<const bf16x8*>(p.B + b_off[i] + woff) : zero8;
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int row = lrow + LR * i;
            *reinterpret_cast<bf16x8*>(sA + buf * STAGE + row * GG_BK + ((chunk ^ (row & 7)) << 3)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int row = lrow + LR * i;
            *reinterpret_cast<bf16x8*>(sB + buf * STAGE + row * GG_BK + ((chunk ^ (row & 7)) << 3)) = rb[i];
        }
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[FM], fb[FN];
            const int kc = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int row = wm * TM + i * 16 + (lane & 15);
                fa[i] = *reinterpret_cast<const bf16x8*>(sA + buf * STAGE + row * GG_BK + ((kc ^ (row & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int row = wn * TN + j * 16 + (lane & 15);
                fb[j] = *reinterpret_cast<const bf16x8*>(sB + buf * STAGE + row * GG_BK + ((kc ^ (row & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
    };

    if (nsteps > 0) {
        int t = 0, c0 = 0;
        load_step(0, 0);
        store_step(0);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const bool more = s + 1 < nsteps;
            if (more) {
                if (SMALLC) {
                    ++t;
                } else {
                    c0 += GG_BK;
                    if (c0 >= p.C) {
                        c0 = 0;
                        ++t;
                    }
                }
                load_step(t, c0);
            }
            compute(s & 1);
            if (more) store_step((s + 1) & 1);
            __syncthreads();
        }
    }

    // ---- epilogue: registers -> (bias) -> bf16 -> LDS -> 16-byte row segments + channel statistics
    constexpr int LDC = BN + 8;
    bf16* sC = reinterpret_cast<bf16*>(smem);  // [BM][LDC]; all K-loop LDS reads are behind a barrier
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int nl = wn * TN + j * 16 + (lane >> 4) * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) {
            const int n = n0 + nl;
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = (n + e < p.N) ? p.bias[n + e] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int ml = wm * TM + i * 16 + (lane & 15);
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = f32_to_bf16(acc[i][j][e] + bv[e]);
            *reinterpret_cast<bf16x4*>(sC + ml * LDC + nl) = v;
        }
    }
    __syncthreads();

    constexpr int CH = BN / 8;      // 16-byte chunks per tile row
    constexpr int RPP = NT / CH;    // rows per pass
    constexpr int NPASS = BM / RPP;
    const int ch = tid % CH, prow = tid / CH;
    const int ncol = n0 + ch * 8;
    float s1[8], s2[8], pv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s1[e] = s2[e] = 0.f;
        pv[e] = (p.pivot != nullptr && p.stats != nullptr && ncol < p.N) ? p.pivot[ncol + e] : 0.f;
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int row = prow + ps * RPP;
        const int m = m0 + row;
        if (m < p.M && ncol < p.N) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(sC + row * LDC + ch * 8);
            const uint32_t img = fastdiv(m, p.div_ab);
            const uint32_t rem = m - img * (p.OA * p.OB);
            const uint32_t a = fastdiv(rem, p.div_b);
            const uint32_t b = rem - a * p.OB;
            const size_t pix = (static_cast<size_t>(img) * p.Hout + a * p.ostr + p.oph) * p.Wout + b * p.ostr + p.opw;
            if (p.addend != nullptr) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(p.addend + pix * p.ldo + ncol);
                const unsigned mb = p.addend_bits != nullptr ? p.addend_bits[(pix * p.ldo + ncol) >> 3] : 0xffu;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    v[e] = f32_to_bf16(bf16_to_f32(v[e]) + (((mb >> e) & 1u) ? bf16_to_f32(av[e]) : 0.f));
            }
            if (p.act == 2) {
                const bf16x8 hv = *reinterpret_cast<const bf16x8*>(p.act_src + pix * p.ldo + ncol);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = f32_to_bf16(bf16_to_f32(v[e]) * bf16_to_f32(hv[e]));
            }
            if (p.act == 1) {
                bf16x8 gv;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float g, d;
                    pdt_gelu_both(bf16_to_f32(v[e]), g, d);
                    gv[e] = f32_to_bf16(g);
                    v[e] = f32_to_bf16(d);
                }
                *reinterpret_cast<bf16x8*>(p.out2 + pix * p.ldo + ncol) = gv;
            }
            *reinterpret_cast<bf16x8*>(p.Out + pix * p.ldo + ncol) = v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = bf16_to_f32(v[e]) - pv[e];
                s1[e] += f;
                s2[e] += f * f;
            }
        }
    }
    if (p.stats != nullptr) {
        // Workgroup totals, then one atomic per (statistic, channel) into slot tile_m % PDT_STAT_SLOTS.
        // Spreading over slots keeps same-address contention low; a finalize kernel folds the slots.
        __syncthreads();  // sC fully consumed
        float* sS = reinterpret_cast<float*>(smem);  // [RPP][2][BN]
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sS[(prow * 2 + 0) * BN + ch * 8 + e] = s1[e];
            sS[(prow * 2 + 1) * BN + ch * 8 + e] = s2[e];
        }
        __syncthreads();
        float* slot = p.stats + static_cast<size_t>(tile_m % PDT_STAT_SLOTS) * 2 * p.N;
        for (int idx = tid; idx < 2 * BN; idx += NT) {
            const int which = idx / BN, col = idx - which * BN;
            float tot = 0.f;
            for (int r = 0; r < RPP; ++r) tot += sS[(r * 2 + which) * BN + col];
            if (n0 + col < p.N) {
                __hip_atomic_fetch_add(slot + which * p.N + n0 + col, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // the pivot travels with the sums (every workgroup of a column tile stores the same value)
                if (which == 0 && p.pivot != nullptr)
                    p.stats[static_cast<size_t>(PDT_STAT_SLOTS) * 2 * p.N + n0 + col] = p.pivot[n0 + col];
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool SMALLC = false>
static int launch_gather_gemm(const GatherGemmArgs& a, hipStream_t stream) {
    const int grid = pdt_cdiv(a.M, BM) * pdt_cdiv(a.N, BN);
    constexpr size_t lds_loop = 2ull * (BM + BN) * GG_BK * sizeof(bf16);
    constexpr size_t lds_epi = static_cast<size_t>(BM) * (BN + 8) * sizeof(bf16);
    constexpr size_t lds_stats = ((WM * WN * 64) / (BN / 8)) * 2ull * BN * sizeof(float);
    constexpr size_t lds_full = lds_loop > lds_epi ? (lds_loop > lds_stats ? lds_loop : lds_stats)
                                                   : (lds_epi > lds_stats ? lds_epi : lds_stats);
    // A single K-step touches only the first staging buffer: request half the loop LDS so twice as many
    // workgroups are resident (these launches are bound by their loads and stores, not by MFMA).
    constexpr size_t lds_half = (lds_loop / 2) > lds_epi ? ((lds_loop / 2) > lds_stats ? lds_loop / 2 : lds_stats)
                                                         : (lds_epi > lds_stats ? lds_epi : lds_stats);
    const size_t lds = (!SMALLC && a.T * (a.C / GG_BK) <= 1) ? lds_half : lds_full;
    hipLaunchKernelGGL((gather_gemm_kernel<BM, BN, WM, WN, SMALLC>), dim3(grid), dim3(WM * WN * 64), lds, stream, a);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Ring variant: persistent workgroups, operands staged by LDS-DMA into a ring of NS stages.
//
// The kernel above keeps one K-step of loads in flight per workgroup and nothing during its epilogue, so
// launches with few K-steps per tile (the 1x1 convolutions) run at the latency of their loads.  Here
//   * a workgroup walks a strided sequence of output tiles; the K-steps of all its tiles form one
//     stream, so the loads of the next tile are in flight while the current tile is being written;
//   * every stage is filled by `global_load_lds_dwordx4` (no staging registers, no ds_write); the LDS
//     image is lane-linear per instruction (8 rows x 128 B), so the XOR swizzle is applied to the
//     per-lane SOURCE address and again by the fragment reads;
//   * NS - 2 stages stay in flight across the single barrier of a K-step: counted `s_waitcnt vmcnt`
//     for this wave's own loads of the stage about to be read, then a raw `s_barrier`;
//   * rows outside the image or beyond M / N read a page of zeros instead of being predicated off
//     (an LDS-DMA lane that does not execute would leave stale data in the stage);
//   * the epilogue is private to a wave (16 rows at a time through 2 KB of LDS, no workgroup barrier)
//     and BatchNorm partial sums stay in registers across tiles: one round of atomics per workgroup.
//
// vmcnt also counts the epilogue's stores and loads; they are younger than the loads a wait targets, so
// the constant used in the K-loop is at worst stricter than needed, never too weak.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void global_cvoid_t;

template <int N>
__device__ __forceinline__ void wait_vm_and_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// EPI selects what the epilogue reads besides the accumulators (bit mask): 1 bias, 2 addend (+ mask bits),
// 4 BatchNorm-backward sums with the ReLU mask recomputed from bn_y, 8 the same with the mask from bn_bits,
// 16 result times act_src (a stored gelu'), 32 gelu(result) to out2 and gelu'(result) in place of the result, 64 (with 8) sums for a second BatchNorm
// that shares dz (the shortcut BatchNorm of a projected block tail), 128 no channel statistics at all (a launch
// without a statistics request: the partial sums carried across tiles would only cost registers - on the ping-pong
// schedule, which runs at the 256-register limit, they made the compiler spill, and every reload of a spilled
// value in the epilogue is a vmcnt(0) wait that also drains the stores and the operand prefetch in flight).
// It is a template parameter because an ordinary load anywhere on the path makes the compiler drain the
// DMA queue (vmcnt(0)) at its use.
//
// PLAIN: single tap at offset (0, 0) that is in bounds for every output pixel (linear layers, 1x1 convolutions of
// any stride).  The source address of a DMA piece is then the same row pointer advanced by BK elements per
// K-step, so the loader keeps running 64-bit pointers (one add per piece and stage instead of a dozen vector
// ALU instructions of bounds checks and index arithmetic) and issues unconditionally - past its last tile from
// the zero page - which puts loads and MFMAs of a K-step into one basic block for the scheduler to interleave.
//
// PP ("ping-pong", 256x256 tile, 2 x 4 waves, two stages): the K-step is cut into four phases of 16 MFMAs (one
// quadrant of the wave's 128x64 output each), every phase bracketed by two workgroup barriers, and the wave rows run
// one barrier apart: while the waves of row 0 issue the MFMAs of a phase, the waves of row 1 - one per SIMD next to
// them - issue their fragment reads and LDS-DMA requests, and vice versa, so the matrix pipe of every SIMD always has
// a wave feeding it while the other wave's LDS traffic is hidden behind it (with one schedule for all eight waves,
// both waves of a SIMD read at the same time and then both compete for the matrix pipe).  Operands are staged in
// units of 128 rows (the rows of one phase: the lower or upper half of every wave's A rows or B columns), one unit
// per phase, six units ahead of the reads, with a counted wait that leaves four units in flight across the barriers.
//
// XF (single-tap launches on the two-stage tiles): BatchNorm + ReLU of the producing layer applied to the A operand.
// A stage is transformed in place in LDS right after it has landed - every thread rewrites four 16-byte chunks with
// the per-channel scale / shift read from a table in LDS - followed by a second barrier, so each element is
// transformed once per tile (doing it on the MFMA fragments would repeat the arithmetic in all four waves of a wave
// row).  These launches are bound by their output stores; the extra LDS pass and barrier hide behind them.
// 16 bytes per lane from `base + soff + voff` straight into LDS through a buffer instruction: `soff` is wave-uniform, a
// lane whose `voff` lies beyond the descriptor's 2 GiB range reads nothing and has zeros written for it.
__device__ __forceinline__ void buffer_load_lds_16(bf16* base, lds_void_t* dst, uint32_t voff, int soff) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, 0);
}

template <int BM, int BN, int WM, int WN, int NS, int EPI, int BK = GG_BK, bool PLAIN = false, bool PP = false, bool XF = false>
__global__ __launch_bounds__(WM * WN * 64) void gather_gemm_ring_kernel(const GatherGemmArgs p) {
    constexpr int NW = WM * WN;
    static_assert(!XF || (PLAIN && !PP && BK == 64), "operand transform: single-tap launches on the plain ring schedule");
    static_assert(!PP || (BM == 256 && BN == 256 && WM == 2 && WN == 4 && NS == 2 && BK == 64), "ping-pong schedule: 256x256, 2x4 waves");
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int PR = 512 / BK;                            // rows per 1 KB DMA piece (8 rows of 128 B or 16 of 64 B)
    constexpr int CPR = BK / 8;                             // 16-byte chunks per row
    constexpr int AI = (BM / PR) / NW, BI = (BN / PR) / NW; // LDS-DMA instructions per wave and stage
    constexpr int LPW = AI + BI;
    constexpr int STAGE = (BM + BN) * BK;                  // elements per ring stage: [A tile | B tile]
    constexpr int EP = TN;                                 // pitch of the wave-private epilogue rows: 16-byte slots XOR-ed with the row, and in rows 8-15 the
                                                           // two 8-byte halves of a slot exchanged - the 16 lanes of a ds_write_b64 group (one row each, banks modulo
                                                           // 128 bytes) then cover every bank once
    constexpr int CH = TN / 8;                             // 16-byte chunks per epilogue row
    constexpr int RPP = 64 / CH;                           // rows per epilogue pass
    static_assert(BK == 32 || BK == 64, "K-step of 32 or 64");
    static_assert((BM / PR) % NW == 0 && (BN / PR) % NW == 0, "whole DMA instructions per wave");
    static_assert(NS >= 2 && (NS - 2) * LPW <= 63, "vmcnt is a 6-bit counter");
    static_assert(16 % RPP == 0, "epilogue passes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* ring = reinterpret_cast<bf16*>(smem);  // [NS][STAGE], then [NW][16][EP]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    bf16* epi = ring + NS * STAGE + wave * 16 * EP;
    float* xf_tab = reinterpret_cast<float*>(ring + NS * STAGE + NW * 16 * EP);  // XF: [2][C] scale, shift
    // Ping-pong schedule with a bias: the TN bias values of this wave's columns are fetched once per tile by LDS-DMA
    // (wave-private row behind the epilogue staging area), issued in front of the tile's K-loop and retired by its
    // counted waits long before the epilogue reads them.  Loading them with ordinary loads inside the epilogue - 16 per
    // lane for each of the 8 row groups - made every group wait for vmcnt(0): for its own loads, for the stores of the
    // group before it and for the next tile's operand prefetch.
    constexpr bool BIAS_LDS = PP && (EPI & 1);
    static_assert(!BIAS_LDS || TN == 64, "one bias value per lane");
    float* bias_row = reinterpret_cast<float*>(ring + NS * STAGE + NW * 16 * EP) + wave * TN;
    if constexpr (XF) {
        for (int i = tid; i < 2 * p.C; i += NW * 64) xf_tab[i] = p.xf[2 * p.C + i];
        __syncthreads();
    }

    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int total_tiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int first = xcd_remap(blockIdx.x, G);
    const int my_tiles = (total_tiles - first + G - 1) / G;
    const int csteps = p.C / BK;
    const int nsteps = p.T * csteps;
    const int total_steps = my_tiles * nsteps;

    // ---- loader: a cursor over (tile, tap, channel block) that runs NS - 1 stages ahead of the MFMAs
    // Chunk swizzle, applied to the DMA source address and again by the fragment reads.  128-byte rows: chunk ^
    // (row & 7).  64-byte rows (BK = 32): four rows share a 256-byte bank line, and the lane groups of
    // ds_read_b128 pair rows {0-3, 12-15} with rows {4-11} of the next chunk; XOR-ing the chunk with
    // f((row >> 2) & 3), f = {0, 2, 3, 1}, puts the 16 lanes of every group on 16 different 16-byte slots.
    auto swz = [](int row) { return BK == 64 ? (row & 7) : ((0x78 >> (((row >> 2) & 3) * 2)) & 3); };
    const int lrow = lane / CPR;                              // row of this lane inside a DMA piece
    const int kchunk = ((lane % CPR) ^ swz(lrow)) << 3;       // swizzled source chunk (pieces start at multiples of 16 rows)
    // Offset of this lane's chunk inside an activation row.  Packed rows (few input channels, see gather_gemm_kernel):
    // a K-step of 64 spans 64 / tw kernel rows of tw contiguous elements each, so chunk j lies (j / (tw/8)) image rows
    // further down; a "tap" is then one K-step (dh = its first kernel row) and C = 64 makes it a single channel block.
    const int kchunk_a = p.tw ? ((kchunk >> 3) / (p.tw >> 3)) * p.Win * p.cs + (((kchunk >> 3) % (p.tw >> 3)) << 3) : kchunk;
    int ld_tile = first, ld_t = 0, ld_c0 = 0, ld_issued = 0, ld_slot = 0;
    // Gather loader (multi-tap launches): buffer instructions with LDS destination.  Per tile, a resource descriptor
    // whose base is the input pixel of the tile's first row (minus the largest backward reach of a tap) and, per DMA
    // piece, this lane's byte offset from it and its pixel's input coordinates.  A K-step adds a wave-uniform byte
    // offset (tap displacement + channel block) in the instruction's scalar offset, so the issue sequence is the bounds
    // test and one select per piece: a lane whose tap falls outside the image gets an offset beyond the descriptor's
    // range and the hardware writes zeros for it - no page of zeros, no 64-bit address arithmetic per lane.
    constexpr uint32_t OUT_OF_RANGE = 0x80000000u;
    bf16* ga_base = const_cast<bf16*>(p.A);  // (descriptor bases; the descriptors are formed where they are used)
    bf16* gb_base = const_cast<bf16*>(p.B);
    uint32_t a_lane[AI], b_lane[BI];
    int a_h[AI], a_w[AI];
    int halo = 0;  // elements: how far the most backward tap reaches in front of an output pixel's own input pixel
    if constexpr (!PLAIN) {
        for (int t = 0; t < p.T; ++t) halo = max(halo, -(p.tap_dh[t] * p.Win + p.tap_dw[t]) * p.cs);
    }
    // First tile row of this wave's DMA piece i.  Plain ring: consecutive pieces.  Ping-pong: pieces 2i' and 2i'+1 of
    // staging unit i >> 1 (A: rows [64u, 64u + 64) of both wave rows; B: columns [32v, 32v + 32) of all four wave columns).
    auto row_a = [&](int i) {
        if constexpr (PP) {
            const int q = 2 * wave + (i & 1);
            return (q >> 3) * 128 + (i >> 1) * 64 + (q & 7) * 8;
        } else {
            return (wave * AI + i) * PR;
        }
    };
    auto row_b = [&](int i) {
        if constexpr (PP) {
            const int q = 2 * wave + (i & 1);
            return (q >> 2) * 64 + (i >> 1) * 32 + (q & 3) * 8;
        } else {
            return (wave * BI + i) * PR;
        }
    };
    auto pixel_in = [&](int m, int& ah, int& aw) -> size_t {
        const uint32_t img = fastdiv(m, p.div_ab);
        const uint32_t rem = m - img * (p.OA * p.OB);
        const uint32_t a = fastdiv(rem, p.div_b);
        const uint32_t b = rem - a * p.OB;
        ah = a * p.istr;
        aw = b * p.istr;
        return static_cast<size_t>(img) * (p.Hin * p.Win) + static_cast<uint32_t>(ah * p.Win + aw);
    };
    auto loader_tile = [&](int tile) {
        const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
        int h0, w0;
        const size_t pix0 = pixel_in(min(tile_m * BM, p.M - 1), h0, w0);  // (wave-uniform)
        ga_base = const_cast<bf16*>(p.A) + (static_cast<ptrdiff_t>(pix0 * static_cast<uint32_t>(p.cs)) - halo);
        gb_base = const_cast<bf16*>(p.B) + static_cast<size_t>(tile_n * BN) * p.ldb;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int m = tile_m * BM + row_a(i) + lrow;
            // (one assignment per array element, outside the branch: stores to different elements in the two arms get
            // merged into one store through a selected address, which keeps the arrays in scratch memory)
            int ah = -(1 << 24), aw = 0;
            uint32_t lane_off = OUT_OF_RANGE;
            if (m < p.M) {
                const size_t pixel = pixel_in(m, ah, aw);
                lane_off = (static_cast<uint32_t>(pixel - pix0) * static_cast<uint32_t>(p.cs) + kchunk_a) * 2u;
            }
            a_lane[i] = lane_off;
            a_h[i] = ah;
            a_w[i] = aw;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int r = row_b(i) + lrow;
            b_lane[i] = tile_n * BN + r < p.N ? (static_cast<uint32_t>(r) * static_cast<uint32_t>(p.ldb) + kchunk) * 2u : OUT_OF_RANGE;
        }
    };
    // pieces [a_lo, a_hi) of the A tile and [b_lo, b_hi) of the B tile of the stage under the cursor
    // The tap under the cursor, read from the argument block when the cursor moves onto it - a K-step (ping-pong: a phase)
    // before its first use - instead of inside the issue sequence, where the scalar loads' latency sat in front of the
    // DMA requests of every K-step.
    int ld_dh = p.tap_dh[0], ld_dw = p.tap_dw[0], ld_wo = p.tap_woff[0];
    auto issue_pieces = [&](int a_lo, int a_hi, int b_lo, int b_hi) {
        const int dh = ld_dh, dw = ld_dw;
        const int a_soff = (halo + (dh * p.Win + dw) * p.cs + ld_c0) * 2;  // (wave-uniform, never negative)
        const int b_soff = (ld_wo + ld_c0) * 2;
        bf16* sA = ring + ld_slot * STAGE;
        bf16* sB = sA + BM * BK;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            if (i < a_lo || i >= a_hi) continue;
            const int ih = a_h[i] + dh, iw = a_w[i] + dw;
            const bool ok = static_cast<unsigned>(ih) < static_cast<unsigned>(p.Hin) &&
                            static_cast<unsigned>(iw) < static_cast<unsigned>(p.Win);
            buffer_load_lds_16(ga_base, (lds_void_t*)(sA + row_a(i) * BK), ok ? a_lane[i] : OUT_OF_RANGE, a_soff);
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            if (i < b_lo || i >= b_hi) continue;
            buffer_load_lds_16(gb_base, (lds_void_t*)(sB + row_b(i) * BK), b_lane[i], b_soff);
        }
    };
    auto issue_stage = [&]() {
        issue_pieces(0, AI, 0, BI);
        ++ld_issued;
        ld_slot = ld_slot + 1 == NS ? 0 : ld_slot + 1;
    };
    // Moves the loader's cursor to the next (channel block, tap, tile).  Past the last tile the row indices fall
    // beyond M, so every A piece comes from the zero page and the loader can keep issuing unconditionally.
    auto advance_cursor = [&]() {
        ld_c0 += BK;
        if (ld_c0 >= p.C) {
            ld_c0 = 0;
            if (++ld_t == p.T) {
                ld_t = 0;
                ld_tile += G;
                loader_tile(ld_tile);
            }
            ld_dh = p.tap_dh[ld_t];
            ld_dw = p.tap_dw[ld_t];
            ld_wo = p.tap_woff[ld_t];
        }
    };

    // PLAIN loader state: a wave-uniform address per operand (origin of the tile's rows + K position, held in scalar
    // registers and advanced by scalar adds) and this lane's byte offsets from it, which change once per tile.  The DMA
    // instructions take "scalar base + 32-bit vector offset", so a K-step costs no vector arithmetic and the loader holds
    // 8 vector registers (running 64-bit pointers and per-piece strides held 24 and cost 16-24 instructions per K-step).
    // Rows beyond M (columns beyond N) re-read the last valid row: what they produce is neither stored nor counted in
    // the statistics, and past its last tile the loader re-reads tile 0, which nothing consumes.
    const char* a_base = nullptr;
    const char* b_base = nullptr;
    uint32_t pa_lane[AI], pb_lane[BI];
    // output row m reads input pixel m (linear layers, stride-1 1x1 convolutions): no index arithmetic in the loader
    const bool a_identity = p.istr == 1 && p.OA == p.Hin && p.OB == p.Win;
    auto pixel_of = [&](int m) -> size_t {
        if (a_identity) return static_cast<uint32_t>(m);
        const uint32_t img = fastdiv(m, p.div_ab);
        const uint32_t rem = m - img * (p.OA * p.OB);
        const uint32_t a = fastdiv(rem, p.div_b);
        const uint32_t b = rem - a * p.OB;
        return static_cast<size_t>(img) * (p.Hin * p.Win) + (a * p.istr) * p.Win + b * p.istr;
    };
    auto plain_tile = [&](int tile) {
        const int t = tile < total_tiles ? tile : 0;
        const int tile_m = t / tiles_n, tile_n = t - tile_m * tiles_n;
        const int mt = tile_m * BM, nt = tile_n * BN;
        const size_t pix0 = pixel_of(mt);
        a_base = reinterpret_cast<const char*>(p.A + pix0 * static_cast<uint32_t>(p.C));
        b_base = reinterpret_cast<const char*>(p.B + static_cast<size_t>(nt) * p.ldb + p.tap_woff[0]);
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int m = min(mt + row_a(i) + lrow, p.M - 1);
            pa_lane[i] = (static_cast<uint32_t>(pixel_of(m) - pix0) * static_cast<uint32_t>(p.C) + kchunk) * 2u;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int n = min(nt + row_b(i) + lrow, p.N - 1) - nt;
            pb_lane[i] = (static_cast<uint32_t>(n) * static_cast<uint32_t>(p.ldb) + kchunk) * 2u;
        }
    };
    auto plain_pieces = [&](int a_lo, int a_hi, int b_lo, int b_hi) {
        bf16* sA = ring + ld_slot * STAGE;
        bf16* sB = sA + BM * BK;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            if (i < a_lo || i >= a_hi) continue;
            __builtin_amdgcn_global_load_lds((global_cvoid_t*)(a_base + pa_lane[i]), (lds_void_t*)(sA + row_a(i) * BK), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            if (i < b_lo || i >= b_hi) continue;
            __builtin_amdgcn_global_load_lds((global_cvoid_t*)(b_base + pb_lane[i]), (lds_void_t*)(sB + row_b(i) * BK), 16, 0, 0);
        }
    };
    auto plain_issue = [&]() {
        plain_pieces(0, AI, 0, BI);
        ld_slot = ld_slot + 1 == NS ? 0 : ld_slot + 1;
    };
    // (called after the last piece of a stage has been issued)
    auto plain_advance = [&]() {
        ld_c0 += BK;
        a_base += BK * 2;
        b_base += BK * 2;
        if (ld_c0 >= p.C) {
            ld_c0 = 0;
            ld_tile += G;
            plain_tile(ld_tile);
        }
    };

    f32x4 acc[FM][FN];
    // `first` (a std::bool_constant): the first K-step of a tile starts from zero - its first 32-deep slice feeds the
    // MFMAs a constant-zero accumulator operand, so the 128 accumulator registers are never cleared with moves (a
    // quarter of the vector instructions a tile with a single K-step issues outside its MFMAs).
    // On the plain schedule of the 256x256 tile a second copy of the pinned K-step costs ~150 bytes of scratch per lane,
    // so the single-tap flavours there run their first K-step unpinned (the scheduler is left to place its reads), and
    // the multi-tap, operand-transform and bias flavours - which spill inside the tile loop even so - clear their
    // accumulators with moves as before.
    constexpr bool PEEL = PP || !(BM == 256 && BN == 256) || (PLAIN && !XF && (EPI & 1) == 0);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int slot, auto first) {
        constexpr bool FIRST = decltype(first)::value;
        const bf16* sA = ring + slot * STAGE;
        const bf16* sB = sA + BM * BK;
        // every 32-deep slice of the K-step has its own fragment registers, so the reads of the next slice (or
        // of the second half of the rows) are issued under the MFMAs of the current one (schedule pinned below)
        constexpr int KK = BK / 32;
        bf16x8 fa[KK][FM], fb[KK][FN];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int kc = kk * 4 + (lane >> 4);
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int row = wn * TN + j * 16 + (lane & 15);
                fb[kk][j] = *reinterpret_cast<const bf16x8*>(sB + row * BK + ((kc ^ swz(row)) << 3));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int row = wm * TM + i * 16 + (lane & 15);
                fa[kk][i] = *reinterpret_cast<const bf16x8*>(sA + row * BK + ((kc ^ swz(row)) << 3));
            }
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[kk][i], (FIRST && kk == 0) ? zero4 : acc[i][j], 0, 0, 0);
        // Pinned schedule.  Reads are issued in the order of the loops above (per 32-deep slice: B fragments, then
        // A fragments by row block), MFMAs row block by row block; a fragment is read a few MFMA groups before
        // its use and is dead right after it, which keeps the live fragment registers far below KK * (FM + FN).
        constexpr int READS = KK * (FM + FN), GROUPS = KK * FM;
        constexpr int LEAD = FN + 2;                                   // reads issued before the first MFMA
        constexpr int PER = (READS - LEAD + GROUPS - 2) / (GROUPS - 1) > 1 ? 2 : 1;  // reads after each MFMA group
        if constexpr (FIRST && BM == 256 && BN == 256) return;  // (unpinned: see PEEL)
        __builtin_amdgcn_sched_group_barrier(0x100, LEAD, 0);
#pragma unroll
        for (int gi = 0; gi < GROUPS; ++gi) {
            __builtin_amdgcn_sched_group_barrier(0x008, FN, 0);
            // (a request beyond the last read simply matches nothing)
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (PER == 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    };

    // BatchNorm partial sums of this lane's 8 channels, carried across tiles
    constexpr bool NOSTATS = (EPI & 128) != 0;
    static_assert(!NOSTATS || (EPI & 12) == 0, "BatchNorm-backward sums are statistics");
    const int ech = lane % CH, erow = lane / CH;
    float s1[8], s2[8], s3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = s3[e] = 0.f;
    // Forward statistics are accumulated about a pivot so that the variance is never a small difference of large sums.
    // Inside the row groups the pivot is the first value the lane sees for each of its channels (f0f, exact in bf16; no
    // load in the hot path); when the sums are flushed they are re-based to the launch's pivot p.pivot[c]:
    // sum(y - P) = sum(y - f0) + n (f0 - P), sum((y - P)^2) = sum((y - f0)^2) + 2 (f0 - P) sum(y - f0) + n (f0 - P)^2.
    float f0f[8];  // (as floats: converting a packed pivot for every chunk cost as much as the sums themselves)
#pragma unroll
    for (int e = 0; e < 8; ++e) f0f[e] = 0.f;
    float fcnt = 0.f;
    bool f0_fresh = true;  // the next chunk this lane stages becomes its pivot (set by every flush; wave-uniform)
    int stats_n0 = -1;
    auto flush_stats = [&]() {
        if (NOSTATS || p.stats == nullptr || stats_n0 < 0) return;
        if constexpr ((EPI & (12 | 48)) == 0) {
            const int pc = stats_n0 + wn * TN + ech * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float P = (p.pivot != nullptr && pc < p.N) ? p.pivot[pc + e] : 0.f;
                const float dl = f0f[e] - P;
                s2[e] += dl * (2.f * s1[e] + fcnt * dl);
                s1[e] += fcnt * dl;
                // the pivot travels with the sums: last slot, first row (every flushing workgroup stores the same value)
                if (p.pivot != nullptr && lane < CH && pc < p.N) p.stats[static_cast<size_t>(PDT_STAT_SLOTS) * 2 * p.N + pc + e] = P;
            }
            fcnt = 0.f;
            f0_fresh = true;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = CH; o < 64; o <<= 1) {
                s1[e] += __shfl_xor(s1[e], o, 64);
                s2[e] += __shfl_xor(s2[e], o, 64);
                if (EPI & 64) s3[e] += __shfl_xor(s3[e], o, 64);
            }
        }
        const int ncol = stats_n0 + wn * TN + ech * 8;
        if (lane < CH && ncol < p.N) {
            float* slot = p.stats + static_cast<size_t>(blockIdx.x % PDT_STAT_SLOTS) * p.stat_rows * p.N;
            if (EPI & 12) {
                // The row groups accumulate the raw moments sum(dz) and sum(dz * y); the normalised sum BatchNorm's backward
                // wants is rstd * (sum(dz * y) - mean * sum(dz)).  Applying mean and rstd here, once per workgroup, keeps
                // the per-channel constants out of the row groups: loaded there they cost a vmcnt(0) wait per pass, which
                // also drained the stores of the pass before and the operands requested ahead for the next groups.
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s2[e] = (s2[e] - p.bn_consts[ncol + e] * s1[e]) * p.bn_consts[p.N + ncol + e];
                    if (EPI & 64) s3[e] = (s3[e] - p.bn_consts2[ncol + e] * s1[e]) * p.bn_consts2[p.N + ncol + e];
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                __hip_atomic_fetch_add(slot + ncol + e, s1[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(slot + p.N + ncol + e, s2[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (EPI & 64) __hip_atomic_fetch_add(slot + 2 * p.N + ncol + e, s3[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = s3[e] = 0.f;
    };

    // ---- ping-pong schedule: staging units in the order they are read (A lower halves, B lower halves, B upper halves,
    // A upper halves of a K-step), one unit per phase; after the last unit of a K-step the cursor moves on
    auto pp_unit = [&](int r) {
        const int a_lo = r == 0 ? 0 : (r == 3 ? 2 : 0), a_hi = r == 0 ? 2 : (r == 3 ? 4 : 0);
        const int b_lo = r == 1 ? 0 : (r == 2 ? 2 : 0), b_hi = r == 1 ? 2 : (r == 2 ? 4 : 0);
        if constexpr (PLAIN) plain_pieces(a_lo, a_hi, b_lo, b_hi);
        else issue_pieces(a_lo, a_hi, b_lo, b_hi);
        if (r == 3) {
            ld_slot ^= 1;
            if constexpr (PLAIN) plain_advance();
            else advance_cursor();
        }
    };
    if constexpr (PP) {
        if constexpr (PLAIN) plain_tile(first);
        else loader_tile(first);
        // six units ahead: the whole first K-step and the first two units of the second
        pp_unit(0); pp_unit(1); pp_unit(2); pp_unit(3); pp_unit(0); pp_unit(1);
        // four units (two DMA instructions each per wave) may stay in flight: the first two have landed
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        if (wm == 1) asm volatile("s_barrier" ::: "memory");  // wave row 1 runs one barrier behind row 0 from here on
    } else if constexpr (PLAIN) {
        plain_tile(first);
#pragma unroll
        for (int k = 0; k < NS - 1; ++k) {
            plain_issue();
            plain_advance();
        }
    } else {
        loader_tile(first);
#pragma unroll
        for (int k = 0; k < NS - 1; ++k) {
            issue_stage();
            advance_cursor();
        }
    }

    int g = 0, slot = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int tile = first + ti * G;
        const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
        const int m0 = tile_m * BM, n0 = tile_n * BN;
        if constexpr (!PEEL) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        if constexpr (BIAS_LDS) {
            const float* src = p.bias + min(n0 + wn * TN + lane, p.N - 1);
            __builtin_amdgcn_global_load_lds((global_cvoid_t*)src, (lds_void_t*)bias_row, 4, 0, 0);
        }
        if constexpr (PP) {
            auto pp_step = [&](auto first) {
                constexpr bool FIRST = decltype(first)::value;
                const bf16* sA = ring + slot * STAGE;
                const bf16* sB = sA + BM * BK;
                bf16x8 fa[2][4], fbl[2][2], fbh[2][2];
                auto read_a = [&](int half) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int row = wm * TM + (half * 4 + i) * 16 + (lane & 15);
                            fa[kk][i] = *reinterpret_cast<const bf16x8*>(sA + row * BK + (((kk * 4 + (lane >> 4)) ^ swz(row)) << 3));
                        }
                };
                auto read_b = [&](int half, bf16x8(&fb)[2][2]) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int row = wn * TN + (half * 2 + j) * 16 + (lane & 15);
                            fb[kk][j] = *reinterpret_cast<const bf16x8*>(sB + row * BK + (((kk * 4 + (lane >> 4)) ^ swz(row)) << 3));
                        }
                };
                // One phase: [this phase's fragment reads and DMA unit were issued by the caller] -> counted wait that
                // retires the unit the next phase reads -> barrier -> 16 MFMAs of one output quadrant -> barrier.
                auto mfma_quadrant = [&](int ih, int jh, bf16x8(&fb)[2][2]) {
                    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                acc[ih * 4 + i][jh * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    fb[kk][j], fa[kk][i], (FIRST && kk == 0) ? zero4 : acc[ih * 4 + i][jh * 2 + j], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_barrier" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                };
                // phase 1: lower A rows x lower B columns
                pp_unit(2);
                read_b(0, fbl);
                read_a(0);
                mfma_quadrant(0, 0, fbl);
                // phase 2: lower A rows x upper B columns
                pp_unit(3);
                read_b(1, fbh);
                mfma_quadrant(0, 1, fbh);
                // phase 3: upper A rows x upper B columns
                pp_unit(0);
                read_a(1);
                mfma_quadrant(1, 1, fbh);
                // phase 4: upper A rows x lower B columns (fragments already in registers)
                pp_unit(1);
                mfma_quadrant(1, 0, fbl);
                slot ^= 1;
            };
            pp_step(std::true_type{});
            for (int s = 1; s < nsteps; ++s) pp_step(std::false_type{});
            g += nsteps;
        } else {
        auto ring_step = [&](int s, auto first) {
            if constexpr (PLAIN) {
                // NS - 1 stages are always in flight here (the loader never stops issuing)
                wait_vm_and_barrier<(NS - 2) * LPW>();
                plain_issue();
                if constexpr (XF) {
                    bf16* sAx = ring + slot * STAGE;
                    const int cbase = s * BK;
                    // A thread's chunks of a stage lie 64 rows apart (NW * 64 threads cover 64 rows of 8 chunks), so they
                    // share the physical position and the row's swizzle: one set of constants per thread and K-step
                    // instead of one per chunk (the table reads were four fifths of the transform's LDS traffic).
                    static_assert((NW * 64) % 64 == 0 && BK == 64, "a thread's chunks share their channels");
                    const int pos = tid & 7;
                    const int cb = cbase + ((pos ^ ((tid >> 3) & 7)) << 3);  // channels of this thread's physical chunks
                    float sc[8], sh[8];
                    *reinterpret_cast<f32x4*>(sc) = *reinterpret_cast<const f32x4*>(xf_tab + cb);
                    *reinterpret_cast<f32x4*>(sc + 4) = *reinterpret_cast<const f32x4*>(xf_tab + cb + 4);
                    *reinterpret_cast<f32x4*>(sh) = *reinterpret_cast<const f32x4*>(xf_tab + p.C + cb);
                    *reinterpret_cast<f32x4*>(sh + 4) = *reinterpret_cast<const f32x4*>(xf_tab + p.C + cb + 4);
#pragma unroll
                    for (int i = 0; i < (BM * 8) / (NW * 64); ++i) {
                        const int row = (tid >> 3) + i * ((NW * 64) >> 3);
                        bf16* a = sAx + row * BK + (pos << 3);
                        bf16x8 v = *reinterpret_cast<const bf16x8*>(a);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = f32_to_bf16(fmaxf(bf16_to_f32(v[e]) * sc[e] + sh[e], 0.f));
                        *reinterpret_cast<bf16x8*>(a) = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
                compute(slot, first);
                plain_advance();
            } else {
                // stage g was issued NS - 1 steps ago; NS - 2 younger stages stay in flight
                wait_vm_and_barrier<(NS - 2) * LPW>();
                // every wave has finished reading the stage of step g - 1: refill it
                issue_stage();
                compute(slot, first);
                advance_cursor();
            }
            slot = slot + 1 == NS ? 0 : slot + 1;
        };
        if constexpr (PEEL) {
            if (nsteps > 0) {
                ring_step(0, std::true_type{});
                for (int s = 1; s < nsteps; ++s) ring_step(s, std::false_type{});
            } else {  // a launch without taps (the untouched phases of a strided data gradient) stores zeros (+ addend)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = zero4;
            }
        } else {
            for (int s = 0; s < nsteps; ++s) ring_step(s, std::false_type{});
        }
        g += nsteps;
        }

        // ---- epilogue (wave-private): 16 rows at a time -> bf16 -> LDS -> 16-byte row segments
        if constexpr (PP) {
            // Both wave rows write their part of the tile at the same time: row 0 gives up its one-barrier lead here
            // (this barrier pairs with the last barrier of row 1's K-loop) and row 1 falls back behind it after the
            // epilogue.  Left one barrier apart, each row would wait at its next barrier for the other's whole epilogue.
            if (wm == 0) asm volatile("s_barrier" ::: "memory");
        }
        if (n0 != stats_n0) {
            flush_stats();
            stats_n0 = n0;
        }
        const int ncol = n0 + wn * TN + ech * 8;
        float msc[8], msh[8];  // EPI & 4: scale and shift of this lane's 8 channels, for the recomputed ReLU mask
        if constexpr ((EPI & 12) == 4) {
            const int nc = min(ncol, p.N - 8);
            *reinterpret_cast<f32x4*>(msc) = *reinterpret_cast<const f32x4*>(p.bn_consts + 2 * p.N + nc);
            *reinterpret_cast<f32x4*>(msc + 4) = *reinterpret_cast<const f32x4*>(p.bn_consts + 2 * p.N + nc + 4);
            *reinterpret_cast<f32x4*>(msh) = *reinterpret_cast<const f32x4*>(p.bn_consts + 3 * p.N + nc);
            *reinterpret_cast<f32x4*>(msh + 4) = *reinterpret_cast<const f32x4*>(p.bn_consts + 3 * p.N + nc + 4);
        }
        // (the statistics-free flavour is only launched for densely stored outputs: no branch on it in the row groups)
        const bool dense_out = NOSTATS || (p.ostr == 1 && p.oph == 0 && p.opw == 0 && p.OA == p.Hout && p.OB == p.Wout);
        // Output pixels of this lane's rows in a 16-row group and the loads of the epilogue operands that share the
        // output's layout (shortcut gradient, BatchNorm inputs, pre-activation).  Two register sets: the operands of
        // group i + 1 are requested before group i is staged and processed, so their latency is covered by a whole
        // group's work instead of one LDS round trip.
        constexpr int NPS = 16 / RPP;
        size_t pixo_[3][NPS];
        bool rok_[3][NPS];
        bf16x8 yv_[3][NPS], av_[3][NPS], y2v_[3][NPS];
        unsigned amb_[3][NPS], bnb_[3][NPS];
        // Densely stored outputs: row m of the tile is row m of Out, so a group's offsets are the tile's first offset plus
        // a wave-uniform multiple of the row pitch (one 64-bit multiply per tile instead of one per pass).  Rows beyond M
        // are not stored; where the epilogue also loads operands of the output's layout they read the last valid row.
        constexpr bool EPI_LOADS = (EPI & (2 | 12 | 16 | 64)) != 0;
        const size_t dense_base = static_cast<size_t>(static_cast<uint32_t>(m0 + wm * TM + erow)) * static_cast<uint32_t>(p.ldo) + min(ncol, p.N - 8);
        const size_t dense_last = static_cast<size_t>(static_cast<uint32_t>(p.M - 1)) * static_cast<uint32_t>(p.ldo) + min(ncol, p.N - 8);
        auto fetch = [&](int i, int st) {
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int m = m0 + wm * TM + i * 16 + ps * RPP + erow;
                rok_[st][ps] = m < p.M && ncol < p.N;
                const int mc = min(m, p.M - 1);
                if (dense_out) {  // output pixel m is row m of Out (linear layers, stride-1 convolutions): no index arithmetic
                    const size_t o = dense_base + static_cast<size_t>(static_cast<uint32_t>((i * 16 + ps * RPP) * p.ldo));
                    pixo_[st][ps] = (EPI_LOADS && m >= p.M) ? dense_last : o;
                } else {
                    const uint32_t img = fastdiv(mc, p.div_ab);
                    const uint32_t rem = mc - img * (p.OA * p.OB);
                    const uint32_t a = fastdiv(rem, p.div_b);
                    const uint32_t b = rem - a * p.OB;
                    pixo_[st][ps] = ((static_cast<size_t>(img) * p.Hout + a * p.ostr + p.oph) * p.Wout + b * p.ostr + p.opw) * p.ldo +
                                    min(ncol, p.N - 8);
                }
                // Operands with the layout of the output are read once: streamed past the caches when they are large.
                // The loads are inline assembly and their completion is awaited by hand (counted vmcnt in front of the
                // group that uses them, see below): for compiler-issued loads the waits came out as vmcnt(0) in front
                // of every group, which also waited for the groups requested ahead - the prefetch bought nothing.
                // Every request issues the same number of loads (LPP per pass) whatever the run-time arguments are.
                auto ld8 = [&](const bf16* q) {
                    // (no branch around the load: a value merged from two assembly statements may be copied at the merge,
                    // before it has arrived; the streaming hint is right for read-once operands of any size)
                    u32x4 r;
                    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r) : "v"(q) : "memory");
                    return __builtin_bit_cast(bf16x8, r);
                };
                auto ld1 = [&](const unsigned char* q) {
                    unsigned r;
                    asm volatile("global_load_ubyte %0, %1, off" : "=v"(r) : "v"(q) : "memory");
                    return r;
                };
                if (EPI & 12) yv_[st][ps] = ld8(p.bn_y + pixo_[st][ps]);
                if (EPI & 64) y2v_[st][ps] = ld8(p.bn_y2 + pixo_[st][ps]);
                if (EPI & 16) yv_[st][ps] = ld8(p.act_src + pixo_[st][ps]);
                if (EPI & 8) bnb_[st][ps] = ld1(p.bn_bits + (pixo_[st][ps] >> 3));
                if (EPI & 2) {
                    av_[st][ps] = ld8(p.addend + pixo_[st][ps]);
                    // (without mask bits the byte comes from the zero page and is ignored)
                    amb_[st][ps] = ld1(p.addend_bits != nullptr ? p.addend_bits + (pixo_[st][ps] >> 3)
                                                                : reinterpret_cast<const unsigned char*>(p.zeros));
                }
            }
        };
        // Groups requested ahead (group g lives in register set g % 3): none with three operand tensors (they do not fit
        // the register file twice), one for the tiles whose accumulators fill half the registers, two for the smaller tiles.
        constexpr int AHEAD = (EPI & 64) ? 0 : (FM * FN <= 16 ? 2 : 1);
        // loads one request issues per pass, and per group
        constexpr int LPP = ((EPI & 12) ? 1 : 0) + ((EPI & 64) ? 1 : 0) + ((EPI & 16) ? 1 : 0) + ((EPI & 8) ? 1 : 0) + ((EPI & 2) ? 2 : 0);
        constexpr int LPG = LPP * NPS;
        static_assert(2 * LPG <= 63, "vmcnt is a 6-bit counter");
#pragma unroll
        for (int g = 0; g < AHEAD && g < FM; ++g) fetch(g, g % 3);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if (!NOSTATS && p.out_nt == 3) break;  // (measurement aid: no epilogue at all, see pdt_set_out_nt_bytes; the
                                                   // statistics-free flavour is not launched in measurement mode)
            if (i + AHEAD < FM) fetch(i + AHEAD, (i + AHEAD) % 3);
            size_t(&pixo)[NPS] = pixo_[i % 3];
            bool(&rok)[NPS] = rok_[i % 3];
            bf16x8(&yv)[NPS] = yv_[i % 3];
            bf16x8(&av)[NPS] = av_[i % 3];
            bf16x8(&y2v)[NPS] = y2v_[i % 3];
            unsigned(&amb)[NPS] = amb_[i % 3];
            unsigned(&bnb)[NPS] = bnb_[i % 3];
            if constexpr (LPG > 0) {
                // The operands of this group have arrived once at most the loads of the groups requested after it are
                // outstanding (loads return in order; younger stores only make the wait stricter).
                const int later = min(AHEAD, FM - 1 - i);
                if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPG) : "memory");
                else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPG) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (EPI & 2) {
#pragma unroll
                    for (int ps = 0; ps < NPS; ++ps) amb[ps] = p.addend_bits != nullptr ? amb[ps] : 0xffu;
                }
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int nl = j * 16 + (lane >> 4) * 4;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if constexpr (BIAS_LDS) {
                    bv = *reinterpret_cast<const f32x4*>(bias_row + nl);  // (an LDS read per fragment: no registers held across the tile)
                } else if (EPI & 1) {
                    const int n = n0 + wn * TN + nl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[e] = p.bias[min(n + e, p.N - 1)];
                }
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = f32_to_bf16(acc[i][j][e] + bv[e]);
                const int er = lane & 15;
                *reinterpret_cast<bf16x4*>(epi + er * EP + ((((nl >> 3) ^ er) & (CH - 1)) << 3) + ((nl ^ (er >> 1)) & 4)) = v;
            }
#pragma unroll
            for (int ps = 0; ps < NPS; ++ps) {
                const int row = ps * RPP + erow;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(epi + row * EP + (((ech ^ row) & (CH - 1)) << 3));
                // rows 8-15 were staged with the two 8-byte halves of every 16-byte slot exchanged (see the staging layout)
                if (RPP <= 8 ? ((ps * RPP) & 8) != 0 : (row & 8) != 0) v = __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3);
                if constexpr (!NOSTATS && (EPI & (12 | 48)) == 0) {
                    // The local pivot: whatever this lane stages first after a flush, a row beyond M included (such a row
                    // holds finite values nobody uses - any finite pivot is exact, a close one only keeps the sums small).
                    // Only the first chunk of a tile can be that one: a scalar branch at one place in the unrolled code.
                    if (i == 0 && ps == 0 && f0_fresh) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f0f[e] = bf16_to_f32(v[e]);
                        f0_fresh = false;
                    }
                }
                if (rok[ps]) {
                    const size_t o = pixo[ps];
                    if (EPI & 2) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            v[e] = f32_to_bf16(bf16_to_f32(v[e]) + (((amb[ps] >> e) & 1u) ? bf16_to_f32(av[ps][e]) : 0.f));
                    }
                    if (EPI & 16) {  // act_src holds gelu' of the pre-activation, stored by the forward launch (EPI & 32)
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = f32_to_bf16(bf16_to_f32(v[e]) * bf16_to_f32(yv[ps][e]));
                    }
                    if (EPI & 32) {
                        // gelu(result) goes to out2, gelu'(result) takes the result's place in Out: the backward pass
                        // needs the derivative, never the pre-activation itself
                        bf16x8 gv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float g, d;
                            pdt_gelu_both(bf16_to_f32(v[e]), g, d);
                            gv[e] = f32_to_bf16(g);
                            v[e] = f32_to_bf16(d);
                        }
                        *reinterpret_cast<bf16x8*>(p.out2 + o) = gv;
                    }
                    if (!NOSTATS && p.out_nt == 2) {
                        // (measurement aid: no output store at all, see pdt_set_out_nt_bytes)
                    } else if (p.out_nt) {
                        // the empty statements keep the optimizer from merging the two branches' stores into one plain
                        // store (which loses the cache policy)
                        asm volatile("" ::: "memory");
                        __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p.Out + o));
                        asm volatile("" ::: "memory");
                    } else {
                        *reinterpret_cast<bf16x8*>(p.Out + o) = v;
                    }
                    if (EPI & 48) {
                        // linear layers: no channel statistics
                    } else if (EPI & 12) {
                        // BatchNorm-backward sums of the gradient just written (its bf16 value, as a separate pass
                        // over the stored tensor would see it)
                        // raw moments; mean and rstd are applied when the sums are flushed (see flush_stats)
                        unsigned mb = 0xffu;
                        if (EPI & 8) {
                            mb = bnb[ps];  // requested with the other operands: a load here would be waited for on the spot
                        } else {
                            mb = 0;  // ReLU mask recomputed from the BatchNorm input (scale / shift loaded once per tile)
#pragma unroll
                            for (int e = 0; e < 8; ++e) mb |= ((bf16_to_f32(yv[ps][e]) * msc[e] + msh[e]) > 0.f ? 1u : 0u) << e;
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float dz = ((mb >> e) & 1u) ? bf16_to_f32(v[e]) : 0.f;
                            s1[e] += dz;
                            s2[e] += dz * bf16_to_f32(yv[ps][e]);
                            if (EPI & 64) s3[e] += dz * bf16_to_f32(y2v[ps][e]);
                        }
                    } else if (!NOSTATS && p.stats != nullptr) {
                        fcnt += 1.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = bf16_to_f32(v[e]) - f0f[e];
                            s1[e] += f;
                            s2[e] += f * f;
                        }
                    }
                }
            }
        }
        if constexpr (PP) {
            if (wm == 1) asm volatile("s_barrier" ::: "memory");  // row 1 is one barrier behind row 0 again
        }
    }
    flush_stats();
    if constexpr (PP) {
        if (wm == 0) asm volatile("s_barrier" ::: "memory");  // pairs with the last barrier of wave row 1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the loader's trailing stages
}

// CUs the persistent grids are sized for.  PDT_RESERVE_CUS=k leaves k of them out: the 256-register ring kernels fill
// a CU's register file, so a CU on which a collective's workgroup is resident cannot take its ring workgroup until that
// one leaves, and with one workgroup per CU and tiles dealt out statically the launch then runs a second round; a
// multi-GPU job that sees this can trade k CUs of every launch for it.
static int pdt_num_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        const char* e = getenv("PDT_RESERVE_CUS");
        const int k = e ? atoi(e) : 0;
        cus = k > 0 && k < n ? n - k : n;
    }
    return cus;
}

template <int BM, int BN, int WM, int WN, int NS, int EPI, int BK, bool PLAIN, bool PP, bool XF = false>
static int launch_gather_gemm_ring_plain(const GatherGemmArgs& a, hipStream_t stream);

template <int BM, int BN, int WM, int WN, int NS, int EPI, int BK, bool PP = false>
static int launch_gather_gemm_ring_epi(const GatherGemmArgs& a, hipStream_t stream) {
    // one tap at (0, 0), in bounds for every output pixel: the pointer-walking loader applies
    const bool plain = a.T == 1 && a.tap_dh[0] == 0 && a.tap_dw[0] == 0 && (a.OA - 1) * a.istr < a.Hin && (a.OB - 1) * a.istr < a.Win;
    if (plain) return launch_gather_gemm_ring_plain<BM, BN, WM, WN, NS, EPI, BK, true, PP>(a, stream);
    return launch_gather_gemm_ring_plain<BM, BN, WM, WN, NS, EPI, BK, false, PP>(a, stream);
}

template <int BM, int BN, int WM, int WN, int NS, int EPI, int BK, bool PLAIN, bool PP, bool XF>
static int launch_gather_gemm_ring_plain(const GatherGemmArgs& a, hipStream_t stream) {
    if ((a.C % BK) != 0) return -1;
    // (+ one row of bias values per wave on the ping-pong schedule, see the kernel)
    constexpr size_t lds_base = (static_cast<size_t>(NS) * (BM + BN) * BK + static_cast<size_t>(WM * WN) * 16 * (BN / WN)) * sizeof(bf16) +
                                ((PP && (EPI & 1)) ? static_cast<size_t>(WM * WN) * (BN / WN) * sizeof(float) : 0);
    static_assert(lds_base <= 160 * 1024, "ring does not fit the LDS of a CU");
    const size_t lds = lds_base + (XF ? static_cast<size_t>(a.C) * 2 * sizeof(float) : 0);  // + the scale / shift table
    if (lds > 160 * 1024) return -1;
    auto kernel = gather_gemm_ring_kernel<BM, BN, WM, WN, NS, EPI, BK, PLAIN, PP, XF>;
    static bool configured = false;
    if (!configured) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(XF ? 160 * 1024 : lds_base));
        if (e != hipSuccess) return static_cast<int>(e);
        configured = true;
    }
    const int per_cu = static_cast<int>((160 * 1024) / lds);
    const long long tiles = static_cast<long long>(pdt_cdiv(a.M, BM)) * pdt_cdiv(a.N, BN);
    // PDT_GRID_ROUNDS=k: k workgroups per CU slot, each with 1/k of the tiles.  The slots still hold one workgroup at a
    // time, so the hardware's dispatcher hands the later workgroups to whichever CUs come free first - a CU held by a
    // collective's workgroup loses 1/k of a share instead of a whole one, at the price of k - 1 more pipeline fills per CU.
    static const int rounds = [] { const char* e = getenv("PDT_GRID_ROUNDS"); const int k = e ? atoi(e) : 1; return k >= 1 && k <= 16 ? k : 1; }();
    const long long cap = static_cast<long long>(pdt_num_cus()) * per_cu * rounds;
    const int grid = static_cast<int>(tiles < cap ? tiles : cap);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(WM * WN * 64), lds, stream, a);
    return pdt_launch_status();
}

template <int BM, int BN, int WM, int WN, int NS, int BK = GG_BK, bool PP = false>
static int launch_gather_gemm_ring(const GatherGemmArgs& a, hipStream_t stream) {
    if (a.zeros == nullptr) return -1;
    if (a.xf != nullptr) {
        // operand transform: plain epilogue (+ statistics), single tap in bounds, two-stage tile, not ping-pong
        const bool plain = a.T == 1 && a.tap_dh[0] == 0 && a.tap_dw[0] == 0 && (a.OA - 1) * a.istr < a.Hin && (a.OB - 1) * a.istr < a.Win;
        if constexpr (NS == 2 && !PP && BK == 64) {
            if (!plain || a.act != 0 || a.bn_y != nullptr || a.bias != nullptr || a.addend != nullptr) return -1;
            return launch_gather_gemm_ring_plain<BM, BN, WM, WN, NS, 0, BK, true, false, true>(a, stream);
        } else {
            return -1;
        }
    }
    if (a.act != 0) {
        if (a.bn_y != nullptr || a.addend != nullptr || a.stats != nullptr) return -1;
        if (a.act == 1 && a.bias != nullptr) return launch_gather_gemm_ring_epi<BM, BN, WM, WN, NS, 33, BK, PP>(a, stream);
        if (a.act == 2 && a.bias == nullptr) return launch_gather_gemm_ring_epi<BM, BN, WM, WN, NS, 16, BK, PP>(a, stream);
        return -1;
    }
    if (a.bn_y != nullptr) {
        // BatchNorm-backward sums: after a shortcut addend with the mask bits of a block tail, or alone with the
        // mask recomputed (plain BatchNorm + ReLU between two convolutions)
        if (a.bias != nullptr || a.stats == nullptr || a.bn_consts == nullptr || a.stat_rows != 3) return -1;
        if (a.bn_y2 != nullptr) {
            if (a.addend == nullptr || a.bn_bits == nullptr || a.bn_consts2 == nullptr) return -1;
            return launch_gather_gemm_ring_epi<BM, BN, WM, WN, NS, 74, BK, PP>(a, stream);
        }
        if (a.addend != nullptr && a.bn_bits != nullptr) return launch_gather_gemm_ring_epi<BM, BN, WM, WN, NS, 10, BK, PP>(a, stream);
        if (a.addend == nullptr && a.bn_bits == nullptr) return launch_gather_gemm_ring_epi<BM, BN, WM, WN, NS, 4, BK, PP>(a, stream);
        return -1;
    }
    if constexpr (PP) {
        // linear layers: no statistics accumulators in the kernel (EPI bit 128), single-tap form only
        const bool plain = a.T == 1 && a.tap_dh[0] == 0 && a.tap_dw[0] == 0 && (a.OA - 1) * a.istr < a.Hin && (a.OB - 1) * a.istr < a.Win;
        static const bool nostats = std::getenv("PDT_GEMM_NOSTATS") == nullptr || std::atoi(std::getenv("PDT_GEMM_NOSTATS")) != 0;  // (A/B switch)
        const bool dense = a.ostr == 1 && a.oph == 0 && a.opw == 0 && a.OA == a.Hout && a.OB == a.Wout;
        if (a.stats == nullptr && plain && dense && nostats && a.out_nt <= 1) {
            if (a.bias != nullptr && a.addend != nullptr) return launch_gather_gemm_ring_plain<BM, BN, WM, WN, NS, 128 | 3, BK, true, PP>(a, stream);
            if (a.bias != nullptr) return launch_gather_gemm_ring_plain<BM, BN, WM, WN, NS, 128 | 1, BK, true, PP>(a, stream);
            if (a.addend == nullptr) return launch_gather_gemm_ring_plain<BM, BN, WM, WN, NS, 128, BK, true, PP>(a, stream);
        }
    }
    if (a.bias != nullptr && a.addend != nullptr) return launch_gather_gemm_ring_epi<BM, BN, WM, WN, NS, 3, BK, PP>(a, stream);
    if (a.bias != nullptr) return launch_gather_gemm_ring_epi<BM, BN, WM, WN, NS, 1, BK, PP>(a, stream);
    if (a.addend != nullptr) return launch_gather_gemm_ring_epi<BM, BN, WM, WN, NS, 2, BK, PP>(a, stream);
    return launch_gather_gemm_ring_epi<BM, BN, WM, WN, NS, 0, BK, PP>(a, stream);
}

// ------------------------------------------------------------------------------------------------
// 3x3 convolution, stride 1, padding 1, 64 input and 64 output channels (forward and data gradient of the first
// ResNet stage): the nine taps share one copy of the input rows in LDS.
//
// The tap-table kernels above fetch the [pixels][64] operand of every tap separately - nine trips of the same rows
// through the L2 -> LDS path per output tile, with a K-step of 64 channels that lasts less than the path's latency, so
// a two-stage ring cannot cover it (these layers ran at 0.5 PFLOP/s).  Here a workgroup owns a CONTIGUOUS range of
// output rows and keeps a window of input rows in a circular LDS buffer: output row m needs input rows m + dh W + dw,
// i.e. the rows [m0 - W - 1, m0 + 256 + W] for a tile of 256, which overlap the next tile's by all but 256 rows, so
// every input row is fetched once per workgroup.  The tap is a shift of the fragment's row address; a tap that falls
// outside the image (the shifted row then belongs to the neighbouring image row or image) reads a chunk of zeros
// instead.  The 3x3x64x64 weights (72 KB) stay in LDS for the whole launch.
//
//   LDS: [560 input rows of 128 B, circular][128 B of zeros][9 taps x 64 x 64 weights][8 waves x 16 x 64 staging]
//   tile of 256 output rows, 8 waves x (32 rows x 64 channels), 9 taps x 16 MFMAs per wave
//   refill of the window for the next tile in three portions (128, 64, 64 rows) as the taps of the current tile retire
//   the rows they were the last to need: taps are worked off in the order dh = -1, 0, +1, and a workgroup barrier
//   after each dh separates the last reads of a row from the DMA that overwrites it.
// Needs 24 <= W <= 63 (the halo W + 1 fits the 64 rows kept on either side; the refill schedule assumes that the
// dh = +1 taps read nothing below row m0 + 16).
// ------------------------------------------------------------------------------------------------
constexpr int HALO_CAP = 560;                         // rows of the circular input buffer
constexpr int HALO_Z_OFF = HALO_CAP * 128;            // 128 bytes of zeros
constexpr int HALO_B_OFF = HALO_Z_OFF + 128;          // weights: [9][64 output channels][64 input channels]
constexpr int HALO_EPI_OFF = HALO_B_OFF + 9 * 8192;   // per-wave staging rows of the epilogue
constexpr int HALO_LDS = HALO_EPI_OFF + 8 * 2048;
static_assert(HALO_LDS <= 160 * 1024, "the halo kernel's buffers fit the LDS of a CU");

// EPI 0: forward statistics of the stored values (if p.stats);  EPI 4: BatchNorm-backward sums with the ReLU mask
// recomputed from bn_y (see gather_gemm_ring_kernel)
// XF (EPI 0): the input is relu(A * scale[c] + shift[c]) - the BatchNorm + ReLU of the layer before, applied to its raw
// output on the way in.  Every lane rewrites the 16-byte chunks its own DMA requests fetched, right after the counted
// wait that retires them and in front of the barrier that publishes them: each input element is transformed once per
// workgroup (in the tap-table kernels it would be once per tap), rows outside the tensor stay zero, and a tap outside
// the image still reads the chunk of zeros.
template <int EPI, bool XF = false>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(const GatherGemmArgs p, const int tiles_per_wg) {
    static_assert(!XF || EPI == 0, "operand transform: forward launches");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr uint32_t OUT_OF_RANGE = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int total_tiles = (p.M + 255) >> 8;
    const int tile_begin = blockIdx.x * tiles_per_wg;
    const int ntiles = min(tiles_per_wg, total_tiles - tile_begin);
    const int m_begin = tile_begin << 8;
    const int W = p.Win, H = p.Hin;

    // ---- loader: 1 KB pieces (8 rows of 128 bytes); lane -> (row, physical chunk), fetching the chunk the swizzle puts there
    const int lrow = lane >> 3;
    const int kchunk = ((lane & 7) ^ lrow) << 3;
    {   // weights, in the order the taps are worked off: tap k = (k / 3 - 1, k % 3 - 1); wave w fetches output channels 8w ..
        bf16* bbase = const_cast<bf16*>(p.B);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            int woff = 0;
            for (int t = 0; t < 9; ++t)
                if (p.tap_dh[t] == k / 3 - 1 && p.tap_dw[t] == k % 3 - 1) woff = p.tap_woff[t];
            const uint32_t voff = static_cast<uint32_t>((8 * wave + lrow) * p.ldb + woff + kchunk) * 2u;
            buffer_load_lds_16(bbase, (lds_void_t*)(smem + HALO_B_OFF + k * 8192 + wave * 1024), voff, 0);
        }
    }
    // input rows are addressed relative to row m_begin - 64, which sits at physical row 0
    bf16* abase = const_cast<bf16*>(p.A) + (static_cast<ptrdiff_t>(m_begin) - 64) * p.cs;
    // (returns the LDS byte of this lane's chunk, or -1 if the row is outside the tensor: what the transform needs later)
    auto issue_rows = [&](int rel0, int phys0) -> int {  // 8 rows from relative row rel0 (a multiple of 8) to physical row phys0
        const int rel = rel0 + lrow;
        const int m = m_begin - 64 + rel;
        const bool ok = m >= 0 && m < p.M;
        const uint32_t voff = ok ? static_cast<uint32_t>(rel * p.cs + kchunk) * 2u : OUT_OF_RANGE;
        buffer_load_lds_16(abase, (lds_void_t*)(smem + phys0 * 128), voff, 0);
        return ok ? phys0 * 128 + lane * 16 : -1;
    };
    float xsc[8], xsh[8];  // XF: scale / shift of the 8 channels of this lane's chunk (the same logical chunk in every piece)
    if constexpr (XF) {
        *reinterpret_cast<f32x4*>(xsc) = *reinterpret_cast<const f32x4*>(p.xf + 2 * 64 + kchunk);
        *reinterpret_cast<f32x4*>(xsc + 4) = *reinterpret_cast<const f32x4*>(p.xf + 2 * 64 + kchunk + 4);
        *reinterpret_cast<f32x4*>(xsh) = *reinterpret_cast<const f32x4*>(p.xf + 3 * 64 + kchunk);
        *reinterpret_cast<f32x4*>(xsh + 4) = *reinterpret_cast<const f32x4*>(p.xf + 3 * 64 + kchunk + 4);
    }
    auto xform = [&](int at) {
        if (at < 0) return;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + at);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f32_to_bf16(fmaxf(bf16_to_f32(v[e]) * xsc[e] + xsh[e], 0.f));
        *reinterpret_cast<bf16x8*>(smem + at) = v;
    };
    auto xform3 = [&](int a0, int a1, int a2) {  // three chunks: all reads first, so that the LDS round trips overlap
        const int at[3] = {a0, a1, a2};
        bf16x8 v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (at[k] >= 0) v[k] = *reinterpret_cast<const bf16x8*>(smem + at[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (at[k] < 0) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[k][e] = f32_to_bf16(fmaxf(bf16_to_f32(v[k][e]) * xsc[e] + xsh[e], 0.f));
            *reinterpret_cast<bf16x8*>(smem + at[k]) = v[k];
        }
    };
    int pend_a[2] = {-1, -1}, pend_b = -1, pend_c = -1;  // XF: chunks requested and not yet transformed
    int ld_rel = 0, ld_phys = 0;  // the next row to fetch (relative / physical); both multiples of 8
    auto refill = [&](int pieces_per_wave, int* at) {  // 8 waves x pieces x 8 rows
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k >= pieces_per_wave) continue;
            const int q = (wave + 8 * k) * 8;
            int ph = ld_phys + q;
            ph = ph >= HALO_CAP ? ph - HALO_CAP : ph;
            at[k] = issue_rows(ld_rel + q, ph);
        }
        ld_rel += pieces_per_wave * 64;
        ld_phys += pieces_per_wave * 64;
        ld_phys = ld_phys >= HALO_CAP ? ld_phys - HALO_CAP : ld_phys;
    };
    int first_at[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) first_at[k] = issue_rows((wave + 8 * k) * 8, (wave + 8 * k) * 8);  // rows [m_begin - 64, m_begin + 320)
    ld_rel = ld_phys = 384;
    if (tid < 32) *reinterpret_cast<uint32_t*>(smem + HALO_Z_OFF + 4 * tid) = 0u;
    if constexpr (XF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 6; ++k) xform(first_at[k]);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- fragment addressing
    // weights: output channel 16 j + lr, chunk (2 kk' ...): byte offset of the kk = 0 fragment of j = 0, tap 0
    const uint32_t b_off0 = HALO_B_OFF + lr * 128 + ((lg ^ (lr & 7)) << 4);
    // ---- epilogue state
    const int ech = lane & 7, erow = lane >> 3;
    bf16* epi = reinterpret_cast<bf16*>(smem + HALO_EPI_OFF + wave * 2048);
    float s1[8], s2[8], f0f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = f0f[e] = 0.f;
    float fcnt = 0.f;
    bool f0_fresh = true;
    float msc[8], msh[8];
    if constexpr (EPI == 4) {
        *reinterpret_cast<f32x4*>(msc) = *reinterpret_cast<const f32x4*>(p.bn_consts + 2 * 64 + ech * 8);
        *reinterpret_cast<f32x4*>(msc + 4) = *reinterpret_cast<const f32x4*>(p.bn_consts + 2 * 64 + ech * 8 + 4);
        *reinterpret_cast<f32x4*>(msh) = *reinterpret_cast<const f32x4*>(p.bn_consts + 3 * 64 + ech * 8);
        *reinterpret_cast<f32x4*>(msh + 4) = *reinterpret_cast<const f32x4*>(p.bn_consts + 3 * 64 + ech * 8 + 4);
    }
    const bool want_stats = EPI == 4 || p.stats != nullptr;

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[2][4];
    int base_phys = 0;  // physical row of input row m0 - 64
    for (int ti = 0; ti < ntiles; ++ti) {
        const int m0 = m_begin + (ti << 8);
        const bool more = ti + 1 < ntiles;
        // this lane's two fragment rows: which taps stay inside the image, and the row's place in the buffer
        bool hm[2], hp[2], wm_[2], wp[2];
        int rowp[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t r = static_cast<uint32_t>(m0 + 32 * wave + 16 * i + lr);
            const uint32_t img = fastdiv(r, p.div_ab);
            const uint32_t rem = r - img * (p.OA * p.OB);
            const uint32_t h = fastdiv(rem, p.div_b);
            const uint32_t w = rem - h * p.OB;
            hm[i] = h > 0;
            hp[i] = static_cast<int>(h) < H - 1;
            wm_[i] = w > 0;
            wp[i] = static_cast<int>(w) < W - 1;
            rowp[i] = base_phys + 64 + 32 * wave + 16 * i + lr;
        }
        // Fragments of tap k + 1 are requested before the MFMAs of tap k are issued (two register sets), so a tap costs
        // one counted wait on the LDS queue instead of one per fragment.  Tap 3 is requested in front of the first
        // barrier: its rows (up to m0 + 254) are in place since the tile began, and the barriers only keep the DMA that
        // OVERWRITES rows behind the last wave that reads them.
        bf16x8 fa[2][2][2], fb[2][2][4];
        auto load = [&](auto kc, auto bc) {
            constexpr int k = decltype(kc)::value, buf = decltype(bc)::value;
            constexpr int dh = k / 3 - 1, dw = k % 3 - 1;
            const int delta = dh * W + dw;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool ok = (dh < 0 ? hm[i] : (dh > 0 ? hp[i] : true)) && (dw < 0 ? wm_[i] : (dw > 0 ? wp[i] : true));
                int ph = rowp[i] + delta;
                ph = ph >= HALO_CAP ? ph - HALO_CAP : ph;
                uint32_t a0 = static_cast<uint32_t>(ph * 128 + ((lg ^ (ph & 7)) << 4));
                a0 = ok ? a0 : static_cast<uint32_t>(HALO_Z_OFF);
                fa[buf][0][i] = *reinterpret_cast<const bf16x8*>(smem + a0);
                fa[buf][1][i] = *reinterpret_cast<const bf16x8*>(smem + (a0 ^ 64u));  // channels 32 .. 63: chunk index + 4
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                fb[buf][0][j] = *reinterpret_cast<const bf16x8*>(smem + b_off0 + k * 8192 + j * 2048);
                fb[buf][1][j] = *reinterpret_cast<const bf16x8*>(smem + (b_off0 ^ 64u) + k * 8192 + j * 2048);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mma = [&](auto kc, auto bc) {
            constexpr int k = decltype(kc)::value, buf = decltype(bc)::value;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[buf][kk][j], fa[buf][kk][i],
                                                                            (k == 0 && kk == 0) ? zero4 : acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        using std::integral_constant;
        constexpr integral_constant<int, 0> B0{};
        constexpr integral_constant<int, 1> B1{};
        load(integral_constant<int, 0>{}, B0);
        load(integral_constant<int, 1>{}, B1);
        mma(integral_constant<int, 0>{}, B0);
        if (more) refill(2, pend_a);  // (behind the first MFMAs; nobody reads the rows it overwrites since the tile began)
        load(integral_constant<int, 2>{}, B0);
        mma(integral_constant<int, 1>{}, B1);
        load(integral_constant<int, 3>{}, B1);
        mma(integral_constant<int, 2>{}, B0);
        // The last portion of the previous tile's refill (rows that the taps from number 5 on read) has landed: what this
        // wave has issued since are the previous tile's 4 stores and, if there is a next tile, this tile's first portion
        // (2 requests).  (Only the launch's very last tile can store fewer rows, and nothing is refilled after it.)
        if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if constexpr (XF) {
            __builtin_amdgcn_sched_barrier(0);
            xform(pend_c);
            pend_c = -1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // and nobody reads rows below m0 - 1 any more
        // BatchNorm-backward sums: this lane's pre-normalisation activations of the tile, requested six taps before
        // their use (rows beyond M read the last row)
        u32x4 yraw[2][2];
        if constexpr (EPI == 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int m = min(m0 + 32 * wave + 16 * i + ps * 8 + erow, p.M - 1);
                    const bf16* q = p.bn_y + static_cast<size_t>(static_cast<uint32_t>(m)) * static_cast<uint32_t>(p.ldo) + ech * 8;
                    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(yraw[i][ps]) : "v"(q) : "memory");
                }
        }
        if (more) refill(1, &pend_b);
        load(integral_constant<int, 4>{}, B0);
        mma(integral_constant<int, 3>{}, B1);
        load(integral_constant<int, 5>{}, B1);
        mma(integral_constant<int, 4>{}, B0);
        load(integral_constant<int, 6>{}, B0);
        mma(integral_constant<int, 5>{}, B1);
        asm volatile("s_barrier" ::: "memory");  // nobody reads rows below m0 + W - 1 any more
        if (more) refill(1, &pend_c);
        load(integral_constant<int, 7>{}, B1);
        mma(integral_constant<int, 6>{}, B0);
        load(integral_constant<int, 8>{}, B0);
        mma(integral_constant<int, 7>{}, B1);
        mma(integral_constant<int, 8>{}, B0);
        // All but the last portion of the next tile's rows (and bn_y) have landed - waited for here, in front of this
        // tile's stores, so that the wait never covers a store.  The last portion (one request per wave, three taps old)
        // holds rows that only the taps behind the next tile's first barrier read: it is awaited there.
        if (more) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (XF) {
            if (more) xform3(pend_a[0], pend_a[1], pend_b);
        }

        // ---- epilogue: 16 rows at a time -> bf16 -> wave-private LDS rows -> 16-byte row segments (staging layout as in
        // gather_gemm_ring_kernel)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nl = j * 16 + lg * 4;
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = f32_to_bf16(acc[i][j][e]);
                *reinterpret_cast<bf16x4*>(epi + lr * 64 + ((((nl >> 3) ^ lr) & 7) << 3) + ((nl ^ (lr >> 1)) & 4)) = v;
            }
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int row = ps * 8 + erow;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(epi + row * 64 + (((ech ^ row) & 7) << 3));
                if (ps == 1) v = __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3);
                if constexpr (EPI == 0) {
                    if (i == 0 && ps == 0 && f0_fresh) {  // the lane's pivot: the first values it sees (see the ring kernel)
#pragma unroll
                        for (int e = 0; e < 8; ++e) f0f[e] = bf16_to_f32(v[e]);
                        f0_fresh = false;
                    }
                }
                const int m = m0 + 32 * wave + 16 * i + row;
                if (m < p.M) {
                    bf16* dst = p.Out + static_cast<size_t>(static_cast<uint32_t>(m)) * static_cast<uint32_t>(p.ldo) + ech * 8;
                    if (p.out_nt) {
                        asm volatile("" ::: "memory");
                        __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(dst));
                        asm volatile("" ::: "memory");
                    } else {
                        *reinterpret_cast<bf16x8*>(dst) = v;
                    }
                    if constexpr (EPI == 4) {
                        const bf16x8 yv = __builtin_bit_cast(bf16x8, yraw[i][ps]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float y = bf16_to_f32(yv[e]);
                            const float dz = (y * msc[e] + msh[e]) > 0.f ? bf16_to_f32(v[e]) : 0.f;
                            s1[e] += dz;
                            s2[e] += dz * y;
                        }
                    } else if (want_stats) {
                        fcnt += 1.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = bf16_to_f32(v[e]) - f0f[e];
                            s1[e] += f;
                            s2[e] += f * f;
                        }
                    }
                }
            }
        }
        base_phys += 256;
        base_phys = base_phys >= HALO_CAP ? base_phys - HALO_CAP : base_phys;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every wave is done with this tile's rows, and everybody's refill has landed
    }

    // ---- the workgroup's sums -> one of the launch's accumulator slots
    if (!want_stats) return;
    const int ncol = ech * 8;
    if constexpr (EPI == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float P = p.pivot != nullptr ? p.pivot[ncol + e] : 0.f;
            const float dl = f0f[e] - P;
            s2[e] += dl * (2.f * s1[e] + fcnt * dl);
            s1[e] += fcnt * dl;
            if (p.pivot != nullptr && lane < 8) p.stats[static_cast<size_t>(PDT_STAT_SLOTS) * 2 * 64 + ncol + e] = P;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            s1[e] += __shfl_xor(s1[e], o, 64);
            s2[e] += __shfl_xor(s2[e], o, 64);
        }
    }
    if (lane < 8) {
        float* slot = p.stats + static_cast<size_t>(blockIdx.x % PDT_STAT_SLOTS) * p.stat_rows * 64;
        if constexpr (EPI == 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s2[e] = (s2[e] - p.bn_consts[ncol + e] * s1[e]) * p.bn_consts[64 + ncol + e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __hip_atomic_fetch_add(slot + ncol + e, s1[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(slot + 64 + ncol + e, s2[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the launch is a 3x3 / stride 1 / padding 1 convolution of 64 -> 64 channels in the form conv3x3_halo_kernel handles
static bool conv3x3_halo_applies(const GatherGemmArgs& a) {
    static const bool enabled = std::getenv("PDT_CONV3X3_HALO") == nullptr || std::atoi(std::getenv("PDT_CONV3X3_HALO")) != 0;  // (A/B switch)
    if (!enabled || a.tw != 0 || a.T != 9 || a.C != 64 || a.N != 64 || a.cs != 64 || a.ldo != 64) return false;
    if (a.istr != 1 || a.ostr != 1 || a.oph != 0 || a.opw != 0) return false;
    if (a.OA != a.Hin || a.OB != a.Win || a.Hout != a.Hin || a.Wout != a.Win || a.Win < 24 || a.Win > 63) return false;
    if (a.bias != nullptr || a.addend != nullptr || a.act != 0 || a.out_nt > 1 || (a.xf != nullptr && a.bn_y != nullptr)) return false;
    if (a.bn_y != nullptr && (a.bn_bits != nullptr || a.bn_y2 != nullptr || a.stats == nullptr || a.bn_consts == nullptr || a.stat_rows != 3))
        return false;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        const int dh = a.tap_dh[t], dw = a.tap_dw[t];
        if (dh < -1 || dh > 1 || dw < -1 || dw > 1 || a.tap_woff[t] < 0 || a.tap_woff[t] + 64 > a.ldb) return false;
        seen |= 1u << ((dh + 1) * 3 + dw + 1);
    }
    return seen == 0x1ffu;
}

static int launch_conv3x3_halo(const GatherGemmArgs& a, hipStream_t stream) {
    const int total = pdt_cdiv(a.M, 256);
    const int cus = pdt_num_cus();
    const int per = pdt_cdiv(total, cus);
    const int grid = pdt_cdiv(total, per);
    // 32-bit byte offsets from the workgroup's first row
    if ((static_cast<long long>(per) * 256 + 1024) * 128 >= (1ll << 31)) return -1;
    auto kernel = a.bn_y != nullptr ? conv3x3_halo_kernel<4> : (a.xf != nullptr ? conv3x3_halo_kernel<0, true> : conv3x3_halo_kernel<0>);
    static bool configured[3] = {false, false, false};
    const int which = a.bn_y != nullptr ? 1 : (a.xf != nullptr ? 2 : 0);
    if (!configured[which]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, HALO_LDS);
        if (e != hipSuccess) return static_cast<int>(e);
        configured[which] = true;
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), HALO_LDS, stream, a, per);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// The 7x7 / stride 2 stem on 4-element pixels (packed rows, see gather_gemm_kernel), input rows shared in LDS.
//
// In packed-row form a kernel row of an output pixel is 32 contiguous elements (64 bytes) of the zero-bordered input,
// starting 16 bytes (one stride of two 4-element pixels) after its left neighbour's: the tap-table kernel fetches those
// 64 bytes per pixel and kernel row - fourteen times the bytes the input holds - through the L2 -> LDS path, which is
// what bounds it.  Here the input is streamed through LDS once, as it sits in memory: a circular 64 KB image of the
// byte range a few tiles need (rows of all images are contiguous, so the window of a tile of 256 consecutive output
// pixels is one contiguous range), and an A fragment is read straight out of it - lane (pixel, 16-byte chunk) reads
// at row byte + 16 (ow + chunk), so the 16 pixels of a fragment read 16 consecutive chunks and no swizzle is needed.
// The weights (64 x 7 x 32) stay in LDS at a row pitch of 464 bytes (a pitch of 448 would put four rows on a bank).
//   tile of 256 output pixels, 8 waves x (32 pixels x 64 channels), 7 kernel rows x 8 MFMAs per wave
//   every tile issues exactly three DMA requests per wave (real 1 KB pieces of the window two tiles ahead, or dummies
//   into a trash area), so one counted wait per tile retires the previous tile's requests
// ------------------------------------------------------------------------------------------------
constexpr int ROWS_B_OFF = 65536;                    // weights: 64 rows of 464 bytes
constexpr int ROWS_B_PITCH = 464;
constexpr int ROWS_EPI_OFF = ROWS_B_OFF + 64 * ROWS_B_PITCH;  // 95232
constexpr int ROWS_TRASH_OFF = ROWS_EPI_OFF + 8 * 2048;       // one KB per wave for the dummy requests
constexpr int ROWS_LDS = ROWS_TRASH_OFF + 8 * 1024;
constexpr int ROWS_LA = 2;    // the window of tile t + 2 is requested at the top of tile t and awaited at the end of tile t + 1
constexpr int ROWS_REQ = 3;   // requests per wave and tile: 24 KB, more than the window moves per tile (<= 20 KB across an image boundary)

__global__ __launch_bounds__(512) void conv_rows7_kernel(const GatherGemmArgs p, const int tiles_per_wg, const long long in_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr uint32_t OUT_OF_RANGE = 0x80000000u;
    constexpr int T = 7;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int total_tiles = (p.M + 255) >> 8;
    const int tile_begin = blockIdx.x * tiles_per_wg;
    const int ntiles = min(tiles_per_wg, total_tiles - tile_begin);
    const int m_begin = tile_begin << 8;
    const uint32_t RB = static_cast<uint32_t>(p.Win) * p.cs * 2u;  // bytes per input row (a multiple of 16)

    // byte offset (from the start of the input) of kernel row 0 of output pixel m, at its first pixel
    auto row_byte = [&](int m, uint32_t& ow) -> long long {
        const uint32_t mm = static_cast<uint32_t>(min(m, p.M - 1));
        const uint32_t img = fastdiv(mm, p.div_ab);
        const uint32_t rem = mm - img * (p.OA * p.OB);
        const uint32_t oh = fastdiv(rem, p.div_b);
        ow = rem - oh * p.OB;
        return static_cast<long long>(img * p.Hin + oh * p.istr) * RB;
    };
    // window of tile t (clamped to the launch): [first byte of its first pixel's first kernel row, last byte of its last pixel's last)
    uint32_t dummy;
    const long long org = row_byte(m_begin, dummy) & ~1023ll;  // LDS position 0 holds input byte `org`
    auto need_end = [&](int t) -> uint32_t {  // relative to org
        const int last = min(m_begin + (min(t, ntiles - 1) << 8) + 255, p.M - 1);
        return static_cast<uint32_t>(row_byte(last, dummy) + static_cast<long long>(T) * RB - org);
    };
    bf16* abase = const_cast<bf16*>(p.A) + org / 2;
    // The window is requested in groups of 8 pieces (1 KB per wave).  Every call issues exactly ROWS_REQ requests per
    // wave: the next groups that start below the end of tile t's window, and dummies into a trash area once the cursor
    // has passed it - a fixed count per tile is what lets one counted wait retire "everything but this tile's requests".
    uint32_t ld_off = 0;  // first byte (from org) of the next group
    auto issue_tile = [&](int t) {
        const uint32_t end = need_end(t);
#pragma unroll
        for (int k = 0; k < ROWS_REQ; ++k) {
            const bool real = ld_off < end;  // (wave-uniform)
            const uint32_t off = ld_off + 1024u * wave + 16u * lane;
            const uint32_t voff = (real && org + off + 16 <= in_bytes) ? off : OUT_OF_RANGE;
            char* dst = real ? smem + ((ld_off + 1024u * wave) & 0xffffu) : smem + ROWS_TRASH_OFF + wave * 1024;
            buffer_load_lds_16(abase, (lds_void_t*)dst, voff, 0);
            if (real) ld_off += 8192u;
        }
    };

    // ---- weights -> LDS (plain loads: the LDS pitch differs from the memory pitch)
    for (int c = tid; c < 64 * (T * 4); c += 512) {
        const int n = c / (T * 4), q = c - n * (T * 4);
        *reinterpret_cast<bf16x8*>(smem + ROWS_B_OFF + n * ROWS_B_PITCH + q * 16) =
            *reinterpret_cast<const bf16x8*>(p.B + static_cast<size_t>(n) * p.ldb + q * 8);
    }
    // the windows of the first ROWS_LA tiles (at most 56 KB, see the launcher)
    for (int rep = 0; rep < (7 + ROWS_REQ - 1) / ROWS_REQ; ++rep) issue_tile(ROWS_LA - 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    const uint32_t b_off0 = ROWS_B_OFF + lr * ROWS_B_PITCH + lg * 16;
    const int ech = lane & 7, erow = lane >> 3;
    bf16* epi = reinterpret_cast<bf16*>(smem + ROWS_EPI_OFF + wave * 2048);
    float s1[8], s2[8], f0f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = f0f[e] = 0.f;
    float fcnt = 0.f;
    bool f0_fresh = true;
    const bool want_stats = p.stats != nullptr;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[2][4];

    for (int ti = 0; ti < ntiles; ++ti) {
        const int m0 = m_begin + (ti << 8);
        uint32_t a_byte[2];  // LDS-relative byte of this lane's chunk of kernel row 0, per 16-pixel block (before the wrap)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint32_t ow;
            const long long rb = row_byte(m0 + 32 * wave + 16 * i + lr, ow);
            a_byte[i] = static_cast<uint32_t>(rb - org) + 16u * (ow + lg);
        }
        bf16x8 fa[2][2], fb[2][4];
        auto load = [&](const int kh, const int buf) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[buf][i] = *reinterpret_cast<const bf16x8*>(smem + ((a_byte[i] + kh * RB) & 0xffffu));
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[buf][j] = *reinterpret_cast<const bf16x8*>(smem + b_off0 + j * (16 * ROWS_B_PITCH) + kh * 64);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mma = [&](const int kh, const int buf) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[buf][j], fa[buf][i], kh == 0 ? zero4 : acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        load(0, 0);
#pragma unroll
        for (int kh = 0; kh < T; ++kh) {
            if (kh + 1 < T) load(kh + 1, (kh + 1) & 1);
            mma(kh, kh & 1);
            if (kh == 0) {  // this tile's requests, behind its first MFMAs (their address arithmetic runs under the matrix pipe)
                issue_tile(ti + ROWS_LA);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the previous tile's requests have landed (this tile's may stay in flight); in front of the stores
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ROWS_REQ) : "memory");
        __builtin_amdgcn_sched_barrier(0);

        // ---- epilogue (see conv3x3_halo_kernel)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nl = j * 16 + lg * 4;
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = f32_to_bf16(acc[i][j][e]);
                *reinterpret_cast<bf16x4*>(epi + lr * 64 + ((((nl >> 3) ^ lr) & 7) << 3) + ((nl ^ (lr >> 1)) & 4)) = v;
            }
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int row = ps * 8 + erow;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(epi + row * 64 + (((ech ^ row) & 7) << 3));
                if (ps == 1) v = __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3);
                if (i == 0 && ps == 0 && f0_fresh) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f0f[e] = bf16_to_f32(v[e]);
                    f0_fresh = false;
                }
                const int m = m0 + 32 * wave + 16 * i + row;
                if (m < p.M) {
                    bf16* dst = p.Out + static_cast<size_t>(static_cast<uint32_t>(m)) * static_cast<uint32_t>(p.ldo) + ech * 8;
                    if (p.out_nt) {
                        asm volatile("" ::: "memory");
                        __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(dst));
                        asm volatile("" ::: "memory");
                    } else {
                        *reinterpret_cast<bf16x8*>(dst) = v;
                    }
                    if (want_stats) {
                        fcnt += 1.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = bf16_to_f32(v[e]) - f0f[e];
                            s1[e] += f;
                            s2[e] += f * f;
                        }
                    }
                }
            }
        }
        asm volatile("s_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!want_stats) return;
    const int ncol = ech * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float P = p.pivot != nullptr ? p.pivot[ncol + e] : 0.f;
        const float dl = f0f[e] - P;
        s2[e] += dl * (2.f * s1[e] + fcnt * dl);
        s1[e] += fcnt * dl;
        if (p.pivot != nullptr && lane < 8) p.stats[static_cast<size_t>(PDT_STAT_SLOTS) * 2 * 64 + ncol + e] = P;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            s1[e] += __shfl_xor(s1[e], o, 64);
            s2[e] += __shfl_xor(s2[e], o, 64);
        }
    }
    if (lane < 8) {
        float* slot = p.stats + static_cast<size_t>(blockIdx.x % PDT_STAT_SLOTS) * p.stat_rows * 64;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __hip_atomic_fetch_add(slot + ncol + e, s1[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(slot + 64 + ncol + e, s2[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the packed-row launch (before its conversion into K-steps) is the form conv_rows7_kernel handles
static bool conv_rows7_applies(const GatherGemmArgs& a, int smallc, int T) {
    static const bool enabled = std::getenv("PDT_CONV_ROWS") == nullptr || std::atoi(std::getenv("PDT_CONV_ROWS")) != 0;  // (A/B switch)
    if (!enabled || smallc != 32 || T != 7 || a.C != 4 || a.N != 64 || a.ldo != 64 || a.istr != 2 || a.ldb < 7 * 32) return false;
    if (a.ostr != 1 || a.oph != 0 || a.opw != 0 || a.OA != a.Hout || a.OB != a.Wout) return false;
    if (a.bias != nullptr || a.addend != nullptr || a.act != 0 || a.xf != nullptr || a.bn_y != nullptr || a.out_nt > 1) return false;
    if ((a.Win * 4 * 2) % 16 != 0) return false;
    // the kernel rows of an output pixel are inside the input: (OA - 1) * 2 + 7 <= Hin, and 64 bytes from pixel 2 (OB - 1) fit the row
    if ((a.OA - 1) * 2 + 7 > a.Hin || ((a.OB - 1) * 2) * 4 + 32 > a.Win * 4) return false;
    // window of three tiles + what a tile can move + one group of over-fetch must fit the 64 KB image (see the kernel)
    const long long RB = static_cast<long long>(a.Win) * 8;
    const long long steps3 = (3 * 256 - 1 + a.OB - 1) / a.OB;   // output-row steps inside three tiles
    const long long rows = steps3 * 2 + (a.Hin - 2 * (a.OA - 1)) + 7;  // + the jump at an image boundary + the kernel rows
    if (rows * RB + 1024 + 8192 > 65536) return false;
    const long long steps1 = (256 - 1 + a.OB - 1) / a.OB;
    if ((steps1 * 2 + (a.Hin - 2 * (a.OA - 1))) * RB > ROWS_REQ * 8192) return false;  // a tile's requests keep up with the window
    return true;
}

static int launch_conv_rows7(const GatherGemmArgs& a, hipStream_t stream) {
    const int total = pdt_cdiv(a.M, 256);
    const int cus = pdt_num_cus();
    const int per = pdt_cdiv(total, cus);
    const int grid = pdt_cdiv(total, per);
    const long long images = a.M / (static_cast<long long>(a.OA) * a.OB);
    const long long in_bytes = (images * a.Hin + 1) * a.Win * 8;  // (+ the spare row the caller keeps behind the last image)
    if ((static_cast<long long>(per) * 256 / a.OB + 8) * 2 * a.Win * 8 >= (1ll << 31)) return -1;
    static bool configured = false;
    if (!configured) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rows7_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS_LDS);
        if (e != hipSuccess) return static_cast<int>(e);
        configured = true;
    }
    hipLaunchKernelGGL(conv_rows7_kernel, dim3(grid), dim3(512), ROWS_LDS, stream, a, per, in_bytes);
    return pdt_launch_status();
}

// Outputs of at least this many bytes are written with non-temporal stores by the ring kernels.  A negative value is a
// measurement aid: the ring kernels then skip the output store altogether (what the epilogue's stores cost a launch).
static long long g_out_nt_bytes = 64ll << 20;
PDT_API void pdt_set_out_nt_bytes(long long bytes) { g_out_nt_bytes = bytes; }

// Host entry.  Shapes are validated here; the kernel assumes them.
//   tile: 0 = 128x128, 1 = 128x64 (narrow N), 2 = 256x128... (see switch)
// Returns 0 on success, -1 on invalid arguments, otherwise a hipError_t.
PDT_API int pdt_gather_gemm(const void* A, const void* B, void* Out, float* stats, const float* bias,
                            const void* addend, const void* addend_bits, int M, int N,
                            int ldb, int ldo, int C, int T, int Hin, int Win, int istr, int OA, int OB, int Hout,
                            int Wout, int ostr, int oph, int opw, const int* tap_dh, const int* tap_dw,
                            const int* tap_woff, int tile, int smallc, const void* zeros, const void* bn_y,
                            const float* bn_consts, const void* bn_bits, const void* bn_y2, const float* bn_consts2,
                            int act, void* out2, const void* act_src, const float* xf, const float* pivot, void* stream) {
    // the operand transform lives in the 256x256 two-stage kernel and in the shared-rows 3x3 kernel (tiles 14 / 15, checked below)
    if (xf != nullptr && (smallc || (tile != 16 && tile != 14 && tile != 15))) return -1;
    if (act < 0 || act > 2 || (act == 1 && (out2 == nullptr || bias == nullptr)) || (act == 2 && act_src == nullptr)) return -1;
    if (M <= 0 || N <= 0 || T < 0 || T > GG_MAX_TAPS || C <= 0) return -1;
    if (!smallc && (C % GG_BK) != 0) return -1;
    // packed rows: `smallc` is the row width tw; pixels are C (4 or 8) elements apart; B rows cover whole K-steps
    if (smallc && (T == 0 || (smallc != 32 && smallc != 64) || (C != 4 && C != 8) || ldb < (T * smallc + GG_BK - 1) / GG_BK * GG_BK))
        return -1;
    if ((ldo % 8) != 0 || (ldb % 8) != 0 || OA <= 0 || OB <= 0) return -1;
    GatherGemmArgs a;
    a.A = static_cast<const bf16*>(A);
    a.B = static_cast<const bf16*>(B);
    a.Out = static_cast<bf16*>(Out);
    a.stats = stats;
    a.bias = bias;
    a.addend = static_cast<const bf16*>(addend);
    a.addend_bits = static_cast<const uint8_t*>(addend_bits);
    a.zeros = static_cast<const bf16*>(zeros);
    a.bn_y = static_cast<const bf16*>(bn_y);
    a.bn_consts = bn_consts;
    a.bn_bits = static_cast<const uint8_t*>(bn_bits);
    a.bn_y2 = static_cast<const bf16*>(bn_y2);
    a.bn_consts2 = bn_consts2;
    if (bn_y2 != nullptr && bn_y == nullptr) return -1;
    a.stat_rows = bn_y != nullptr ? 3 : 2;
    a.xf = xf;
    a.pivot = pivot;
    a.act = act;
    a.out_nt = g_out_nt_bytes < 0 ? (g_out_nt_bytes == -2 ? 3 : 2) : (static_cast<long long>(M) * N * 2 >= g_out_nt_bytes);
    a.out2 = static_cast<bf16*>(out2);
    a.act_src = static_cast<const bf16*>(act_src);
    if (bn_y != nullptr && (smallc || tile < 10)) return -1;  // only the ring kernels have this epilogue
    a.M = M; a.N = N; a.ldb = ldb; a.ldo = ldo; a.C = C; a.T = T;
    a.Hin = Hin; a.Win = Win; a.istr = istr;
    a.Hout = Hout; a.Wout = Wout; a.ostr = ostr; a.oph = oph; a.opw = opw;
    a.OA = OA; a.OB = OB;
    a.div_ab = make_fastdiv(static_cast<uint32_t>(OA) * OB);
    a.div_b = make_fastdiv(OB);
    a.tw = smallc;
    a.cs = C;
    for (int t = 0; t < T; ++t) {
        if (!smallc && (tap_woff[t] % 8) != 0) return -1;
        a.tap_dh[t] = tap_dh[t];
        a.tap_dw[t] = tap_dw[t];
        a.tap_woff[t] = tap_woff[t];
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (smallc && tile >= 13 && conv_rows7_applies(a, smallc, T)) return launch_conv_rows7(a, s);
    if (smallc && tile >= 13 && zeros != nullptr) {
        // packed rows on the ring kernels: one "tap" per K-step of 64 / tw kernel rows, a single 64-wide channel block.
        // The last K-step may reach one image row past the kernel (zero weights there): the caller keeps one readable
        // row behind the last image.
        const int rows_per_step = GG_BK / smallc, steps = (T * smallc + GG_BK - 1) / GG_BK;
        a.T = steps;
        a.C = GG_BK;
        for (int t = 0; t < steps; ++t) {
            a.tap_dh[t] = t * rows_per_step;
            a.tap_dw[t] = 0;
            a.tap_woff[t] = t * GG_BK;
        }
    } else if (smallc) {
        return N <= 64 ? launch_gather_gemm<128, 64, 4, 1, true>(a, s) : launch_gather_gemm<128, 128, 2, 2, true>(a, s);
    }
    if ((tile == 14 || tile == 15) && conv3x3_halo_applies(a)) return launch_conv3x3_halo(a, s);
    if (xf != nullptr && tile != 16) return -1;
    switch (tile) {
        case 0: return launch_gather_gemm<128, 128, 2, 2>(a, s);
        case 1: return launch_gather_gemm<128, 64, 4, 1>(a, s);
        case 2: return launch_gather_gemm<64, 64, 2, 2>(a, s);
        case 3: return launch_gather_gemm<128, 128, 4, 2>(a, s);  // 8 waves, 32x64 per wave
        case 4: return launch_gather_gemm<128, 128, 2, 4>(a, s);  // 8 waves, 64x32 per wave
        case 5: return launch_gather_gemm<128, 64, 4, 2>(a, s);   // 8 waves, 32x32 per wave
        case 6: return launch_gather_gemm<256, 128, 4, 2>(a, s);  // 8 waves, 64x64 per wave
        // ring variants (persistent, LDS-DMA): need the page of zeros
        case 13: return launch_gather_gemm_ring<256, 128, 4, 2, 3>(a, s);  // 8 waves, 256-row tile
        case 14: return launch_gather_gemm_ring<256, 64, 8, 1, 3>(a, s);   // narrow N, 32x64 per wave
        case 15: return launch_gather_gemm_ring<512, 64, 8, 1, 2>(a, s);   // narrow N, 64x64 per wave: half the weight-fragment reads per FLOP
        case 16: return launch_gather_gemm_ring<256, 256, 2, 4, 2, 64>(a, s);  // 128x64 per wave, K-step 64, 2 stages
        case 17: return launch_gather_gemm_ring<256, 256, 2, 4, 2, 64, true>(a, s);  // the same tile on the ping-pong schedule
        default: return -1;
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient:  dW[n][woff_t + c] += sum_m dY[m][n] * A(m, t, c)
//
// The reduction runs over output pixels m, which is the slow axis of both operands, so both MFMA
// operands are read from LDS with the transposing ds_read_b64_tr_b16 (tiles are staged exactly as
// they sit in memory: [m][n] and [m][c]).  The M range is split across workgroups (grid.y) and the
// fp32 partial tiles are combined with global atomic adds into a zero-initialised dW.
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    const bf16* dY;
    const bf16* X;
    float* dW;
    const bf16* zeros;  // >= 16 bytes of zeros (ring variant)
    float* dbias;       // ring variant: column sums of dY (the bias gradient) are added here, or nullptr
    const float* xf;    // ring variant, single tap: the X operand is relu(X * scale[c] + shift[c]); [4][C] BatchNorm constants
    int M, N, ldy, C, T, ldw;
    int Hin, Win, istr, OA, OB;
    FastDiv div_ab, div_b, div_c;
    int m_per_split;
    long long split_stride;  // elements between the destinations of consecutive pixel-range splits; 0: every split adds
                             // into dW itself.  Non-zero (deterministic mode): split s adds into dW + s * split_stride, a
                             // zeroed slab of its own, so every element has one adder and the caller sums the slabs in order
    int cp, tw;  // packed-row mode (few input channels): elements per pixel and per kernel row; C = T * tw columns
    int tap_dh[GG_MAX_TAPS];
    int tap_dw[GG_MAX_TAPS];
    int tap_woff[GG_MAX_TAPS];
};

// ring weight-gradient kernels with at least this many stages request the next stage behind the K-step's first MFMAs
#ifndef PDT_WGRAD_ISSUE_MID_MIN_STAGES
#define PDT_WGRAD_ISSUE_MID_MIN_STAGES 2
#endif
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

__device__ __forceinline__ bf16x8 lds_read_tr8(const bf16* base_lo, const bf16* base_hi) {
    // two 4(k) x 16(col) transposed block reads -> this lane's 8 consecutive k of one column
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(base_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(base_hi));
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

// SMALLC: packed rows of a few-channel convolution (the stem), see gather_gemm_kernel: the "channel" axis of
// the X operand is (kernel row, position in the row) and every chunk is one contiguous 16-byte load.
template <int BT, bool SMALLC>  // BT x BT output tile (n x c), BT in {64, 128}
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
    constexpr int KM = 64;             // pixels per K-step
    constexpr int PITCH = BT + 16;     // bf16 elements per LDS row (pad 32 B: conflict-free transposed reads)
    constexpr int TW = BT / 2;         // per-wave tile (2 x 2 waves)
    constexpr int F = TW / 16;         // fragments per wave per dim
    constexpr int CPR = BT / 8;        // 16-byte chunks per row
    constexpr int RPP = 256 / CPR;     // rows per loader pass
    constexpr int NP = KM / RPP;       // loader passes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* sY = reinterpret_cast<bf16*>(smem);  // [2][KM][PITCH]
    bf16* sX = sY + 2 * KM * PITCH;            // [2][KM][PITCH]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (p.N + BT - 1) / BT, tiles_c = (p.C + BT - 1) / BT;
    // All workgroups of one pixel range (every tile and tap) read the same rows of dY and X: give them
    // consecutive logical ids and map those to one XCD, so the rows are fetched into that L2 once.
    const int lin = xcd_remap(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
    int bid = lin % gridDim.x;
    const int split = lin / gridDim.x;
    const int ntap = SMALLC ? 1 : p.T;
    const int tap = bid % ntap;
    bid /= ntap;
    const int tc = bid % tiles_c, tn = bid / tiles_c;
    const int n0 = tn * BT, c0 = tc * BT;
    const int ms = split * p.m_per_split;
    const int me = min(p.M, ms + p.m_per_split);
    const int dh = SMALLC ? 0 : p.tap_dh[tap], dw = SMALLC ? 0 : p.tap_dw[tap];

    const int chunk = tid % CPR, lrow = tid / CPR;
    const bool y_ok = n0 + chunk * 8 < p.N;
    const bool x_ok = c0 + chunk * 8 < p.C;

    f32x4 acc[F][F];
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
        for (int j = 0; j < F; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 ry[NP], rx[NP];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    int s_kh = 0, s_piece = 0;  // packed rows: kernel row and 16-byte piece of this thread's 8 columns
    if (SMALLC) {
        const int cc = (c0 >> 3) + chunk, cpr = p.tw >> 3;
        s_kh = min(cc / cpr, p.T - 1);
        s_piece = cc % cpr;
    }
    auto load_step = [&](int mbase) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int m = mbase + lrow + i * RPP;
            const bool mok = m < me;
            ry[i] = (mok && y_ok) ? *reinterpret_cast<const bf16x8*>(p.dY + static_cast<size_t>(m) * p.ldy + n0 + chunk * 8)
                                  : zero8;
            if (SMALLC) {
                if (mok && x_ok) {
                    const uint32_t img = fastdiv(m, p.div_ab);
                    const uint32_t rem = m - img * (p.OA * p.OB);
                    const uint32_t a = fastdiv(rem, p.div_b);
                    const uint32_t b = rem - a * p.OB;
                    const uint32_t off = ((img * p.Hin + a * p.istr + s_kh) * p.Win + b * p.istr) * p.cp + s_piece * 8;
                    rx[i] = *reinterpret_cast<const bf16x8*>(p.X + off);
                } else {
                    rx[i] = zero8;
                }
                continue;
            }
            bool ok = mok && x_ok;
            uint32_t off = 0;
            if (ok) {
                const uint32_t img = fastdiv(m, p.div_ab);
                const uint32_t rem = m - img * (p.OA * p.OB);
                const uint32_t a = fastdiv(rem, p.div_b);
                const uint32_t b = rem - a * p.OB;
                const int ih = static_cast<int>(a) * p.istr + dh, iw = static_cast<int>(b) * p.istr + dw;
                ok = static_cast<unsigned>(ih) < static_cast<unsigned>(p.Hin) &&
                     static_cast<unsigned>(iw) < static_cast<unsigned>(p.Win);
                off = (img * (p.Hin * p.Win) + ih * p.Win + iw) * p.C + c0 + chunk * 8;
            }
            rx[i] = ok ? *reinterpret_cast<const bf16x8*>(p.X + off) : zero8;
        }
    };
    auto store_step = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = lrow + i * RPP;
            *reinterpret_cast<bf16x8*>(sY + (buf * KM + row) * PITCH + chunk * 8) = ry[i];
            *reinterpret_cast<bf16x8*>(sX + (buf * KM + row) * PITCH + chunk * 8) = rx[i];
        }
    };
    // Transposed fragment addressing: lane (16g + 4q + pp) supplies the address of block row q,
    // columns 4pp..4pp+3.  Which tile row plays "k = 8g + j" is free as long as both operands agree
    // (k is only summed over), so lane group g takes rows 4g..4g+3 and 16+4g..16+4g+3 of each
    // 32-row sub-step: a 32-lane half then touches 8 consecutive rows, whose 32-byte segments fall
    // on distinct banks with the padded pitch.
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    auto compute = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < KM / 32; ++kk) {
            const int krow = kk * 32 + 4 * g + q;
            bf16x8 fy[F], fx[F];
#pragma unroll
            for (int i = 0; i < F; ++i) {
                const bf16* b0 = sY + (buf * KM + krow) * PITCH + wr * TW + i * 16 + 4 * pp;
                fy[i] = lds_read_tr8(b0, b0 + 16 * PITCH);
            }
#pragma unroll
            for (int j = 0; j < F; ++j) {
                const bf16* b0 = sX + (buf * KM + krow) * PITCH + wc * TW + j * 16 + 4 * pp;
                fx[j] = lds_read_tr8(b0, b0 + 16 * PITCH);
            }
#pragma unroll
            for (int i = 0; i < F; ++i)
#pragma unroll
                for (int j = 0; j < F; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[i], fx[j], acc[i][j], 0, 0, 0);
        }
    };

    const int nsteps = (me - ms + KM - 1) / KM;
    if (nsteps > 0) {
        load_step(ms);
        store_step(0);
        __syncthreads();
        for (int s = 0; s < nsteps; ++s) {
            const bool more = s + 1 < nsteps;
            if (more) load_step(ms + (s + 1) * KM);
            compute(s & 1);
            if (more) store_step((s + 1) & 1);
            __syncthreads();
        }
    }

    // D[i = n][j = c]: lane holds rows (lane>>4)*4 + e, column lane&15
    const int woff = SMALLC ? 0 : p.tap_woff[tap];
#pragma unroll
    for (int i = 0; i < F; ++i)
#pragma unroll
        for (int j = 0; j < F; ++j) {
            const int c = c0 + wc * TW + j * 16 + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + wr * TW + i * 16 + (lane >> 4) * 4 + e;
                if (n < p.N && c < p.C)
                    __hip_atomic_fetch_add(p.dW + static_cast<size_t>(split) * p.split_stride + static_cast<size_t>(n) * p.ldw + woff + c,
                                           acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
}

// ------------------------------------------------------------------------------------------------
// Ring variant of the weight gradient: both operand tiles ([64 pixels][BN] of dY and [64 pixels][BC]
// of the gathered input) are filled by LDS-DMA into a ring of NS stages, NS - 2 of which stay in flight
// across the one barrier of a K-step (see gather_gemm_ring_kernel for the protocol).  Rows are not padded
// (an LDS-DMA instruction writes 1 KB linearly); instead the 32-byte units of a row are XOR-ed with
// row & 7, applied to the per-lane source address and again by the transposed fragment reads: the 8
// consecutive rows a 32-lane half of `ds_read_b64_tr_b16` touches then fall on 8 distinct bank groups.
// ------------------------------------------------------------------------------------------------
// MODE 0: the column axis of the weight tile is the flattened (tap, channel) index, so a tile may span several
//         taps (a 64-channel 3x3 layer puts four taps into one 256-wide tile and reads dY once for them).
// MODE 1: packed rows of a few-channel convolution (see gather_gemm_kernel): column = (kernel row, position).
// PLAIN (MODE 0 only): one tap at (0, 0), stride 1, output grid = input grid, i.e. output pixel m reads input
// pixel m (linear layers, 1x1 stride-1 convolutions).  Both operands are then walked with running pointers that
// advance by 64 rows per stage, and the loader issues unconditionally (rows beyond the split's range come
// from the zero page), which leaves one basic block per K-step (see gather_gemm_ring_kernel).
// `block` is the workgroup's index inside its problem's grid of gx x (nblocks / gx) workgroups.
// XF (PLAIN only): the X operand is the BatchNorm + ReLU of the stored tensor, recomputed on the way in: every X stage
// is transformed in place in LDS after it has landed (a thread rewrites its 16-byte chunks with the scale / shift of
// their channels from a table in LDS), then a second barrier.  Rows beyond the pixel range hold relu(shift) instead of
// zero after that, which is harmless: the dY rows they are multiplied with are zero.
//
// PP ("ping-pong", 256x256 tile, 2 x 4 waves, two stages, PLAIN only; see gather_gemm_ring_kernel): a K-step is cut
// into four phases of 16 MFMAs - (32-pixel slice, half of the wave's dY fragments) - each bracketed by two workgroup
// barriers, and wave row 1 runs one barrier behind wave row 0, so on every SIMD one wave issues MFMAs while the other
// issues its transposed fragment reads and LDS-DMA requests.  Operands are staged in four units of 32 pixel rows per
// K-step, in the order {X rows 0-31, dY rows 0-31, X rows 32-63, dY rows 32-63}, one unit per phase, six units
// ahead of the reads; a counted wait leaves four units in flight across the barriers.  With that order a unit is
// overwritten no earlier than a full phase after its last reader's MFMAs, for either wave row.
template <int BN, int BC, int WN, int WC, int NS, int MODE, bool PLAIN, bool XF = false, bool PP = false>
__device__ __forceinline__ void wgrad_ring_body(const WgradArgs& p, int block, int gx, int nblocks) {
    static_assert(!XF || (PLAIN && MODE == 0), "operand transform: single-tap launches");
    static_assert(!PP || (PLAIN && !XF && MODE == 0 && BN == 256 && BC == 256 && WN == 2 && WC == 4 && NS == 2),
                  "ping-pong schedule: 256x256, 2x4 waves, single tap");
    constexpr int KM = 64;                       // pixels per K-step
    constexpr int NW = WN * WC;
    constexpr int TN = BN / WN, TC = BC / WC;    // per-wave tile
    constexpr int FN = TN / 16, FC = TC / 16;
    constexpr int YCH = BN / 8, XCH = BC / 8;    // 16-byte chunks per row
    constexpr int YR = 64 / YCH, XR = 64 / XCH;  // rows per 1 KB DMA piece
    constexpr int YI = (KM / YR) / NW, XI = (KM / XR) / NW;  // pieces per wave and stage
    constexpr int LPW = YI + XI;
    constexpr int STAGE = KM * (BN + BC);        // elements: [Y tile | X tile]
    static_assert((KM / YR) % NW == 0 && (KM / XR) % NW == 0, "whole DMA pieces per wave");
    static_assert(NS >= 2 && (NS - 2) * LPW <= 63, "vmcnt is a 6-bit counter");
    static_assert(YR <= 8 && XR <= 8 && BN >= 64 && BC >= 64, "rows of at least 128 bytes");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* ring = reinterpret_cast<bf16*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int KC = p.T * p.C;  // columns of the weight gradient (MODE 1: C is the row width tw)
    const int tiles_c = (KC + BC - 1) / BC;
    // workgroups of one pixel range share their operand rows: consecutive logical ids, one XCD (see wgrad_kernel)
    const int lin = xcd_remap(block, nblocks);
    const int bid = lin % gx;
    const int split = lin / gx;
    const int tc = bid % tiles_c, tn = bid / tiles_c;
    const int n0 = tn * BN, c0 = tc * BC;
    const int ms = split * p.m_per_split;
    const int me = min(p.M, ms + p.m_per_split);
    const int nsteps = (me - ms + KM - 1) / KM;
    float* xf_tab = reinterpret_cast<float*>(ring + NS * STAGE);  // XF: [2][BC] scale, shift of channels c0 .. c0 + BC
    if constexpr (XF) {
        for (int i = tid; i < 2 * BC; i += NW * 64) {
            const int which = i / BC, c = c0 + (i - which * BC);
            xf_tab[i] = c < p.C ? p.xf[(2 + which) * p.C + c] : 0.f;
        }
        __syncthreads();
    }

    // 32-byte units of a row are XOR-ed with a function of the row so that the 8 consecutive rows one half of a
    // transposed read touches fall on 8 distinct bank groups: row & 7 for rows of 256 bytes or more (every row
    // starts on bank 0), (row >> 1) & 3 for 128-byte rows (two rows per 256-byte bank line).
    auto swz_y = [](int row) { return BN == 64 ? ((row >> 1) & 3) : (row & 7); };
    auto swz_x = [](int row) { return BC == 64 ? ((row >> 1) & 3) : (row & 7); };

    // loader: lane -> (row inside a piece, physical chunk); the logical chunk it fetches undoes the swizzle
    const int y_lr = lane / YCH, y_pc = lane % YCH;
    const int x_lr = lane / XCH, x_pc = lane % XCH;
    // First tile row of this wave's DMA piece i.  Plain ring: consecutive pieces.  Ping-pong: pieces 2i' and 2i' + 1 of
    // the 32-row staging unit i >> 1 (sixteen 2-row pieces per unit, two per wave).
    auto y_row0 = [&](int i) {
        if constexpr (PP) return (i >> 1) * 32 + (2 * wave + (i & 1)) * YR;
        else return (wave * YI + i) * YR;
    };
    auto x_row0 = [&](int i) {
        if constexpr (PP) return (i >> 1) * 32 + (2 * wave + (i & 1)) * XR;
        else return (wave * XI + i) * XR;
    };
    int x_dh[XI], x_dw[XI], x_cc[XI];  // per piece: tap offset and channel (MODE 0) / kernel row, -, position (MODE 1)
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int row = x_row0(i) + x_lr;
        const int j = c0 + ((((x_pc >> 1) ^ swz_x(row)) << 1) | (x_pc & 1)) * 8;  // first of this lane's 8 columns
        if (j >= KC) {
            x_dh[i] = -(1 << 24);
            x_dw[i] = 0;
            x_cc[i] = 0;
        } else if (MODE == 1) {
            const int cpr = p.tw >> 3, cc = j >> 3;
            x_dh[i] = cc / cpr;
            x_dw[i] = 0;
            x_cc[i] = (cc % cpr) * 8;
        } else {
            const int tap = fastdiv(j, p.div_c);
            x_dh[i] = p.tap_dh[tap];
            x_dw[i] = p.tap_dw[tap];
            x_cc[i] = j - tap * p.C;
        }
    }
    int ld_issued = 0, ld_slot = 0;
    auto issue_stage = [&]() {
        const int mbase = ms + ld_issued * KM;
        bf16* sY = ring + ld_slot * STAGE;
        bf16* sX = sY + KM * BN;
#pragma unroll
        for (int i = 0; i < YI; ++i) {
            const int row = (wave * YI + i) * YR + y_lr;
            const int c = ((((y_pc >> 1) ^ swz_y(row)) << 1) | (y_pc & 1)) * 8;
            const int m = mbase + row;
            const bool ok = m < me && n0 + c < p.N;
            const bf16* src = ok ? p.dY + static_cast<size_t>(m) * p.ldy + n0 + c : p.zeros;
            __builtin_amdgcn_global_load_lds((global_cvoid_t*)src, (lds_void_t*)(sY + (wave * YI + i) * YR * BN), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int row = (wave * XI + i) * XR + x_lr;
            const int m = mbase + row;
            bool ok = m < me && x_dh[i] > -(1 << 23);
            uint32_t off = 0;
            if (ok) {
                const uint32_t img = fastdiv(m, p.div_ab);
                const uint32_t rem = m - img * (p.OA * p.OB);
                const uint32_t a = fastdiv(rem, p.div_b);
                const uint32_t b = rem - a * p.OB;
                const int ih = static_cast<int>(a) * p.istr + x_dh[i], iw = static_cast<int>(b) * p.istr + x_dw[i];
                if (MODE == 1) {
                    off = ((img * p.Hin + ih) * p.Win + iw) * p.cp + x_cc[i];
                } else {
                    ok = static_cast<unsigned>(ih) < static_cast<unsigned>(p.Hin) &&
                         static_cast<unsigned>(iw) < static_cast<unsigned>(p.Win);
                    off = (img * (p.Hin * p.Win) + ih * p.Win + iw) * p.C + x_cc[i];
                }
            }
            const bf16* src = ok ? p.X + off : p.zeros;
            __builtin_amdgcn_global_load_lds((global_cvoid_t*)src, (lds_void_t*)(sX + (wave * XI + i) * XR * BC), 16, 0, 0);
        }
        ++ld_issued;
        ld_slot = ld_slot + 1 == NS ? 0 : ld_slot + 1;
    };

    // PLAIN loader state
    const bf16* y_ptr[YI];
    const bf16* x_ptr[XI];
    size_t y_adv[YI], x_adv[XI];
    if constexpr (PLAIN) {
#pragma unroll
        for (int i = 0; i < YI; ++i) {
            const int row = y_row0(i) + y_lr;
            const int c = ((((y_pc >> 1) ^ swz_y(row)) << 1) | (y_pc & 1)) * 8;
            const bool ok = n0 + c < p.N;
            y_ptr[i] = ok ? p.dY + static_cast<size_t>(ms + row) * p.ldy + n0 + c : p.zeros;
            y_adv[i] = ok ? static_cast<size_t>(KM) * p.ldy : 0;
        }
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int row = x_row0(i) + x_lr;
            const bool ok = x_dh[i] > -(1 << 23);
            x_ptr[i] = ok ? p.X + static_cast<size_t>(ms + row) * p.C + x_cc[i] : p.zeros;
            x_adv[i] = ok ? static_cast<size_t>(KM) * p.C : 0;
        }
    }
    auto plain_issue = [&]() {
        const int mbase = ms + ld_issued * KM;
        bf16* sY = ring + ld_slot * STAGE;
        bf16* sX = sY + KM * BN;
#pragma unroll
        for (int i = 0; i < YI; ++i) {
            const bool ok = mbase + (wave * YI + i) * YR + y_lr < me;
            __builtin_amdgcn_global_load_lds((global_cvoid_t*)(ok ? y_ptr[i] : p.zeros), (lds_void_t*)(sY + (wave * YI + i) * YR * BN), 16, 0, 0);
            y_ptr[i] += y_adv[i];
        }
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const bool ok = mbase + (wave * XI + i) * XR + x_lr < me;
            __builtin_amdgcn_global_load_lds((global_cvoid_t*)(ok ? x_ptr[i] : p.zeros), (lds_void_t*)(sX + (wave * XI + i) * XR * BC), 16, 0, 0);
            x_ptr[i] += x_adv[i];
        }
        ++ld_issued;
        ld_slot = ld_slot + 1 == NS ? 0 : ld_slot + 1;
    };
    // ping-pong: staging unit r of the K-step under the loader's cursor (0: X rows 0-31, 1: dY rows 0-31, 2: X rows
    // 32-63, 3: dY rows 32-63); after the last unit the cursor moves to the next K-step
    auto pp_unit = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const int mbase = ms + ld_issued * KM;
        bf16* sY = ring + ld_slot * STAGE;
        bf16* sX = sY + KM * BN;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            constexpr int half = r >> 1;
            const int i = 2 * half + k;
            if constexpr (r & 1) {
                const bool ok = mbase + y_row0(i) + y_lr < me;
                __builtin_amdgcn_global_load_lds((global_cvoid_t*)(ok ? y_ptr[i] : p.zeros), (lds_void_t*)(sY + y_row0(i) * BN), 16, 0, 0);
                y_ptr[i] += y_adv[i];
            } else {
                const bool ok = mbase + x_row0(i) + x_lr < me;
                __builtin_amdgcn_global_load_lds((global_cvoid_t*)(ok ? x_ptr[i] : p.zeros), (lds_void_t*)(sX + x_row0(i) * BC), 16, 0, 0);
                x_ptr[i] += x_adv[i];
            }
        }
        if constexpr (r == 3) {
            ++ld_issued;
            ld_slot ^= 1;
        }
    };

    f32x4 acc[FN][FC];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Bias gradient = column sums of dY = dY^T * 1: one more MFMA per dY fragment against a fragment of ones.
    // All workgroups of a row tile and all waves of a wave row see the same dY, so the extra work is dealt
    // out: column tile tc takes the K-steps s with s % tiles_c == tc, wave column wc the fragments i with
    // i % WC == wc; everything is added to dbias atomically at the end.
    const bool do_bias = p.dbias != nullptr;
    constexpr int NB = (FN + WC - 1) / WC;  // fragments i with i % WC == wc: their sums live in accb[i / WC]
    f32x4 accb[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = f32_to_bf16(1.f);

    // transposed fragment addressing as in wgrad_kernel: lane (16g + 4q + pp) supplies block row q, columns
    // 4pp..4pp+3; lane group g takes rows 4g..4g+3 and 16+4g..16+4g+3 of each 32-row sub-step.
    // The reads are inline assembly: through the builtin the compiler cannot tell them from the pending
    // LDS-DMA writes and drains the whole DMA queue (vmcnt(0)) in front of every K-step.  Their completion
    // is therefore awaited by hand (lgkmcnt(0), then a scheduling fence so no MFMA is hoisted above it).
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_void_t*)ring));
    // A K-step is worked off in units of (32-pixel slice kk, group of dY fragments): the fragment reads of the next unit
    // are issued before the MFMAs of the current one and awaited with a counted lgkmcnt, so within a wave the LDS
    // latency hides behind 16 MFMAs instead of being paid at the head of every slice.
    constexpr bool WGRAD_ISSUE_MID = NS >= PDT_WGRAD_ISSUE_MID_MIN_STAGES;
    auto compute = [&](int slot, const bool bias_step) {
        constexpr int HG = FN >= 4 ? 2 : 1;  // groups of dY fragments per slice
        constexpr int GI = FN / HG;
        constexpr int UNITS = 2 * HG;
        const uint32_t yb = lds_base + static_cast<uint32_t>(slot * STAGE) * 2u;
        const uint32_t xb = yb + KM * BN * 2u;
        s16x4 ylo[2][GI], yhi[2][GI], xlo[2][FC], xhi[2][FC];
        auto issue_x = [&](int kk) {
            const int krow = kk * 32 + 4 * g + q;  // krow and krow + 16 swizzle alike
            const int swx = swz_x(krow);
#pragma unroll
            for (int j = 0; j < FC; ++j) {
                const int u = (wc * TC + j * 16) >> 4;
                const uint32_t a = xb + static_cast<uint32_t>(krow * BC + ((u ^ swx) << 4) + 4 * pp) * 2u;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xlo[kk][j]) : "v"(a) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(xhi[kk][j]) : "v"(a), "n"(16 * BC * 2) : "memory");
            }
        };
        auto issue_y = [&](int kk, int h, int buf) {
            const int krow = kk * 32 + 4 * g + q;
            const int swy = swz_y(krow);
#pragma unroll
            for (int ii = 0; ii < GI; ++ii) {
                const int u = (wr * TN + (h * GI + ii) * 16) >> 4;
                const uint32_t a = yb + static_cast<uint32_t>(krow * BN + ((u ^ swy) << 4) + 4 * pp) * 2u;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ylo[buf][ii]) : "v"(a) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(yhi[buf][ii]) : "v"(a), "n"(16 * BN * 2) : "memory");
            }
        };
        issue_x(0);
        issue_y(0, 0, 0);
        auto unit = [&](auto un_c) {
            constexpr int un = decltype(un_c)::value;
            constexpr int kk = un / HG, h = un % HG, buf = un & 1;
            if constexpr (un + 1 < UNITS) {
                constexpr int kn = (un + 1) / HG, hn = (un + 1) % HG;
                if constexpr (kn != kk) issue_x(kn);
                issue_y(kn, hn, buf ^ 1);
                // everything older than the reads just issued has to be back (the counter holds 15 at most)
                constexpr int AHEAD = (kn != kk ? 2 * FC : 0) + 2 * GI;
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(AHEAD < 15 ? AHEAD : 15) : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ii = 0; ii < GI; ++ii) {
                const int i = h * GI + ii;
                const s16x8 yv = {ylo[buf][ii][0], ylo[buf][ii][1], ylo[buf][ii][2], ylo[buf][ii][3],
                                  yhi[buf][ii][0], yhi[buf][ii][1], yhi[buf][ii][2], yhi[buf][ii][3]};
                if (bias_step && (i % WC) == wc)
                    accb[i / WC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, yv), ones, accb[i / WC], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < FC; ++j) {
                    const s16x8 xv = {xlo[kk][j][0], xlo[kk][j][1], xlo[kk][j][2], xlo[kk][j][3],
                                      xhi[kk][j][0], xhi[kk][j][1], xhi[kk][j][2], xhi[kk][j][3]};
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, yv), __builtin_bit_cast(bf16x8, xv),
                                                                         acc[i][j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        unit(std::integral_constant<int, 0>{});
        if constexpr (!PLAIN && WGRAD_ISSUE_MID) {
            // Gather loader: the next stage is requested here, behind the K-step's first MFMAs, instead of right after the
            // barrier - its index arithmetic (two divisions per piece) then runs under the matrix pipe and not in front
            // of the K-step's first fragment reads.
            if (ld_issued < nsteps) issue_stage();
            __builtin_amdgcn_sched_barrier(0);
        }
        unit(std::integral_constant<int, 1>{});
        if constexpr (UNITS == 4) {
            unit(std::integral_constant<int, 2>{});
            unit(std::integral_constant<int, 3>{});
        }
    };

    int slot = 0;
    int bias_turn = do_bias ? tc : -1;  // counts down to this workgroup's next bias K-step
    if constexpr (PP) {
        using U0 = std::integral_constant<int, 0>;
        using U1 = std::integral_constant<int, 1>;
        using U2 = std::integral_constant<int, 2>;
        using U3 = std::integral_constant<int, 3>;
        constexpr int GI = FN / 2;  // dY fragments of a phase
        // six units ahead: the whole first K-step and the first two units of the second
        pp_unit(U0{}); pp_unit(U1{}); pp_unit(U2{}); pp_unit(U3{}); pp_unit(U0{}); pp_unit(U1{});
        // four units (two DMA instructions each per wave) may stay in flight: the first two have landed
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        if (wr == 1) asm volatile("s_barrier" ::: "memory");  // wave row 1 runs one barrier behind row 0 from here on
        for (int s = 0; s < nsteps; ++s) {
            const bool bias_step = bias_turn == 0;
            const uint32_t yb = lds_base + static_cast<uint32_t>(slot * STAGE) * 2u;
            const uint32_t xb = yb + KM * BN * 2u;
            // fragment registers alternate between phases (dY) and slices (X): a read never targets registers that MFMAs
            // issued just before still name as sources
            s16x4 ylo[2][GI], yhi[2][GI], xlo[2][FC], xhi[2][FC];
            // One phase: DMA unit for a later K-step, this phase's fragment reads -> counted wait that retires the unit
            // the next phase reads -> barrier -> reads back -> 16 MFMAs at raised priority -> barrier.
            auto phase = [&](auto phc, auto unitc) {
                constexpr int ph = decltype(phc)::value;
                constexpr int kk = ph >> 1, h = ph & 1, buf = ph & 1;
                pp_unit(unitc);
                const int krow = kk * 32 + 4 * g + q;  // krow and krow + 16 swizzle alike
                if constexpr (h == 0) {
                    const int swx = swz_x(krow);
#pragma unroll
                    for (int j = 0; j < FC; ++j) {
                        const int u = (wc * TC + j * 16) >> 4;
                        const uint32_t a = xb + static_cast<uint32_t>(krow * BC + ((u ^ swx) << 4) + 4 * pp) * 2u;
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xlo[kk][j]) : "v"(a) : "memory");
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(xhi[kk][j]) : "v"(a), "n"(16 * BC * 2) : "memory");
                    }
                }
                const int swy = swz_y(krow);
#pragma unroll
                for (int ii = 0; ii < GI; ++ii) {
                    const int u = (wr * TN + (h * GI + ii) * 16) >> 4;
                    const uint32_t a = yb + static_cast<uint32_t>(krow * BN + ((u ^ swy) << 4) + 4 * pp) * 2u;
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ylo[buf][ii]) : "v"(a) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(yhi[buf][ii]) : "v"(a), "n"(16 * BN * 2) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ii = 0; ii < GI; ++ii) {
                    // the reads return in order (X fragments first, then the dY fragments one by one): a group of MFMAs
                    // waits for its own dY fragment only, the later ones land under it
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (GI - 1 - ii)) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    const int i = h * GI + ii;
                    const s16x8 yv = {ylo[buf][ii][0], ylo[buf][ii][1], ylo[buf][ii][2], ylo[buf][ii][3],
                                      yhi[buf][ii][0], yhi[buf][ii][1], yhi[buf][ii][2], yhi[buf][ii][3]};
                    if (bias_step && (i % WC) == wc)
                        accb[i / WC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, yv), ones, accb[i / WC], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < FC; ++j) {
                        const s16x8 xv = {xlo[kk][j][0], xlo[kk][j][1], xlo[kk][j][2], xlo[kk][j][3],
                                          xhi[kk][j][0], xhi[kk][j][1], xhi[kk][j][2], xhi[kk][j][3]};
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, yv), __builtin_bit_cast(bf16x8, xv),
                                                                             acc[i][j], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            phase(U0{}, U2{});
            phase(U1{}, U3{});
            phase(U2{}, U0{});
            phase(U3{}, U1{});
            if (do_bias) bias_turn = bias_step ? tiles_c - 1 : bias_turn - 1;
            slot ^= 1;
        }
        if (wr == 0) asm volatile("s_barrier" ::: "memory");  // pairs with the last barrier of wave row 1
    } else {
#pragma unroll
    for (int k = 0; k < NS - 1; ++k) {
        if constexpr (PLAIN) plain_issue();
        else if (ld_issued < nsteps) issue_stage();
    }
    for (int s = 0; s < nsteps; ++s) {
        if constexpr (PLAIN) {
            wait_vm_and_barrier<(NS - 2) * LPW>();  // NS - 1 stages are always in flight here
            plain_issue();
            if constexpr (XF) {
                bf16* sXx = ring + slot * STAGE + KM * BN;
#pragma unroll
                for (int i = 0; i < (KM * XCH + NW * 64 - 1) / (NW * 64); ++i) {
                    const int q = tid + i * (NW * 64);
                    if ((KM * XCH) % (NW * 64) != 0 && q >= KM * XCH) break;
                    const int row = q / XCH, pc = q % XCH;
                    const int cb = ((((pc >> 1) ^ swz_x(row)) << 1) | (pc & 1)) << 3;  // channels (relative to c0) of this chunk
                    bf16* a = sXx + row * BC + (pc << 3);
                    bf16x8 v = *reinterpret_cast<const bf16x8*>(a);
                    float sc[8], sh[8];
                    *reinterpret_cast<f32x4*>(sc) = *reinterpret_cast<const f32x4*>(xf_tab + cb);
                    *reinterpret_cast<f32x4*>(sc + 4) = *reinterpret_cast<const f32x4*>(xf_tab + cb + 4);
                    *reinterpret_cast<f32x4*>(sh) = *reinterpret_cast<const f32x4*>(xf_tab + BC + cb);
                    *reinterpret_cast<f32x4*>(sh + 4) = *reinterpret_cast<const f32x4*>(xf_tab + BC + cb + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = f32_to_bf16(fmaxf(bf16_to_f32(v[e]) * sc[e] + sh[e], 0.f));
                    *reinterpret_cast<bf16x8*>(a) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        } else {
            if (ld_issued - (s + 1) >= NS - 2) wait_vm_and_barrier<(NS - 2) * LPW>();
            else wait_vm_and_barrier<0>();
            if constexpr (!WGRAD_ISSUE_MID) {
                if (ld_issued < nsteps) issue_stage();
            }
        }
        const bool bias_step = bias_turn == 0;
        compute(slot, bias_step);
        if (do_bias) bias_turn = bias_step ? tiles_c - 1 : bias_turn - 1;
        slot = slot + 1 == NS ? 0 : slot + 1;
    }
    }

    if constexpr (PLAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the loader's trailing zero-page stages
    if (do_bias && (lane & 15) == 0) {  // every column of accb holds the same sums: take column 0
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            if ((i % WC) != wc) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + wr * TN + i * 16 + (lane >> 4) * 4 + e;
                if (n < p.N) __hip_atomic_fetch_add(p.dbias + n, accb[i / WC][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // D[i = n][j = column]: lane holds rows (lane>>4)*4 + e, column lane&15
#pragma unroll
    for (int j = 0; j < FC; ++j) {
        const int col = c0 + wc * TC + j * 16 + (lane & 15);
        if (col >= KC) continue;
        int wcol = col;
        if (MODE == 0) {
            const int tap = fastdiv(col, p.div_c);
            wcol = p.tap_woff[tap] + col - tap * p.C;
        }
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + wr * TN + i * 16 + (lane >> 4) * 4 + e;
                if (n < p.N)
                    __hip_atomic_fetch_add(p.dW + static_cast<size_t>(split) * p.split_stride + static_cast<size_t>(n) * p.ldw + wcol,
                                           acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    }
}

template <int BN, int BC, int WN, int WC, int NS, int MODE, bool PLAIN = false, bool XF = false, bool PP = false>
__global__ __launch_bounds__(WN * WC * 64) void wgrad_ring_kernel(const WgradArgs p) {
    wgrad_ring_body<BN, BC, WN, WC, NS, MODE, PLAIN, XF, PP>(p, blockIdx.x + blockIdx.y * gridDim.x, gridDim.x, gridDim.x * gridDim.y);
}

// Two independent weight-gradient problems of the same tile shape in one launch (workgroups [0, n0) belong to the
// first).  A workgroup's fp32 atomics are fire-and-forget, so when the first problem's workgroups retire the
// second problem's main loops start on the freed CUs while those atomics are still draining at the L2: the
// atomic tail of the first problem - a quarter of its run time on transformer shapes - is no longer exposed.
template <int BN, int BC, int WN, int WC, int NS, bool PP = false>
__global__ __launch_bounds__(WN * WC * 64) void wgrad_ring_dual_kernel(const WgradArgs p0, const WgradArgs p1, int gx0, int n0,
                                                                      int gx1, int n1) {
    const int b = blockIdx.x;
    if (b < n0) wgrad_ring_body<BN, BC, WN, WC, NS, 0, true, false, PP>(p0, b, gx0, n0);
    else wgrad_ring_body<BN, BC, WN, WC, NS, 0, true, false, PP>(p1, b - n0, gx1, n1);
}

template <int BN, int BC, int WN, int WC, int NS, int MODE, bool PLAIN, bool XF = false, bool PP = false>
static int launch_wgrad_ring_plain(const WgradArgs& a, int gy, hipStream_t stream);

template <int BN, int BC, int WN, int WC, int NS, int MODE = 0>
static int launch_wgrad_ring(const WgradArgs& a, int gy, hipStream_t stream) {
    if constexpr (MODE == 0) {
        // output pixel m reads input pixel m: the pointer-walking loader applies
        if (a.T == 1 && a.tap_dh[0] == 0 && a.tap_dw[0] == 0 && a.istr == 1 && a.OA == a.Hin && a.OB == a.Win) {
            if (a.xf != nullptr) return launch_wgrad_ring_plain<BN, BC, WN, WC, NS, MODE, true, true>(a, gy, stream);
            return launch_wgrad_ring_plain<BN, BC, WN, WC, NS, MODE, true>(a, gy, stream);
        }
    }
    if (a.xf != nullptr) return -1;  // the operand transform needs the single-tap form
    return launch_wgrad_ring_plain<BN, BC, WN, WC, NS, MODE, false>(a, gy, stream);
}

template <int BN, int BC, int WN, int WC, int NS, int MODE, bool PLAIN, bool XF, bool PP>
static int launch_wgrad_ring_plain(const WgradArgs& a, int gy, hipStream_t stream) {
    constexpr size_t lds = static_cast<size_t>(NS) * 64 * (BN + BC) * sizeof(bf16) + (XF ? 2 * BC * sizeof(float) : 0);
    static_assert(lds <= 160 * 1024, "ring does not fit the LDS of a CU");
    if (a.zeros == nullptr) return -1;
    auto kernel = wgrad_ring_kernel<BN, BC, WN, WC, NS, MODE, PLAIN, XF, PP>;
    static bool configured = false;
    if (!configured) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return static_cast<int>(e);
        configured = true;
    }
    const int gx = pdt_cdiv(a.N, BN) * pdt_cdiv(a.T * a.C, BC);
    hipLaunchKernelGGL(kernel, dim3(gx, gy), dim3(WN * WC * 64), lds, stream, a);
    return pdt_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3 / stride 1 / padding 1 / 64 -> 64 convolution, input rows shared between the nine taps.
//
// The tap-table kernel above stages the [64 pixels][64] input operand of every tap separately (nine trips through the
// L2 -> LDS path) and needs three 256-column tiles for the 576 columns, each re-reading dY.  Here the reduction runs
// over PADDED pixel indices q = (img (H + 2) + hp) (W + 2) + wp, as if every image had a one-pixel border: dY and X
// are laid out in LDS in that numbering, border positions hold zeros (the DMA's range check writes them), and tap
// (dh, dw) of pixel q is simply row q + dh (W + 2) + dw of the same array - never outside the image's padded grid, so
// there is no bounds logic in the main loop at all; a border position contributes nothing because its dY row is zero.
// The price is (H + 2)(W + 2) / (H W) - 1 more K-steps (7 % at 56 x 56).
//
//   workgroup = one contiguous range of padded pixels (a split), all 64 x 576 columns: 8 waves as 2 (32 output
//   channels) x 4 (9 column blocks of 16 = consecutive (tap, channel quarter) pairs), 72 accumulator registers
//   LDS: X rows in a ring of eight 64-row blocks, dY in a ring of four 128-row stages
//   per K-step of 128 pixels: one barrier, 6 DMA requests per wave (128 new X rows and the dY rows two steps ahead)
// ------------------------------------------------------------------------------------------------
// C = 64: blocks of 64 rows, a K-step of 128 pixels, one workgroup for all 64 output channels.
// C = 128 (and 128 output channels): rows of 256 bytes, blocks of 32 rows (W + 3 <= 32), a K-step of 64 pixels, and two
// workgroups per pixel range (blockIdx.y), each with 64 of the output channels - 64 x 1152 columns are 144 accumulator
// registers per lane.
template <int C>
struct WgradHalo {
    static constexpr int BR = C == 64 ? 64 : 32;        // rows per block; a block is 8 KB
    static constexpr int KS = 2 * BR;                   // pixels per K-step
    static constexpr int KK = KS / 32;                  // 32-pixel slices per K-step
    static constexpr int RB = 2 * C;                    // bytes per X row
    static constexpr int UNITS = C / 16;                // 16-channel units per tap
    static constexpr int NJ = 9 * UNITS / 4;            // column blocks per wave
    static constexpr int GJ = 8 / (2 * KK);             // column blocks per group of 8 fragment reads
    static constexpr int NG = NJ / GJ;                  // groups per K-step (9)
    static constexpr int YP = KS / 64;                  // dY pieces (64 rows of 128 bytes) per K-step
    static constexpr int X_BYTES = 11 * 8192;           // ring of eight blocks, blocks 0-2 mirrored behind block 7
    static constexpr int Y_STAGE = KS * 128;
    static constexpr int LDS = X_BYTES + 4 * Y_STAGE;
    static constexpr int REQ = 4 + YP;                  // DMA requests per wave and K-step
};

// XF: the X operand is relu(X * scale[c] + shift[c]) (see conv3x3_halo_kernel): a lane rewrites the chunks its own requests
// fetched - both copies of a mirrored block - after the counted wait that retires them and in front of the K-step's
// barrier; border positions and rows outside the tensor stay zero.
template <int C, bool XF = false>
__global__ __launch_bounds__(512) void wgrad3x3_halo_kernel(const WgradArgs p, const int P) {
    using G = WgradHalo<C>;
    constexpr int BR = G::BR, KS = G::KS, KK = G::KK, RB = G::RB, NJ = G::NJ, GJ = G::GJ, NG = G::NG;
    static_assert(NG == 9 && NJ % GJ == 0, "nine groups of eight reads per K-step");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr uint32_t OUT_OF_RANGE = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int H = p.Hin, W = p.Win, Wp = W + 2, HpWp = (H + 2) * Wp, HW = H * W;
    const int split = blockIdx.x;
    const int nh = blockIdx.y;  // which 64 output channels
    const int ps = split * p.m_per_split, pe = min(P, ps + p.m_per_split);  // (ps is a multiple of KS)
    const int nsteps = (pe - ps + KS - 1) / KS;
    const int blk0 = ps / BR;
    // 32-bit byte offsets from the first pixel of the image the range starts in
    const int img_lo = max(ps - BR, 0) / HpWp;
    bf16* xbase = const_cast<bf16*>(p.X) + static_cast<size_t>(img_lo) * HW * C;
    bf16* ybase = const_cast<bf16*>(p.dY) + static_cast<size_t>(img_lo) * HW * p.ldy + nh * 64;

    // element offset of padded pixel q from the base pixel, or -1 for a border position / a pixel outside the tensor
    auto pixel_of = [&](int q) -> int {
        if (q < 0 || q >= P) return -1;
        const uint32_t img = fastdiv(static_cast<uint32_t>(q), p.div_ab);
        const uint32_t rem = static_cast<uint32_t>(q) - img * HpWp;
        const uint32_t hp = fastdiv(rem, p.div_b);
        const uint32_t wp = rem - hp * Wp;
        if (hp - 1u >= static_cast<uint32_t>(H) || wp - 1u >= static_cast<uint32_t>(W)) return -1;
        return (static_cast<int>(img) - img_lo) * HW + static_cast<int>(hp - 1u) * W + static_cast<int>(wp - 1u);
    };
    // ---- loader: 1 KB pieces; lane -> (row, physical 16-byte chunk); the 32-byte units of a row are XOR-ed with
    // (row >> 1) & 3 (128-byte rows) or row & 7 (256-byte rows), see wgrad_ring_body
    // X block b = padded rows [ps + BR (b - 1), ps + BR b), at ring position (blk0 + b - 1) mod 8.  The window of a K-step
    // is four consecutive blocks; so that it never wraps (which would put a wrap test into every fragment address), ring
    // positions 0-2 exist a second time behind position 7.  Every block is requested twice - the second request goes to
    // the mirror position if there is one and to the same place otherwise - so that a K-step always issues the same
    // number of requests (the counted waits below rely on that).
    uint32_t x_real = 0;  // XF: bit b & 31 = this lane's row of block b is a real pixel (blocks are transformed at most 6 behind their request)
    auto issue_x = [&](int b) {
        constexpr int CPR = RB / 16, RPP = 1024 / RB;  // chunks per row, rows per piece
        const int lrow = lane / CPR, pc = lane % CPR;
        const int pos = (blk0 + b - 1) & 7;
        const int q = ps + BR * (b - 1) + RPP * wave + lrow;
        const int sw = C == 64 ? ((q >> 1) & 3) : (q & 7);
        const int chunk = (((pc >> 1) ^ sw) << 1) | (pc & 1);
        const int m = pixel_of(q);
        const uint32_t voff = m >= 0 ? static_cast<uint32_t>(m * C + chunk * 8) * 2u : OUT_OF_RANGE;
        char* dst = smem + pos * 8192 + wave * 1024;
        buffer_load_lds_16(xbase, (lds_void_t*)dst, voff, 0);
        buffer_load_lds_16(xbase, (lds_void_t*)(dst + (pos < 3 ? 8 * 8192 : 0)), voff, 0);
        if constexpr (XF) x_real = (x_real & ~(1u << (b & 31))) | ((m >= 0 ? 1u : 0u) << (b & 31));  // remembered for the transform
    };
    float xsc[8], xsh[8];  // XF: scale / shift of the 8 channels of this lane's chunk (the same logical chunk in every block)
    if constexpr (XF) {
        constexpr int CPR = RB / 16, RPP = 1024 / RB;
        const int lrow = lane / CPR, pc = lane % CPR;
        const int q = RPP * wave + lrow;  // (block starts are multiples of 8 rows: the swizzle sees only these bits)
        const int sw = C == 64 ? ((q >> 1) & 3) : (q & 7);
        const int c0 = ((((pc >> 1) ^ sw) << 1) | (pc & 1)) * 8;
        *reinterpret_cast<f32x4*>(xsc) = *reinterpret_cast<const f32x4*>(p.xf + 2 * C + c0);
        *reinterpret_cast<f32x4*>(xsc + 4) = *reinterpret_cast<const f32x4*>(p.xf + 2 * C + c0 + 4);
        *reinterpret_cast<f32x4*>(xsh) = *reinterpret_cast<const f32x4*>(p.xf + 3 * C + c0);
        *reinterpret_cast<f32x4*>(xsh + 4) = *reinterpret_cast<const f32x4*>(p.xf + 3 * C + c0 + 4);
    }
    // this lane's chunks of X blocks b and b + 1, where they landed (and their mirrors): both reads first, so that the two
    // LDS round trips overlap
    auto xform_x2 = [&](int b) {
        char* at[2];
        bool real[2], mir[2];
        bf16x8 v[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int pos = (blk0 + b + k - 1) & 7;
            at[k] = smem + pos * 8192 + wave * 1024 + lane * 16;
            mir[k] = pos < 3;
            real[k] = ((x_real >> ((b + k) & 31)) & 1u) != 0;
            if (real[k]) v[k] = *reinterpret_cast<const bf16x8*>(at[k]);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!real[k]) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[k][e] = f32_to_bf16(fmaxf(bf16_to_f32(v[k][e]) * xsc[e] + xsh[e], 0.f));
            *reinterpret_cast<bf16x8*>(at[k]) = v[k];
            if (mir[k]) *reinterpret_cast<bf16x8*>(at[k] + 8 * 8192) = v[k];
        }
    };
    auto issue_y = [&](int ks) {  // dY stage of K-step ks: padded rows [ps + KS ks, + KS), 64 of the channels; rows beyond the split are zeros
        const int lrow = lane >> 3, pc = lane & 7;
#pragma unroll
        for (int k = 0; k < G::YP; ++k) {
            const int q = ps + KS * ks + 64 * k + 8 * wave + lrow;
            const int chunk = (((pc >> 1) ^ ((q >> 1) & 3)) << 1) | (pc & 1);
            const int m = q < pe ? pixel_of(q) : -1;
            const uint32_t voff = m >= 0 ? static_cast<uint32_t>(m * p.ldy + chunk * 8) * 2u : OUT_OF_RANGE;
            buffer_load_lds_16(ybase, (lds_void_t*)(smem + G::X_BYTES + (ks & 3) * G::Y_STAGE + k * 8192 + wave * 1024), voff, 0);
        }
    };
    // K-step ks reads X blocks 2 ks .. 2 ks + 3 and dY stage ks; both are requested two K-steps ahead: REQ requests per
    // wave and K-step (two X blocks twice, the dY pieces), so "all but the last REQ" is what a K-step waits for.
    issue_x(0); issue_x(1); issue_x(2); issue_x(3); issue_x(4); issue_x(5);
    issue_y(0); issue_y(1);
    if constexpr (XF) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        for (int b2 = 0; b2 < 6; b2 += 2) xform_x2(b2);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    f32x4 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- transposed fragment reads (see wgrad_ring_body): lane (16 g + 4 q4 + pp) reads 4 columns of row 4 g + q4
    const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_void_t*)smem));
    const int krow = 4 * g + q4;  // row inside a 32-pixel slice (+ 16 for the second read of a fragment)
    uint32_t y_off[2];  // dY fragments of this wave's two blocks of 16 output channels, slice 0 of stage 0
#pragma unroll
    for (int i = 0; i < 2; ++i) y_off[i] = lds0 + G::X_BYTES + static_cast<uint32_t>(krow * 64 + (((2 * wr + i) ^ ((krow >> 1) & 3)) << 4) + 4 * pp) * 2u;
    // This wave's column blocks: block NJ wc + j = (tap, 16-channel unit).  The fragment of tap (dh, dw) is the fragment
    // of the rows shifted by dh (W + 2) + dw; a K-step starts at a multiple of 64 rows, so a lane's offset from the window's
    // first row - swizzle included - is the same in every K-step.
    uint32_t x_off[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int cb = NJ * wc + j, tap = cb / G::UNITS, u = cb % G::UNITS;
        const int r = krow + (tap / 3 - 1) * Wp + (tap % 3 - 1);  // relative to the K-step's first pixel; > -BR
        const int sw = C == 64 ? ((r >> 1) & 3) : (r & 7);
        x_off[j] = lds0 + static_cast<uint32_t>((BR + r) * RB + ((u ^ sw) << 5) + 8 * pp);
    }

    for (int ks = 0; ks < nsteps; ++ks) {
        if (ks > 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::REQ) : "memory");
            if constexpr (XF) {
                __builtin_amdgcn_sched_barrier(0);
                if (ks >= 2) {  // the blocks requested two K-steps ago have landed (blocks 0-5 were done in front of the loop)
                    xform_x2(2 * ks + 2);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const uint32_t xs = static_cast<uint32_t>(((blk0 + 2 * ks - 1) & 7) * 8192);  // the window's first row (block 2 ks)
        const uint32_t ys = static_cast<uint32_t>((ks & 3) * G::Y_STAGE);
        s16x4 ylo[KK][2], yhi[KK][2], xlo[3][GJ][KK], xhi[3][GJ][KK];  // dY: [slice][channel block]; X: [buffer][block in group][slice]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t a = y_off[i] + ys;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ylo[kk][i]) : "v"(a), "n"(kk * 4096) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(yhi[kk][i]) : "v"(a), "n"(kk * 4096 + 2048) : "memory");
            }
        }
        auto read_x = [&](const int gr, const int buf) {  // (constants after inlining): 8 reads
#pragma unroll
            for (int jj = 0; jj < GJ; ++jj) {
                const uint32_t a = x_off[gr * GJ + jj] + xs;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(xlo[buf][jj][kk]) : "v"(a), "n"(kk * 32 * RB) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(xhi[buf][jj][kk]) : "v"(a), "n"(kk * 32 * RB + 16 * RB) : "memory");
                }
            }
        };
        auto mma = [&](const int gr, const int buf) {
#pragma unroll
            for (int jj = 0; jj < GJ; ++jj)
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const s16x8 xv = {xlo[buf][jj][kk][0], xlo[buf][jj][kk][1], xlo[buf][jj][kk][2], xlo[buf][jj][kk][3],
                                      xhi[buf][jj][kk][0], xhi[buf][jj][kk][1], xhi[buf][jj][kk][2], xhi[buf][jj][kk][3]};
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const s16x8 yv = {ylo[kk][i][0], ylo[kk][i][1], ylo[kk][i][2], ylo[kk][i][3],
                                          yhi[kk][i][0], yhi[kk][i][1], yhi[kk][i][2], yhi[kk][i][3]};
                        acc[i][gr * GJ + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, yv), __builtin_bit_cast(bf16x8, xv),
                                                                                      acc[i][gr * GJ + jj], 0, 0, 0);
                    }
                }
        };
        // fragment reads return in order, 8 per group; a group is requested two groups ahead of its MFMAs (the counter
        // holds 15: waiting for "at most 15 outstanding" is one read stricter than needed).  The dY fragments (4 KK
        // reads) are in front of everything.
        read_x(0, 0);
        read_x(1, 1);
#pragma unroll
        for (int gr = 0; gr < NG; ++gr) {
            if (gr + 2 < NG) read_x(gr + 2, (gr + 2) % 3);
            if (gr + 2 < NG) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
            else if (gr + 1 < NG) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma(gr, gr % 3);
            __builtin_amdgcn_sched_barrier(0);
            if (gr == 0) {
                // This K-step's requests (for two K-steps ahead), behind the first MFMAs rather than right after the
                // barrier: their address arithmetic (a row's real pixel, four times) then runs under the matrix pipe
                // instead of in front of the K-step's first fragment read.
                issue_x(2 * ks + 6);
                issue_x(2 * ks + 7);
                issue_y(ks + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the loader's trailing requests

    // D[n][column]: lane holds output channels (lane >> 4) * 4 + e of its block, column lane & 15
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int cb = NJ * wc + j, tap = cb / G::UNITS;
        int woff = 0;
        for (int t = 0; t < 9; ++t)
            if (p.tap_dh[t] == tap / 3 - 1 && p.tap_dw[t] == tap % 3 - 1) woff = p.tap_woff[t];
        const int col = woff + (cb % G::UNITS) * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = nh * 64 + (2 * wr + i) * 16 + (lane >> 4) * 4 + e;
                __hip_atomic_fetch_add(p.dW + static_cast<size_t>(split) * p.split_stride + static_cast<size_t>(n) * p.ldw + col, acc[i][j][e],
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the 7x7 / stride 2 stem on 4-element pixels (packed rows), input streamed through LDS once.
//
//   dW[n][kh * 32 + e] = sum over output pixels of dY[pixel][n] * in[(2 oh + kh) row][8 ow + e]
// The reduction runs over output pixels in a padded numbering with 128 positions per output row (positions >= OB have a
// zero dY row), and a K-step is one output row: its seven input rows are wave-uniform, and lane (pixel r,
// 4 elements) of a transposed fragment read addresses "row byte + 16 r + ..." - one scalar per kernel row plus a
// per-lane constant.  The input sits in LDS as in conv_rows7_kernel (a circular 64 KB image of the byte range around
// the K-step), with its first 3 KB mirrored behind the end so that a fragment never wraps.
//   8 waves = 4 (16 output channels) x 2 (elements 0-15 / 16-31 of every kernel row): 7 accumulators, 28 MFMAs per K-step
//   per K-step and wave 4 DMA requests: one 1 KB piece of the input window (or a dummy), its mirror / duplicate, and the
//   dY rows three steps ahead (two pieces) - a fixed count, so "all but the last eight" retires what the step reads.
// The kernel is bound by reading dY (128 bytes per output pixel) from HBM.
// ------------------------------------------------------------------------------------------------
constexpr int WR_Y_OFF = 65536 + 3072;
constexpr int WR_TRASH_OFF = WR_Y_OFF + 4 * 16384;
constexpr int WR_LDS = WR_TRASH_OFF + 8 * 1024;
constexpr int WR_LAX = 5;  // the input window is requested this many K-steps ahead (3 are needed; the rest absorbs image boundaries)

__global__ __launch_bounds__(512) void wgrad_rows7_kernel(const WgradArgs p, const int total_steps, const int steps_per_wg,
                                                          const long long in_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr uint32_t OUT_OF_RANGE = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int split = blockIdx.x;
    const int ks0 = split * steps_per_wg;
    const int nsteps = min(steps_per_wg, total_steps - ks0);
    const long long RB = static_cast<long long>(p.Win) * p.cp * 2;
    // K-step s (launch-wide index) = output row s = img * OA + oh (128 positions); its first input row
    auto in_row = [&](int s) -> long long {
        const uint32_t line = static_cast<uint32_t>(min(s, total_steps - 1));
        const uint32_t img = fastdiv(line, p.div_ab);
        return static_cast<long long>(img) * p.Hin + static_cast<long long>(line - img * p.OA) * p.istr;
    };
    const long long org = (in_row(ks0) * RB) & ~1023ll;  // LDS position 0 holds input byte `org`
    // (8 rows: the positions 112-127 of the last kernel row's fragment read on into the row behind it - times a zero dY row, but
    // what they read must be data or zeros, never LDS that nothing has written)
    auto need_end = [&](int s) -> uint32_t { return static_cast<uint32_t>((in_row(min(s, ks0 + nsteps - 1)) + 8) * RB - org); };
    bf16* xbase = const_cast<bf16*>(p.X) + org / 2;
    const long long m_base = static_cast<long long>(ks0) * p.OB;
    bf16* ybase = const_cast<bf16*>(p.dY) + m_base * p.ldy;

    const int lrow = lane >> 3, pc = lane & 7;
    uint32_t ld_off = 0;  // first byte (from org) of the next group of 8 pieces
    auto issue_x = [&](int s) {  // two requests: the next piece of the window if step s needs it (else a dummy), and its mirror / duplicate
        const bool real = ld_off < need_end(s);  // (wave-uniform)
        const uint32_t off = ld_off + 1024u * wave;
        const uint32_t pos = off & 0xffffu;
        const uint32_t voff = (real && org + off + 16 * lane + 16 <= in_bytes) ? off + 16u * lane : OUT_OF_RANGE;
        char* dst = real ? smem + pos : smem + WR_TRASH_OFF + wave * 1024;
        buffer_load_lds_16(xbase, (lds_void_t*)dst, voff, 0);
        buffer_load_lds_16(xbase, (lds_void_t*)((real && pos < 3072u) ? dst + 65536 : dst), voff, 0);
        if (real) ld_off += 8192u;
    };
    auto issue_y = [&](int ks) {  // dY rows of local K-step ks: positions 0-127 of its output row (zeros beyond OB / the range), two requests
        const int s = ks0 + ks;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = 8 * wave + lrow;  // row inside the 64-row half (the swizzle's period divides 64)
            const int ow = 64 * k + r;
            const int chunk = (((pc >> 1) ^ ((r >> 1) & 3)) << 1) | (pc & 1);
            const bool ok = ks < nsteps && ow < p.OB;
            const long long m = static_cast<long long>(s) * p.OB + ow - m_base;
            const uint32_t voff = ok ? static_cast<uint32_t>(m * p.ldy + chunk * 8) * 2u : OUT_OF_RANGE;
            buffer_load_lds_16(ybase, (lds_void_t*)(smem + WR_Y_OFF + (ks & 3) * 16384 + k * 8192 + wave * 1024), voff, 0);
        }
    };
    for (int rep = 0; rep < 8; ++rep) issue_x(ks0 + WR_LAX);  // (at most 8 groups are real: the launcher checks the window's size)
    issue_y(0); issue_y(1); issue_y(2);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    f32x4 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
    const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_void_t*)smem));
    const int krow = 4 * g + q4;
    const uint32_t y_off = lds0 + WR_Y_OFF + static_cast<uint32_t>(krow * 64 + ((wr ^ ((krow >> 1) & 3)) << 4) + 4 * pp) * 2u;
    const uint32_t x_lane = lds0 + static_cast<uint32_t>(16 * krow + 32 * wc + 8 * pp);

    for (int ks = 0; ks < nsteps; ++ks) {
        // per K-step and wave: X, X', Y, Y' - the dY rows of this step were the last two requests of step ks - 3
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        const long long rb = in_row(ks0 + ks) * RB - org;  // position 0 of the output row in kernel row 0
        const uint32_t ys = static_cast<uint32_t>((ks & 3) * 16384);
        s16x4 ylo[4], yhi[4], xlo[7][4], xhi[7][4];  // [..][32-position slice]
        {
            const uint32_t a = y_off + ys;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ylo[0]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(yhi[0]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(ylo[1]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:6144" : "=v"(yhi[1]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(ylo[2]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:10240" : "=v"(yhi[2]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:12288" : "=v"(ylo[3]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:14336" : "=v"(yhi[3]) : "v"(a) : "memory");
        }
        auto read_x = [&](const int kh) {
            const uint32_t a = x_lane + (static_cast<uint32_t>(rb + kh * RB) & 0xffffu);
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xlo[kh][0]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:256" : "=v"(xhi[kh][0]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(xlo[kh][1]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:768" : "=v"(xhi[kh][1]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(xlo[kh][2]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1280" : "=v"(xhi[kh][2]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1536" : "=v"(xlo[kh][3]) : "v"(a) : "memory");
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1792" : "=v"(xhi[kh][3]) : "v"(a) : "memory");
        };
        auto mma = [&](const int kh) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const s16x8 yv = {ylo[kk][0], ylo[kk][1], ylo[kk][2], ylo[kk][3], yhi[kk][0], yhi[kk][1], yhi[kk][2], yhi[kk][3]};
                const s16x8 xv = {xlo[kh][kk][0], xlo[kh][kk][1], xlo[kh][kk][2], xlo[kh][kk][3],
                                  xhi[kh][kk][0], xhi[kh][kk][1], xhi[kh][kk][2], xhi[kh][kk][3]};
                acc[kh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, yv), __builtin_bit_cast(bf16x8, xv), acc[kh], 0, 0, 0);
            }
        };
        // reads are awaited in order, 8 per kernel row, two kernel rows behind the newest request (the counter holds 15)
        read_x(0);
        read_x(1);
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            if (kh + 2 < 7) read_x(kh + 2);
            if (kh + 2 < 7) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");      // 16 requests are younger than kernel row kh's
            else if (kh + 1 < 7) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma(kh);
            __builtin_amdgcn_sched_barrier(0);
            if (kh == 0) {  // this K-step's requests, behind the first MFMAs (their address arithmetic runs under the matrix pipe)
                issue_x(ks0 + ks + WR_LAX);
                issue_y(ks + 3);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // D[n][column]: lane holds output channels (lane >> 4) * 4 + e of its block, column lane & 15
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
        const int col = kh * 32 + 16 * wc + (lane & 15);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = wr * 16 + (lane >> 4) * 4 + e;
            __hip_atomic_fetch_add(p.dW + static_cast<size_t>(split) * p.split_stride + static_cast<size_t>(n) * p.ldw + col, acc[kh][e],
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// tile 19: packed-row weight gradient of a 7-row kernel with 32-element rows on 4-element pixels, stride 2, 64 output channels
static int launch_wgrad_rows7(WgradArgs a, int T, int splits, hipStream_t stream) {
    // (a.C: elements per pixel, a.tw: row width on entry)
    if (T != 7 || a.tw != 32 || a.cp != 4 || a.N != 64 || a.istr != 2 || a.OB > 128 || a.OB <= 64 || a.ldw < 7 * 32 || a.xf != nullptr ||
        a.dbias != nullptr || (a.M % (a.OA * a.OB)) != 0 || (a.Win * 8) % 16 != 0)
        return -1;
    const long long RB = static_cast<long long>(a.Win) * 8;
    if ((a.OA - 1) * 2 + 7 > a.Hin || 2096 > 2 * RB) return -1;  // (a fragment reaches 2096 bytes from the row's start: positions up to 127)
    const long long jump = a.Hin - 2 * (a.OA - 1);  // input rows from an image's last output row to the next image's first
    // window of a K-step (8 rows) + WR_LAX steps of lookahead + an image boundary + one group of over-fetch inside the
    // 64 KB image; the prologue's 8 groups cover the first window
    if ((8 + WR_LAX * 2 + jump) * RB + 1024 + 8192 > 65536) return -1;
    if ((2 + jump) * RB > (WR_LAX - 3 + 1) * 8192) return -1;  // one group per K-step catches up with a boundary within the lookahead's slack
    const long long images = a.M / (static_cast<long long>(a.OA) * a.OB);
    const long long total = images * a.OA;
    if (total >= (1ll << 30)) return -1;
    const int per = pdt_cdiv(total, splits);
    const int grid = pdt_cdiv(total, per);
    if ((static_cast<long long>(per) + 8) * (2 * RB > a.OB * a.ldy * 2ll ? 2 * RB : a.OB * a.ldy * 2ll) >= (1ll << 31)) return -1;
    const long long in_bytes = (images * a.Hin + 1) * RB;  // (+ the spare row behind the last image)
    a.div_ab = make_fastdiv(a.OA);
    static bool configured = false;
    if (!configured) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_rows7_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WR_LDS);
        if (e != hipSuccess) return static_cast<int>(e);
        configured = true;
    }
    hipLaunchKernelGGL(wgrad_rows7_kernel, dim3(grid), dim3(512), WR_LDS, stream, a, static_cast<int>(total), per, in_bytes);
    return pdt_launch_status();
}

// tile 18: the launch must be the 3x3 / stride 1 form with 64 -> 64 or 128 -> 128 channels; `splits` pixel ranges
template <int C>
static int launch_wgrad3x3_halo_c(WgradArgs a, int splits, hipStream_t stream) {
    using G = WgradHalo<C>;
    const int Wp = a.Win + 2, HpWp = (a.Hin + 2) * Wp;
    const long long P = static_cast<long long>(a.M / (a.Hin * a.Win)) * HpWp;
    if (P >= (1ll << 31) - 1024) return -1;
    int kps = pdt_cdiv(P, splits);
    kps = pdt_cdiv(kps, G::KS) * G::KS;
    const int grid = pdt_cdiv(P, kps);
    if ((static_cast<long long>(kps) + 2ll * HpWp + 1024) * (a.ldy > C ? a.ldy : C) * 2 >= (1ll << 31)) return -1;
    a.m_per_split = kps;
    a.div_ab = make_fastdiv(HpWp);
    a.div_b = make_fastdiv(Wp);
    if (a.xf != nullptr && C != 64) return -1;  // (the 128-channel flavour has no registers left for the transform)
    auto kernel = (C == 64 && a.xf != nullptr) ? wgrad3x3_halo_kernel<C, C == 64> : wgrad3x3_halo_kernel<C, false>;
    static bool configured[2] = {false, false};
    bool& conf = configured[a.xf != nullptr ? 1 : 0];
    if (!conf) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return static_cast<int>(e);
        conf = true;
    }
    hipLaunchKernelGGL(kernel, dim3(grid, C / 64), dim3(512), G::LDS, stream, a, static_cast<int>(P));
    return pdt_launch_status();
}

static int launch_wgrad3x3_halo(const WgradArgs& a, int splits, hipStream_t stream) {
    if (a.T != 9 || (a.C != 64 && a.C != 128) || a.N != a.C || a.istr != 1 || a.OA != a.Hin || a.OB != a.Win || a.Win < 1 || a.tw != 0 ||
        a.dbias != nullptr || (a.M % (a.Hin * a.Win)) != 0)
        return -1;
    if (a.Win + 3 > WgradHalo<64>::BR && a.C == 64) return -1;   // the halo W + 3 fits one block
    if (a.Win + 3 > WgradHalo<128>::BR && a.C == 128) return -1;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        const int dh = a.tap_dh[t], dw = a.tap_dw[t];
        if (dh < -1 || dh > 1 || dw < -1 || dw > 1 || a.tap_woff[t] < 0 || a.tap_woff[t] + a.C > a.ldw) return -1;
        seen |= 1u << ((dh + 1) * 3 + dw + 1);
    }
    if (seen != 0x1ffu) return -1;
    return a.C == 64 ? launch_wgrad3x3_halo_c<64>(a, splits, stream) : launch_wgrad3x3_halo_c<128>(a, splits, stream);
}

static bool wgrad_is_plain(const WgradArgs& a) {
    return a.T == 1 && a.tap_dh[0] == 0 && a.tap_dw[0] == 0 && a.istr == 1 && a.OA == a.Hin && a.OB == a.Win;
}

template <int BN, int BC, int WN, int WC, int NS, bool PP = false>
static int launch_wgrad_ring_dual(const WgradArgs& a0, int gy0, const WgradArgs& a1, int gy1, hipStream_t stream) {
    constexpr size_t lds = static_cast<size_t>(NS) * 64 * (BN + BC) * sizeof(bf16);
    auto kernel = wgrad_ring_dual_kernel<BN, BC, WN, WC, NS, PP>;
    static bool configured = false;
    if (!configured) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return static_cast<int>(e);
        configured = true;
    }
    const int gx0 = pdt_cdiv(a0.N, BN) * pdt_cdiv(a0.T * a0.C, BC), gx1 = pdt_cdiv(a1.N, BN) * pdt_cdiv(a1.T * a1.C, BC);
    hipLaunchKernelGGL(kernel, dim3(gx0 * gy0 + gx1 * gy1), dim3(WN * WC * 64), lds, stream, a0, a1, gx0, gx0 * gy0, gx1, gx1 * gy1);
    return pdt_launch_status();
}

// Launch grouping: between pdt_wgrad_group_begin() and pdt_wgrad_group_end() ring-kernel launches are recorded
// instead of issued; the end call issues pairs of compatible problems (same tile shape, both single-tap) as one
// dual launch and everything else one by one, in recording order.  The operands must stay valid until then.
struct WgradPending {
    WgradArgs a;
    int tile, gy;
};
static thread_local bool wgrad_grouping = false;
static thread_local int wgrad_npending = 0;
static thread_local WgradPending wgrad_pending[4];

static int wgrad_launch_one(const WgradArgs& a, int tile, int gy, hipStream_t s);

PDT_API void pdt_wgrad_group_begin() {
    wgrad_grouping = true;
    wgrad_npending = 0;
}

PDT_API int pdt_wgrad_group_end(void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    wgrad_grouping = false;
    int rc = 0, i = 0;
    while (i < wgrad_npending && rc == 0) {
        const WgradPending& u = wgrad_pending[i];
        if (i + 1 < wgrad_npending && wgrad_pending[i + 1].tile == u.tile && wgrad_is_plain(u.a) && wgrad_is_plain(wgrad_pending[i + 1].a)) {
            const WgradPending& v = wgrad_pending[i + 1];
            int r = -2;
            if (u.tile == 10) r = launch_wgrad_ring_dual<128, 128, 2, 4, 4>(u.a, u.gy, v.a, v.gy, s);
            else if (u.tile == 11) r = launch_wgrad_ring_dual<256, 128, 4, 2, 3>(u.a, u.gy, v.a, v.gy, s);
            else if (u.tile == 12) r = launch_wgrad_ring_dual<128, 256, 2, 4, 3>(u.a, u.gy, v.a, v.gy, s);
            else if (u.tile == 16) r = launch_wgrad_ring_dual<256, 256, 2, 4, 2>(u.a, u.gy, v.a, v.gy, s);
            else if (u.tile == 17) r = launch_wgrad_ring_dual<256, 256, 2, 4, 2, true>(u.a, u.gy, v.a, v.gy, s);
            if (r != -2) {
                rc = r;
                i += 2;
                continue;
            }
        }
        rc = wgrad_launch_one(u.a, u.tile, u.gy, s);
        ++i;
    }
    wgrad_npending = 0;
    return rc;
}

// dW must be zero (or hold a value to accumulate onto) before the call.
PDT_API int pdt_wgrad(const void* dY, const void* X, float* dW, int M, int N, int ldy, int C, int T, int ldw, int Hin,
                      int Win, int istr, int OA, int OB, const int* tap_dh, const int* tap_dw, const int* tap_woff,
                      int splits, int tile, int smallc, const void* zeros, float* dbias, const float* xf,
                      long long split_stride, void* stream) {
    // rows of dY are read in 8-column chunks: ldy must cover N rounded up to 8 (columns >= N are ignored)
    if (M <= 0 || N <= 0 || C <= 0 || T <= 0 || T > GG_MAX_TAPS || (ldy % 8) != 0 || ldy < (N + 7) / 8 * 8) return -1;
    if (!smallc && (C % 8) != 0) return -1;
    if (splits <= 0 || OA <= 0 || OB <= 0) return -1;
    WgradArgs a;
    a.dY = static_cast<const bf16*>(dY);
    a.X = static_cast<const bf16*>(X);
    a.dW = dW;
    a.zeros = static_cast<const bf16*>(zeros);
    a.dbias = dbias;
    a.xf = xf;
    a.split_stride = split_stride;
    if (split_stride != 0 && (wgrad_grouping || dbias != nullptr)) return -1;  // per-split slabs: single launches, no bias sums
    if (xf != nullptr && (tile < 10 || tile == 15 || smallc || wgrad_grouping)) return -1;  // ring kernels, single launches
    if (dbias != nullptr && (tile < 10 || tile == 15)) return -1;  // only the MODE 0 ring kernels sum the bias gradient
    a.M = M; a.N = N; a.ldy = ldy; a.C = C; a.T = T; a.ldw = ldw;
    a.Hin = Hin; a.Win = Win; a.istr = istr; a.OA = OA; a.OB = OB;
    a.div_ab = make_fastdiv(static_cast<uint32_t>(OA) * OB);
    a.div_b = make_fastdiv(OB);
    a.cp = 0;
    a.tw = 0;
    a.div_c = make_fastdiv(C);
    if (smallc) {
        // packed rows: `smallc` is the row width tw, C the elements per pixel on entry, T the number of kernel rows;
        // the kernel sees (kernel row, position in row) as its channel axis
        if ((smallc != 32 && smallc != 64) || (C != 4 && C != 8) || (tile >= 10 && tile != 15 && tile != 19)) return -1;
        a.cp = C;
        a.tw = smallc;
        if (ldw < T * smallc) return -1;
        if (tile == 15) {
            a.C = smallc;  // ring kernel: columns = T rows of tw
        } else {
            a.C = T * smallc;
        }
    } else if (tile == 15) {
        return -1;
    }
    int mps = pdt_cdiv(M, splits);
    mps = pdt_cdiv(mps, 64) * 64;
    a.m_per_split = mps;
    const int gy = pdt_cdiv(M, mps);
    for (int t = 0; t < T; ++t) {
        a.tap_dh[t] = tap_dh[t];
        a.tap_dw[t] = tap_dw[t];
        a.tap_woff[t] = tap_woff[t];
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // tile 17 = the 256x256 tile on the ping-pong schedule; it needs the single-tap form and no operand transform
    if (tile == 17 && (!wgrad_is_plain(a) || xf != nullptr)) tile = 16;
    if (wgrad_grouping && (tile == 10 || tile == 11 || tile == 12 || tile == 16 || tile == 17) && wgrad_npending < 4) {
        if (a.zeros == nullptr) return -1;
        wgrad_pending[wgrad_npending].a = a;
        wgrad_pending[wgrad_npending].tile = tile;
        wgrad_pending[wgrad_npending].gy = gy;
        ++wgrad_npending;
        return 0;
    }
    if (tile == 18) {
        if (smallc || wgrad_grouping) return -1;
        return launch_wgrad3x3_halo(a, splits, s);
    } else if (tile == 19) {
        if (!smallc || wgrad_grouping) return -1;
        return launch_wgrad_rows7(a, T, splits, s);
    } else if (tile == 15) {
        return launch_wgrad_ring<64, 256, 2, 4, 3, 1>(a, gy, s);
    } else if (smallc) {
        constexpr int BT = 64;
        const int gx = pdt_cdiv(N, BT) * pdt_cdiv(a.C, BT);
        const size_t lds = 4ull * 64 * (BT + 16) * sizeof(bf16);
        hipLaunchKernelGGL((wgrad_kernel<BT, true>), dim3(gx, gy), dim3(256), lds, s, a);
    } else if (tile == 0) {
        constexpr int BT = 128;
        const int gx = pdt_cdiv(N, BT) * pdt_cdiv(C, BT) * T;
        const size_t lds = 4ull * 64 * (BT + 16) * sizeof(bf16);
        hipLaunchKernelGGL((wgrad_kernel<BT, false>), dim3(gx, gy), dim3(256), lds, s, a);
    } else if (tile == 1) {
        constexpr int BT = 64;
        const int gx = pdt_cdiv(N, BT) * pdt_cdiv(C, BT) * T;
        const size_t lds = 4ull * 64 * (BT + 16) * sizeof(bf16);
        hipLaunchKernelGGL((wgrad_kernel<BT, false>), dim3(gx, gy), dim3(256), lds, s, a);
    } else if (tile == 10) {
        return launch_wgrad_ring<128, 128, 2, 4, 4>(a, gy, s);
    } else if (tile == 11) {
        return launch_wgrad_ring<256, 128, 4, 2, 3>(a, gy, s);
    } else if (tile == 12) {
        return launch_wgrad_ring<128, 256, 2, 4, 3>(a, gy, s);
    } else if (tile == 16) {
        return launch_wgrad_ring<256, 256, 2, 4, 2>(a, gy, s);
    } else if (tile == 17) {
        return launch_wgrad_ring_plain<256, 256, 2, 4, 2, 0, true, false, true>(a, gy, s);
    } else if (tile == 13) {
        return launch_wgrad_ring<64, 256, 2, 4, 3>(a, gy, s);
    } else if (tile == 14) {
        return launch_wgrad_ring<256, 64, 4, 2, 3>(a, gy, s);
    } else {
        return -1;
    }
    return pdt_launch_status();
}

static int wgrad_launch_one(const WgradArgs& a, int tile, int gy, hipStream_t s) {
    if (tile == 10) return launch_wgrad_ring<128, 128, 2, 4, 4>(a, gy, s);
    if (tile == 11) return launch_wgrad_ring<256, 128, 4, 2, 3>(a, gy, s);
    if (tile == 12) return launch_wgrad_ring<128, 256, 2, 4, 3>(a, gy, s);
    if (tile == 16) return launch_wgrad_ring<256, 256, 2, 4, 2>(a, gy, s);
    if (tile == 17) return launch_wgrad_ring_plain<256, 256, 2, 4, 2, 0, true, false, true>(a, gy, s);
    return -1;
}
