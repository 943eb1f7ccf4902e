// Fused multi-head attention (flash-style, head dim 64) on MFMA for packed QKV activations.
//
//   qkv : [B, S, 3, H, 64] bf16  (the output of the fused QKV projection, used in place)
//   out : [B, S, H, 64]    bf16
//   lse : [B, H, S]        fp32  log-sum-exp of the scaled scores (natural log), kept for backward
//
// The S x S score matrix is never written.  A workgroup (4 waves) owns 64 queries (forward, dQ) or
// 64 keys (dK/dV); each wave owns 16 of them and walks the other sequence in tiles of 64 staged in
// LDS.  The MFMA orientation is always chosen so that the reduction index of the *next* product sits
// on the accumulator registers of the current one: a 16x16 score tile leaves each lane with four
// values of one column, which - packed to bf16 - is exactly the B-operand fragment of the following
// MFMA once the other operand is read with the matching row permutation (transposing
// ds_read_b64_tr_b16).  Softmax statistics are therefore per-lane scalars and P / dS never touch LDS.
//
// Not exercised by the reference (CNN only, /root/reference/src/main.py:49); serves ViT-B/16 and GPT-2.
#include "common.h"
#include <cstdlib>

#define AT_D 64
#define AT_T 64           // tile length along the streamed sequence
#define AT_PITCH 80       // padded LDS pitch (elements) for tiles read with the transposing load
#define AT_NEG (-1.0e30f)

typedef __attribute__((address_space(3))) s16x4 at_lds_s16x4;

struct AttnArgs {
    const bf16* qkv;
    bf16* out;          // forward: O; backward kernels: unused
    float* lse;
    const bf16* dout;   // [B, S, H, 64]
    const bf16* o;      // [B, S, H, 64]
    float* delta;       // [B, H, S]
    bf16* dqkv;         // [B, S, 3, H, 64]
    int B, S, H;
    float scale_log2;   // softmax scale * log2(e)
    float scale;
    int causal;
};

__device__ __forceinline__ size_t qkv_off(const AttnArgs& p, int b, int s, int which, int h) {
    return (((static_cast<size_t>(b) * p.S + s) * 3 + which) * p.H + h) * AT_D;
}

// Staging of a 64 x 64 tile (row r from `src + r*stride` for r < valid, zeros otherwise) into LDS, either with
// XOR-swizzled 16-byte chunks (row-major fragment reads) or in a plain padded layout (transposed reads).
// The same staging split in two, so the global loads of tile t+1 are in flight while tile t is being
// computed: tile_load right after the barrier that publishes tile t, tile_store_* after the next one.
struct TileRegs {
    bf16x8 v[2];
};

__device__ __forceinline__ TileRegs tile_load(const bf16* src, size_t stride, int valid, int tid) {
    const int chunk = tid & 7, lrow = tid >> 3;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    TileRegs r;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = lrow + 32 * i;
        r.v[i] = row < valid ? *reinterpret_cast<const bf16x8*>(src + row * stride + chunk * 8) : zero8;
    }
    return r;
}

__device__ __forceinline__ void tile_store_swz(bf16* dst, const TileRegs& r, int tid) {
    const int chunk = tid & 7, lrow = tid >> 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = lrow + 32 * i;
        *reinterpret_cast<bf16x8*>(dst + row * AT_D + ((chunk ^ (row & 7)) << 3)) = r.v[i];
    }
}

__device__ __forceinline__ void tile_store_pad(bf16* dst, const TileRegs& r, int tid) {
    const int chunk = tid & 7, lrow = tid >> 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = lrow + 32 * i;
        *reinterpret_cast<bf16x8*>(dst + row * AT_PITCH + chunk * 8) = r.v[i];
    }
}

// row-major fragment: lane (g, i) gets tile[row0 + i][32*ks + 8g .. +7]
__device__ __forceinline__ bf16x8 frag_row(const bf16* tile, int row0, int ks, int lane) {
    const int row = row0 + (lane & 15), kc = ks * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(tile + row * AT_D + ((kc ^ (row & 7)) << 3));
}

// transposed fragment: lane (g, i) gets tile[k0 + kappa(g, j)][col0 + i], kappa(g, j) = 4g + j (j < 4), 16 + 4g + j - 4
__device__ __forceinline__ bf16x8 frag_tr(const bf16* tile, int k0, int col0, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const bf16* b0 = tile + (k0 + 4 * g + q) * AT_PITCH + col0 + 4 * pp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_s16x4*)(b0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_s16x4*)(b0 + 16 * AT_PITCH));
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

// two adjacent 16-row accumulator tiles -> bf16 B-operand fragment with the kappa row order
__device__ __forceinline__ bf16x8 pack_pair(const f32x4& a, const f32x4& b) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r[e] = f32_to_bf16(a[e]);
        r[e + 4] = f32_to_bf16(b[e]);
    }
    return r;
}

// 2^x by the bare hardware instruction: arguments here are <= ~8 and results below 2^-126 may flush to zero
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Reductions across the four 16-lane groups (same lane & 15) with the gfx950 row-swap instructions: after
// v_permlane16_swap (v_permlane32_swap) of two copies of v, one copy holds the even rows (lower half) in every
// row (half) and the other the odd rows (upper half), so one max / add finishes a butterfly step on the vector ALU;
// the generic shuffle goes through the LDS crossbar instead (ds_bpermute, an order of magnitude more latency).
__device__ __forceinline__ void swap_rows16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap_rows32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float group_max(float v) {
    float a = v, b = v;
    swap_rows16(a, b);
    a = b = fmaxf(a, b);
    swap_rows32(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float group_sum(float v) {
    float a = v, b = v;
    swap_rows16(a, b);
    a = b = a + b;
    swap_rows32(a, b);
    return a + b;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// QF query fragments of 16 per wave: a workgroup owns 64 * QF queries.  With QF = 2 every K / V fragment read
// from LDS feeds two MFMAs, and the barriers and the staging of a tile are amortised over twice the work.
template <int QF>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs p) {
    // K / V tiles are double-buffered in LDS: tile t + 1 is stored while tile t is being read, so one barrier
    // per tile suffices and the LDS stores are a whole iteration ahead of their first read.
    __shared__ __attribute__((aligned(16))) bf16 sKb[2][AT_T * AT_D];
    __shared__ __attribute__((aligned(16))) bf16 sVb[2][AT_T * AT_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int q0 = blockIdx.x * (64 * QF);
    int qrow[QF];  // this lane's queries
#pragma unroll
    for (int f = 0; f < QF; ++f) qrow[f] = q0 + (wave * QF + f) * 16 + i;
    const size_t srow = static_cast<size_t>(3) * p.H * AT_D;  // elements between consecutive sequence positions

    bf16x8 qf[QF][2];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const bf16* qp = p.qkv + qkv_off(p, b, min(qrow[f], p.S - 1), 0, h);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[f][ks] = qrow[f] < p.S ? *reinterpret_cast<const bf16x8*>(qp + ks * 32 + g * 8) : zero8;
    }
    // The softmax denominators are accumulated by the matrix unit as well: one more MFMA per 32 keys against a
    // fragment of ones leaves sum_k P[k][query] in every row of `l` (all four registers of every lane group hold
    // the query's total), which takes 16 additions and a cross-lane reduction per tile off the vector ALU.
    f32x4 o[QF][4], l[QF];
    float m[QF];
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = f32_to_bf16(1.f);
#pragma unroll
    for (int f = 0; f < QF; ++f) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[f][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        m[f] = AT_NEG;
        l[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int kend = p.causal ? min(p.S, q0 + 64 * QF) : p.S;
    // keys this wave needs under a causal mask: up to its own last query (wave-uniform); later tiles are skipped
    // (a wave whose first query lies beyond S has nothing to do at all)
    const int wave_kend = q0 + wave * QF * 16 >= p.S ? 0 : (p.causal ? min(p.S, q0 + (wave * QF + QF) * 16) : p.S);
    TileRegs rk = tile_load(p.qkv + qkv_off(p, b, 0, 1, h), srow, p.S, tid);
    TileRegs rv = tile_load(p.qkv + qkv_off(p, b, 0, 2, h), srow, p.S, tid);
    tile_store_swz(sKb[0], rk, tid);
    tile_store_pad(sVb[0], rv, tid);
    if (AT_T < kend) {
        rk = tile_load(p.qkv + qkv_off(p, b, AT_T, 1, h), srow, p.S - AT_T, tid);
        rv = tile_load(p.qkv + qkv_off(p, b, AT_T, 2, h), srow, p.S - AT_T, tid);
    }
    int buf = 0;
    for (int k0 = 0; k0 < kend; k0 += AT_T, buf ^= 1) {
        // tile k0 is complete in buffer `buf`, and every wave is done reading the other buffer (tile k0 - 64)
        __syncthreads();
        const bf16* sK = sKb[buf];
        const bf16* sV = sVb[buf];
        if (k0 + AT_T < kend) {  // registers hold tile k0 + 64 (loaded during the previous iteration): park it in LDS
            tile_store_swz(sKb[buf ^ 1], rk, tid);
            tile_store_pad(sVb[buf ^ 1], rv, tid);
            if (k0 + 2 * AT_T < kend) {  // and fetch the one after it, in flight during this tile's MFMAs
                rk = tile_load(p.qkv + qkv_off(p, b, k0 + 2 * AT_T, 1, h), srow, p.S - k0 - 2 * AT_T, tid);
                rv = tile_load(p.qkv + qkv_off(p, b, k0 + 2 * AT_T, 2, h), srow, p.S - k0 - 2 * AT_T, tid);
            }
        }
        if (k0 >= wave_kend) continue;  // every key of this tile is masked for every query of this wave
        f32x4 s[QF][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int f = 0; f < QF; ++f) s[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 kf = frag_row(sK, t * 16, ks, lane);
#pragma unroll
                for (int f = 0; f < QF; ++f) s[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[f][ks], s[f][t], 0, 0, 0);
            }
        }
        // Only the last key tile (keys beyond S) and, with a causal mask, the tiles that reach past the first
        // query of the workgroup need per-element masking; the branch is uniform for the workgroup.
        const bool masked = (k0 + AT_T > p.S) || (p.causal && k0 + AT_T > q0);
#pragma unroll
        for (int f = 0; f < QF; ++f) {
            float mx = AT_NEG;  // maximum of the raw scores; the softmax scale (> 0) is folded into the exponent's fma
            if (masked) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = k0 + t * 16 + g * 4 + e;
                        const bool ok = key < p.S && (!p.causal || key <= qrow[f]);
                        s[f][t][e] = ok ? s[f][t][e] : AT_NEG;
                        mx = fmaxf(mx, s[f][t][e]);
                    }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) mx = fmaxf(mx, s[f][t][e]);
            }
            mx = group_max(mx) * p.scale_log2;
            // The running maximum is raised only when a score exceeds it by more than 2^6: probabilities are then
            // bounded by 64 instead of 1 (harmless in fp32 / bf16) and the rescaling of the output accumulators,
            // which costs a round trip through the vector ALU, is skipped for most tiles.  Wave-uniform decision.
            const bool raise = __any(mx > m[f] + 6.0f);
            if (raise) {
                const float mn = fmaxf(m[f], mx);
                const float alpha = fast_exp2(m[f] - mn);
                l[f][0] *= alpha;
                m[f] = mn;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[f][dt][e] *= alpha;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    s[f][t][e] = fast_exp2(fmaf(s[f][t][e], p.scale_log2, -m[f]));  // masked scores are -1e30: 2^-huge = 0
        }
        bf16x8 pf[QF][2];
#pragma unroll
        for (int f = 0; f < QF; ++f)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                pf[f][kp] = pack_pair(s[f][2 * kp], s[f][2 * kp + 1]);
                l[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[f][kp], l[f], 0, 0, 0);
            }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                const bf16x8 vf = frag_tr(sV, kp * 32, dt * 16, lane);
#pragma unroll
                for (int f = 0; f < QF; ++f) o[f][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[f][kp], o[f][dt], 0, 0, 0);
            }
    }
#pragma unroll
    for (int f = 0; f < QF; ++f) {
        if (qrow[f] < p.S) {
            const float lt = l[f][0];
            const float inv = lt > 0.f ? 1.f / lt : 0.f;
            bf16* op = p.out + ((static_cast<size_t>(b) * p.S + qrow[f]) * p.H + h) * AT_D;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = f32_to_bf16(o[f][dt][e] * inv);
                *reinterpret_cast<bf16x4*>(op + dt * 16 + g * 4) = v;
            }
            if (g == 0)
                p.lse[(static_cast<size_t>(b) * p.H + h) * p.S + qrow[f]] = (m[f] + log2f(fmaxf(lt, 1e-30f))) * 0.6931471805599453f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// delta[b,h,s] = sum_d dO * O   (8 lanes x 16 bytes per (b, s, h) row of 64: a wave reads 8 rows = 1 KB per load)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_delta_kernel(const AttnArgs p) {
    const long long rows = static_cast<long long>(p.B) * p.S * p.H;
    const long long row = blockIdx.x * 32ll + (threadIdx.x >> 3);
    const int part = threadIdx.x & 7;
    float v = 0.f;
    if (row < rows) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(p.dout + row * AT_D + part * 8);
        const bf16x8 c = *reinterpret_cast<const bf16x8*>(p.o + row * AT_D + part * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v += bf16_to_f32(a[e]) * bf16_to_f32(c[e]);
    }
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    if (part == 0 && row < rows) {
        const int h = static_cast<int>(row % p.H);
        const long long bs = row / p.H;
        const int s = static_cast<int>(bs % p.S), b = static_cast<int>(bs / p.S);
        p.delta[(static_cast<size_t>(b) * p.H + h) * p.S + s] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// dK, dV: workgroup = 64 keys, wave = 16 keys (key on the lane), streams query tiles.
//   S[q][key] = Q K^T, dP[q][key] = dO V^T  (accumulators: q on registers, key on lanes)
//   dV^T[d][key] += dO^T[d][q] P[q][key],  dK^T[d][key] += Q^T[d][q] dS[q][key]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(16))) bf16 sQ[AT_T * AT_D];
    __shared__ __attribute__((aligned(16))) bf16 sQt[AT_T * AT_PITCH];
    __shared__ __attribute__((aligned(16))) bf16 sDO[AT_T * AT_D];
    __shared__ __attribute__((aligned(16))) bf16 sDOt[AT_T * AT_PITCH];
    __shared__ float sLse[AT_T], sDelta[AT_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int k0 = blockIdx.x * 64;
    const int key = k0 + wave * 16 + i;
    const size_t srow = static_cast<size_t>(3) * p.H * AT_D;
    const size_t orow = static_cast<size_t>(p.H) * AT_D;

    bf16x8 kf[2], vf[2];  // B operands: K^T / V^T fragments of this lane's key
    {
        const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const int kc = min(key, p.S - 1);
        const bf16* kp = p.qkv + qkv_off(p, b, kc, 1, h);
        const bf16* vp = p.qkv + qkv_off(p, b, kc, 2, h);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[ks] = key < p.S ? *reinterpret_cast<const bf16x8*>(kp + ks * 32 + g * 8) : zero8;
            vf[ks] = key < p.S ? *reinterpret_cast<const bf16x8*>(vp + ks * 32 + g * 8) : zero8;
        }
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int qstart = p.causal ? k0 : 0;  // queries before the first key of this tile never attend to it
    TileRegs rq, rdo;
    float r_lse = 0.f, r_delta = 0.f;
    auto prefetch = [&](int q0) {
        rq = tile_load(p.qkv + qkv_off(p, b, q0, 0, h), srow, p.S - q0, tid);
        rdo = tile_load(p.dout + ((static_cast<size_t>(b) * p.S + q0) * p.H + h) * AT_D, orow, p.S - q0, tid);
        if (tid < AT_T) {
            const int qq = q0 + tid;
            const size_t off = (static_cast<size_t>(b) * p.H + h) * p.S + min(qq, p.S - 1);
            r_lse = qq < p.S ? p.lse[off] * 1.4426950408889634f : 0.f;
            r_delta = qq < p.S ? p.delta[off] : 0.f;
        }
    };
    if (qstart < p.S) prefetch(qstart);
    for (int q0 = qstart; q0 < p.S; q0 += AT_T) {
        __syncthreads();  // every wave is done with the previous tile
        tile_store_swz(sQ, rq, tid);
        tile_store_pad(sQt, rq, tid);
        tile_store_swz(sDO, rdo, tid);
        tile_store_pad(sDOt, rdo, tid);
        if (tid < AT_T) {
            sLse[tid] = r_lse;
            sDelta[tid] = r_delta;
        }
        __syncthreads();
        if (q0 + AT_T < p.S) prefetch(q0 + AT_T);  // in flight during this tile's MFMAs
        // per-element masking only where it can matter: queries or keys beyond S, the causal diagonal (uniform)
        const bool masked = (q0 + AT_T > p.S) || (k0 + AT_T > p.S) || (p.causal && q0 < k0 + AT_T);
        f32x4 pt[4], ds[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(sQ, t * 16, ks, lane), kf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(sDO, t * 16, ks, lane), vf[ks], dp, 0, 0, 0);
            }
            if (masked) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ql = t * 16 + g * 4 + e, qq = q0 + ql;
                    const bool ok = qq < p.S && key < p.S && (!p.causal || key <= qq);
                    const float pv = ok ? fast_exp2(s[e] * p.scale_log2 - sLse[ql]) : 0.f;
                    pt[t][e] = pv;
                    ds[t][e] = pv * (dp[e] - sDelta[ql]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ql = t * 16 + g * 4 + e;
                    const float pv = fast_exp2(s[e] * p.scale_log2 - sLse[ql]);
                    pt[t][e] = pv;
                    ds[t][e] = pv * (dp[e] - sDelta[ql]);
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp) {
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(sDOt, kp * 32, dt * 16, lane),
                                                                 pack_pair(pt[2 * kp], pt[2 * kp + 1]), dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(sQt, kp * 32, dt * 16, lane),
                                                                 pack_pair(ds[2 * kp], ds[2 * kp + 1]), dk[dt], 0, 0, 0);
            }
    }
    if (key < p.S) {
        bf16* dkp = p.dqkv + qkv_off(p, b, key, 1, h);
        bf16* dvp = p.dqkv + qkv_off(p, b, key, 2, h);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x4 a, c;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = f32_to_bf16(dk[dt][e] * p.scale);
                c[e] = f32_to_bf16(dv[dt][e]);
            }
            *reinterpret_cast<bf16x4*>(dkp + dt * 16 + g * 4) = a;
            *reinterpret_cast<bf16x4*>(dvp + dt * 16 + g * 4) = c;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dQ: workgroup = 64 queries, wave = 16 queries (query on the lane), streams key tiles.
//   S^T[key][q] = K Q^T, dP^T[key][q] = V dO^T;  dQ^T[d][q] += K^T[d][key] dS^T[key][q]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(16))) bf16 sK[AT_T * AT_D];
    __shared__ __attribute__((aligned(16))) bf16 sKt[AT_T * AT_PITCH];
    __shared__ __attribute__((aligned(16))) bf16 sV[AT_T * AT_D];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int q0 = blockIdx.x * 64;
    const int qrow = q0 + wave * 16 + i;
    const size_t srow = static_cast<size_t>(3) * p.H * AT_D;

    bf16x8 qf[2], dof[2];
    float lse2 = 0.f, delta = 0.f;
    {
        const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const int qc = min(qrow, p.S - 1);
        const bf16* qp = p.qkv + qkv_off(p, b, qc, 0, h);
        const bf16* dop = p.dout + ((static_cast<size_t>(b) * p.S + qc) * p.H + h) * AT_D;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = qrow < p.S ? *reinterpret_cast<const bf16x8*>(qp + ks * 32 + g * 8) : zero8;
            dof[ks] = qrow < p.S ? *reinterpret_cast<const bf16x8*>(dop + ks * 32 + g * 8) : zero8;
        }
        if (qrow < p.S) {
            const size_t off = (static_cast<size_t>(b) * p.H + h) * p.S + qrow;
            lse2 = p.lse[off] * 1.4426950408889634f;
            delta = p.delta[off];
        }
    }
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int kend = p.causal ? min(p.S, q0 + 64) : p.S;
    TileRegs rk = tile_load(p.qkv + qkv_off(p, b, 0, 1, h), srow, p.S, tid);
    TileRegs rv = tile_load(p.qkv + qkv_off(p, b, 0, 2, h), srow, p.S, tid);
    for (int k0 = 0; k0 < kend; k0 += AT_T) {
        __syncthreads();  // every wave is done with the previous tile
        tile_store_swz(sK, rk, tid);
        tile_store_pad(sKt, rk, tid);
        tile_store_swz(sV, rv, tid);
        __syncthreads();
        if (k0 + AT_T < kend) {  // next tile: in flight during this tile's MFMAs
            rk = tile_load(p.qkv + qkv_off(p, b, k0 + AT_T, 1, h), srow, p.S - k0 - AT_T, tid);
            rv = tile_load(p.qkv + qkv_off(p, b, k0 + AT_T, 2, h), srow, p.S - k0 - AT_T, tid);
        }
        const bool masked = (k0 + AT_T > p.S) || (q0 + 64 > p.S) || (p.causal && k0 + AT_T > q0);
        f32x4 ds[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(sK, t * 16, ks, lane), qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row(sV, t * 16, ks, lane), dof[ks], dp, 0, 0, 0);
            }
            if (masked) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = k0 + t * 16 + g * 4 + e;
                    const bool ok = key < p.S && qrow < p.S && (!p.causal || key <= qrow);
                    const float pv = ok ? fast_exp2(s[e] * p.scale_log2 - lse2) : 0.f;
                    ds[t][e] = pv * (dp[e] - delta);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) ds[t][e] = fast_exp2(s[e] * p.scale_log2 - lse2) * (dp[e] - delta);
            }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int kp = 0; kp < 2; ++kp)
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr(sKt, kp * 32, dt * 16, lane),
                                                                 pack_pair(ds[2 * kp], ds[2 * kp + 1]), dq[dt], 0, 0, 0);
    }
    if (qrow < p.S) {
        bf16* dqp = p.dqkv + qkv_off(p, b, qrow, 0, h);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = f32_to_bf16(dq[dt][e] * p.scale);
            *reinterpret_cast<bf16x4*>(dqp + dt * 16 + g * 4) = a;
        }
    }
}

static int attn_check(int B, int S, int H, int D) {
    if (B <= 0 || S <= 0 || H <= 0 || D != AT_D) return -1;
    if (static_cast<long long>(B) * H > 65535ll) return -1;  // grid.y limit
    if (static_cast<long long>(B) * S * 3 * H * D >= (1ll << 40)) return -1;
    return 0;
}

PDT_API int pdt_attention_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int D, float scale, int causal,
                              void* stream) {
    if (attn_check(B, S, H, D) != 0) return -1;
    AttnArgs a = {};
    a.qkv = static_cast<const bf16*>(qkv);
    a.out = static_cast<bf16*>(out);
    a.lse = lse;
    a.B = B; a.S = S; a.H = H;
    a.scale = scale;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.causal = causal;
    // 32 queries per wave pay for short non-causal sequences (ViT: +12 %); on long or causal ones the higher
    // register footprint costs more occupancy than the halved LDS traffic gains (measured: bench/attn_bench.py)
    static int qf_env = -1;
    if (qf_env < 0) {
        const char* e = getenv("PDT_ATTN_QF");
        qf_env = e ? atoi(e) : 0;
    }
    const int qf = qf_env ? qf_env : ((S > 64 && S <= 512 && !causal) ? 2 : 1);
    if (qf == 2) hipLaunchKernelGGL(attn_fwd_kernel<2>, dim3(pdt_cdiv(S, 128), B * H), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(attn_fwd_kernel<1>, dim3(pdt_cdiv(S, 64), B * H), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return pdt_launch_status();
}

PDT_API int pdt_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv,
                              int B, int S, int H, int D, float scale, int causal, void* stream) {
    if (attn_check(B, S, H, D) != 0) return -1;
    AttnArgs a = {};
    a.qkv = static_cast<const bf16*>(qkv);
    a.o = static_cast<const bf16*>(o);
    a.dout = static_cast<const bf16*>(dout);
    a.lse = const_cast<float*>(lse);
    a.delta = delta;
    a.dqkv = static_cast<bf16*>(dqkv);
    a.B = B; a.S = S; a.H = H;
    a.scale = scale;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.causal = causal;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long rows = static_cast<long long>(B) * S * H;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(pdt_cdiv(rows, 32)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(pdt_cdiv(S, 64), B * H), dim3(256), 0, s, a);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(pdt_cdiv(S, 64), B * H), dim3(256), 0, s, a);
    return pdt_launch_status();
}
