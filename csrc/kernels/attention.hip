// Fused multi-head attention (flash-style, head dim 64) on MFMA for packed QKV activations.
//
//   qkv  : [B, S, 3, H, 64] bf16  (the output of the fused QKV projection, used in place)
//   out  : [B, S, H, 64]    bf16
//   lse  : [B, H, S]        fp32  log2-sum-exp2 of the scaled scores (log2 units), kept for backward
//   dqkv : [B, S, 3, H, 64] bf16  (produced packed, so autograd never splits or concatenates)
//
// The S x S score matrix is never written.  All three kernels share one structure:
//   * a workgroup (4 waves) owns 128 rows of one sequence (queries in forward and dQ, keys in dK/dV), a wave 32 of
//     them with the row on the MFMA lane (v_mfma_f32_32x32x16_bf16), and streams the other sequence in tiles of 64;
//   * the MFMA orientation is chosen so that the reduction index of the *next* product sits on the accumulator
//     registers of the current one: a 32x32 score tile leaves each lane with 16 values of one column, which - packed
//     to bf16 - are the B-operand fragments of the following MFMAs once the other operand is read with the matching
//     row permutation (transposing ds_read_b64_tr_b16).  P / dS never touch LDS, and whatever is constant per owned
//     row (running maximum, row sum, log-sum-exp, delta) is a per-lane scalar;
//   * tiles arrive by LDS-DMA (global_load_lds_dwordx4, no staging registers) in a ring of three, two tiles ahead of
//     the MFMAs, with one counted wait and one barrier per tile; one LDS image serves row reads and transposed reads;
//   * the grid is one-dimensional and XCD-aware: consecutive workgroup ids go to the eight XCDs in turn, so the id is
//     taken apart as (id % 8, id / 8) and all blocks of one (batch, head) land on one XCD, whose L2 then serves the
//     tiles they share; under a causal mask the blocks with the most tiles are launched first.
//
// Not exercised by the reference (CNN only, /root/reference/src/main.py:49); serves ViT-B/16 and GPT-2.
#include "common.h"
#include <cstdlib>

#define AT_D 64
#define AT_T 64           // tile length along the streamed sequence
#define AT_BR 128         // rows of the owned sequence per workgroup
#define AT_RING 3         // tiles in LDS per streamed operand
#define AT_TILE (AT_T * AT_D)
#define AT_NEG (-1.0e30f)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void at_lds_void;
typedef const __attribute__((address_space(1))) void at_global_cvoid;

struct AttnArgs {
    const bf16* qkv;
    bf16* out;          // forward: O; backward kernels: unused
    float* lse;         // [B, H, S], log2 units
    const bf16* dout;   // [B, S, H, 64]
    const bf16* o;      // [B, S, H, 64]
    float* delta;       // [B, H, S]: MINUS sum_d dO * O, written by the dQ kernel, read by the dK/dV kernel
    bf16* dqkv;         // [B, S, 3, H, 64]
    int B, S, H;
    float scale_log2;   // softmax scale * log2(e)
    float scale;
    int causal;
    int nblk;           // blocks of AT_BR rows per sequence
};

__device__ __forceinline__ size_t qkv_off(const AttnArgs& p, int b, int s, int which, int h) {
    return (((static_cast<size_t>(b) * p.S + s) * 3 + which) * p.H + h) * AT_D;
}

// 2^x by the bare hardware instruction: arguments here are bounded above and results below 2^-126 may flush to zero
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Exchange between lane l and lane l ^ 32 (the two lanes that share a row of a 32x32 accumulator tile): after
// v_permlane32_swap of two copies of v, one copy holds the lower half's values in both halves and the other the
// upper half's, so one max / add finishes the reduction on the vector ALU.
__device__ __forceinline__ void swap_rows32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float other_half_max(float v) {
    float a = v, b = v;
    swap_rows32(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float other_half_sum(float v) {
    float a = v, b = v;
    swap_rows32(a, b);
    return a + b;
}

// ------------------------------------------------------------------------------------------------
// LDS image of a 64 x 64 bf16 tile: [row][128 B], the 16-byte slots of row r XOR-ed with
//   swz(r) = 4 * ((r >> 1) & 1) + ((r >> 2) & 3).
// Row reads (ds_read_b128, 32 rows per instruction, serviced in the lane groups {0-3, 12-15, 20-27} and
// {4-11, 16-19, 28-31} of each half wave): the 8 rows of equal parity in a group get 8 different slots.
// Transposed reads (ds_read_b64_tr_b16, a half wave reads a 4-row x 32-column block): rows r and r + 2 differ in
// bit 2 of the slot, rows r and r + 1 in the 128-byte parity, so the block covers all 64 banks once.
// The image of an LDS-DMA piece is lane-linear (lane -> row lane / 8, slot lane % 8), so the permutation is applied
// to the source address: slot s of row r receives chunk s ^ swz(r).  Rows past the end of the sequence repeat the
// last row: their scores are masked, and the data stays finite.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t at_swz(uint32_t row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

struct TileDma {
    uint32_t row[2];  // row of the tile this lane fetches in each of its wave's two pieces (8 pieces of 8 rows per tile)
    uint32_t off[2];  // element offset of the lane's source chunk inside the row
};

__device__ __forceinline__ TileDma tile_dma_setup(int wave, int lane) {
    TileDma d;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t row = (2 * wave + j) * 8 + (lane >> 3), slot = lane & 7;
        d.row[j] = row;
        d.off[j] = (slot ^ at_swz(row)) << 3;
    }
    return d;
}

// rows [r0, r0 + 64) of a [*, row_stride] matrix (clamped to last_row) into one tile; two LDS-DMA instructions per wave
__device__ __forceinline__ void tile_dma_issue(const TileDma& d, const bf16* base, uint32_t row_stride, int r0, int last_row, bf16* tile,
                                               int wave) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t off = static_cast<uint32_t>(min(r0 + static_cast<int>(d.row[j]), last_row)) * row_stride;
        __builtin_amdgcn_global_load_lds((at_global_cvoid*)(base + off + d.off[j]), (at_lds_void*)(tile + (2 * wave + j) * 512), 16, 0, 0);
    }
}

// Per-lane addressing of the fragment reads (LDS byte offsets inside a tile).
struct FragAddr {
    uint32_t row[4];    // row-major A fragment of k-step ks, row lane % 32 (+ 4096 per 32-row tile)
    uint32_t tr[2][2];  // transposed A fragment: [column tile dt][4-row block blk] (+ 2048 per 16 rows)
};

__device__ __forceinline__ FragAddr frag_addr_setup(int lane) {
    FragAddr f;
    const int hi = lane >> 5, lq = lane & 31, l16 = lane & 15, half = (lane >> 4) & 1, piece = l16 & 3;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f.row[ks] = lq * 128 + (((2 * ks + hi) ^ at_swz(lq)) << 4);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const uint32_t row = 4 * hi + (l16 >> 2) + 8 * blk;
            const uint32_t slot = (4 * dt + 2 * half + (piece >> 1)) ^ at_swz(row);
            f.tr[dt][blk] = row * 128 + (slot << 4) + (piece & 1) * 8;
        }
    return f;
}

// Every LDS read of a kernel that has LDS-DMA in flight is inline assembly: the compiler cannot tell an ordinary LDS
// load from the pending DMA writes and would drain the whole DMA queue (vmcnt(0)) in front of it, which takes away
// the two tiles of prefetch.  Completion is therefore awaited by hand: LDS reads return in order, so a counted
// lgkmcnt(N) releases all but the youngest N; the scheduling fence keeps the consumers behind the wait.
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int N>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// A[i][k] = tile[32 rt + i][16 ks + k], ks = 0..3: rows of the tile as four 32 x 16 A operands (4 reads)
struct RowFrags {
    u32x4 v[4];
};
__device__ __forceinline__ void frag_rows_issue(RowFrags& f, uint32_t tile_lds, const FragAddr& fa, int rt) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.v[ks]) : "v"(tile_lds + fa.row[ks]), "n"(4096 * rt) : "memory");
}
__device__ __forceinline__ bf16x8 frag_rows_get(const RowFrags& f, int ks) { return __builtin_bit_cast(bf16x8, f.v[ks]); }

// four consecutive floats of a per-row vector in LDS (the same address in every lane of a half wave: a broadcast)
__device__ __forceinline__ void lds_read_f32x4(f32x4& v, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory"); }

// Transposed A fragments of column tile dt for the two 16-row steps j of row tile rt (4 reads):
// A[i][k] = tile[32 rt + 16 j + row(k)][32 dt + i], where row(k) follows the accumulator row order of the 32x32
// MFMA, so that registers 8 j .. 8 j + 7 of a packed accumulator tile are the matching B operand.
struct TrFrags {
    s16x4 lo[2], hi[2];
};
__device__ __forceinline__ void frag_tr_issue(TrFrags& v, uint32_t tile_lds, const FragAddr& fa, int dt, int rt) {
    const uint32_t a0 = tile_lds + fa.tr[dt][0], a1 = tile_lds + fa.tr[dt][1];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v.lo[j]) : "v"(a0), "n"((32 * rt + 16 * j) * 128) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v.hi[j]) : "v"(a1), "n"((32 * rt + 16 * j) * 128) : "memory");
    }
}
__device__ __forceinline__ bf16x8 frag_tr_get(const TrFrags& v, int j) {
    const s16x8 r = {v.lo[j][0], v.lo[j][1], v.lo[j][2], v.lo[j][3], v.hi[j][0], v.hi[j][1], v.hi[j][2], v.hi[j][3]};
    return __builtin_bit_cast(bf16x8, r);
}

// registers 8 j .. 8 j + 7 of an accumulator tile as the B operand of the next product
__device__ __forceinline__ bf16x8 pack_half(const f32x16& a, int j) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = f32_to_bf16(a[8 * j + e]);
    return r;
}

__device__ __forceinline__ uint32_t lds_addr(const void* p) { return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((at_lds_void*)p)); }

// acc^T[column][row] (two 32-column tiles, the row on the lane) times `mult` -> bf16 rows of 64 in a wave-private LDS
// area (16-byte slots XOR-ed by row & 7), then whole 128-byte rows to global memory: dst(row) is the row's address.
template <typename DstFn>
__device__ __forceinline__ void store_rows32(bf16* so, const f32x16* acc, float mult, int lane, int rows_valid, DstFn dst) {
    const int hi = lane >> 5, lq = lane & 31;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = f32_to_bf16(acc[dt][4 * r4 + e] * mult);
            *reinterpret_cast<bf16x4*>(so + lq * AT_D + (((4 * dt + r4) ^ (lq & 7)) << 3) + 4 * hi) = v;
        }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int row = pass * 8 + (lane >> 3), slot = lane & 7;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(so + row * AT_D + ((slot ^ (row & 7)) << 3));
        if (row < rows_valid) *reinterpret_cast<bf16x8*>(dst(row) + slot * 8) = v;
    }
}

// (block, batch * head) of a workgroup; false when the id is padding (batch * head rounded up to whole XCD rounds)
__device__ __forceinline__ bool at_block_of(const AttnArgs& p, int& blk, int& b, int& h) {
    const int xcd = blockIdx.x & 7, wid = blockIdx.x >> 3;
    const int bh = (wid / p.nblk) * 8 + xcd;
    if (bh >= p.B * p.H) return false;
    blk = wid % p.nblk;
    b = bh / p.H;
    h = bh % p.H;
    return true;
}

// ------------------------------------------------------------------------------------------------
// forward:  S^T[key][q] = K . Q^T  (A = K rows, B = Q fragments in registers);  O^T[d][q] += V^T . P^T
// A tile takes one fused multiply-add (softmax scale and running maximum), one exponential, one addition and half
// a conversion per score; there is NO maximum in the common path: the maximum is taken on the first tile only, and
// a later tile whose partial row sum exceeds 2^30 (a score ~30 octaves above the running maximum) is recomputed on
// the careful path, which raises the maximum and rescales O.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(1024))) bf16 sKb[AT_RING][AT_TILE];
    __shared__ __attribute__((aligned(1024))) bf16 sVb[AT_RING][AT_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lq = lane & 31;
    int qb, b, h;
    if (!at_block_of(p, qb, b, h)) return;
    if (p.causal) qb = p.nblk - 1 - qb;  // the last query blocks carry the most key tiles
    const int q0 = qb * AT_BR, qw0 = q0 + 32 * wave, qrow = qw0 + lq;
    const uint32_t srow = static_cast<uint32_t>(3 * p.H * AT_D);

    // number of key tiles of the workgroup and of this wave (a wave without a valid query computes nothing)
    const int nt = p.causal ? min(p.S - 1, q0 + AT_BR - 1) / AT_T + 1 : (p.S + AT_T - 1) / AT_T;
    const int wave_nt = qw0 >= p.S ? 0 : (p.causal ? min(p.S - 1, qw0 + 31) / AT_T + 1 : nt);
    const bf16* kbase = p.qkv + qkv_off(p, b, 0, 1, h);
    const bf16* vbase = p.qkv + qkv_off(p, b, 0, 2, h);
    const TileDma dma = tile_dma_setup(wave, lane);
    const FragAddr fa = frag_addr_setup(lane);

    bf16x8 qf[4];  // B operand of S^T: Q[q][16 ks + 8 hi .. + 7]
    {
        const bf16* qp = p.qkv + qkv_off(p, b, min(qrow, p.S - 1), 0, h) + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
        if (t < nt) {
            tile_dma_issue(dma, kbase, srow, t * AT_T, p.S - 1, sKb[t], wave);
            tile_dma_issue(dma, vbase, srow, t * AT_T, p.S - 1, sVb[t], wave);
        }

    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) o[0][e] = o[1][e] = 0.f;
    float lsum = 0.f;  // this lane's share (32 of every 64 keys) of the query's row sum
    float nm = 0.f;    // minus the running maximum, in log2 units
    const int key_limit = p.causal ? min(qrow, p.S - 1) : p.S - 1;  // last key this lane's query attends to
    const uint32_t sk_lds = lds_addr(&sKb[0][0]), sv_lds = lds_addr(&sVb[0][0]);

    int buf = 0;
    // Start of tile t: this wave's pieces of the tile have landed (those of tile t + 1 may still be in flight); after
    // the barrier everybody's have, and nobody reads tile t - 1 any more: its buffer takes tile t + 2.
    auto next_tile = [&](int t) {
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + 2 < nt) {
            const int nb = buf == 0 ? AT_RING - 1 : buf - 1;
            tile_dma_issue(dma, kbase, srow, (t + 2) * AT_T, p.S - 1, sKb[nb], wave);
            tile_dma_issue(dma, vbase, srow, (t + 2) * AT_T, p.S - 1, sVb[nb], wave);
        }
        const int cur = buf;
        buf = buf == AT_RING - 1 ? 0 : buf + 1;
        return cur;
    };
    for (int t = 0; t < wave_nt; ++t) {
        const int k0 = t * AT_T;
        const int cur = next_tile(t);
        const uint32_t sk_tile = sk_lds + static_cast<uint32_t>(cur) * (AT_TILE * 2), sv_tile = sv_lds + static_cast<uint32_t>(cur) * (AT_TILE * 2);
        const bool masked = (k0 + AT_T > p.S) || (p.causal && k0 + AT_T - 1 > qw0);
        const int rel = key_limit - k0 - 4 * hi;  // last valid key offset inside the tile, minus this half wave's share
        // The tile is handled as two halves of 32 keys, software-pipelined so that the matrix pipe always has work
        // queued while the vector ALU turns scores into probabilities:
        //   S(0) S(1) | softmax(0) | PV(0) | softmax(1) | PV(1)
        f32x16 acc0, acc1;
        auto scores = [&](f32x16& acc, const RowFrags& kf) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_get(kf, ks), qf[ks], acc, 0, 0, 0);
        };
        // scores of a half (keys k0 + koff ...) -> probabilities relative to the running maximum, in place; returns the
        // lane's partial row sum
        auto probs = [&](f32x16& acc, const int koff, const bool careful, const bool first) {
            if (masked) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = (koff + (r & 3) + 8 * (r >> 2)) <= rel ? acc[r] : AT_NEG;
            }
            if (careful) {
                float mx = AT_NEG;
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[r]);
                mx = other_half_max(mx) * p.scale_log2;
                // very first half: the maximum becomes the reference whatever its sign; later: only ever raised
                const float cand = mx < -1.0e29f ? -nm : mx;
                const float m_new = first ? cand : fmaxf(cand, -nm);
                if (!first) {
                    const float alpha = fast_exp2(-nm - m_new);
                    lsum *= alpha;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        o[0][e] *= alpha;
                        o[1][e] *= alpha;
                    }
                }
                nm = -m_new;
            }
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[r] = fast_exp2(fmaf(acc[r], p.scale_log2, nm));  // masked scores are -1e30: 2^-huge = 0
                sum += acc[r];
            }
            return sum;
        };
        // common path: no maximum; a half whose partial row sum is beyond 2^30 (or not a number) is recomputed carefully
        auto softmax_half = [&](f32x16& acc, const int kt) {
            const bool first = t == 0 && kt == 0;
            float ls = probs(acc, 32 * kt, first, first);
            if (!first && __builtin_expect(__any(!(ls <= 1073741824.f)), 0)) {
                RowFrags kf;
                frag_rows_issue(kf, sk_tile, fa, kt);
                lds_wait<0>();
                scores(acc, kf);
                ls = probs(acc, 32 * kt, true, false);
            }
            lsum += ls;
        };
        TrFrags vf0[2], vf1[2];  // V^T fragments of the two key halves, per column tile
        {
            RowFrags kf0, kf1;
            frag_rows_issue(kf0, sk_tile, fa, 0);
            frag_rows_issue(kf1, sk_tile, fa, 1);
            lds_wait<4>();
            scores(acc0, kf0);
            lds_wait<0>();
            scores(acc1, kf1);
        }
        frag_tr_issue(vf0[0], sv_tile, fa, 0, 0);
        frag_tr_issue(vf0[1], sv_tile, fa, 1, 0);
        softmax_half(acc0, 0);
        {
            const bf16x8 pb0 = pack_half(acc0, 0), pb1 = pack_half(acc0, 1);
            frag_tr_issue(vf1[0], sv_tile, fa, 0, 1);
            frag_tr_issue(vf1[1], sv_tile, fa, 1, 1);
            lds_wait<8>();
            o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_get(vf0[0], 0), pb0, o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_get(vf0[1], 0), pb0, o[1], 0, 0, 0);
            o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_get(vf0[0], 1), pb1, o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_get(vf0[1], 1), pb1, o[1], 0, 0, 0);
        }
        softmax_half(acc1, 1);
        {
            const bf16x8 pb0 = pack_half(acc1, 0), pb1 = pack_half(acc1, 1);
            lds_wait<0>();
            o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_get(vf1[0], 0), pb0, o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_get(vf1[1], 0), pb0, o[1], 0, 0, 0);
            o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_get(vf1[0], 1), pb1, o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_get(vf1[1], 1), pb1, o[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int t = wave_nt; t < nt; ++t) next_tile(t);  // tiles only the later waves of the workgroup need
    // every wave is done with the K / V tiles: the first buffers become the staging area of the output rows
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    const float lt = other_half_sum(lsum);
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    if (hi == 0 && qrow < p.S) p.lse[(static_cast<size_t>(b) * p.H + h) * p.S + qrow] = log2f(fmaxf(lt, 1e-30f)) - nm;
    bf16* orow0 = p.out + ((static_cast<size_t>(b) * p.S + qw0) * p.H + h) * AT_D;
    const size_t ostride = static_cast<size_t>(p.H) * AT_D;
    store_rows32(&sKb[0][0] + wave * (32 * AT_D), o, inv, lane, p.S - qw0, [&](int row) { return orow0 + row * ostride; });
}

// ------------------------------------------------------------------------------------------------
// dQ (and delta): workgroup = 128 queries, the query on the lane, streams K / V tiles.
//   S^T[key][q] = K . Q^T,  dP^T[key][q] = V . dO^T - delta[q]  (delta goes in as the initial accumulator: 16 equal
//   registers per lane, set once),  dS^T = P^T * dP^T,  dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
// delta[q] = sum_d dO[q][d] O[q][d] is formed from the dO fragments the wave holds anyway and written (negated) for
// the dK/dV kernel, which runs after this one.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(1024))) bf16 sKb[AT_RING][AT_TILE];
    __shared__ __attribute__((aligned(1024))) bf16 sVb[AT_RING][AT_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lq = lane & 31;
    int qb, b, h;
    if (!at_block_of(p, qb, b, h)) return;
    if (p.causal) qb = p.nblk - 1 - qb;
    const int q0 = qb * AT_BR, qw0 = q0 + 32 * wave, qrow = qw0 + lq;
    const uint32_t srow = static_cast<uint32_t>(3 * p.H * AT_D);
    const int nt = p.causal ? min(p.S - 1, q0 + AT_BR - 1) / AT_T + 1 : (p.S + AT_T - 1) / AT_T;
    const int wave_nt = qw0 >= p.S ? 0 : (p.causal ? min(p.S - 1, qw0 + 31) / AT_T + 1 : nt);
    const bf16* kbase = p.qkv + qkv_off(p, b, 0, 1, h);
    const bf16* vbase = p.qkv + qkv_off(p, b, 0, 2, h);
    const TileDma dma = tile_dma_setup(wave, lane);
    const FragAddr fa = frag_addr_setup(lane);

    // the first two tiles are requested before the per-row operands, so the prologue pays one memory latency, not two
#pragma unroll
    for (int t = 0; t < 2; ++t)
        if (t < nt) {
            tile_dma_issue(dma, kbase, srow, t * AT_T, p.S - 1, sKb[t], wave);
            tile_dma_issue(dma, vbase, srow, t * AT_T, p.S - 1, sVb[t], wave);
        }
    bf16x8 qf[4], dof[4];
    float nl = 0.f;    // minus the row's log-sum-exp (log2 units)
    f32x16 ndelta;
    {
        const int qc = min(qrow, p.S - 1);
        const bf16* qp = p.qkv + qkv_off(p, b, qc, 0, h) + 8 * hi;
        const size_t orow = ((static_cast<size_t>(b) * p.S + qc) * p.H + h) * AT_D + 8 * hi;
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
            dof[ks] = *reinterpret_cast<const bf16x8*>(p.dout + orow + 16 * ks);
            const bf16x8 ov = *reinterpret_cast<const bf16x8*>(p.o + orow + 16 * ks);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += bf16_to_f32(dof[ks][e]) * bf16_to_f32(ov[e]);
        }
        dl = other_half_sum(dl);
        const size_t off = (static_cast<size_t>(b) * p.H + h) * p.S + qc;
        if (hi == 0 && qrow < p.S) p.delta[off] = -dl;
        nl = -p.lse[off];
#pragma unroll
        for (int e = 0; e < 16; ++e) ndelta[e] = -dl;
    }
    f32x16 dq[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[0][e] = dq[1][e] = 0.f;
    const int key_limit = p.causal ? min(qrow, p.S - 1) : p.S - 1;
    const int wave_last_key = p.causal ? min(qw0 + 31, p.S - 1) : p.S - 1;
    const uint32_t sk_lds = lds_addr(&sKb[0][0]), sv_lds = lds_addr(&sVb[0][0]);

    int buf = 0;
    auto next_tile = [&](int t) {  // see attn_fwd_kernel
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + 2 < nt) {
            const int nb = buf == 0 ? AT_RING - 1 : buf - 1;
            tile_dma_issue(dma, kbase, srow, (t + 2) * AT_T, p.S - 1, sKb[nb], wave);
            tile_dma_issue(dma, vbase, srow, (t + 2) * AT_T, p.S - 1, sVb[nb], wave);
        }
        const int cur = buf;
        buf = buf == AT_RING - 1 ? 0 : buf + 1;
        return cur;
    };
    for (int t = 0; t < wave_nt; ++t) {
        const int k0 = t * AT_T;
        const int cur = next_tile(t);
        const uint32_t sk_tile = sk_lds + static_cast<uint32_t>(cur) * (AT_TILE * 2), sv_tile = sv_lds + static_cast<uint32_t>(cur) * (AT_TILE * 2);
        const bool masked = (k0 + AT_T > p.S) || (p.causal && k0 + AT_T - 1 > qw0);
        const int rel = key_limit - k0 - 4 * hi;
        // two halves of 32 keys, one after the other: S^T and dP^T of a half, its vector-ALU work, its share of dQ^T
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt == 1 && k0 + 32 > wave_last_key) break;  // no query of the wave reaches the second half
            f32x16 s, dp;
            {
                RowFrags kf, vf;
                frag_rows_issue(kf, sk_tile, fa, kt);
                frag_rows_issue(vf, sv_tile, fa, kt);
                lds_wait<4>();
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_get(kf, ks), qf[ks], s, 0, 0, 0);
                lds_wait<0>();
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_get(vf, ks), dof[ks], ks == 0 ? ndelta : dp, 0, 0, 0);
            }
            TrFrags kt_f[2];  // K^T fragments per column tile, in flight during the vector-ALU work
            frag_tr_issue(kt_f[0], sk_tile, fa, 0, kt);
            frag_tr_issue(kt_f[1], sk_tile, fa, 1, kt);
            if (masked) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = (32 * kt + (r & 3) + 8 * (r >> 2)) <= rel ? s[r] : AT_NEG;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = fast_exp2(fmaf(s[r], p.scale_log2, nl)) * dp[r];
            const bf16x8 ds0 = pack_half(s, 0), ds1 = pack_half(s, 1);
            lds_wait<0>();
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_get(kt_f[dt], 0), ds0, dq[dt], 0, 0, 0);
                dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_get(kt_f[dt], 1), ds1, dq[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (int t = wave_nt; t < nt; ++t) next_tile(t);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    bf16* drow0 = p.dqkv + qkv_off(p, b, min(qw0, p.S - 1), 0, h);
    store_rows32(&sKb[0][0] + wave * (32 * AT_D), dq, p.scale, lane, p.S - qw0, [&](int row) { return drow0 + static_cast<size_t>(row) * srow; });
}

// ------------------------------------------------------------------------------------------------
// dK, dV: workgroup = 128 keys, the key on the lane, streams Q / dO tiles (and their log-sum-exp / delta rows).
//   S[q][key] = Q . K^T,  dP[q][key] = dO . V^T - delta[q]  (the tile's -delta values are the initial accumulator),
//   dV^T[d][key] += dO^T[d][q] . P[q][key],   dK^T[d][key] += Q^T[d][q] . dS[q][key]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const AttnArgs p) {
    __shared__ __attribute__((aligned(1024))) bf16 sQb[AT_RING][AT_TILE];
    __shared__ __attribute__((aligned(1024))) bf16 sDOb[AT_RING][AT_TILE];
    __shared__ __attribute__((aligned(1024))) float sRow[AT_RING][4][AT_T];  // per tile: lse, -delta, and two unused rows (waves 2, 3)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, lq = lane & 31;
    int kb, b, h;
    if (!at_block_of(p, kb, b, h)) return;  // under a causal mask the first key blocks carry the most query tiles: natural order
    const int k0 = kb * AT_BR, kw0 = k0 + 32 * wave, key = kw0 + lq;
    const uint32_t srow = static_cast<uint32_t>(3 * p.H * AT_D), orow = static_cast<uint32_t>(p.H * AT_D);
    const int nt = (p.S + AT_T - 1) / AT_T;
    const int t0 = p.causal ? k0 / AT_T : 0;                             // first query tile that reaches this key block
    const int wave_t0 = kw0 >= p.S ? nt : (p.causal ? kw0 / AT_T : 0);  // ... and this wave's keys
    const bf16* qbase = p.qkv + qkv_off(p, b, 0, 0, h);
    const bf16* dobase = p.dout + (static_cast<size_t>(b) * p.S * p.H + h) * AT_D;
    const float* rowbase = (wave & 1 ? p.delta : p.lse) + (static_cast<size_t>(b) * p.H + h) * p.S;
    const TileDma dma = tile_dma_setup(wave, lane);
    const FragAddr fa = frag_addr_setup(lane);

    bf16x8 kf[4], vf[4];  // B operands: K^T / V^T fragments of this lane's key
    {
        const int kc = min(key, p.S - 1);
        const bf16* kp = p.qkv + qkv_off(p, b, kc, 1, h) + 8 * hi;
        const bf16* vp = p.qkv + qkv_off(p, b, kc, 2, h) + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(kp + 16 * ks);
            vf[ks] = *reinterpret_cast<const bf16x8*>(vp + 16 * ks);
        }
    }
    auto issue = [&](int t, int slot) {
        tile_dma_issue(dma, qbase, srow, t * AT_T, p.S - 1, sQb[slot], wave);
        tile_dma_issue(dma, dobase, orow, t * AT_T, p.S - 1, sDOb[slot], wave);
        // 64 floats of lse (even waves) or -delta (odd waves): one 4-byte LDS-DMA per lane
        __builtin_amdgcn_global_load_lds((at_global_cvoid*)(rowbase + min(t * AT_T + lane, p.S - 1)), (at_lds_void*)&sRow[slot][wave][0], 4, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (t0 + i < nt) issue(t0 + i, i);
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) dk[0][e] = dk[1][e] = dv[0][e] = dv[1][e] = 0.f;
    const uint32_t sq_lds = lds_addr(&sQb[0][0]), sdo_lds = lds_addr(&sDOb[0][0]), srow_lds = lds_addr(&sRow[0][0][0]);

    int buf = 0;
    auto next_tile = [&](int t) {  // see attn_fwd_kernel; five DMA instructions per wave and tile here
        if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + 2 < nt) issue(t + 2, buf == 0 ? AT_RING - 1 : buf - 1);
        const int cur = buf;
        buf = buf == AT_RING - 1 ? 0 : buf + 1;
        return cur;
    };
    for (int t = t0; t < min(wave_t0, nt); ++t) next_tile(t);  // query tiles only the earlier waves of the workgroup need
    for (int t = max(t0, wave_t0); t < nt; ++t) {
        const int q0 = t * AT_T;
        const int cur = next_tile(t);
        const uint32_t sq_tile = sq_lds + static_cast<uint32_t>(cur) * (AT_TILE * 2), sdo_tile = sdo_lds + static_cast<uint32_t>(cur) * (AT_TILE * 2);
        const uint32_t row_tile = srow_lds + static_cast<uint32_t>(cur) * (4 * AT_T * 4) + 16 * hi;  // this half's first float4
        // per-element masking only where it can matter: queries beyond S, the causal diagonal (wave-uniform)
        const bool masked = (q0 + AT_T > p.S) || (p.causal && q0 < kw0 + 31);
        const int lo = (p.causal ? key : 0) - q0 - 4 * hi, up = p.S - 1 - q0 - 4 * hi;  // valid query offsets: [lo, up]
        // the tile is processed as two halves of 32 queries: accumulator register r of a half is its query
        // (r & 3) + 8 (r >> 2) + 4 hi, so a lane needs four float4 of the half's lse / -delta rows
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            // a half none of whose queries sees any key of the wave: past the end, or wholly before the wave's keys
            if (qt == 0 ? (p.causal && q0 + 31 < kw0) : (q0 + 32 >= p.S)) continue;
            f32x16 s, dp;
            {
                RowFrags qf;
                f32x4 nd[4];
                frag_rows_issue(qf, sq_tile, fa, qt);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) lds_read_f32x4(nd[r4], row_tile + AT_T * 4 + (32 * qt + 8 * r4) * 4);
                lds_wait<4>();
#pragma unroll
                for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_get(qf, ks), kf[ks], s, 0, 0, 0);
                RowFrags dof;
                frag_rows_issue(dof, sdo_tile, fa, qt);
                lds_wait<0>();
#pragma unroll
                for (int e = 0; e < 16; ++e) dp[e] = nd[e >> 2][e & 3];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows_get(dof, ks), vf[ks], dp, 0, 0, 0);
            }
            f32x4 l2[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) lds_read_f32x4(l2[r4], row_tile + (32 * qt + 8 * r4) * 4);
            TrFrags dot[2], qt_f[2];  // dO^T / Q^T fragments per column tile
            frag_tr_issue(dot[0], sdo_tile, fa, 0, qt);
            frag_tr_issue(qt_f[0], sq_tile, fa, 0, qt);
            lds_wait<8>();  // lse
            if (masked) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qo = 32 * qt + (r & 3) + 8 * (r >> 2);
                    s[r] = (qo >= lo && qo <= up) ? s[r] : AT_NEG;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = fast_exp2(fmaf(s[r], p.scale_log2, -l2[r >> 2][r & 3]));
                dp[r] *= s[r];
            }
            bf16x8 pb[2], dsb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                pb[j] = pack_half(s, j);
                dsb[j] = pack_half(dp, j);
            }
            frag_tr_issue(dot[1], sdo_tile, fa, 1, qt);
            frag_tr_issue(qt_f[1], sq_tile, fa, 1, qt);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                if (dt == 0) lds_wait<8>();
                else lds_wait<0>();
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_get(dot[dt], j), pb[j], dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr_get(qt_f[dt], j), dsb[j], dk[dt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    bf16* so = &sQb[0][0] + wave * (32 * AT_D);
    bf16* krow0 = p.dqkv + qkv_off(p, b, min(kw0, p.S - 1), 1, h);
    bf16* vrow0 = p.dqkv + qkv_off(p, b, min(kw0, p.S - 1), 2, h);
    store_rows32(so, dk, p.scale, lane, p.S - kw0, [&](int row) { return krow0 + static_cast<size_t>(row) * srow; });
    store_rows32(so, dv, 1.f, lane, p.S - kw0, [&](int row) { return vrow0 + static_cast<size_t>(row) * srow; });
}

static int attn_check(int B, int S, int H, int D) {
    if (B <= 0 || S <= 0 || H <= 0 || D != AT_D) return -1;
    // 32-bit element offsets inside one batch element's [S, 3, H, 64] slab, and a grid of at most 2^31 - 1 workgroups
    if (static_cast<long long>(S) * 3 * H * D >= (1ll << 31)) return -1;
    if (static_cast<long long>(pdt_cdiv(S, AT_BR)) * 8 * pdt_cdiv(static_cast<long long>(B) * H, 8) >= (1ll << 31)) return -1;
    if (static_cast<long long>(B) * S * 3 * H * D >= (1ll << 40)) return -1;
    return 0;
}

static AttnArgs attn_args(int B, int S, int H, float scale, int causal) {
    AttnArgs a = {};
    a.B = B; a.S = S; a.H = H;
    a.scale = scale;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.causal = causal;
    a.nblk = pdt_cdiv(S, AT_BR);
    return a;
}

static dim3 attn_grid(const AttnArgs& a) { return dim3(static_cast<unsigned>(a.nblk) * 8u * static_cast<unsigned>(pdt_cdiv(a.B * a.H, 8))); }

PDT_API int pdt_attention_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int D, float scale, int causal,
                              void* stream) {
    if (attn_check(B, S, H, D) != 0) return -1;
    AttnArgs a = attn_args(B, S, H, scale, causal);
    a.qkv = static_cast<const bf16*>(qkv);
    a.out = static_cast<bf16*>(out);
    a.lse = lse;
    hipLaunchKernelGGL(attn_fwd_kernel, attn_grid(a), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return pdt_launch_status();
}

// `delta` is scratch of B * H * S floats: written by the dQ kernel, consumed by the dK/dV kernel behind it.
PDT_API int pdt_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv,
                              int B, int S, int H, int D, float scale, int causal, void* stream) {
    if (attn_check(B, S, H, D) != 0) return -1;
    AttnArgs a = attn_args(B, S, H, scale, causal);
    a.qkv = static_cast<const bf16*>(qkv);
    a.o = static_cast<const bf16*>(o);
    a.dout = static_cast<const bf16*>(dout);
    a.lse = const_cast<float*>(lse);
    a.delta = delta;
    a.dqkv = static_cast<bf16*>(dqkv);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, attn_grid(a), dim3(256), 0, s, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, attn_grid(a), dim3(256), 0, s, a);
    return pdt_launch_status();
}
