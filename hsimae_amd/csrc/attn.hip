// Small-sequence multi-head attention on MFMA, forward and backward, for the HSIMAE blocks.
//
// Every attention on this path is tiny: per sample 14/27/54 kept tokens (encoder, head dim 16) or
// 54/108/216 tokens (decoder, head dim 8).  The spatial stack attends within one kept band-group, the
// spectral stack within one kept position, the fusion/decoder blocks over the whole sample.  All three
// are the same kernel with a per-token class mask (mode 1: class = i / len_l, mode 2: class = i % len_l),
// so no rearrange / transposed copies of the token grid are ever materialised (Models.py:553-564).
//
// One workgroup = one sample x 4 heads, one wave per head.  Scores are computed TRANSPOSED
// (S^T = K Q^T: key on the accumulator row, query on the lane) so the softmax reductions are in-lane plus
// two cross-group shuffles, and the bf16-packed probabilities are directly the B operand of the next
// MFMA (O^T = V^T P^T) with a permuted key order that the V^T LDS image is read in.  The backward
// recomputes the scores in both orientations instead of transposing dS through LDS.
#include "common.h"
#include "kernels.h"
#ifndef HS_BLK_PRIO
#define HS_BLK_PRIO 0      /* s_setprio level of waves 4-7 in blk128_fwd / blk128_bwd (round 6) */
#endif

// column offset of k inside a q|k|v row (v: twice that): the storage width when rows are stored wider than d
#define KVO(p) ((p).kv_off ? (p).kv_off : (p).d)
#include <cstdlib>

#ifndef HS_NT_C
#define HS_NT_C 1      /* u / q|k|v / o saved by blk128_fwd for the backward: read ~10 ms later (step -0.5 %) */
#endif
#ifndef HS_BF_XCOPY
#define HS_BF_XCOPY 1          /* blk128_fwd_kernel: residual from an fp32 LDS copy of x (1) or re-read from L2 (0) */
#endif
#ifndef HS_BB_DELTA_PDP
#define HS_BB_DELTA_PDP 1        /* blk128_bwd_kernel: delta = sum_j P dP inside the core (1: O is not read) or rowsum(dO * O) (0) */
#endif
#ifndef HS_BB_WQ_RESIDENT
#define HS_BB_WQ_RESIDENT 1   /* blk128_bwd_kernel<RC>: forward Wqkv fragments resident (1) or streamed from L2 per group (0) */
#endif
#ifndef HS_BF_EARLY_WAIT
#define HS_BF_EARLY_WAIT 1     /* blk128_fwd_kernel: the same in front of its output stores */
#endif
#ifndef HS_BB_EARLY_WAIT
#define HS_BB_EARLY_WAIT 1     /* blk128_bwd_kernel: one explicit vmcnt(0) in front of the LayerNorm epilogue (see there) */
#endif
#ifndef HS_NT_E
#define HS_NT_E 0      /* dq|dk|dv rows of blk128_bwd as streaming stores */
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------------
// Head dim 16 (every encoder attention): second-generation kernels.
//   * K = 16 MFMAs (`v_mfma_f32_16x16x16_bf16`): the head dim fills the contraction exactly, no zero-padded half;
//   * row-major LDS images only: the operands that need the token index on the contraction axis (V^T, K^T, Q^T, dO^T)
//     are read with `ds_read_b64_tr_b16` instead of being stored a second time transposed with 2-byte writes
//     (1,300 ds_write_b16 per wave in the first backward kernel);
//   * backward in ONE pass over the (query tile, key tile) grid: scores key-major (directly the B operand of dq^T),
//     P and dS transposed through a 1.5 KB per-wave LDS tile for dk^T / dv^T — no second score / exp pass.
// Same work decomposition: one workgroup = one sample x 4 heads, one wave per head.
typedef __attribute__((ext_vector_type(4))) short s16x4_;
typedef __attribute__((address_space(3))) bf16x4* lds_b64_p;
__device__ __forceinline__ f32x4 mfma_k16(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_, a), __builtin_bit_cast(s16x4_, b), c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 tr4(const bf16_t* a) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_p)(a)); }
__device__ __forceinline__ bf16x4 cvt4(f32x4 v) {
    bf16x4 r;
    r[0] = (bf16_t)v[0]; r[1] = (bf16_t)v[1]; r[2] = (bf16_t)v[2]; r[3] = (bf16_t)v[3];
    return r;
}

__device__ __forceinline__ bf16x4 zero4_() { bf16x4 z; z[0] = z[1] = z[2] = z[3] = (bf16_t)0.f; return z; }

constexpr int RS16 = 24;          // image row stride (elements) at head dim 16: 16 + 8 pad => conflict-free 8-byte row reads

// HD = 8 (the decoder's heads, layer-at-a-time decoder: 216-token sequences): the same kernels with the head dim
// zero-extended to the K = 16 MFMA.  The images hold the head's 8 columns only (16-byte rows); the zero half of every operand
// whose CONTRACTION runs over the head dim (the row fragments of q, k, v, dO) is supplied in registers (lane groups 2, 3 = 0),
// and the transposed fragments' columns 8..15 (the next row's data) only reach output rows d >= 8, which nobody stores.
// (With the zeros stored in the images a 224-row backward needed 29 KB per head: two heads per workgroup, 4 waves per CU;
// compact images: four heads per workgroup, two workgroups per CU.)
template <int NT, int HD = 16, int HPW = 4>
struct Lay16 {
    static constexpr int RS = HD == 16 ? RS16 : 8;                        // HD = 8: compact 16-byte rows (see below)
    static constexpr int TRS = HD == 16 ? RS16 : 16;                      // row stride of the 16 x 16 P / dS transposition tiles
    static constexpr int ROWS = NT * 16;
    static constexpr int IMG = ROWS * RS;                                 // elements per image
    static constexpr int CLS = ROWS * 4;
    static constexpr int FWD_WAVE = 3 * IMG * 2;                          // Q K V
    static constexpr int BWD_WAVE = 4 * IMG * 2 + 2 * ROWS * 4 + 2 * 16 * TRS * 2;    // Q K V dO | lse delta | T(P, dS)
};

// Rows [0, Ts) of the workgroup's HPW heads -> row-major per-wave images (rows [Ts, ROWS) zero: P = 0 there, but 0 * NaN = NaN),
// by all its threads: consecutive lanes take consecutive 16-byte pieces of
// a token's HPW * HD contiguous columns (128 B at head dim 16), instead of every wave fetching / writing 32-byte pieces of its
// own head at the row stride.  `img0` = wave 0's image of this matrix, `wstride` = elements between two waves' images.
template <int NT, int HD, int HPW>
__device__ __forceinline__ void load16_wg(const bf16_t* src, int ld, int Ts, int nheads, bf16_t* img0, int wstride) {
    constexpr int RS = HD == 16 ? RS16 : 8, PCS = HD / 8, PPR = HPW * PCS;
    for (int idx = threadIdx.x; idx < NT * 16 * PPR; idx += 64 * HPW) {
        const int tok = idx / PPR, pc = idx - tok * PPR, w = pc / PCS, sub = pc - w * PCS;
        bf16x8 v = zero8();
        if (tok < Ts && w < nheads) v = *reinterpret_cast<const bf16x8*>(src + (size_t)tok * ld + w * HD + sub * 8);
        *reinterpret_cast<bf16x8*>(img0 + w * wstride + tok * RS + sub * 8) = v;
    }
}
template <int NT, int HD, int HPW>
__device__ __forceinline__ void store16_wg(bf16_t* dst, int ld, int Ts, int nheads, const bf16_t* img0, int wstride) {
    constexpr int RS = HD == 16 ? RS16 : 8, PCS = HD / 8, PPR = HPW * PCS;
    for (int idx = threadIdx.x; idx < NT * 16 * PPR; idx += 64 * HPW) {
        const int tok = idx / PPR, pc = idx - tok * PPR, w = pc / PCS, sub = pc - w * PCS;
        if (tok < Ts && w < nheads)
            *reinterpret_cast<bf16x8*>(dst + (size_t)tok * ld + w * HD + sub * 8) = *reinterpret_cast<const bf16x8*>(img0 + w * wstride + tok * RS + sub * 8);
    }
}

template <int NT, int HD = 16, int HPW = 4, bool MODE0 = false>      // MODE0: one class (p.mode == 0), chosen at launch
__global__ __launch_bounds__(64 * HPW) void attn16_fwd_kernel(AttnParams p) {
    using L = Lay16<NT, HD, HPW>;
    constexpr int RS16 = L::RS;                 // (shadows the head-dim-16 constant: every image access below uses the layout's stride)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hgroups = (p.heads + HPW - 1) / HPW;
    const int sample = blockIdx.x / hgroups, head = (blockIdx.x % hgroups) * HPW + wave;
    const bool active = head < p.heads;
    int* cls = reinterpret_cast<int*>(smem);
    bf16_t* Qi = reinterpret_cast<bf16_t*>(smem + L::CLS + wave * L::FWD_WAVE);
    bf16_t* Ki = Qi + L::IMG;
    bf16_t* Vi = Ki + L::IMG;
    for (int i = threadIdx.x; i < L::ROWS; i += 64 * HPW) {
        int c = -1;
        if (i < p.Ts) c = (p.mode == 1) ? i / p.len_l : (p.mode == 2) ? i % p.len_l : 0;
        cls[i] = c;
    }
    const size_t row_base = (size_t)sample * p.Ts;
    const int head0 = (blockIdx.x % hgroups) * HPW, nheads = min(HPW, p.heads - head0);
    bf16_t* img0 = reinterpret_cast<bf16_t*>(smem + L::CLS);
    constexpr int WSTR = L::FWD_WAVE / 2;
    {
        const bf16_t* base = p.qkv + row_base * p.ld + head0 * HD;
        load16_wg<NT, HD, HPW>(base, p.ld, p.Ts, nheads, img0, WSTR);
        load16_wg<NT, HD, HPW>(base + KVO(p), p.ld, p.Ts, nheads, img0 + L::IMG, WSTR);
        load16_wg<NT, HD, HPW>(base + 2 * KVO(p), p.ld, p.Ts, nheads, img0 + 2 * L::IMG, WSTR);
    }
    lds_barrier();

    const int c16 = lane & 15, g = lane >> 4, q4 = c16 >> 2, p4 = c16 & 3;
    const float sc = (HD == 16 ? 0.25f : 0.35355339059327373f) * 1.4426950408889634f;      // hd^-0.5 * log2(e)
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    for (int qt = 0; qt < NT && active; ++qt) {
        if (qt * 16 >= p.Ts) break;
        const int query = qt * 16 + c16;
        const int qcls = cls[query];
        const bf16x4 bq = (HD == 16 || g < 2) ? *reinterpret_cast<const bf16x4*>(Qi + query * RS16 + 4 * g) : zero4_();
        f32x4 s[NT];
        float m = -INFINITY;
        if constexpr (MODE0) {
            // one class (the decoder, the fusion blocks): the only mask is the sequence end, and only key tiles that reach past it
            // need it (the per-element class compare below was a quarter of this loop's VALU work)
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const bf16x4 ak = (HD == 16 || g < 2) ? *reinterpret_cast<const bf16x4*>(Ki + (kt * 16 + c16) * RS16 + 4 * g) : zero4_();
                s[kt] = mfma_k16(ak, bq, z4);
                if ((kt + 1) * 16 > p.Ts) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[kt][r] = (kt * 16 + g * 4 + r < p.Ts) ? s[kt][r] * sc : -INFINITY;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[kt][r] *= sc;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, s[kt][r]);
            }
        } else {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const bf16x4 ak = (HD == 16 || g < 2) ? *reinterpret_cast<const bf16x4*>(Ki + (kt * 16 + c16) * RS16 + 4 * g) : zero4_();
            s[kt] = mfma_k16(ak, bq, z4);
            const int4 kc = *reinterpret_cast<const int4*>(cls + kt * 16 + g * 4);
            const int kcl[4] = {kc.x, kc.y, kc.z, kc.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = (kcl[r] >= 0) && (kcl[r] == qcls);
                s[kt][r] = ok ? s[kt][r] * sc : -INFINITY;
                m = fmaxf(m, s[kt][r]);
            }
        }
        }
        m = rows_max(m);
        if (m == -INFINITY) m = 0.f;
        float lsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[kt][r] - m);
                s[kt][r] = e;
                lsum += e;
            }
        lsum = rows_sum(lsum);
        const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
        f32x4 o = z4;                                           // o^T[d = 4g + r][query c16]
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const bf16x4 av = tr4(Vi + (kt * 16 + 4 * g + q4) * RS16 + 4 * p4);      // V^T[d = c16][key 4g + j]
            o = mfma_k16(av, cvt4(s[kt]), o);
        }
        if (query < p.Ts) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)(o[r] * inv);
            // in place over this head's q rows of the finished query tile (only this wave reads them); leaves as whole row segments below
            if (g * 4 < HD) *reinterpret_cast<bf16x4*>(Qi + query * RS16 + g * 4) = ov;
            if (g == 0 && p.lse) p.lse[(row_base + query) * p.heads + head] = m + __builtin_amdgcn_logf(fmaxf(lsum, 1e-30f));
        }
    }
    lds_barrier();
    store16_wg<NT, HD, HPW>(p.o + row_base * p.ldo + head0 * HD, p.ldo, p.Ts, nheads, img0, WSTR);
}

template <int NT, int HD = 16, int HPW = 4, bool MODE0 = false>
__global__ __launch_bounds__(64 * HPW) void attn16_bwd_kernel(AttnParams p) {
    using L = Lay16<NT, HD, HPW>;
    constexpr int RS16 = L::RS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hgroups = (p.heads + HPW - 1) / HPW;
    const int sample = blockIdx.x / hgroups, head = (blockIdx.x % hgroups) * HPW + wave;
    const bool active = head < p.heads;
    int* cls = reinterpret_cast<int*>(smem);
    bf16_t* Qi = reinterpret_cast<bf16_t*>(smem + L::CLS + wave * L::BWD_WAVE);
    bf16_t* Ki = Qi + L::IMG;
    bf16_t* Vi = Ki + L::IMG;
    bf16_t* Di = Vi + L::IMG;
    float* lse = reinterpret_cast<float*>(Di + L::IMG);
    float* delta = lse + L::ROWS;
    bf16_t* Tp = reinterpret_cast<bf16_t*>(delta + L::ROWS);
    constexpr int TRS = L::TRS;
    bf16_t* Td = Tp + 16 * TRS;
    for (int i = threadIdx.x; i < L::ROWS; i += 64 * HPW) {
        int c = -1;
        if (i < p.Ts) c = (p.mode == 1) ? i / p.len_l : (p.mode == 2) ? i % p.len_l : 0;
        cls[i] = c;
    }
    const size_t row_base = (size_t)sample * p.Ts;
    const int head0 = (blockIdx.x % hgroups) * HPW, nheads = min(HPW, p.heads - head0);
    bf16_t* img0 = reinterpret_cast<bf16_t*>(smem + L::CLS);
    constexpr int WSTR = L::BWD_WAVE / 2;
    {
        const bf16_t* base = p.qkv + row_base * p.ld + head0 * HD;
        load16_wg<NT, HD, HPW>(base, p.ld, p.Ts, nheads, img0, WSTR);
        load16_wg<NT, HD, HPW>(base + KVO(p), p.ld, p.Ts, nheads, img0 + L::IMG, WSTR);
        load16_wg<NT, HD, HPW>(base + 2 * KVO(p), p.ld, p.Ts, nheads, img0 + 2 * L::IMG, WSTR);
        load16_wg<NT, HD, HPW>(p.dout + row_base * p.lddo + head0 * HD, p.lddo, p.Ts, nheads, img0 + 3 * L::IMG, WSTR);
    }
    // delta_i = rowsum(dO * O) = sum_j P_ij dP_ij.  With up to four key tiles per query tile (every encoder attention) it is formed
    // inside the core from the P and dP tiles themselves (PDP): O is not read, and the delta no longer carries the bf16 rounding of
    // the saved O (round 4: what the backward adds to the q / k gradients at config 1 went 7e-3 -> 3e-3 in the d = 128 kernel).
    // Longer sequences (the 216-token decoder: 14 key tiles = 112 registers of P and dP) keep the prologue form.
    constexpr bool PDP = NT <= 4;
    if (active) {
        for (int tok = lane; tok < L::ROWS; tok += 64) {
            float acc = 0.f, l = 1e30f;                          // rows past Ts: exp2(s - 1e30) = 0
            if (tok < p.Ts) {
                if constexpr (!PDP) {
                    const bf16_t* orow = p.o + (row_base + tok) * p.ldo + head * HD;
                    const bf16_t* drow = p.dout + (row_base + tok) * p.lddo + head * HD;
#pragma unroll
                    for (int e = 0; e < HD; e += 8) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(orow + e);
                        const bf16x8 b = *reinterpret_cast<const bf16x8*>(drow + e);
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc += bf2f(a[i]) * bf2f(b[i]);
                    }
                }
                l = p.lse[(row_base + tok) * p.heads + head];
            }
            if constexpr (!PDP) delta[tok] = acc;
            lse[tok] = l;
        }
    }
    lds_barrier();

    const int c16 = lane & 15, g = lane >> 4, q4 = c16 >> 2, p4 = c16 & 3;
    const float scale = HD == 16 ? 0.25f : 0.35355339059327373f, sc = scale * 1.4426950408889634f;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int troff = (4 * g + q4) * RS16 + 4 * p4, ttoff = (4 * g + q4) * TRS + 4 * p4;
    f32x4 dkT[NT], dvT[NT];                       // [d = 4g + r][key c16], accumulated over the query tiles
    bf16x4 KT[NT];                                // K^T[d = c16][key 4g + j]
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) { dkT[kt] = z4; dvT[kt] = z4; KT[kt] = tr4(Ki + kt * 16 * RS16 + troff); }
    for (int qt = 0; qt < NT && active; ++qt) {
        if (qt * 16 >= p.Ts) break;
        const int query = qt * 16 + c16;
        const int qcls = cls[query];
        const bf16x4 bq = (HD == 16 || g < 2) ? *reinterpret_cast<const bf16x4*>(Qi + query * RS16 + 4 * g) : zero4_();
        const bf16x4 bdo = (HD == 16 || g < 2) ? *reinterpret_cast<const bf16x4*>(Di + query * RS16 + 4 * g) : zero4_();
        const float lqn = -lse[query];
        const bf16x4 QT = tr4(Qi + qt * 16 * RS16 + troff);      // Q^T[d = c16][query 4g + j]
        const bf16x4 DT = tr4(Di + qt * 16 * RS16 + troff);
        f32x4 dqT = z4;
        // P (and dP) of one (query tile, key tile) pair: S^T[key 4g + r][query c16]
        auto pair_p = [&](int kt, f32x4& pv, f32x4& dp) {
            const bf16x4 ak = (HD == 16 || g < 2) ? *reinterpret_cast<const bf16x4*>(Ki + (kt * 16 + c16) * RS16 + 4 * g) : zero4_();
            const bf16x4 av = (HD == 16 || g < 2) ? *reinterpret_cast<const bf16x4*>(Vi + (kt * 16 + c16) * RS16 + 4 * g) : zero4_();
            const f32x4 s = mfma_k16(ak, bq, z4);
            dp = mfma_k16(av, bdo, z4);
            if constexpr (MODE0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(fminf(fmaf(s[r], sc, lqn), 0.f));
            } else {
                const int4 kc = *reinterpret_cast<const int4*>(cls + kt * 16 + g * 4);
                const int kcl[4] = {kc.x, kc.y, kc.z, kc.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[r] = ((kcl[r] >= 0) && (kcl[r] == qcls)) ? __builtin_amdgcn_exp2f(fmaf(s[r], sc, lqn)) : 0.f;
            }
        };
        f32x4 pvs[PDP ? NT : 1], dps[PDP ? NT : 1];
        float dl = 0.f;
        if constexpr (PDP) {
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                pair_p(kt, pvs[kt], dps[kt]);
                // (MODE0: a padded key has K = V = 0 rows, so dP = 0 and its clamped finite P adds nothing)
#pragma unroll
                for (int r = 0; r < 4; ++r) dl = fmaf(pvs[kt][r], dps[kt][r], dl);
            }
            dl = rows_sum(dl);
        } else {
            dl = delta[query];
        }
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x4 pv, dp, ds;
            if constexpr (PDP) { pv = pvs[kt]; dp = dps[kt]; } else pair_p(kt, pv, dp);
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[r] = pv[r] * (dp[r] - dl);
            const bf16x4 pb = cvt4(pv), dsb = cvt4(ds);
            dqT = mfma_k16(KT[kt], dsb, dqT);
            *reinterpret_cast<bf16x4*>(Tp + c16 * TRS + 4 * g) = pb;
            *reinterpret_cast<bf16x4*>(Td + c16 * TRS + 4 * g) = dsb;
            asm volatile("" ::: "memory");
            const bf16x4 Bp = tr4(Tp + ttoff), Bds = tr4(Td + ttoff);       // [k = query 4g + j][col key c16]
            dkT[kt] = mfma_k16(QT, Bds, dkT[kt]);
            dvT[kt] = mfma_k16(DT, Bp, dvT[kt]);
        }
        {   // dq in place over this head's q rows of the finished query tile; dk / dv below, once every read of K / V is done.
            // The gradients leave as whole row segments (store16_wg) instead of 8-byte pieces at the row stride.
            bf16x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (bf16_t)(dqT[r] * scale);
            if (g * 4 < HD) *reinterpret_cast<bf16x4*>(Qi + query * RS16 + g * 4) = v;
        }
    }
    if (active) {
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const int key = kt * 16 + c16;
            if (g * 4 < HD) {
                bf16x4 vk, vv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { vk[r] = (bf16_t)(dkT[kt][r] * scale); vv[r] = (bf16_t)dvT[kt][r]; }
                *reinterpret_cast<bf16x4*>(Ki + key * RS16 + g * 4) = vk;
                *reinterpret_cast<bf16x4*>(Vi + key * RS16 + g * 4) = vv;
            }
        }
    }
    lds_barrier();
    bf16_t* dq_base = p.dqkv + row_base * p.ld + head0 * HD;
    store16_wg<NT, HD, HPW>(dq_base, p.ld, p.Ts, nheads, img0, WSTR);
    store16_wg<NT, HD, HPW>(dq_base + KVO(p), p.ld, p.Ts, nheads, img0 + L::IMG, WSTR);
    store16_wg<NT, HD, HPW>(dq_base + 2 * KVO(p), p.ld, p.Ts, nheads, img0 + 2 * L::IMG, WSTR);
}

constexpr int FS = 128 + 8;        // full-row image stride (elements) of the padded (HS_BF_SWZ = 0) layout of blk128_fwd_kernel

// ---------------------------------------------------------------------------------------------------------------
// The attention half of an encoder Block in ONE persistent kernel (d = 128, 8 heads, <= 32 tokens per sample):
//   u = LN1(x);  q|k|v = u Wqkv^T + b;  o = softmax(q k^T / 4) v per head;  x1 = x + rs * (o Wp^T + bp)
// (Models.py:303-304 with Attention.forward :192-219).  Replaced lnqkv_kernel + attn128_fwd_kernel (rounds 1-3; removed in round 6): q|k|v never make
// the round trip through HBM between the two (they are still written once, for the backward) and x is read once
// (it is both the LayerNorm input and the residual).  One workgroup walks samples; wave h owns head h end to end:
// its 48 columns of Wqkv and its 16 columns of Wp stay in registers for the whole launch, its q / k / v columns go
// from the MFMA accumulators (operands swapped: a lane owns 4 consecutive columns of one token) into the LDS images
// the attention reads, so the only workgroup barriers are LN1 -> products and attention -> projection.
// Per-phase cycle accounting for scripts/phase_blk128_bwd.py (compiled only with -DHS_PHASE_TIMING; never in the shipped library)
#ifdef HS_PHASE_TIMING
}  // namespace
__device__ unsigned long long hs_phase_cycles_b128[32];      // [0, 16): blk128_bwd_kernel, [16, 32): blk128_fwd_kernel
extern "C" __attribute__((visibility("default"))) int hsimae_debug_phases_b128(unsigned long long* out, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hs_phase_cycles_b128), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(hs_phase_cycles_b128), z, sizeof(z)); }
    return rc;
}
namespace {
#define PHB_DECL unsigned long long ph_t0 = __builtin_readcyclecounter(), ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PHB(i) { const unsigned long long ph_t = __builtin_readcyclecounter(); ph_acc[i] += ph_t - ph_t0; ph_t0 = ph_t; }
#define PHB_FLUSH() if (threadIdx.x == 64 * 5) { for (int i = 0; i < 12; ++i) atomicAdd(&hs_phase_cycles_b128[i], ph_acc[i]); }
#define PHF_FLUSH() if (threadIdx.x == 64 * 5) { for (int i = 0; i < 12; ++i) atomicAdd(&hs_phase_cycles_b128[16 + i], ph_acc[i]); }
#else
#define PHB_DECL
#define PHB(i)
#define PHB_FLUSH()
#define PHF_FLUSH()
#endif
// XOR swizzle of the 16-byte chunks of an unpadded 256-byte image row (found for blk128_bwd_kernel in round 4 by exhaustive search
// over the XOR-linear maps of the row's low 4 bits; scripts/micro/lds_audit_blk128.py): chunk c of row r sits at chunk c ^ bsw(r).
__device__ __forceinline__ int bsw(int row) { return ((row & 1) << 1) ^ ((row & 2) << 1) ^ ((row & 4) << 1) ^ (((row >> 3) & 1) * 9); }
#ifndef HS_BF_SWZ
#define HS_BF_SWZ 1          /* blk128_fwd_kernel: images as unpadded swizzled 256-byte rows (round 5; 0 = 272-byte pitch of rounds 3-4) */
#endif
struct Blk128Args {
    const float* x; const float* n1w; const float* n1b;
    const bf16_t* wqkv; const float* bqkv; const bf16_t* wp; const float* pb;
    bf16_t* u; bf16_t* qkv; bf16_t* o; float* lse; float* x1; const float* rowscale;
    int Ts, nsamples, mode, len_l;
};

// SPW samples are walked per iteration ("slots" of NT m-tiles each in the images): one sample is a chain of ~30 dependent
// LDS / MFMA steps, so a wave with a single sample in hand is parked 75 % of the time; two independent chains per wave
// and half the barriers per sample is what the 2 waves per SIMD can still interleave.
template <int NT, int SPW>
struct LayB {
    static constexpr int ROWS = NT * 16;                 // rows of one slot
    static constexpr int RT = SPW * ROWS;                // rows of the images
    static constexpr int PITCH = HS_BF_SWZ ? 128 : FS;   // image row (elements)
    static constexpr int IMG = RT * PITCH;
    static constexpr int XS = 132;                       // fp32 copy of the group's x rows (the residual): row stride
    // cls | U | Q K V | O | lse | vectors (gamma | beta | bqkv | bp) | X
    static constexpr int TOTAL = RT * 4 + 5 * IMG * 2 + RT * 8 * 4 + 768 * 4 + HS_BF_XCOPY * RT * XS * 4;
};

template <int NT, int SPW>
__global__ __launch_bounds__(512, 2) void blk128_fwd_kernel(Blk128Args p) {
    using L = LayB<NT, SPW>;
    constexpr int ROWS = L::ROWS, RT = L::RT, MTT = SPW * NT;
    constexpr int PASSES = (RT * 16 + 511) / 512;               // LayerNorm passes: 16 lanes per row, 32 rows per pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, head = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int* cls = reinterpret_cast<int*>(smem);
    bf16_t* Uf = reinterpret_cast<bf16_t*>(smem + RT * 4);
    bf16_t* Qf = Uf + L::IMG;
    bf16_t* Kf = Qf + L::IMG;
    bf16_t* Vf = Kf + L::IMG;
    bf16_t* Of = Vf + L::IMG;
    float* lse_s = reinterpret_cast<float*>(Of + L::IMG);       // [RT][8]
    const int c16 = lane & 15, g = lane >> 4, q4 = c16 >> 2, p4 = c16 & 3, hc = head * 16;
    const float sc = 0.25f * 1.4426950408889634f;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // Image addressing (elements).  Round 5 (3.4 bank-conflict cycles per LDS instruction at the counters, against
    // 0.5-1.1 in the backward kernels): the layout of blk128_bwd_kernel — unpadded 256-byte rows, 16-byte chunk c of row r at
    // c ^ bsw(r) — under which the 16-byte row fragments (8 -> 4 LDS cycles per wave-instruction against the 272-byte pitch), the
    // 8-byte head-row reads and the transposed reads are conflict-free (scripts/micro/lds_audit_blk128.py lists every pattern of
    // both kernels under both layouts); the 8-byte head-row writes stay 2-way.
    constexpr int PITCH = L::PITCH;
    const int fr = bsw(c16), ft = bsw(4 * g + q4);
    auto wide = [&](int irow, int pc) -> int { return HS_BF_SWZ ? irow * 128 + ((pc ^ bsw(irow)) << 3) : irow * FS + pc * 8; };          // 16-byte piece pc of row irow
    auto fragoff = [&](int mt, int ks) -> int {                      // MFMA operand: row 16 mt + c16, columns 32 ks + 8 g ..
        return HS_BF_SWZ ? (mt * 16 + c16) * 128 + (((4 * ks + g) ^ fr) << 3) : (mt * 16 + c16) * FS + ks * 32 + g * 8; };
    auto cell = [&](int row0) -> int {                               // this head's columns 4 g .. of row row0 + c16 (row0 % 16 == 0)
        return HS_BF_SWZ ? (row0 + c16) * 128 + (((2 * head + (g >> 1)) ^ fr) << 3) + (g & 1) * 4 : (row0 + c16) * FS + hc + 4 * g; };
    auto trof = [&](int row0) -> int {                               // transposed read: rows row0 + 4 g + q4, this head's columns 4 p4 ..
        return HS_BF_SWZ ? (row0 + 4 * g + q4) * 128 + (((2 * head + (p4 >> 1)) ^ ft) << 3) + (p4 & 1) * 4 : (row0 + 4 * g + q4) * FS + hc + 4 * p4; };

    if (HS_BLK_PRIO > 0 && head >= 4) __builtin_amdgcn_s_setprio(HS_BLK_PRIO);      // static priority for the younger half (round 6, see fused_dec.hip HS_DEC_PRIO)
    // this wave's weights: n-tiles head (q), 8 + head (k), 16 + head (v) of the packed [384][128] image, n-tile head of Wp
    bf16x8 wq[3][4], wpj[4];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wq[m][ks] = *reinterpret_cast<const bf16x8*>(p.wqkv + ((size_t)((m * 8 + head) * 4 + ks) * 64 + lane) * 8);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wpj[ks] = *reinterpret_cast<const bf16x8*>(p.wp + ((size_t)(head * 4 + ks) * 64 + lane) * 8);
    // vectors every sample needs live in LDS, not in registers (the weights take 64)
    float* vec_s = lse_s + RT * 8;                                // gamma[128] | beta[128] | bqkv[384] | bp[128]
    float* XR = vec_s + 768;                                      // [RT][XS] (HS_BF_XCOPY)
    for (int i = threadIdx.x; i < 768; i += 512)
        vec_s[i] = i < 128 ? p.n1w[i] : i < 256 ? p.n1b[i - 128] : i < 640 ? p.bqkv[i - 256] : p.pb[i - 640];
    for (int i = threadIdx.x; i < RT; i += 512) {                 // class of an image row: -1 = padding; slots never mix
        const int slot = i / ROWS, r = i - slot * ROWS;
        int c = -1;
        if (r < p.Ts) c = slot * 64 + ((p.mode == 1) ? r / p.len_l : (p.mode == 2) ? r % p.len_l : 0);
        cls[i] = c;
    }

    const int lc8 = (threadIdx.x & 15) * 8;
    float xn[PASSES][8];                                          // next group's row pieces, in flight during this group
    auto fetch = [&](int first) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int irow = ps * 32 + (threadIdx.x >> 4), slot = irow / ROWS, r = irow - slot * ROWS;
#pragma unroll
            for (int e = 0; e < 8; ++e) xn[ps][e] = 0.f;
            if (irow < RT && first + slot < p.nsamples && r < p.Ts) {
                const float* src = p.x + ((size_t)(first + slot) * p.Ts + r) * 128 + lc8;
                const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
                xn[ps][0] = a.x; xn[ps][1] = a.y; xn[ps][2] = a.z; xn[ps][3] = a.w;
                xn[ps][4] = b.x; xn[ps][5] = b.y; xn[ps][6] = b.z; xn[ps][7] = b.w;
            }
        }
    };
    fetch(blockIdx.x * SPW);
    lds_barrier();                                                // vec_s / cls visible
    // The class / padding mask of a (query tile, key tile) pair is the same for every sample and slot: built once, as the
    // accumulator the score MFMA starts from (0 or -inf) — the per-element compares and selects leave the sample loop.
    f32x4 cm[NT][NT];
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        const int qc = cls[qt * 16 + c16];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kc = cls[kt * 16 + g * 4 + r];
                cm[qt][kt][r] = (kc >= 0 && kc == qc) ? 0.f : -INFINITY;
            }
    }
    PHB_DECL
    if (HS_BF_EARLY_WAIT) __builtin_amdgcn_s_waitcnt(0x0F70);       // the first group's rows (wanted at once): no load pending on EITHER way into the loop head
    for (int first = blockIdx.x * SPW; first < p.nsamples; first += gridDim.x * SPW) {
        // global row of image row i (slot-major), or -1
        auto grow = [&](int irow) -> int64_t {
            const int slot = irow / ROWS, r = irow - slot * ROWS;
            return (first + slot < p.nsamples && r < p.Ts) ? (int64_t)(first + slot) * p.Ts + r : -1;
        };
        // ---- LN1 -> U image (+ u to HBM: the q / k / v weight gradients' operand)
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int irow = ps * 32 + (threadIdx.x >> 4);
            if (irow < RT) {
#if HS_BF_XCOPY
                *reinterpret_cast<float4*>(XR + irow * L::XS + lc8) = make_float4(xn[ps][0], xn[ps][1], xn[ps][2], xn[ps][3]);
                *reinterpret_cast<float4*>(XR + irow * L::XS + lc8 + 4) = make_float4(xn[ps][4], xn[ps][5], xn[ps][6], xn[ps][7]);
#endif
                float sm = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += xn[ps][e];
                sm = lanes_sum<16>(sm);
                const float mean = sm * (1.f / 128.f);
                float vq = 0.f, f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { f[e] = xn[ps][e] - mean; vq += f[e] * f[e]; }
                vq = lanes_sum<16>(vq);
                const float rstd = rsqrtf(vq * (1.f / 128.f) + 1e-5f);
                {
                    const float4 g0 = *reinterpret_cast<const float4*>(vec_s + lc8), g1 = *reinterpret_cast<const float4*>(vec_s + lc8 + 4);
                    const float4 b0 = *reinterpret_cast<const float4*>(vec_s + 128 + lc8), b1 = *reinterpret_cast<const float4*>(vec_s + 128 + lc8 + 4);
                    const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = f[e] * rstd * gm[e] + bt[e];
                }
                const int64_t gr = grow(irow);
                const bf16x8 ub = gr >= 0 ? cvt8(f) : zero8();
                *reinterpret_cast<bf16x8*>(Uf + wide(irow, threadIdx.x & 15)) = ub;
                if (gr >= 0) HS_NT(HS_NT_C, reinterpret_cast<bf16x8*>(p.u + gr * 128 + lc8), ub);
            }
        }
        PHB(0)
        fetch(first + gridDim.x * SPW);                           // next group's rows fly during this one
        PHB(1)
        lds_barrier();
        PHB(2)
        // ---- q | k | v of this head: transposed accumulators -> 8-byte writes into the images
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            bf16_t* img = Qf + m * L::IMG;
            const f32x4 bias = *reinterpret_cast<const f32x4*>(vec_s + 256 + m * 128 + hc + 4 * g);
#pragma unroll
            for (int mt = 0; mt < MTT; ++mt) {
                f32x4 acc = bias;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    acc = mfma16(wq[m][ks], *reinterpret_cast<const bf16x8*>(Uf + fragoff(mt, ks)), acc);
                *reinterpret_cast<bf16x4*>(img + cell(mt * 16)) = cvt4(acc);
            }
        }
        PHB(3)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // own writes before own reads (only this wave touches these columns)
        // ---- attention of this head (one head per wave) slot by slot, O into its own image
#pragma unroll
        for (int slot = 0; slot < SPW; ++slot) {
            const int r0 = slot * ROWS;
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                if (qt * 16 >= p.Ts) break;
                const int query = r0 + qt * 16 + c16;
                const bf16x4 bqf = *reinterpret_cast<const bf16x4*>(Qf + cell(r0 + qt * 16));
                f32x4 sv[NT];
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    const bf16x4 ak = *reinterpret_cast<const bf16x4*>(Kf + cell(r0 + kt * 16));
                    sv[kt] = mfma_k16(ak, bqf, cm[qt][kt]);           // masked pairs start (and stay) at -inf
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sv[kt][r] *= sc;
                        m = fmaxf(m, sv[kt][r]);
                    }
                }
                m = rows_max(m);
                if (m == -INFINITY) m = 0.f;
                float lsum = 0.f;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __builtin_amdgcn_exp2f(sv[kt][r] - m);
                        sv[kt][r] = e;
                        lsum += e;
                    }
                lsum = rows_sum(lsum);
                const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
                f32x4 o = z4;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
                    o = mfma_k16(tr4(Vf + trof(r0 + kt * 16)), cvt4(sv[kt]), o);
                bf16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)(o[r] * inv);
                *reinterpret_cast<bf16x4*>(Of + cell(r0 + qt * 16)) = ov;
                if (g == 0) lse_s[query * 8 + head] = m + __builtin_amdgcn_logf(fmaxf(lsum, 1e-30f));
            }
        }
#if HS_BF_XCOPY
        // ---- residual pieces of this wave's output tiles, from the fp32 copy the LayerNorm phase left in LDS (read BEFORE the
        //      barrier: the next group's LayerNorm phase, which overwrites the copy, starts only after it).  Re-read from global
        //      memory they were L2 hits, but 22 % of this kernel's traffic through the CU's memory pipeline, which is what it waits for
        f32x4 xr[MTT];
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) xr[mt] = *reinterpret_cast<const f32x4*>(XR + (mt * 16 + c16) * L::XS + hc + 4 * g);
#endif
        PHB(4)
        lds_barrier();
        PHB(5)
        // (as in blk128_bwd_kernel, HS_BB_EARLY_WAIT: the next group's x rows — issued in front of the q|k|v products — are waited for
        //  HERE, explicitly, so that the group's o / x1 stores below are not what the next iteration's first vmcnt wait drains)
        if (HS_BF_EARLY_WAIT) __builtin_amdgcn_s_waitcnt(0x0F70);
        int64_t orow[MTT];
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) orow[mt] = grow(mt * 16 + c16);
#if !HS_BF_XCOPY
        // ---- residual pieces of this wave's output tiles (L2-hot: the LayerNorm read the same rows)
        f32x4 xr[MTT];
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) xr[mt] = orow[mt] >= 0 ? *reinterpret_cast<const f32x4*>(p.x + orow[mt] * 128 + hc + 4 * g) : z4;
#endif
        // ---- saved activations leave as whole rows: q|k|v (48 pieces per row), o, lse
        //      (round 4: moving the q|k|v stores into the next group's LayerNorm phase moved their cost with them — 727 M cycles per
        //       step either way: the waves wait for the memory system wherever the stores are issued)
        if (p.qkv)                 // nullptr: the backward recomputes q|k|v from u (blk128_bwd_kernel<RC>), nothing to save
        for (int idx = threadIdx.x; idx < RT * 48; idx += 512) {
            const int irow = idx / 48, pc = idx - irow * 48;
            const int64_t gr = grow(irow);
            if (gr >= 0)
                HS_NT(HS_NT_C, reinterpret_cast<bf16x8*>(p.qkv + gr * 384 + pc * 8),
                      *reinterpret_cast<const bf16x8*>(Qf + (pc >> 4) * L::IMG + wide(irow, pc & 15)));
        }
        for (int idx = threadIdx.x; idx < RT * 16; idx += 512) {
            const int irow = idx >> 4, pc = idx & 15;
            const int64_t gr = grow(irow);
            if (gr >= 0) HS_NT(HS_NT_C, reinterpret_cast<bf16x8*>(p.o + gr * 128 + pc * 8), *reinterpret_cast<const bf16x8*>(Of + wide(irow, pc)));
        }
        for (int idx = threadIdx.x; idx < RT * 2; idx += 512) {
            const int irow = idx >> 1;
            const int64_t gr = grow(irow);
            if (gr >= 0) *reinterpret_cast<float4*>(p.lse + gr * 8 + (idx & 1) * 4) = *reinterpret_cast<const float4*>(lse_s + idx * 4);
        }
        PHB(6)
        // ---- projection: this wave's 16 output columns; transposed accumulators -> x1 leaves as 16-byte pieces
        //      (rows c16, columns hc + 4 g ..: two heads complete a 128-byte line)
        const f32x4 pbias = *reinterpret_cast<const f32x4*>(vec_s + 640 + hc + 4 * g);
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) {
            f32x4 pa = pbias;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                pa = mfma16(wpj[ks], *reinterpret_cast<const bf16x8*>(Of + fragoff(mt, ks)), pa);
            if (orow[mt] >= 0) {
                const float rs = p.rowscale ? p.rowscale[orow[mt]] : 1.f;        // DropPath: x + scale * attn(x)
                f32x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = fmaf(pa[r], rs, xr[mt][r]);
                *reinterpret_cast<f32x4*>(p.x1 + orow[mt] * 128 + hc + 4 * g) = ov;
            }
        }
        PHB(7)
        // no barrier here: the next group's LayerNorm writes only U (its readers passed the barrier above), and its
        // q|k|v image writes come after its own first barrier, which every wave reaches after the row stores above
    }
    PHF_FLUSH()
}

template <int NT, int SPW>
int launch_blk128(const Blk128Args& a, hipStream_t s) {
    using L = LayB<NT, SPW>;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blk128_fwd_kernel<NT, SPW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::TOTAL); attr_set = true; }
    // persistent, one 8-wave workgroup per CU: the weights (64 registers) + the attention state need ~156 registers;
    // forced to 128 (two workgroups per CU) the kernel spills 24-68 of them and is 20 % slower.  HSIMAE_BLK128_WGS overrides.
    static int wgs = 0;
    if (!wgs) wgs = 256;
    const int groups = (a.nsamples + SPW - 1) / SPW;
    hipLaunchKernelGGL((blk128_fwd_kernel<NT, SPW>), dim3(groups < wgs ? groups : wgs), dim3(512), (size_t)L::TOTAL, s, a);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// The attention half of an encoder Block, BACKWARD, in one persistent kernel (d = 128, 8 heads, <= 32 tokens; round 4):
//   dO = dx1 Wp;  dq|dk|dv = attention backward per head;  du = dq|dk|dv Wqkv;  dx = dx1 + LayerNorm1-backward(du; x),  dgamma / dbeta
// (autograd of Models.py:303-304 with Attention.forward :192-219).  Replaced attn128_bwd_kernel + lnbwd_dma_kernel (rounds 1-3; removed in round 6): dq|dk|dv
// are written once (the q / k / v weight gradients' operand) and not read back, x and dx1 are read once.  Mirror image of
// blk128_fwd_kernel: wave h owns head h in the attention and output columns 16 h .. 16 h + 15 of both products, whose weight
// slices (4 + 12 packed fragments = 64 registers) stay in registers for the whole launch; two samples per iteration; the next
// group's inputs are fetched into registers during this one's phases.
// RC (the default schedule): q|k|v are RECOMPUTED from the saved LayerNorm output u (bf16, kept anyway for the q / k / v weight
// gradients) exactly as blk128_fwd_kernel computed them — same operands, same MFMA order, hence the same bf16 values — so the
// forward does not store them (85 MB per launch at C2) and this kernel reads 128 instead of 384 columns per row; the 12 weight
// fragments per wave come from L2 once per group.  !RC reads the q|k|v the forward saved.
// LDS layout of blk128_bwd_kernel (round 4; bank model scripts/micro/lds_banks.py, audit scripts/micro/lds_audit_blk128.py): unpadded
// 256-byte image rows, 16-byte chunk c of row r at chunk c ^ bsw(r).  Found by exhaustive search over the XOR-linear maps of the
// row's low 4 bits: 16-byte row fragments (8 -> 4 cycles per wave-instruction against the 272-byte pitch of the forward kernel's
// images), 8-byte head-row reads and the transposed reads of 8 consecutive rows are conflict-free for every head and k-step; the
// 8-byte head-row WRITES stay 2-way (16 rows x 8 bytes meet in one 32-bank half: no layout of 256-byte rows avoids it).
constexpr int BIR = 128;                   // image row (elements)
constexpr int BTS = 16;                    // P / dS transposition tile row (elements), chunks rotated by the row group (fused_dec.hip TTS)
struct Blk128BwdArgs {
    const bf16_t* qkv; const bf16_t* o; const float* lse;        // saved by the forward: [rows][384] (!RC) | [rows][128] | [rows][8]
    const bf16_t* u; const bf16_t* wqkv; const float* bqkv;      // RC: LayerNorm-1 output [rows][128], packed Wqkv [n = 384][k = 128], bias
    const bf16_t* dx1b; const float* dx1;                        // the projection's dY as bf16 (DropPath factor folded in) | the residual gradient
    const float* x; const float* gamma;                          // block input, LayerNorm-1 weight
    const bf16_t* wpT; const bf16_t* wqkvT;                      // packed images of Wp^T [n = 128][k = 128] and Wqkv^T [n = 128][k = 384]
    bf16_t* dqkv; float* dx; float* dgamma; float* dbeta;
    const float* det_base; long long* det_acc;
    int Ts, nsamples, mode, len_l, accumulate;
};

template <int NT, int SPW, bool RC>
struct LayBB {
    static constexpr int ROWS = NT * 16, RT = SPW * ROWS, IMG = RT * BIR;
    static constexpr int DUS = 132;                              // fp32 du tile row stride
    static constexpr int TT = 8 * 2 * 16 * BTS;                  // per-wave P / dS transposition tiles (elements)
    static constexpr int RED = 2 * 512 * 8 * 4;                  // final dgamma / dbeta reduction (bytes), over the du tile
    static constexpr int DUB = RT * DUS * 4 > RED ? RT * DUS * 4 : RED;
    // cls | Q K V dO dX | T | lse delta | du | gamma (+ bqkv) | RC: U
    static constexpr int TOTAL = RT * 4 + 5 * IMG * 2 + TT * 2 + 2 * 8 * RT * 4 + DUB + 512 * 4 + (RC ? IMG * 2 : 0);
};

template <int NT, int SPW, bool RC>
__global__ __launch_bounds__(512) void blk128_bwd_kernel(Blk128BwdArgs p) {
    using L = LayBB<NT, SPW, RC>;
    constexpr int ROWS = L::ROWS, RT = L::RT, MTT = SPW * NT, DUS = L::DUS;
    constexpr int PASSES = (RT * 16 + 511) / 512;               // wide layout: 16 lanes per row, 32 rows per pass
    constexpr int NQ = RC ? 0 : (RT * 48 + 511) / 512;          // saved q|k|v pieces per thread
    constexpr int NS = (RT * 48 + 511) / 512;                   // dq|dk|dv pieces per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, head = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int* cls = reinterpret_cast<int*>(smem);
    bf16_t* Qf = reinterpret_cast<bf16_t*>(smem + RT * 4);
    bf16_t* Kf = Qf + L::IMG;
    bf16_t* Vf = Kf + L::IMG;
    bf16_t* Df = Vf + L::IMG;                                   // O, then dO (per head in place)
    bf16_t* Xf = Df + L::IMG;                                   // dx1 rows (bf16)
    bf16_t* Tp = Xf + L::IMG + head * (2 * 16 * BTS);           // this wave's P / dS transposition tiles
    bf16_t* Td = Tp + 16 * BTS;
    float* lse_s = reinterpret_cast<float*>(Xf + L::IMG + L::TT);       // [8][RT]
    float* dlt_s = lse_s + 8 * RT;                              // [8][RT]
    float* DU = dlt_s + 8 * RT;                                 // [RT][DUS]
    float* gam_s = reinterpret_cast<float*>(reinterpret_cast<char*>(DU) + L::DUB);         // gamma[128] | bqkv[384]
    bf16_t* Uf = reinterpret_cast<bf16_t*>(gam_s + 512);        // RC: LayerNorm-1 output rows
    const int c16 = lane & 15, g = lane >> 4, q4 = c16 >> 2, p4 = c16 & 3, hc = head * 16;
    const float scale = 0.25f, sc = 0.25f * 1.4426950408889634f;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // image addressing (elements): 16-byte chunk c of row r sits at chunk c ^ bsw(r) of its unpadded 256-byte row
    const int fr = bsw(c16), ft = bsw(4 * g + q4);
    const int hcell = c16 * BIR + (((2 * head + (g >> 1)) ^ fr) << 3) + (g & 1) * 4;      // this head's 4 columns 4g.. of row c16 (+ 16 mt rows)
    const int troff = (4 * g + q4) * BIR + (((2 * head + (p4 >> 1)) ^ ft) << 3) + (p4 & 1) * 4;
    const int tw = c16 * BTS + (((g + q4) & 3) << 2);             // tile write: row c16, keys 4g.. at chunk (g + (c16 >> 2)) & 3
    const int ttoff = (4 * g + q4) * BTS + (((p4 + g) & 3) << 2); // tile transpose read: row 4g + q4, chunk p4
    // MFMA operand, row 16 mt + c16, columns 32 ks + 8 g ..: chunk (4 ks + g) ^ fr = 4 (ks ^ (fr >> 2)) + (g ^ (fr & 3)), i.e. the
    // k-step only flips bits 5-6 of ONE per-lane offset.  `fa` is re-materialised (an opaque copy) in every phase that uses it:
    // left to itself hipcc keeps the four per-k-step offsets of every image base alive across the whole group (57 spills)
    const int fa0 = c16 * BIR + ((fr >> 2) << 5) + ((g ^ (fr & 3)) << 3);
    auto frag = [&](const bf16_t* img, int fa, int mt, int ks) -> const bf16x8* {
        return reinterpret_cast<const bf16x8*>(img + mt * 16 * BIR + (fa ^ (ks << 5)));
    };
    auto fresh = [](int v) { asm volatile("" : "+v"(v)); return v; };
    auto wide = [&](int irow, int pc) -> int { return irow * BIR + ((pc ^ bsw(irow)) << 3); };     // 16-byte piece pc of row irow

    if (HS_BLK_PRIO > 0 && head >= 4) __builtin_amdgcn_s_setprio(HS_BLK_PRIO);
    // this wave's weights: n-tile `head` of Wp^T (dO columns of its head) and of Wqkv^T (its 16 du columns)
    bf16x8 wo[4], wu[12];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wo[ks] = *reinterpret_cast<const bf16x8*>(p.wpT + ((size_t)(head * 4 + ks) * 64 + lane) * 8);
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) wu[ks] = *reinterpret_cast<const bf16x8*>(p.wqkvT + ((size_t)(head * 12 + ks) * 64 + lane) * 8);
    float dgam[8], dbet[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { dgam[e] = 0.f; dbet[e] = 0.f; }
    if (threadIdx.x < 128) gam_s[threadIdx.x] = p.gamma[threadIdx.x];
    else if (RC) gam_s[threadIdx.x] = p.bqkv[threadIdx.x - 128];
    for (int i = threadIdx.x; i < RT; i += 512) {
        const int slot = i / ROWS, r = i - slot * ROWS;
        int c = -1;
        if (r < p.Ts) c = slot * 64 + ((p.mode == 1) ? r / p.len_l : (p.mode == 2) ? r % p.len_l : 0);
        cls[i] = c;
    }
    auto grow = [&](int first, int irow) -> int64_t {            // global row of image row irow (slot-major) of the group, or -1
        const int slot = irow / ROWS, r = irow - slot * ROWS;
        return (first + slot < p.nsamples && r < p.Ts) ? (int64_t)(first + slot) * p.Ts + r : -1;
    };
    // a group's inputs as register pieces: q|k|v (NQ), O and dx1 (PASSES each), lse (one float per (row, head) pair)
    struct Pre { bf16x8 q[NQ ? NQ : 1], o[PASSES], d[PASSES], u[RC ? PASSES : 1]; float l[(RT * 8 + 511) / 512]; };
    // (issued in three parts spread over the group's compute phases: every CU bursting its ~200 KB per group at once runs into the
    //  memory system's back-pressure, and a wave blocked at issue computes nothing: 26 % of the kernel in the first version)
    auto fetch_q = [&](int first, Pre& r, int tx, int i0, int i1) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            if (i < i0 || i >= i1) continue;
            const int idx = tx + 512 * i, irow = idx / 48, pc = idx - irow * 48;
            const int64_t gr = irow < RT ? grow(first, irow) : -1;
            r.q[i] = gr >= 0 ? *reinterpret_cast<const bf16x8*>(p.qkv + gr * 384 + pc * 8) : zero8();
        }
    };
    auto fetch_u = [&](int first, Pre& r, int tx) {
        const int lc8 = (tx & 15) * 8;
#pragma unroll
        for (int i = 0; i < (RC ? PASSES : 0); ++i) {
            const int irow = i * 32 + (tx >> 4);
            const int64_t gr = irow < RT ? grow(first, irow) : -1;
            r.u[i] = gr >= 0 ? *reinterpret_cast<const bf16x8*>(p.u + gr * 128 + lc8) : zero8();
        }
    };
    auto fetch_rest = [&](int first, Pre& r, int tx) {
        const int lc8 = (tx & 15) * 8;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int irow = i * 32 + (tx >> 4);
            const int64_t gr = irow < RT ? grow(first, irow) : -1;
#if !HS_BB_DELTA_PDP
            r.o[i] = gr >= 0 ? *reinterpret_cast<const bf16x8*>(p.o + gr * 128 + lc8) : zero8();
#endif
            r.d[i] = gr >= 0 ? *reinterpret_cast<const bf16x8*>(p.dx1b + gr * 128 + lc8) : zero8();
        }
#pragma unroll
        for (int i = 0; i < (RT * 8 + 511) / 512; ++i) {
            const int idx = tx + 512 * i, irow = idx >> 3;
            const int64_t gr = irow < RT ? grow(first, irow) : -1;
            r.l[i] = gr >= 0 ? p.lse[gr * 8 + (idx & 7)] : 1e30f;             // padding rows: exp2(s - 1e30) = 0
        }
    };
    auto commit = [&](const Pre& r, int tx) {
        const int lc8 = (tx & 15) * 8;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int idx = tx + 512 * i, irow = idx / 48, pc = idx - irow * 48;
            if (irow < RT) *reinterpret_cast<bf16x8*>(Qf + (pc >> 4) * L::IMG + wide(irow, pc & 15)) = r.q[i];
        }
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int irow = i * 32 + (tx >> 4);
            if (irow < RT) {
#if !HS_BB_DELTA_PDP
                *reinterpret_cast<bf16x8*>(Df + wide(irow, tx & 15)) = r.o[i];
#endif
                *reinterpret_cast<bf16x8*>(Xf + wide(irow, tx & 15)) = r.d[i];
                if constexpr (RC) *reinterpret_cast<bf16x8*>(Uf + wide(irow, tx & 15)) = r.u[i];
            }
        }
#pragma unroll
        for (int i = 0; i < (RT * 8 + 511) / 512; ++i) {
            const int idx = tx + 512 * i, irow = idx >> 3;
            if (irow < RT) lse_s[(idx & 7) * RT + irow] = r.l[i];
        }
    };
    {
        Pre r0;
        fetch_q(blockIdx.x * SPW, r0, threadIdx.x, 0, NQ);
        fetch_u(blockIdx.x * SPW, r0, threadIdx.x);
        fetch_rest(blockIdx.x * SPW, r0, threadIdx.x);
        lds_barrier();                                            // cls visible
        commit(r0, threadIdx.x);
    }
#if HS_BB_WQ_RESIDENT
    // RC: the 12 fragments of Wqkv this wave's head needs (n-tiles head, 8 + head, 16 + head) stay in registers too: streamed per
    // group they cost the kernel more memory-pipeline time (96 KB per group and CU from L2) than the q|k|v reads they replace
    bf16x8 wq[RC ? 3 : 1][4];
    if constexpr (RC) {
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wq[m][ks] = *reinterpret_cast<const bf16x8*>(p.wqkv + ((size_t)((m * 8 + head) * 4 + ks) * 64 + lane) * 8);
    }
#endif
    PHB_DECL
    if (HS_BB_EARLY_WAIT) __builtin_amdgcn_s_waitcnt(0x0F70);       // the resident weight fragments: no load pending on EITHER way into the loop head
    for (int first = blockIdx.x * SPW; first < p.nsamples; first += gridDim.x * SPW) {
#if !HS_BB_WQ_RESIDENT
        // RC: this group's 12 fragments of Wqkv (n-tiles head, 8 + head, 16 + head), in flight across the barrier.  The image
        // pointer goes through an opaque copy: the loads are loop-invariant, and hoisted they would hold 48 registers for the
        // whole launch
        bf16x8 wq[RC ? 3 : 1][4];
        if constexpr (RC) {
            const bf16_t* wsrc = p.wqkv;
            asm volatile("" : "+s"(wsrc));
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wq[m][ks] = *reinterpret_cast<const bf16x8*>(wsrc + ((size_t)((m * 8 + head) * 4 + ks) * 64 + lane) * 8);
        }
#endif
        lds_barrier();                                            // the group's images are complete
        PHB(0)
        if constexpr (RC) {
            // ---- q | k | v of this head from U (as blk128_fwd_kernel): transposed accumulators -> 8-byte writes into the images
            const int fa = fresh(fa0);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                bf16_t* img = Qf + m * L::IMG;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(gam_s + 128 + m * 128 + hc + 4 * g);
#pragma unroll
                for (int mt = 0; mt < MTT; ++mt) {
                    f32x4 acc = bias;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        acc = mfma16(wq[m][ks], *frag(Uf, fa, mt, ks), acc);
                    *reinterpret_cast<bf16x4*>(img + mt * 16 * BIR + hcell) = cvt4(acc);
                }
            }
        }
        // (every per-thread index below derives from a copy of the thread id the compiler cannot see through: hoisted out of
        //  the sample loop, these ~40 addresses and row numbers spill; recomputed, they cost ~100 VALU instructions per group)
        int tx = threadIdx.x;
        asm volatile("" : "+v"(tx));
        const int lc8 = (tx & 15) * 8;
        // ---- the next group's inputs start to fly (three parts spread over this group's phases)
        Pre nx;
        const int nfirst = first + gridDim.x * SPW;
        const bool more = nfirst < p.nsamples;
        if (more) { fetch_q(nfirst, nx, tx, 0, NQ / 2); fetch_u(nfirst, nx, tx); }
        PHB(1)
        // ---- dO[:, this head's columns] = dx1 Wp, delta = rowsum(dO * O) of this head; dO replaces O in place (own columns)
        const int fa_o = fresh(fa0);
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) {
            f32x4 acc = z4;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = mfma16(wo[ks], *frag(Xf, fa_o, mt, ks), acc);
            bf16_t* cell = Df + mt * 16 * BIR + hcell;
#if HS_BB_DELTA_PDP
            *reinterpret_cast<bf16x4*>(cell) = cvt4(acc);          // delta comes out of the core (sum_j P_ij dP_ij): O is not read at all
#else
            const bf16x4 o4 = *reinterpret_cast<const bf16x4*>(cell), dob = cvt4(acc);
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) v += bf2f(dob[r]) * bf2f(o4[r]);
            v = rows_sum(v);
            if (g == 0) dlt_s[head * RT + mt * 16 + c16] = v;
            *reinterpret_cast<bf16x4*>(cell) = dob;
#endif
        }
        if (more) fetch_q(nfirst, nx, tx, NQ / 2, NQ);
        PHB(2)
        // class / padding mask of a (query tile, key tile) pair, as the accumulator the score MFMA starts from (0 or -inf): rebuilt
        // from LDS per group — 16 registers that would otherwise live (or spill) through the product / epilogue phases
        f32x4 cm[NT][NT];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            const int qc = cls[qt * 16 + c16];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const int4 kc4 = *reinterpret_cast<const int4*>(cls + kt * 16 + g * 4);
                const int kcl[4] = {kc4.x, kc4.y, kc4.z, kc4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) cm[qt][kt][r] = (kcl[r] >= 0 && kcl[r] == qc) ? 0.f : -INFINITY;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // own writes before own reads
        // ---- attention backward of this head (single pass, see attn16_bwd_kernel), slot by slot; dq, dk, dv in place
        const float* lse_h = lse_s + head * RT;
#if !HS_BB_DELTA_PDP
        const float* dlt_h = dlt_s + head * RT;
#endif
#pragma unroll
        for (int slot = 0; slot < SPW; ++slot) {
            const int r0 = slot * ROWS;
            f32x4 dkT[NT], dvT[NT];
            bf16x4 KT[NT];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) { dkT[kt] = z4; dvT[kt] = z4; KT[kt] = tr4(Kf + (r0 + kt * 16) * BIR + troff); }
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                if (qt * 16 >= p.Ts) break;
                const int query = r0 + qt * 16 + c16;
                const int qcell = (r0 + qt * 16) * BIR + hcell;
                const bf16x4 bq = *reinterpret_cast<const bf16x4*>(Qf + qcell);
                const bf16x4 bdo = *reinterpret_cast<const bf16x4*>(Df + qcell);
                const float lqn = -lse_h[query];
                const bf16x4 QT = tr4(Qf + (r0 + qt * 16) * BIR + troff);
                const bf16x4 DT = tr4(Df + (r0 + qt * 16) * BIR + troff);
                f32x4 dqT = z4;
#if HS_BB_DELTA_PDP
                // delta_i = sum_j P_ij dP_ij (= rowsum(dO * O), the form the separate kernel uses, without needing O): with at most
                // NT = 2 key tiles per query tile both P and dP are in hand before dS is formed
                f32x4 pvs[NT], dps[NT];
                float dl = 0.f;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    const bf16x4 ak = *reinterpret_cast<const bf16x4*>(Kf + (r0 + kt * 16) * BIR + hcell);
                    const bf16x4 av = *reinterpret_cast<const bf16x4*>(Vf + (r0 + kt * 16) * BIR + hcell);
                    const f32x4 sv = mfma_k16(ak, bq, cm[qt][kt]);
                    dps[kt] = mfma_k16(av, bdo, z4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pvs[kt][r] = __builtin_amdgcn_exp2f(fmaf(sv[r], sc, lqn));
                        dl = fmaf(pvs[kt][r], dps[kt][r], dl);
                    }
                }
                dl = rows_sum(dl);
#else
                const float dl = dlt_h[query];
#endif
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    f32x4 pv, ds;
#if HS_BB_DELTA_PDP
                    pv = pvs[kt];
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds[r] = pv[r] * (dps[kt][r] - dl);
#else
                    const bf16x4 ak = *reinterpret_cast<const bf16x4*>(Kf + (r0 + kt * 16) * BIR + hcell);
                    const bf16x4 av = *reinterpret_cast<const bf16x4*>(Vf + (r0 + kt * 16) * BIR + hcell);
                    const f32x4 sv = mfma_k16(ak, bq, cm[qt][kt]);
                    const f32x4 dp = mfma_k16(av, bdo, z4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pv[r] = __builtin_amdgcn_exp2f(fmaf(sv[r], sc, lqn));
                        ds[r] = pv[r] * (dp[r] - dl);
                    }
#endif
                    const bf16x4 pb = cvt4(pv), dsb = cvt4(ds);
                    dqT = mfma_k16(KT[kt], dsb, dqT);
                    *reinterpret_cast<bf16x4*>(Tp + tw) = pb;
                    *reinterpret_cast<bf16x4*>(Td + tw) = dsb;
                    asm volatile("" ::: "memory");
                    const bf16x4 Bp = tr4(Tp + ttoff), Bds = tr4(Td + ttoff);
                    dkT[kt] = mfma_k16(QT, Bds, dkT[kt]);
                    dvT[kt] = mfma_k16(DT, Bp, dvT[kt]);
                }
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (bf16_t)(dqT[r] * scale);
                *reinterpret_cast<bf16x4*>(Qf + qcell) = v;
            }
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const int kcell = (r0 + kt * 16) * BIR + hcell;
                bf16x4 vk, vv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { vk[r] = (bf16_t)(dkT[kt][r] * scale); vv[r] = (bf16_t)dvT[kt][r]; }
                *reinterpret_cast<bf16x4*>(Kf + kcell) = vk;
                *reinterpret_cast<bf16x4*>(Vf + kcell) = vv;
            }
            if (slot == 0 && more) fetch_rest(nfirst, nx, tx);
        }
        // ---- the rows the LayerNorm epilogue needs (wide layout) fly during the row stores and the du product
        float xr[PASSES][8], rs[PASSES][8];
        int64_t erow[PASSES];
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int irow = ps * 32 + (tx >> 4);
            erow[ps] = irow < RT ? grow(first, irow) : -1;
#pragma unroll
            for (int e = 0; e < 8; ++e) { xr[ps][e] = 0.f; rs[ps][e] = 0.f; }
            if (erow[ps] >= 0) {
                const float* xp = p.x + erow[ps] * 128 + lc8;
                const float* rp = p.dx1 + erow[ps] * 128 + lc8;
                const float4 a0 = *reinterpret_cast<const float4*>(xp), a1 = *reinterpret_cast<const float4*>(xp + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(rp), b1 = *reinterpret_cast<const float4*>(rp + 4);
                xr[ps][0] = a0.x; xr[ps][1] = a0.y; xr[ps][2] = a0.z; xr[ps][3] = a0.w; xr[ps][4] = a1.x; xr[ps][5] = a1.y; xr[ps][6] = a1.z; xr[ps][7] = a1.w;
                rs[ps][0] = b0.x; rs[ps][1] = b0.y; rs[ps][2] = b0.z; rs[ps][3] = b0.w; rs[ps][4] = b1.x; rs[ps][5] = b1.y; rs[ps][6] = b1.z; rs[ps][7] = b1.w;
                if (p.accumulate) {
                    const float* op = p.dx + erow[ps] * 128 + lc8;
                    const float4 c0 = *reinterpret_cast<const float4*>(op), c1 = *reinterpret_cast<const float4*>(op + 4);
                    rs[ps][0] += c0.x; rs[ps][1] += c0.y; rs[ps][2] += c0.z; rs[ps][3] += c0.w; rs[ps][4] += c1.x; rs[ps][5] += c1.y; rs[ps][6] += c1.z; rs[ps][7] += c1.w;
                }
            }
        }
        PHB(3)
        lds_barrier();                                            // dq | dk | dv images complete
        PHB(4)
        // ---- dq|dk|dv leave as whole rows (the q / k / v weight gradients' operand), a few pieces in front of every m-tile of
        //      du[:, this wave's 16 columns] = dq|dk|dv Wqkv (contraction over the 384 image columns) -> fp32 tile
        auto store_q = [&](int i0, int i1) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                if (i < i0 || i >= i1) continue;
                const int idx = tx + 512 * i, irow = idx / 48, pc = idx - irow * 48;
                const int64_t gr = irow < RT ? grow(first, irow) : -1;
                if (gr >= 0)
                    HS_NT(HS_NT_E, reinterpret_cast<bf16x8*>(p.dqkv + gr * 384 + pc * 8),
                          *reinterpret_cast<const bf16x8*>(Qf + (pc >> 4) * L::IMG + wide(irow, pc & 15)));
            }
        };
        PHB(5)
        const int fa_u = fresh(fa0);
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) {
            store_q(mt * NS / MTT, (mt + 1) * NS / MTT);
            f32x4 acc = z4;
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    acc = mfma16(wu[m * 4 + ks], *frag(Qf + m * L::IMG, fa_u, mt, ks), acc);
            *reinterpret_cast<f32x4*>(DU + (mt * 16 + c16) * DUS + hc + 4 * g) = acc;
            asm volatile("" ::: "memory");
        }
        PHB(6)
        lds_barrier();                                            // du tile complete; every read of the images is done
        PHB(7)
        // One explicit wait for everything in flight — this group's epilogue rows, the next group's inputs (issued a phase or more
        // ago) and the dq|dk|dv row stores — BEFORE the first dx store (round 6).  gfx950 counts loads and stores in one in-order
        // counter and the accesses here sit under per-row predicates, so hipcc fell back to s_waitcnt vmcnt(0) in front of the second
        // pass's arithmetic and again in front of commit(): each waited for the dx stores issued just before it, i.e. for their
        // trip to HBM (2 x ~1 us per 11-us group).  A builtin wait is visible to the wait-count pass: after it nothing older than
        // the dx stores is pending, and they drain under the next group's first phases.
        if (HS_BB_EARLY_WAIT) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), lgkmcnt / expcnt untouched
        // ---- LayerNorm-1 backward + residual gradient, wide layout (16 lanes per row)
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int irow = ps * 32 + (tx >> 4);
            if (irow < RT) {
                const float4 t0 = *reinterpret_cast<const float4*>(DU + irow * DUS + lc8);
                const float4 t1 = *reinterpret_cast<const float4*>(DU + irow * DUS + lc8 + 4);
                const float du[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                float sm = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += xr[ps][e];
                sm = lanes_sum<16>(sm);
                const float mean = sm * (1.f / 128.f);
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { xr[ps][e] -= mean; q += xr[ps][e] * xr[ps][e]; }
                q = lanes_sum<16>(q);
                const float rstd = rsqrtf(q * (1.f / 128.f) + 1e-5f);
                const float4 g0 = *reinterpret_cast<const float4*>(gam_s + lc8), g1 = *reinterpret_cast<const float4*>(gam_s + lc8 + 4);
                const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                float a = 0.f, b = 0.f, t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { xr[ps][e] *= rstd; t[e] = du[e] * gm[e]; a += t[e]; b += t[e] * xr[ps][e]; }
                a = lanes_sum<16>(a); b = lanes_sum<16>(b);
                a *= (1.f / 128.f); b *= (1.f / 128.f);
                if (erow[ps] >= 0) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] = rs[ps][e] + rstd * (t[e] - a - xr[ps][e] * b);
                        dgam[e] += du[e] * xr[ps][e];
                        dbet[e] += du[e];
                    }
                    float* op = p.dx + erow[ps] * 128 + lc8;
                    *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            }
        }
        PHB(8)
        if (more) commit(nx, tx);                                     // images / lse of the next group (their readers are past the barrier above)
        PHB(9)
    }
    PHB_FLUSH()
    // dgamma / dbeta: 32 threads per column octet -> LDS, one commit per column and workgroup
    lds_barrier();
    float* red = DU;
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x * 8 + e] = dgam[e]; red[4096 + threadIdx.x * 8 + e] = dbet[e]; }
    lds_barrier();
    if (threadIdx.x < 256) {
        const int which = threadIdx.x >> 7, c = threadIdx.x & 127, o8 = c >> 3, e = c & 7;
        float sacc = 0.f;
        for (int t2 = o8; t2 < 512; t2 += 16) sacc += red[which * 4096 + t2 * 8 + e];
        hs_gadd(HsDet{p.det_base, p.det_acc}, (which ? p.dbeta : p.dgamma) + c, sacc);
    }
}

template <int NT, int SPW, bool RC>
int launch_blk128_bwd(const Blk128BwdArgs& a, hipStream_t s) {
    using L = LayBB<NT, SPW, RC>;
    static_assert(L::TOTAL <= 160 * 1024, "LDS");
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blk128_bwd_kernel<NT, SPW, RC>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::TOTAL); attr_set = true; }
    static int wgs = 0;
    if (!wgs) wgs = 256;
    const int groups = (a.nsamples + SPW - 1) / SPW;
    hipLaunchKernelGGL((blk128_bwd_kernel<NT, SPW, RC>), dim3(groups < wgs ? groups : wgs), dim3(512), (size_t)L::TOTAL, s, a);
    return (int)hipGetLastError();
}

template <int NT, bool BWD, int HD = 16, int HPW = 4>
int launch_attn16(const AttnParams& p, hipStream_t s) {
    using L = Lay16<NT, HD, HPW>;
    const int hgroups = (p.heads + HPW - 1) / HPW;
    const size_t lds = L::CLS + HPW * (size_t)(BWD ? L::BWD_WAVE : L::FWD_WAVE) + 32;      // + the last transposed read's overhang at HD = 8
    if (lds > 160 * 1024) return HS_EUNSUPPORTED;
    static bool attr_set = false;
    const dim3 grid(p.nsamples * hgroups), blk(64 * HPW);
    if constexpr (BWD) {
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn16_bwd_kernel<NT, HD, HPW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn16_bwd_kernel<NT, HD, HPW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        if (p.mode == 0) hipLaunchKernelGGL((attn16_bwd_kernel<NT, HD, HPW, true>), grid, blk, lds, s, p);
        else hipLaunchKernelGGL((attn16_bwd_kernel<NT, HD, HPW, false>), grid, blk, lds, s, p);
    } else {
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn16_fwd_kernel<NT, HD, HPW, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn16_fwd_kernel<NT, HD, HPW, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        if (p.mode == 0) hipLaunchKernelGGL((attn16_fwd_kernel<NT, HD, HPW, true>), grid, blk, lds, s, p);
        else hipLaunchKernelGGL((attn16_fwd_kernel<NT, HD, HPW, false>), grid, blk, lds, s, p);
    }
    return (int)hipGetLastError();
}

// Stand-alone attention (the layer-at-a-time schedule of every width; at d = 128 the fused attention half, blk128_fwd / blk128_bwd,
// is what a default pass runs).  One kernel family since round 6: per-head workgroups (attn16_*), head dim 16 as it is and head
// dim 8 zero-extended to the K = 16 MFMA.  The first-generation kernels (head dim as the padded K = 32 contraction, images stored
// twice) and the whole-sample d = 128 kernels with the projection fused in (attn128_*: superseded by blk128_*) were removed with
// their switches (HSIMAE_ATTN16_V2, _ATTN8_V2, _ATTN128_V3, _FUSED_PROJ): git history at 96065d4.
template <bool BWD>
int dispatch(const AttnParams& p, hipStream_t s) {
    if (p.nsamples <= 0) return HS_OK;
    if (p.d != p.heads * p.hd || p.ld % 8 || p.ldo % 4) return HS_EDIMS;
    if (BWD && !p.lse) return HS_EUNSUPPORTED;               // the backward needs the forward's logsumexp
    const int nt = (p.Ts + 15) / 16;
    if (p.hd == 16) {
        if (nt <= 1) return launch_attn16<1, BWD>(p, s);
        if (nt <= 2) return launch_attn16<2, BWD>(p, s);
        if (nt <= 3) return launch_attn16<3, BWD>(p, s);
        if (nt <= 4) return launch_attn16<4, BWD>(p, s);
        if (nt <= 7) return launch_attn16<7, BWD>(p, s);
    } else if (p.hd == 8) {                                   // head dim zero-extended to 16
        if (nt <= 4) return launch_attn16<4, BWD, 8, 4>(p, s);
        if (nt <= 7) return launch_attn16<7, BWD, 8, 4>(p, s);
        if (nt <= 14) return launch_attn16<14, BWD, 8, 4>(p, s);
    }
    return HS_EUNSUPPORTED;
}

}  // namespace

int hs_attn_fwd(const AttnParams& p, hipStream_t s) { return dispatch<false>(p, s); }
// shape predicate of the fused attention half at d = 128 (blk128_fwd / blk128_bwd): 8 heads of 16, <= 32 tokens, unpadded rows
bool hs_attn_proj_fusable(const AttnParams& p) {
    return p.lse && p.d == 128 && p.heads == 8 && p.hd == 16 && p.Ts <= 32 && p.ld == 384 && p.ldo == 128;
}
int hs_attn_bwd(const AttnParams& p, hipStream_t s) { return dispatch<true>(p, s); }

// LN1 + q|k|v + attention + projection + residual in one launch (see blk128_fwd_kernel).  Shape predicate only: whether a pass
// uses it is part of the schedule word its forward records (api.hip SC_ATTN_BLOCK, HSIMAE_FUSED_ATTN_BLOCK=0 clears it).
bool hs_attn_block_fusable(int d, int heads, int Ts) {
    AttnParams q = AttnParams();
    q.d = d; q.heads = heads; q.hd = heads ? d / heads : 0; q.Ts = Ts; q.ld = 384; q.ldo = 128; q.lse = reinterpret_cast<float*>(1);
    return hs_attn_proj_fusable(q);
}
int hs_attn_block_fwd(const float* x, const float* n1w, const float* n1b, const hs_bf16* wqkv, const float* bqkv, const hs_bf16* wp,
                      const float* pb, hs_bf16* u, hs_bf16* qkv, hs_bf16* o, float* lse, float* x1, const float* rowscale, int Ts,
                      int nsamples, int mode, int len_l, hipStream_t s) {
    if (nsamples <= 0) return HS_OK;
    if (Ts < 1 || Ts > 32) return HS_EUNSUPPORTED;
    Blk128Args a;
    a.x = x; a.n1w = n1w; a.n1b = n1b; a.wqkv = wqkv; a.bqkv = bqkv; a.wp = wp; a.pb = pb; a.u = u; a.qkv = qkv; a.o = o;
    a.lse = lse; a.x1 = x1; a.rowscale = rowscale; a.Ts = Ts; a.nsamples = nsamples; a.mode = mode; a.len_l = len_l;
    // two samples in hand per iteration (one: 84 instead of 74 us in round 3; three measured 0.5 % slower than two)
    return Ts <= 16 ? launch_blk128<1, 2>(a, s) : launch_blk128<2, 2>(a, s);
}

// dO + attention backward + du + LayerNorm-1 backward in one launch (see blk128_bwd_kernel).  Shape predicate only (api.hip
// SC_ATTN_BLOCK_BWD, HSIMAE_FUSED_ATTN_BLOCK_BWD=0 clears it in the forward's schedule word).
bool hs_attn_block_bwd_fusable(int d, int heads, int Ts) { return hs_attn_block_fusable(d, heads, Ts); }
int hs_attn_block_bwd(const hs_bf16* qkv, const hs_bf16* u, const hs_bf16* wqkv, const float* bqkv, const hs_bf16* o, const float* lse,
                      const hs_bf16* dx1b, const float* dx1, const float* x, const float* gamma, const hs_bf16* wpT, const hs_bf16* wqkvT,
                      hs_bf16* dqkv, float* dx, float* dgamma, float* dbeta, const float* det_base, long long* det_acc, int Ts,
                      int nsamples, int mode, int len_l, int accumulate, hipStream_t s) {
    if (nsamples <= 0) return HS_OK;
    if (Ts < 1 || Ts > 32) return HS_EUNSUPPORTED;
    const bool rc = qkv == nullptr;                  // no saved q|k|v: recompute them from u
    if (rc && !(u && wqkv && bqkv)) return HS_EUNSUPPORTED;
    Blk128BwdArgs a;
    a.qkv = qkv; a.u = u; a.wqkv = wqkv; a.bqkv = bqkv; a.o = o; a.lse = lse; a.dx1b = dx1b; a.dx1 = dx1; a.x = x; a.gamma = gamma;
    a.wpT = wpT; a.wqkvT = wqkvT; a.dqkv = dqkv; a.dx = dx; a.dgamma = dgamma; a.dbeta = dbeta; a.det_base = det_base; a.det_acc = det_acc;
    a.Ts = Ts; a.nsamples = nsamples; a.mode = mode; a.len_l = len_l; a.accumulate = accumulate;
    if (rc) return Ts <= 16 ? launch_blk128_bwd<1, 2, true>(a, s) : launch_blk128_bwd<2, 2, true>(a, s);
    return Ts <= 16 ? launch_blk128_bwd<1, 2, false>(a, s) : launch_blk128_bwd<2, 2, false>(a, s);
}

HS_UNIT_VARIANT_BITS(attn)
