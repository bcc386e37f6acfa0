// The attention half of an encoder Block at D = 256 (HSIMAE-Large: 16 heads of 16, <= 32 kept tokens per sample) in ONE
// persistent kernel — the wide counterpart of blk128_fwd_kernel (attn.hip):
//   u = LN1(x);  q|k|v = u Wqkv^T + b;  o = softmax(q k^T / 4) v per head;  x1 = x + rs * (o Wp^T + bp)
// (Models.py:303-304 with Attention.forward :192-219 at enc_paras = [12, 256, 9], Model_Pretraining.py:130).
//
// Round 3 ran this half layer at a time at D = 256: LN1 + q|k|v GEMM (130 us), attn16_fwd (55 us), proj + residual GEMM (77 us)
// = 262 us per block with q|k|v and o making a round trip through HBM between the launches (byte floor of the three at
// 5 TB/s: 68 + 45 + 57 us).  Why the D = 128 form does not carry over: there wave h keeps head h's 48 columns of Wqkv and 16 of Wp
// in 64 registers for the whole launch; at D = 256 that is 128 registers for 16 waves (4 per SIMD => 128 in all).  Here:
//   * one workgroup = 16 waves = 16 heads, walking groups of SPW = 2 samples (64 image rows, 4 m-tiles);
//   * the weights are STREAMED: wave h fetches its three 16-column n-tiles of Wqkv one k-step (3 KB) ahead of the MFMAs that
//     use it and multiplies it against all 4 m-tiles of the group — 512 KB of L2 -> register traffic per group, 4 MB per CU and
//     launch, against ~0.5 GB of HBM traffic for the whole launch;
//   * operands swapped (weights as A): a lane owns 4 consecutive head dims of one token, which IS the A / B fragment of the
//     K = 16 MFMA whose contraction runs over the head dim — q^T and k^T go from the accumulators into the score MFMAs without
//     touching LDS; q | k | v are also written (8-byte pieces) into a staging image from which (a) V^T is read back with transpose
//     reads and (b) the rows leave for HBM as whole 1,536-byte rows for the backward;
//   * the attention output goes into the image LN1 occupied (dead once every wave has its q | k | v), the projection reads it
//     as row fragments; x is read once for LayerNorm and once more (L2-hot) as the residual.
// LDS: cls 0.25 + U/O image 34.8 + staging 99.3 + lse 4 + vectors 6 = 144.4 KB => one workgroup per CU, 4 waves per SIMD.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

#ifndef HS_NT_C
#define HS_NT_C 1      /* u / q|k|v / o saved for the backward: streaming stores (as blk128_fwd) */
#endif
#ifndef HS_W256_PF2
#define HS_W256_PF2 0  /* q|k|v weight fragments two k-steps ahead instead of one (12 more registers) */
#endif
#ifndef HS_W256_LATE_X
#define HS_W256_LATE_X 1 /* the next group's x rows are fetched after the attention (16 registers less across it) instead of before it */
#endif
// (start stagger of every second workgroup, as blk256_bwd_kernel below: 8.5 us 177.0 vs 176.5, 17 us 181.6 us — neutral to worse,
//  profiles/r05_m_large_stagger.txt; removed)
#ifndef HS_W256_EARLY
#define HS_W256_EARLY 0 /* projection weights (first half) and residual pieces fetched in front of the attention */
#endif

// Per-phase cycle accounting for scripts/phase_wide.py (compiled only with -DHS_PHASE_TIMING; never in the shipped library)
#ifdef HS_PHASE_TIMING
__device__ unsigned long long hs_phase_cycles_wide[16];
extern "C" __attribute__((visibility("default"))) int hsimae_debug_phases_wide(unsigned long long* out, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hs_phase_cycles_wide), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(hs_phase_cycles_wide), z, sizeof(z)); }
    return rc;
}
#define PH_DECL unsigned long long ph_t0 = __builtin_readcyclecounter(), ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PH(i) { const unsigned long long ph_t = __builtin_readcyclecounter(); ph_acc[i] += ph_t - ph_t0; ph_t0 = ph_t; }
#define PH_FLUSH() if (threadIdx.x == 64 * 5) { for (int i = 0; i < 12; ++i) atomicAdd(&hs_phase_cycles_wide[i], ph_acc[i]); }
#else
#define PH_DECL
#define PH(i)
#define PH_FLUSH()
#endif

namespace {

constexpr int DW = 256, HW = 16, HDW = 16;          // width, heads, head dim
constexpr int PU = DW + 16;                         // U / O image row pitch (elements): 16 mod 64 elements => conflict-free 16-byte row fragments
constexpr int PS = 3 * DW + 8;                      // staging image row pitch: 8-byte tile writes 2-way, transpose reads 2-way (scripts/micro/lds_banks.py)
constexpr int NTHW = 1024;
constexpr int KSW = DW / 32;                        // k-steps over the model width

typedef __attribute__((ext_vector_type(4))) short s16x4w;
typedef __attribute__((address_space(3))) bf16x4* lds_b64w;
__device__ __forceinline__ f32x4 mfma_k16w(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4w, a), __builtin_bit_cast(s16x4w, b), c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 tr4w(const bf16_t* a) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64w)(a)); }
__device__ __forceinline__ bf16x4 cvt4w(f32x4 v) {
    bf16x4 r;
    r[0] = (bf16_t)v[0]; r[1] = (bf16_t)v[1]; r[2] = (bf16_t)v[2]; r[3] = (bf16_t)v[3];
    return r;
}

struct Blk256Args {
    const float* x; const float* n1w; const float* n1b;
    const bf16_t* wqkv; const float* bqkv; const bf16_t* wp; const float* pb;
    bf16_t* u; bf16_t* qkv; bf16_t* o; float* lse; float* x1; const float* rowscale;
    int Ts, nsamples, mode, len_l;
};

template <int NT, int SPW>
struct LayW {
    static constexpr int ROWS = NT * 16;                 // rows of one slot (sample)
    static constexpr int RT = SPW * ROWS;                // rows of the images
    static constexpr int OFF_U = RT * 4;                 // after cls
    static constexpr int OFF_S = OFF_U + RT * PU * 2;
    static constexpr int OFF_LSE = OFF_S + RT * PS * 2;
    static constexpr int OFF_VEC = OFF_LSE + RT * HW * 4;
    static constexpr int TOTAL = OFF_VEC + (2 * DW + 3 * DW + DW) * 4;      // gamma | beta | bqkv | bp
    static_assert(TOTAL <= 160 * 1024, "LDS");
};

template <int NT, int SPW>
__global__ __launch_bounds__(NTHW, 1) void blk256_fwd_kernel(Blk256Args p) {
    using L = LayW<NT, SPW>;
    constexpr int ROWS = L::ROWS, RT = L::RT, MTT = SPW * NT;
    constexpr int PASSES = (RT * 32 + NTHW - 1) / NTHW;           // LayerNorm passes: 32 lanes per row, 32 rows per pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, head = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int* cls = reinterpret_cast<int*>(smem);
    bf16_t* Uf = reinterpret_cast<bf16_t*>(smem + L::OFF_U);      // LN1 image, later the attention output image
    bf16_t* Sf = reinterpret_cast<bf16_t*>(smem + L::OFF_S);      // q | k | v rows [RT][PS]
    float* lse_s = reinterpret_cast<float*>(smem + L::OFF_LSE);   // [RT][16]
    float* vec_s = reinterpret_cast<float*>(smem + L::OFF_VEC);   // gamma[256] | beta[256] | bqkv[768] | bp[256]
    const int c16_ = lane & 15, g_ = lane >> 4, hc = head * HDW;
    const float sc = 0.25f * 1.4426950408889634f;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    for (int i = threadIdx.x; i < 6 * DW; i += NTHW)
        vec_s[i] = i < DW ? p.n1w[i] : i < 2 * DW ? p.n1b[i - DW] : i < 5 * DW ? p.bqkv[i - 2 * DW] : p.pb[i - 5 * DW];
    for (int i = threadIdx.x; i < RT; i += NTHW) {                // class of an image row: -1 = padding; slots never mix
        const int slot = i / ROWS, r = i - slot * ROWS;
        int c = -1;
        if (r < p.Ts) c = slot * 64 + ((p.mode == 1) ? r / p.len_l : (p.mode == 2) ? r % p.len_l : 0);
        cls[i] = c;
    }

    float xn[PASSES][8];                                          // next group's row pieces, in flight during this group
    auto fetch = [&](int first) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int irow = ps * 32 + (threadIdx.x >> 5), slot = irow / ROWS, r = irow - slot * ROWS;
#pragma unroll
            for (int e = 0; e < 8; ++e) xn[ps][e] = 0.f;
            if (irow < RT && first + slot < p.nsamples && r < p.Ts) {
                const float* src = p.x + ((unsigned)((first + slot) * p.Ts + r) * DW + (threadIdx.x & 31) * 8);
                const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
                xn[ps][0] = a.x; xn[ps][1] = a.y; xn[ps][2] = a.z; xn[ps][3] = a.w;
                xn[ps][4] = b.x; xn[ps][5] = b.y; xn[ps][6] = b.z; xn[ps][7] = b.w;
            }
        }
    };
    fetch(blockIdx.x * SPW);
    lds_barrier();                                                // vec_s / cls visible
    // this wave's weight fragments: n-tiles head (q), 16 + head (k), 32 + head (v) of the packed [768][256] image
    auto wfrag = [&](int m, int ks) {
        return *reinterpret_cast<const bf16x8*>(p.wqkv + ((size_t)((m * HW + head) * KSW + ks) * 64 + lane) * 8);
    };

    PH_DECL
    for (int first = blockIdx.x * SPW; first < p.nsamples; first += gridDim.x * SPW) {
        PH(11)
        // The per-lane address pieces are laundered once per group: left alone, hipcc hoists ~25 registers of loop-invariant
        // addresses out of this loop and spills them; their scratch reloads then queue behind the group's HBM stores
        // (loads and stores share one in-order counter) — measured 212 vs 188 us per launch.
        int lz = 0;
        asm volatile("" : "+v"(lz));
        const int c16 = c16_ + lz, g = g_ + lz, q4 = c16 >> 2, p4 = c16 & 3, tid = (int)threadIdx.x + lz, lc8 = (tid & 31) * 8;
        // global row of image row i (slot-major), or -1.  Rows and element offsets are 32-bit (the launcher checks the sizes): with
        // 64-bit row numbers hipcc precomputes every per-thread pointer of the copy loops outside the group loop and spills them
        auto grow = [&](int irow) -> int {
            const int slot = irow / ROWS, r = irow - slot * ROWS;
            return (first + slot < p.nsamples && r < p.Ts) ? (first + slot) * p.Ts + r : -1;
        };
        // first two k-steps of this wave's q | k | v weights: in flight under the LayerNorm.  (gfx950 counts loads and stores
        // with ONE in-order counter: a load that is issued behind a store cannot be waited for before that store has landed in
        // HBM.  Every load below that is needed soon is therefore issued in front of the stores of its phase.)
        bf16x8 wc[3], wn[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) { wc[m] = wfrag(m, 0); if (HS_W256_PF2) wn[m] = wfrag(m, 1); }
        // ---- LN1 -> U image (+ u to HBM: the q / k / v weight gradients' operand)
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int irow = ps * 32 + (tid >> 5);
            if (irow < RT) {
                float sm = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += xn[ps][e];
                sm = lanes_sum<32>(sm);
                const float mean = sm * (1.f / DW);
                float vq = 0.f, f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { f[e] = xn[ps][e] - mean; vq += f[e] * f[e]; }
                vq = lanes_sum<32>(vq);
                const float rstd = rsqrtf(vq * (1.f / DW) + 1e-5f);
                {
                    const float4 g0 = *reinterpret_cast<const float4*>(vec_s + lc8), g1 = *reinterpret_cast<const float4*>(vec_s + lc8 + 4);
                    const float4 b0 = *reinterpret_cast<const float4*>(vec_s + DW + lc8), b1 = *reinterpret_cast<const float4*>(vec_s + DW + lc8 + 4);
                    const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = f[e] * rstd * gm[e] + bt[e];
                }
                const int gr = grow(irow);
                const bf16x8 ub = gr >= 0 ? cvt8(f) : zero8();
                *reinterpret_cast<bf16x8*>(Uf + irow * PU + lc8) = ub;
                if (gr >= 0) HS_NT(HS_NT_C, reinterpret_cast<bf16x8*>(p.u + ((unsigned)gr * DW + lc8)), ub);
            }
        }
        PH(0)
        lds_barrier();                                            // B1: U complete
        PH(1)
        // ---- q | k | v of this head over all m-tiles of the group, weights streamed two k-steps ahead
        f32x4 acc[3][MTT];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(vec_s + 2 * DW + m * DW + hc + 4 * g);
#pragma unroll
            for (int mt = 0; mt < MTT; ++mt) acc[m][mt] = bias;
        }
#pragma unroll 1
        for (int ks = 0; ks < KSW; ++ks) {                        // (not unrolled: unrolled, hipcc hoists every fragment load to the top and spills 109 registers)
            bf16x8 wf[3];
            if (ks + 1 + HS_W256_PF2 < KSW) {
#pragma unroll
                for (int m = 0; m < 3; ++m) wf[m] = wfrag(m, ks + 1 + HS_W256_PF2);
            }
#pragma unroll
            for (int mt = 0; mt < MTT; ++mt) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(Uf + (mt * 16 + c16) * PU + ks * 32 + g * 8);
#pragma unroll
                for (int m = 0; m < 3; ++m) acc[m][mt] = mfma16(wc[m], a, acc[m][mt]);
            }
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (HS_W256_PF2) { wc[m] = wn[m]; if (ks + 2 < KSW) wn[m] = wf[m]; }
                else if (ks + 1 < KSW) wc[m] = wf[m];
            }
        }
        PH(2)
        // q^T / k^T stay in registers (lane = token, 4 head dims: the fragment of the K = 16 score MFMA); all three go to staging
        bf16x4 qT[MTT], kT[MTT];
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) {
            qT[mt] = cvt4w(acc[0][mt]); kT[mt] = cvt4w(acc[1][mt]);
            bf16_t* dst = Sf + (mt * 16 + c16) * PS + hc + 4 * g;
            *reinterpret_cast<bf16x4*>(dst) = qT[mt];
            *reinterpret_cast<bf16x4*>(dst + DW) = kT[mt];
            *reinterpret_cast<bf16x4*>(dst + 2 * DW) = cvt4w(acc[2][mt]);
        }
        // loads first (see above): the projection weights of this wave's 16 output columns, the residual pieces of its output
        // tiles (L2-hot: the LayerNorm read the same rows) and the next group's rows — all in flight under the attention
        auto pfrag = [&](int ks) { return *reinterpret_cast<const bf16x8*>(p.wp + ((size_t)(head * KSW + ks) * 64 + lane) * 8); };
        bf16x8 wpj[KSW / 2];                                      // (the second half follows after the attention: 32 more registers here spill)
        f32x4 xr[MTT];
        auto early = [&]() {
#pragma unroll
            for (int ks = 0; ks < KSW / 2; ++ks) wpj[ks] = pfrag(ks);
#pragma unroll
            for (int mt = 0; mt < MTT; ++mt) {
                const int gr = grow(mt * 16 + c16);
                xr[mt] = gr >= 0 ? *reinterpret_cast<const f32x4*>(p.x + ((unsigned)gr * DW + hc + 4 * g)) : z4;
            }
        };
        if (HS_W256_EARLY) early();
        if (!HS_W256_LATE_X) fetch(first + gridDim.x * SPW);
        PH(3)
        lds_barrier();                                            // B2: every wave is done with U; the staging image is complete
        PH(4)
        // ---- saved q | k | v leave as whole rows (three 512-byte thirds of 32 16-byte pieces); the stores drain under the attention
        for (int idx = tid; idx < RT * 32; idx += NTHW) {
            const int irow = idx >> 5, pc = idx & 31;
            const int gr = grow(irow);
            if (gr >= 0) {
#pragma unroll
                for (int m = 0; m < 3; ++m)
                    HS_NT(HS_NT_C, reinterpret_cast<bf16x8*>(p.qkv + ((unsigned)gr * (3 * DW) + m * DW + pc * 8)),
                          *reinterpret_cast<const bf16x8*>(Sf + irow * PS + m * DW + pc * 8));
            }
        }
        PH(5)
        // ---- attention of this head; O into the (dead) U image.  The class / padding mask of a (query tile, key tile) pair is
        //      the same in every slot: built once per query tile as the accumulator the score MFMA starts from (0 or -inf)
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            if (qt * 16 >= p.Ts) break;
            const int qc = cls[qt * 16 + c16];                     // class of this lane's query (slot 0's classes: the same in every slot)
            f32x4 cmk[NT];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const int4 kc = *reinterpret_cast<const int4*>(cls + kt * 16 + g * 4);
                cmk[kt][0] = (kc.x >= 0 && kc.x == qc) ? 0.f : -INFINITY; cmk[kt][1] = (kc.y >= 0 && kc.y == qc) ? 0.f : -INFINITY;
                cmk[kt][2] = (kc.z >= 0 && kc.z == qc) ? 0.f : -INFINITY; cmk[kt][3] = (kc.w >= 0 && kc.w == qc) ? 0.f : -INFINITY;
            }
#pragma unroll
            for (int slot = 0; slot < SPW; ++slot) {
                const int r0 = slot * ROWS;
                const int query = r0 + qt * 16 + c16;
                f32x4 sv[NT];
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    sv[kt] = mfma_k16w(kT[slot * NT + kt], qT[slot * NT + qt], cmk[kt]);     // masked pairs start (and stay) at -inf
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
                    o = mfma_k16w(tr4w(Sf + (r0 + kt * 16 + 4 * g + q4) * PS + 2 * DW + hc + 4 * p4), cvt4w(sv[kt]), o);
                bf16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)(o[r] * inv);
                *reinterpret_cast<bf16x4*>(Uf + query * PU + hc + 4 * g) = ov;
                if (g == 0) lse_s[query * HW + head] = m + __builtin_amdgcn_logf(fmaxf(lsum, 1e-30f));
            }
        }
        PH(6)
        lds_barrier();                                            // B3: O image and logsumexp table complete
        PH(7)
        if (!HS_W256_EARLY) early();
        bf16x8 wpk[KSW / 2];                                      // second half of the projection weights (loads in front of the stores)
#pragma unroll
        for (int ks = 0; ks < KSW / 2; ++ks) wpk[ks] = pfrag(KSW / 2 + ks);
        if (HS_W256_LATE_X) fetch(first + gridDim.x * SPW);      // next group's rows: behind this phase's short loads, in front of its stores
        // ---- o and lse leave as whole rows
        for (int idx = tid; idx < RT * 32; idx += NTHW) {
            const int irow = idx >> 5, pc = idx & 31;
            const int gr = grow(irow);
            if (gr >= 0) HS_NT(HS_NT_C, reinterpret_cast<bf16x8*>(p.o + ((unsigned)gr * DW + pc * 8)), *reinterpret_cast<const bf16x8*>(Uf + irow * PU + pc * 8));
        }
        for (int idx = tid; idx < RT * 4; idx += NTHW) {
            const int irow = idx >> 2;
            const int gr = grow(irow);
            if (gr >= 0) *reinterpret_cast<float4*>(p.lse + ((unsigned)gr * HW + (idx & 3) * 4)) = *reinterpret_cast<const float4*>(lse_s + idx * 4);
        }
        PH(8)
        // ---- projection: this wave's 16 output columns; transposed accumulators -> x1 leaves as 16-byte pieces
        const f32x4 pbias = *reinterpret_cast<const f32x4*>(vec_s + 5 * DW + hc + 4 * g);
        f32x4 pav[MTT];
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) {
            pav[mt] = pbias;
#pragma unroll
            for (int ks = 0; ks < KSW / 2; ++ks)
                pav[mt] = mfma16(wpj[ks], *reinterpret_cast<const bf16x8*>(Uf + (mt * 16 + c16) * PU + ks * 32 + g * 8), pav[mt]);
        }
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) {
            f32x4 pa = pav[mt];
#pragma unroll
            for (int ks = 0; ks < KSW / 2; ++ks)
                pa = mfma16(wpk[ks], *reinterpret_cast<const bf16x8*>(Uf + (mt * 16 + c16) * PU + (KSW / 2 + ks) * 32 + g * 8), pa);
            const int gr = grow(mt * 16 + c16);
            if (gr >= 0) {
                const float rs = p.rowscale ? p.rowscale[gr] : 1.f;              // DropPath: x + scale * attn(x)
                f32x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = fmaf(pa[r], rs, xr[mt][r]);
                *reinterpret_cast<f32x4*>(p.x1 + ((unsigned)gr * DW + hc + 4 * g)) = ov;
            }
        }
        PH(9)
        lds_barrier();                                            // B4: the O image is read; the next group's LayerNorm may overwrite it
        PH(10)
    }
    PH_FLUSH()
}

// ---------------------------------------------------------------------------------------------------------------------------
// The attention half of an encoder Block at D = 256, BACKWARD, in one persistent kernel (round 5; VERDICT r02-r04 "Large: the
// attention half's backward at D = 256") — the wide counterpart of blk128_bwd_kernel (attn.hip):
//   dO = dx1 Wp  ->  attention backward per head (delta = sum_j P dP inside the core, O is not read)  ->  du = dq|dk|dv Wqkv
//   ->  LayerNorm-1 backward + residual gradient, dgamma / dbeta.
// It replaces three launches of round 4 — gemm<A_BF16, E_BF16> (dO, 40 us), attn16_bwd (96 us), gemm<A_BF16, E_LN_BWD> (du +
// LayerNorm backward, 146 us) = 282 us per block with dO and dq|dk|dv making a round trip through HBM between them (1.1 GB per
// block; this kernel: q|k|v, dx1 (bf16 + fp32), x, lse in, dq|dk|dv (the weight gradients' operand) and dx out = 0.74 GB).
//   * one workgroup = 16 waves = 16 heads, groups of SPW = 2 samples (64 image rows, 4 m-tiles), as blk256_fwd_kernel;
//   * wave h owns head h in the attention core and output columns 16 h .. 16 h + 15 of both products; its slices of Wp^T
//     (8 fragments) and Wqkv^T (24) are STREAMED from L2 one k-step ahead (512 KB per group and CU, as the forward);
//   * the forward's saved q|k|v are read (recomputing them from u would stream another 384 KB of weights per group);
//   * LDS: Q | K | V images (dq | dk | dv in place), the dx1 image (dO in place after a barrier), per-wave P / dS transposition
//     tiles, lse, and an fp32 du tile that aliases the Q | K images once their last reader is past: 153 KB, one workgroup per CU.
// Measured on the first version (profiles/r05_l_blk256_bwd_knobs.txt, same box): the LayerNorm epilogue's x / dx1 rows requested in
// FRONT of the du product instead of behind it: 244.0 -> 265.8 us (the du product's weight-fragment waits then sit behind 128 KB of
// HBM loads in the in-order counter) — not kept.  Start stagger (fused_dec.hip HS_DEC_STG_*: every second workgroup, (b >> 3) & 1,
// sleeps HS_W256B_STG x 0.85 us before its first group): 5 us 243.3, 10 us 235.0, 17 us 230.6 / 229.4 (against 239.6 on that box),
// 24 us 246, 31 us 255 us (profiles/r05_m_large_stagger.txt) — kept at 17 us.
#ifndef HS_W256B_EARLY_X
#define HS_W256B_EARLY_X 0
#endif
#ifndef HS_W256B_STG
#define HS_W256B_STG 20
#endif
constexpr int PB = DW + 8;                          // image row pitch of the backward kernel (elements)
constexpr int BTSW = 16;                            // P / dS transposition tile row (elements), chunks rotated by the row group (attn.hip BTS)
constexpr int DUSW = DW + 4;                        // fp32 du tile row stride

struct Blk256BwdArgs {
    const bf16_t* qkv; const float* lse;            // saved by the forward: [rows][768], [rows][16]
    const bf16_t* dx1b; const float* dx1;           // the projection's dY as bf16 (DropPath factor folded in) | the residual gradient
    const float* x; const float* gamma;             // block input, LayerNorm-1 weight
    const bf16_t* wpT; const bf16_t* wqkvT;         // packed images of Wp^T [n = 256][k = 256] and Wqkv^T [n = 256][k = 768]
    bf16_t* dqkv; float* dx; float* dgamma; float* dbeta;
    const float* det_base; long long* det_acc;
    int Ts, nsamples, mode, len_l, accumulate;
};

template <int NT, int SPW>
struct LayWB {
    static constexpr int ROWS = NT * 16, RT = SPW * ROWS, IMG = RT * PB;
    static constexpr int OFF_Q = RT * 4;                                   // after cls
    static constexpr int OFF_X = OFF_Q + 3 * IMG * 2;                      // dx1 image, then dO
    static constexpr int OFF_T = OFF_X + IMG * 2;                          // per-wave transposition tiles
    static constexpr int OFF_LSE = OFF_T + HW * 2 * 16 * BTSW * 2;         // [16][RT]
    static constexpr int OFF_G = OFF_LSE + HW * RT * 4;                    // gamma[256]
    static constexpr int TOTAL = OFF_G + DW * 4;
    static_assert(RT * DUSW * 4 <= 2 * IMG * 2, "the fp32 du tile must fit in the Q | K images");
    static_assert(2 * NTHW * 8 * 4 <= 4 * IMG * 2, "the final dgamma / dbeta reduction must fit in the four images");
    static_assert(TOTAL <= 160 * 1024, "LDS");
};

template <int NT, int SPW>
__global__ __launch_bounds__(NTHW, 1) void blk256_bwd_kernel(Blk256BwdArgs p) {
    using L = LayWB<NT, SPW>;
    constexpr int ROWS = L::ROWS, RT = L::RT, MTT = SPW * NT;
    constexpr int PASSES = (RT * 32 + NTHW - 1) / NTHW;           // wide layout: 32 lanes per row, 32 rows per pass
    constexpr int NQ = (RT * 96 + NTHW - 1) / NTHW;               // q|k|v / dq|dk|dv 16-byte pieces per thread (96 per row)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, head = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int* cls = reinterpret_cast<int*>(smem);
    bf16_t* Qf = reinterpret_cast<bf16_t*>(smem + L::OFF_Q);
    bf16_t* Kf = Qf + L::IMG;
    bf16_t* Vf = Kf + L::IMG;
    bf16_t* Xf = reinterpret_cast<bf16_t*>(smem + L::OFF_X);      // dx1 rows (bf16), then dO (per head in place)
    bf16_t* Tp = reinterpret_cast<bf16_t*>(smem + L::OFF_T) + head * (2 * 16 * BTSW);
    bf16_t* Td = Tp + 16 * BTSW;
    float* lse_s = reinterpret_cast<float*>(smem + L::OFF_LSE);   // [16][RT]
    float* gam_s = reinterpret_cast<float*>(smem + L::OFF_G);
    float* DU = reinterpret_cast<float*>(Qf);                     // [RT][DUSW], over Q | K once nothing reads them any more
    const int c16 = lane & 15, g = lane >> 4, q4 = c16 >> 2, p4 = c16 & 3, hc = head * HDW;
    const float scale = 0.25f, sc = 0.25f * 1.4426950408889634f;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    if (threadIdx.x < DW) gam_s[threadIdx.x] = p.gamma[threadIdx.x];
    for (int i = threadIdx.x; i < RT; i += NTHW) {                // class of an image row: -1 = padding; slots never mix
        const int slot = i / ROWS, r = i - slot * ROWS;
        int c = -1;
        if (r < p.Ts) c = slot * 64 + ((p.mode == 1) ? r / p.len_l : (p.mode == 2) ? r % p.len_l : 0);
        cls[i] = c;
    }
    float dgam[8], dbet[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { dgam[e] = 0.f; dbet[e] = 0.f; }
    const int tw = c16 * BTSW + (((g + q4) & 3) << 2);             // tile write: row c16, keys 4g.. at chunk (g + (c16 >> 2)) & 3
    const int ttoff = (4 * g + q4) * BTSW + (((p4 + g) & 3) << 2); // tile transpose read: row 4g + q4, chunk p4
    auto wpfrag = [&](int ks) { return *reinterpret_cast<const bf16x8*>(p.wpT + ((size_t)(head * KSW + ks) * 64 + lane) * 8); };
    auto wufrag = [&](int ks) { return *reinterpret_cast<const bf16x8*>(p.wqkvT + ((size_t)(head * 3 * KSW + ks) * 64 + lane) * 8); };
    if (HS_W256B_STG > 0 && ((blockIdx.x >> 3) & 1)) {
        for (int i = 0; i < HS_W256B_STG; ++i) __builtin_amdgcn_s_sleep(32);
    }

    for (int first = blockIdx.x * SPW; first < p.nsamples; first += gridDim.x * SPW) {
        // (per-lane address pieces laundered once per group, as in blk256_fwd_kernel: hoisted out of the loop they spill)
        int lz = 0;
        asm volatile("" : "+v"(lz));
        const int tid = (int)threadIdx.x + lz;
        const int hcell = c16 * PB + hc + 4 * g + lz;              // this head's columns 4 g .. of row c16 (+ 16 mt rows)
        const int troff = (4 * g + q4) * PB + hc + 4 * p4 + lz;    // transposed read: rows 4 g + q4, this head's columns 4 p4 ..
        const int fa = c16 * PB + g * 8 + lz;                      // MFMA operand: row c16 (+ 16 mt), columns 8 g .. (+ 32 ks)
        auto grow = [&](int irow) -> int {                         // global row of image row irow (slot-major), or -1
            const int slot = irow / ROWS, r = irow - slot * ROWS;
            return (first + slot < p.nsamples && r < p.Ts) ? (first + slot) * p.Ts + r : -1;
        };
        // ---- the group's q | k | v, dx1 (bf16) and logsumexp rows -> LDS.  The first k-step of Wp^T flies with them.
        bf16x8 wc = wpfrag(0);
        {
            bf16x8 rq[NQ], rd[PASSES];
            float rl;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int idx = tid + NTHW * i, irow = idx / 96, pc = idx - irow * 96;
                const int gr = irow < RT ? grow(irow) : -1;
                rq[i] = gr >= 0 ? *reinterpret_cast<const bf16x8*>(p.qkv + ((unsigned)gr * (3 * DW) + pc * 8)) : zero8();
            }
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int irow = ps * 32 + (tid >> 5);
                const int gr = irow < RT ? grow(irow) : -1;
                rd[ps] = gr >= 0 ? *reinterpret_cast<const bf16x8*>(p.dx1b + ((unsigned)gr * DW + (tid & 31) * 8)) : zero8();
            }
            {
                const int irow = tid >> 4;                         // RT * 16 = 1024 (row, head) pairs: one per thread
                const int gr = irow < RT ? grow(irow) : -1;
                rl = gr >= 0 ? p.lse[(unsigned)gr * HW + (tid & 15)] : 1e30f;      // padding rows: exp2(s - 1e30) = 0
            }
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int idx = tid + NTHW * i, irow = idx / 96, pc = idx - irow * 96;
                if (irow < RT) *reinterpret_cast<bf16x8*>(Qf + (pc >> 5) * L::IMG + irow * PB + (pc & 31) * 8) = rq[i];
            }
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int irow = ps * 32 + (tid >> 5);
                if (irow < RT) *reinterpret_cast<bf16x8*>(Xf + irow * PB + (tid & 31) * 8) = rd[ps];
            }
            if ((tid >> 4) < RT) lse_s[(tid & 15) * RT + (tid >> 4)] = rl;
        }
        lds_barrier();                                             // B1: the group's images are complete
        // ---- dO[:, this head's 16 columns] = dx1 Wp  (K = 256, Wp^T fragments one k-step ahead), held as bf16 until every wave
        //      has read the dx1 image, then written over it (own columns)
        bf16x4 dob[MTT];
        {
            f32x4 acc[MTT];
#pragma unroll
            for (int mt = 0; mt < MTT; ++mt) acc[mt] = z4;
#pragma unroll 1
            for (int ks = 0; ks < KSW; ++ks) {
                bf16x8 wn = wc;
                if (ks + 1 < KSW) wn = wpfrag(ks + 1);
#pragma unroll
                for (int mt = 0; mt < MTT; ++mt)
                    acc[mt] = mfma16(wc, *reinterpret_cast<const bf16x8*>(Xf + mt * 16 * PB + fa + ks * 32), acc[mt]);
                wc = wn;
            }
#pragma unroll
            for (int mt = 0; mt < MTT; ++mt) dob[mt] = cvt4w(acc[mt]);
        }
        // class / padding mask of a (query tile, key tile) pair, as the accumulator the score MFMA starts from (0 or -inf)
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
        lds_barrier();                                             // B2: every wave is done with the dx1 image
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) *reinterpret_cast<bf16x4*>(Xf + mt * 16 * PB + hcell) = dob[mt];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // own writes before own reads (only this wave touches these columns)
        // ---- attention backward of this head (as blk128_bwd_kernel), slot by slot; dq, dk, dv in place
        const float* lse_h = lse_s + head * RT;
#pragma unroll
        for (int slot = 0; slot < SPW; ++slot) {
            const int r0 = slot * ROWS;
            f32x4 dkT[NT], dvT[NT];
            bf16x4 KT[NT];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) { dkT[kt] = z4; dvT[kt] = z4; KT[kt] = tr4w(Kf + (r0 + kt * 16) * PB + troff); }
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                if (qt * 16 >= p.Ts) break;
                const int query = r0 + qt * 16 + c16;
                const int qcell = (r0 + qt * 16) * PB + hcell;
                const bf16x4 bq = *reinterpret_cast<const bf16x4*>(Qf + qcell);
                const bf16x4 bdo = *reinterpret_cast<const bf16x4*>(Xf + qcell);
                const float lqn = -lse_h[query];
                const bf16x4 QT = tr4w(Qf + (r0 + qt * 16) * PB + troff);
                const bf16x4 DT = tr4w(Xf + (r0 + qt * 16) * PB + troff);
                f32x4 dqT = z4;
                // delta_i = sum_j P_ij dP_ij: with at most NT = 2 key tiles per query tile both P and dP are in hand before dS
                f32x4 pvs[NT], dps[NT];
                float dl = 0.f;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    const bf16x4 ak = *reinterpret_cast<const bf16x4*>(Kf + (r0 + kt * 16) * PB + hcell);
                    const bf16x4 av = *reinterpret_cast<const bf16x4*>(Vf + (r0 + kt * 16) * PB + hcell);
                    const f32x4 sv = mfma_k16w(ak, bq, cm[qt][kt]);
                    dps[kt] = mfma_k16w(av, bdo, z4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pvs[kt][r] = __builtin_amdgcn_exp2f(fmaf(sv[r], sc, lqn));
                        dl = fmaf(pvs[kt][r], dps[kt][r], dl);
                    }
                }
                dl = rows_sum(dl);
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    f32x4 ds;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds[r] = pvs[kt][r] * (dps[kt][r] - dl);
                    const bf16x4 pb = cvt4w(pvs[kt]), dsb = cvt4w(ds);
                    dqT = mfma_k16w(KT[kt], dsb, dqT);
                    *reinterpret_cast<bf16x4*>(Tp + tw) = pb;
                    *reinterpret_cast<bf16x4*>(Td + tw) = dsb;
                    asm volatile("" ::: "memory");
                    const bf16x4 Bp = tr4w(Tp + ttoff), Bds = tr4w(Td + ttoff);
                    dkT[kt] = mfma_k16w(QT, Bds, dkT[kt]);
                    dvT[kt] = mfma_k16w(DT, Bp, dvT[kt]);
                }
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (bf16_t)(dqT[r] * scale);
                *reinterpret_cast<bf16x4*>(Qf + qcell) = v;
            }
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const int kcell = (r0 + kt * 16) * PB + hcell;
                bf16x4 vk, vv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { vk[r] = (bf16_t)(dkT[kt][r] * scale); vv[r] = (bf16_t)dvT[kt][r]; }
                *reinterpret_cast<bf16x4*>(Kf + kcell) = vk;
                *reinterpret_cast<bf16x4*>(Vf + kcell) = vv;
            }
        }
        // first k-step of this wave's Wqkv^T slice: in flight over the barrier
        bf16x8 uc = wufrag(0);
        lds_barrier();                                             // B3: dq | dk | dv images complete
        // ---- dq|dk|dv leave as whole rows (the q / k / v weight gradients' operand); du[:, this wave's 16 columns] = dq|dk|dv Wqkv
        //      (contraction over the 768 image columns, Wqkv^T fragments streamed one k-step ahead)
        for (int idx = tid; idx < RT * 96; idx += NTHW) {
            const int irow = idx / 96, pc = idx - irow * 96;
            const int gr = grow(irow);
            if (gr >= 0)
                HS_NT(HS_NT_C, reinterpret_cast<bf16x8*>(p.dqkv + ((unsigned)gr * (3 * DW) + pc * 8)),
                      *reinterpret_cast<const bf16x8*>(Qf + (pc >> 5) * L::IMG + irow * PB + (pc & 31) * 8));
        }
        const int lc8 = (tid & 31) * 8;
        float xr[PASSES][8], rs[PASSES][8];
        int erow[PASSES];
        auto fetch_epi = [&]() {
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                const int irow = ps * 32 + (tid >> 5);
                erow[ps] = irow < RT ? grow(irow) : -1;
#pragma unroll
                for (int e = 0; e < 8; ++e) { xr[ps][e] = 0.f; rs[ps][e] = 0.f; }
                if (erow[ps] >= 0) {
                    const float* xp = p.x + ((unsigned)erow[ps] * DW + lc8);
                    const float* rp = p.dx1 + ((unsigned)erow[ps] * DW + lc8);
                    const float4 a0 = *reinterpret_cast<const float4*>(xp), a1 = *reinterpret_cast<const float4*>(xp + 4);
                    const float4 b0 = *reinterpret_cast<const float4*>(rp), b1 = *reinterpret_cast<const float4*>(rp + 4);
                    xr[ps][0] = a0.x; xr[ps][1] = a0.y; xr[ps][2] = a0.z; xr[ps][3] = a0.w; xr[ps][4] = a1.x; xr[ps][5] = a1.y; xr[ps][6] = a1.z; xr[ps][7] = a1.w;
                    rs[ps][0] = b0.x; rs[ps][1] = b0.y; rs[ps][2] = b0.z; rs[ps][3] = b0.w; rs[ps][4] = b1.x; rs[ps][5] = b1.y; rs[ps][6] = b1.z; rs[ps][7] = b1.w;
                    if (p.accumulate) {
                        const float* op = p.dx + ((unsigned)erow[ps] * DW + lc8);
                        const float4 c0 = *reinterpret_cast<const float4*>(op), c1 = *reinterpret_cast<const float4*>(op + 4);
                        rs[ps][0] += c0.x; rs[ps][1] += c0.y; rs[ps][2] += c0.z; rs[ps][3] += c0.w; rs[ps][4] += c1.x; rs[ps][5] += c1.y; rs[ps][6] += c1.z; rs[ps][7] += c1.w;
                    }
                }
            }
        };
        if (HS_W256B_EARLY_X) fetch_epi();
        f32x4 du[MTT];
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) du[mt] = z4;
#pragma unroll 1
        for (int ks = 0; ks < 3 * KSW; ++ks) {
            bf16x8 un = uc;
            if (ks + 1 < 3 * KSW) un = wufrag(ks + 1);
            const bf16_t* img = Qf + (ks / KSW) * L::IMG + (ks % KSW) * 32;
#pragma unroll
            for (int mt = 0; mt < MTT; ++mt)
                du[mt] = mfma16(uc, *reinterpret_cast<const bf16x8*>(img + mt * 16 * PB + fa), du[mt]);
            uc = un;
        }
        // ---- the rows the LayerNorm epilogue needs (wide layout, 32 lanes per row): in flight over the two barriers
        if (!HS_W256B_EARLY_X) fetch_epi();
        lds_barrier();                                             // B4: every read of the images (row stores, du product) is done
#pragma unroll
        for (int mt = 0; mt < MTT; ++mt) *reinterpret_cast<f32x4*>(DU + (mt * 16 + c16) * DUSW + hc + 4 * g) = du[mt];
        lds_barrier();                                             // B5: du tile complete
        // ---- LayerNorm-1 backward + residual gradient, wide layout
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int irow = ps * 32 + (tid >> 5);
            if (irow < RT) {
                const float4 t0 = *reinterpret_cast<const float4*>(DU + irow * DUSW + lc8);
                const float4 t1 = *reinterpret_cast<const float4*>(DU + irow * DUSW + lc8 + 4);
                const float duv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                float sm = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += xr[ps][e];
                sm = lanes_sum<32>(sm);
                const float mean = sm * (1.f / DW);
                float qv = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { xr[ps][e] -= mean; qv += xr[ps][e] * xr[ps][e]; }
                qv = lanes_sum<32>(qv);
                const float rstd = rsqrtf(qv * (1.f / DW) + 1e-5f);
                const float4 g0 = *reinterpret_cast<const float4*>(gam_s + lc8), g1 = *reinterpret_cast<const float4*>(gam_s + lc8 + 4);
                const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                float a = 0.f, b = 0.f, t[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { xr[ps][e] *= rstd; t[e] = duv[e] * gm[e]; a += t[e]; b += t[e] * xr[ps][e]; }
                a = lanes_sum<32>(a); b = lanes_sum<32>(b);
                a *= (1.f / DW); b *= (1.f / DW);
                if (erow[ps] >= 0) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] = rs[ps][e] + rstd * (t[e] - a - xr[ps][e] * b);
                        dgam[e] += duv[e] * xr[ps][e];
                        dbet[e] += duv[e];
                    }
                    float* op = p.dx + ((unsigned)erow[ps] * DW + lc8);
                    *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            }
        }
        lds_barrier();                                             // B6: the du tile is read; the next group's images may overwrite it
    }
    // dgamma / dbeta: 32 threads per column octet -> LDS, one commit per column and workgroup
    float* red = DU;
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x * 8 + e] = dgam[e]; red[NTHW * 8 + threadIdx.x * 8 + e] = dbet[e]; }
    lds_barrier();
    if (threadIdx.x < 2 * DW) {
        const int which = threadIdx.x >> 8, c = threadIdx.x & 255, o8 = c >> 3, e = c & 7;
        float sacc = 0.f;
        for (int t2 = o8; t2 < NTHW; t2 += 32) sacc += red[which * NTHW * 8 + t2 * 8 + e];
        hs_gadd(HsDet{p.det_base, p.det_acc}, (which ? p.dbeta : p.dgamma) + c, sacc);
    }
}

template <int NT, int SPW>
int launch_blk256_bwd(const Blk256BwdArgs& a, hipStream_t s) {
    using L = LayWB<NT, SPW>;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blk256_bwd_kernel<NT, SPW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::TOTAL); attr_set = true; }
    static int wgs = 0;                       // persistent, one 16-wave workgroup per CU (HSIMAE_BLK256_BWD_WGS overrides)
    if (!wgs) wgs = 256;
    const int groups = (a.nsamples + SPW - 1) / SPW;
    hipLaunchKernelGGL((blk256_bwd_kernel<NT, SPW>), dim3(groups < wgs ? groups : wgs), dim3(NTHW), (size_t)L::TOTAL, s, a);
    return (int)hipGetLastError();
}

template <int NT, int SPW>
int launch_blk256(const Blk256Args& a, hipStream_t s) {
    using L = LayW<NT, SPW>;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(blk256_fwd_kernel<NT, SPW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::TOTAL); attr_set = true; }
    static int wgs = 0;                       // persistent, one 16-wave workgroup per CU (HSIMAE_BLK256_WGS overrides)
    if (!wgs) wgs = 256;
    const int groups = (a.nsamples + SPW - 1) / SPW;
    hipLaunchKernelGGL((blk256_fwd_kernel<NT, SPW>), dim3(groups < wgs ? groups : wgs), dim3(NTHW), (size_t)L::TOTAL, s, a);
    return (int)hipGetLastError();
}

}  // namespace

// Shape predicate (api.hip SC_ATTN_BLOCK256 is the A/B switch).  The kernel's 32-bit element offsets bound the launch size: a
// larger launch answers false here and takes the layer-at-a-time kernels instead of failing in hs_attn_block256_fwd.
bool hs_attn_block256_fusable(int d, int heads, int Ts, int nsamples) {
    return d == DW && heads == HW && Ts >= 1 && Ts <= 32 && (int64_t)nsamples * Ts * 3 * DW < (1ll << 31);
}

// dO + attention backward + du + LayerNorm-1 backward at D = 256 in one launch (blk256_bwd_kernel).  Shape predicate (api.hip
// SC_ATTN_BLOCK256_BWD is the A/B switch); the same 32-bit offset bound as the forward.
bool hs_attn_block256_bwd_fusable(int d, int heads, int Ts, int nsamples) { return hs_attn_block256_fusable(d, heads, Ts, nsamples); }

int hs_attn_block256_bwd(const hs_bf16* qkv, const float* lse, const hs_bf16* dx1b, const float* dx1, const float* x, const float* gamma,
                         const hs_bf16* wpT, const hs_bf16* wqkvT, hs_bf16* dqkv, float* dx, float* dgamma, float* dbeta,
                         const float* det_base, long long* det_acc, int Ts, int nsamples, int mode, int len_l, int accumulate,
                         hipStream_t s) {
    if (nsamples <= 0) return HS_OK;
    if (Ts < 1 || Ts > 32) return HS_EUNSUPPORTED;
    if ((int64_t)nsamples * Ts * 3 * DW >= (1ll << 31)) return HS_EUNSUPPORTED;      // 32-bit element offsets inside the kernel
    Blk256BwdArgs a;
    a.qkv = qkv; a.lse = lse; a.dx1b = dx1b; a.dx1 = dx1; a.x = x; a.gamma = gamma; a.wpT = wpT; a.wqkvT = wqkvT; a.dqkv = dqkv; a.dx = dx;
    a.dgamma = dgamma; a.dbeta = dbeta; a.det_base = det_base; a.det_acc = det_acc;
    a.Ts = Ts; a.nsamples = nsamples; a.mode = mode; a.len_l = len_l; a.accumulate = accumulate;
    return Ts <= 16 ? launch_blk256_bwd<1, 2>(a, s) : launch_blk256_bwd<2, 2>(a, s);
}

int hs_attn_block256_fwd(const float* x, const float* n1w, const float* n1b, const hs_bf16* wqkv, const float* bqkv, const hs_bf16* wp,
                         const float* pb, hs_bf16* u, hs_bf16* qkv, hs_bf16* o, float* lse, float* x1, const float* rowscale, int Ts,
                         int nsamples, int mode, int len_l, hipStream_t s) {
    if (nsamples <= 0) return HS_OK;
    if (Ts < 1 || Ts > 32) return HS_EUNSUPPORTED;
    if ((int64_t)nsamples * Ts * 3 * DW >= (1ll << 31)) return HS_EUNSUPPORTED;      // 32-bit element offsets inside the kernel
    Blk256Args a;
    a.x = x; a.n1w = n1w; a.n1b = n1b; a.wqkv = wqkv; a.bqkv = bqkv; a.wp = wp; a.pb = pb; a.u = u; a.qkv = qkv; a.o = o;
    a.lse = lse; a.x1 = x1; a.rowscale = rowscale; a.Ts = Ts; a.nsamples = nsamples; a.mode = mode; a.len_l = len_l;
    return Ts <= 16 ? launch_blk256<1, 2>(a, s) : launch_blk256<2, 2>(a, s);
}

HS_UNIT_VARIANT_BITS(attn_wide)
