// Row-panel bf16 MFMA GEMM for the HSIMAE linears:  out[M,N] = epi( pro(A)[M,K] * W[N,K]^T ).
//
// Shapes on this path are "M huge (1e5..5e5 token rows), N and K tiny (64..1400)", so the kernel is
// activation-stationary: one workgroup (4 waves) owns a 128-row panel of A, stages it once into LDS
// as bf16 (optionally applying LayerNorm on the way in), and streams the pre-packed weight image
// (common.h "wpk": MFMA B-fragment order, 1 KiB coalesced per wave-instruction) straight from
// L2 into registers.  A is read from HBM exactly once; weights are L2/Infinity-Cache resident.
//
// Wave w of the workgroup owns n-tiles {2w, 2w+1} of each 128-column chunk and all 8 m-tiles:
// 16 accumulators (64 VGPR); the SwiGLU variants carry a second accumulator set for W3.
//
// Epilogue: accumulators are passed through an fp32 LDS tile 32 rows at a time so that every global
// access of the epilogue (output, residual, saved pre-activations) is a 16-B-per-lane row-contiguous
// access; the first version's per-element 2-byte stores made every GEMM store-issue bound (profiles/).
#include "common.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

// Per-phase cycle accounting for scripts/phase_timing.py (compiled only with -DHS_PHASE_TIMING; never in the shipped library)
#ifdef HS_PHASE_TIMING
__device__ unsigned long long hs_phase_cycles_gemm[64];
extern "C" __attribute__((visibility("default"))) int hsimae_debug_phases_gemm(unsigned long long* out, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hs_phase_cycles_gemm), sizeof(unsigned long long) * 64);
    if (reset) { unsigned long long z[64] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(hs_phase_cycles_gemm), z, sizeof(z)); }
    return rc;
}
#define PH_DECL unsigned long long ph_t0 = __builtin_readcyclecounter(), ph_acc[4] = {0, 0, 0, 0};
#define PH(i) { const unsigned long long ph_t = __builtin_readcyclecounter(); ph_acc[i] += ph_t - ph_t0; ph_t0 = ph_t; }
#define PH_FLUSH(base) if (threadIdx.x == 0) { for (int i = 0; i < 4; ++i) atomicAdd(&hs_phase_cycles_gemm[(base) + i], ph_acc[i]); }
#else
#define PH_DECL
#define PH(i)
#define PH_FLUSH(base)
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;

// Rows per epilogue pass (the fp32 exchange tile(s) [PR][TS]).  bf16: 32, or 16 where the 512-deep A panel leaves no room.
// fp8: the e4m3 panel is half the bytes, so the tile could be taller — fewer passes, each with two workgroup barriers (the
// gate epilogue of LN2 + w1|w3 at d = 512 ran 88 passes per workgroup at PR = 16).  HS_EPI_ROWS_F8 (-D) for A/B builds.
#ifndef HS_EPI_ROWS_F8
#define HS_EPI_ROWS_F8 16     /* 32: +0.3 ms, 64: +0.8 ms per Huge step (measured): the taller tiles cost the 64-row kernels an occupancy step */
#endif
template <int KC, int F8 = 0, int BM = 128> struct EpiRows {
    static constexpr int f8rows = HS_EPI_ROWS_F8 > BM ? BM : HS_EPI_ROWS_F8;
    static constexpr int v = F8 ? f8rows : (KC >= 512 ? 16 : 32);
};
constexpr int TS = 132;            // fp32 LDS tile row stride (floats): conflict-free b32 writes

// BM: rows per workgroup.  128 (one workgroup per CU) for the narrow layers, where the weight stream per panel is small;
// 64 (two to three workgroups per CU, so that one panel's staging / epilogue overlaps another's MFMA loop) for K >= 256.
// F8: the MX block-scaled path (hsimae_config.precision = FP8): A is quantised to OCP e4m3 as it is staged, one e8m0
// scale per 32 consecutive K elements of a row, the weights come pre-quantised the same way (pack8_kernel), and the
// products run on v_mfma_scale_f32_16x16x128_f8f6f4 (K = 128 per instruction at twice the bf16 rate), fp32 accumulate.
// Operand maps (pinned with exact data by scripts/micro/mx_layout.hip): lane (r = l & 15, g = l >> 4) holds, in bytes
// 0..15 / 16..31 of its 8-dword operand, k = 16 g + j and k = 64 + 16 g + j of row r (A) / column r (B) of the 128-deep
// step, and supplies in its scale register the e8m0 byte of 32-block g (k in [32 g, 32 g + 32)) of that row / column.
// NCH: 1 = n-chunk outer (the A panel is re-staged for every n-chunk when K spans several chunks); 2 / 4 = k outer with
// that many accumulator sets (every A chunk staged once: the deep-K products, see the k-outer branch below).
template <int AK, int EPI, int KC, int BM, int F8, int NCH>
__global__ __launch_bounds__(256, (EPI == E_LN_BWD || BM <= 64) ? 2 : 1) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(!F8 || KC == 512, "the fp8 path stages 512-deep chunks (one scale dword per row and lane group)");
    static_assert(NCH == 1 || (AK != A_F32_LN && EPI != E_SWIGLU), "k-outer: no LayerNorm prologue, no gate pair");
    constexpr int MT = BM / 16;
    constexpr int LDA = F8 ? KC + 16 : KC + 8;        // LDS row stride (elements): 16-B pad => conflict-free b128 reads
    constexpr int ABYTES = F8 ? BM * LDA : BM * LDA * 2;
    constexpr int SCBYTES = F8 ? BM * 16 : 0;         // fp8: e8m0 scales [row][lane group g][k-step of the chunk]
    constexpr bool DUAL = (EPI == E_SWIGLU);
    constexpr int PR = EpiRows<KC, F8, BM>::v;
    bf16_t* As = reinterpret_cast<bf16_t*>(smem);     // [BM][KC+8] bf16, or [BM][KC+16] e4m3 bytes
    unsigned char* As8 = reinterpret_cast<unsigned char*>(smem);
    unsigned char* Sc = reinterpret_cast<unsigned char*>(smem + ABYTES);
    float* rstat = reinterpret_cast<float*>(smem + ABYTES + SCBYTES);   // [BM][2] mean, rstd
    float* T1 = rstat + 2 * BM;                       // [PR][TS]
    float* T2 = T1 + PR * TS;                         // second tile (SwiGLU pair)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * BM;
    const int KS_total = F8 ? (p.K + 127) / 128 : p.K / 32;      // MFMA k-steps over K
    const int NT_total = p.N / 16;
    const int n_chunks = (NT_total + 7) / 8;
    const int k_chunks = (p.K + KC - 1) / KC;

    // fp8: quantise 8 consecutive values of row r at chunk column c8 (the 4 adjacent lanes that share a 32-block agree
    // on its scale through two shuffles) and store them with the block's e8m0 byte
    auto put_a8 = [&](int r, int c8, const float (&f)[8]) {
        float am = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(f[e]));
        am = lanes_max<4>(am);
        int eb = (int)((__float_as_uint(am) >> 23) & 0xffu) - 8;          // floor(log2(amax)) - emax(e4m3), biased by 127
        eb = min(max(eb, 1), 254);
        const float inv = __uint_as_float((unsigned)(254 - eb) << 23);    // 2^-(eb - 127)
        float q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = fminf(fmaxf(f[e] * inv, -448.f), 448.f);   // v_cvt_pk_fp8_f32 does not saturate (> 464 -> NaN)
        int lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], lo, true);
        int hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], 0, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], hi, true);
        *reinterpret_cast<int2*>(As8 + r * LDA + c8) = make_int2(lo, hi);
        if ((c8 & 31) == 0) Sc[(r * 4 + ((c8 & 127) >> 5)) * 4 + (c8 >> 7)] = (unsigned char)eb;
    };

    // LayerNorm prologue: each row is owned by TPR adjacent lanes (8 columns each).  All loads of a batch of
    // rows are issued before the first reduction, so one memory latency is exposed per batch instead of one per
    // row (the first version walked rows one wave at a time and was latency bound, profiles/).
    auto stage_ln = [&]() {
#ifndef HS_LN_NB
#define HS_LN_NB 4      /* rows staged per batch and thread: 8 held 64 registers and cost the 64-row kernels a wave of occupancy (LN1 + q|k|v at d = 256: 202 -> 137 VGPRs; Large -1.1 %, Huge -1.7 % per step) */
#endif
        constexpr int TPR = KC / 8, RPP = 256 / TPR, NPASS = BM / RPP, NB = NPASS < HS_LN_NB ? NPASS : HS_LN_NB;
        const float* A = reinterpret_cast<const float*>(p.A);
        const int c8 = (tid % TPR) * 8;
        const int lnw = p.ln_width ? p.ln_width : p.K;      // the LayerNorm's width (< K when rows are stored padded)
        const bool cok = c8 < p.K, cv = c8 < lnw;
        float g[8], bt[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { g[e] = cv ? p.gamma[c8 + e] : 0.f; bt[e] = cv ? p.beta[c8 + e] : 0.f; }
        const float invk = 1.f / (float)lnw;
        for (int pb = 0; pb < NPASS; pb += NB) {
            float f[NB][8];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int r = tid / TPR + (pb + i) * RPP;
                const int row = min(row0 + r, p.M - 1);
                float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
                if (cv) {
                    x0 = *reinterpret_cast<const float4*>(A + (size_t)row * p.lda + c8);
                    x1 = *reinterpret_cast<const float4*>(A + (size_t)row * p.lda + c8 + 4);
                }
                f[i][0] = x0.x; f[i][1] = x0.y; f[i][2] = x0.z; f[i][3] = x0.w;
                f[i][4] = x1.x; f[i][5] = x1.y; f[i][6] = x1.z; f[i][7] = x1.w;
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int r = tid / TPR + (pb + i) * RPP;
                float sm = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += f[i][e];
                sm = lanes_sum<TPR>(sm);
                const float mean = sm * invk;
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float dl = cv ? f[i][e] - mean : 0.f; q += dl * dl; }
                q = lanes_sum<TPR>(q);
                const float rstd = rsqrtf(q * invk + 1e-5f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[i][e] = (f[i][e] - mean) * rstd * g[e] + bt[e];
                const bf16x8 val = cv ? cvt8(f[i]) : zero8();
                if (cok && p.u_out && row0 + r < p.M)
                    HS_NT(true, reinterpret_cast<bf16x8*>(p.u_out + (size_t)(row0 + r) * p.ldu + c8), val);   // saved for the backward only
                if constexpr (F8) {
                    if (!cv) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[i][e] = 0.f;
                    }
                    put_a8(r, c8, f[i]);
                } else {
                    *reinterpret_cast<bf16x8*>(As + r * LDA + c8) = val;
                }
            }
        }
    };

    auto stage = [&](int kc) {
        constexpr int TPR = KC / 8;                   // threads per row (8 elements each)
        constexpr int RPP = 256 / TPR;                // rows per pass
        const int c8 = (tid % TPR) * 8;
        const int kcol = kc * KC + c8;
#pragma unroll 4
        for (int r = tid / TPR; r < BM; r += RPP) {
            const int row = min(row0 + r, p.M - 1);
            bf16x8 val = zero8();
            float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (kcol < p.K) {
                if constexpr (AK == A_BF16) {
                    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
                    val = *reinterpret_cast<const bf16x8*>(A + (size_t)row * p.lda + kcol);
                    if constexpr (F8) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) f[i] = bf2f(val[i]);
                    }
                } else {
                    const float* A = reinterpret_cast<const float*>(p.A);
                    const float4 x0 = *reinterpret_cast<const float4*>(A + (size_t)row * p.lda + kcol);
                    const float4 x1 = *reinterpret_cast<const float4*>(A + (size_t)row * p.lda + kcol + 4);
                    f[0] = x0.x; f[1] = x0.y; f[2] = x0.z; f[3] = x0.w; f[4] = x1.x; f[5] = x1.y; f[6] = x1.z; f[7] = x1.w;
                    if constexpr (AK == A_F32) {
                        if (p.a_rowscale) {                       // DropPath: the branch gradient is scale * dY
                            const float rs = p.a_rowscale[row];
#pragma unroll
                            for (int i = 0; i < 8; ++i) f[i] *= rs;
                        }
                    }
                    if constexpr (AK == A_F32_LN) {
                        const float mean = rstat[2 * r], rstd = rstat[2 * r + 1];
                        const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + kcol);
                        const float4 g1 = *reinterpret_cast<const float4*>(p.gamma + kcol + 4);
                        const float4 b0 = *reinterpret_cast<const float4*>(p.beta + kcol);
                        const float4 b1 = *reinterpret_cast<const float4*>(p.beta + kcol + 4);
                        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                        const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                        for (int i = 0; i < 8; ++i) f[i] = (f[i] - mean) * rstd * g[i] + b[i];
                    }
                    val = cvt8(f);
                    if constexpr (AK == A_F32_LN) {
                        if (p.u_out && row0 + r < p.M)
                            HS_NT(true, reinterpret_cast<bf16x8*>(p.u_out + (size_t)(row0 + r) * p.ldu + kcol), val);
                    }
                }
            }
            if constexpr (F8) put_a8(r, c8, f);
            else *reinterpret_cast<bf16x8*>(As + r * LDA + c8) = val;
        }
    };

    // k-outer, bf16 A: the next chunk's rows are loaded into registers BEFORE this chunk's MFMA loop and stored (fp8: quantised)
    // after it, so their HBM latency is covered by the loop instead of exposed between two barriers (the staging phase was
    // 69 % of the d = 512 du / du2 / w2 products and 87 % of the d = 256 LN-backward GEMM: scripts/phase_timing.py)
    constexpr int NPF = (NCH > 1 && AK == A_BF16) ? BM * KC / 8 / 256 : 1;        // 16-byte pieces per thread and chunk
    auto stage_load = [&](int kc, bf16x8 (&pre)[NPF]) {
        constexpr int TPR = KC / 8, RPP = 256 / TPR;
        const int kcol = kc * KC + (tid % TPR) * 8;
        const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int row = min(row0 + tid / TPR + i * RPP, p.M - 1);
            pre[i] = zero8();
            if (kcol < p.K) pre[i] = *reinterpret_cast<const bf16x8*>(A + (size_t)row * p.lda + kcol);
        }
    };
    auto stage_store = [&](const bf16x8 (&pre)[NPF]) {
        constexpr int TPR = KC / 8, RPP = 256 / TPR;
        const int c8 = (tid % TPR) * 8;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int r = tid / TPR + i * RPP;
            if constexpr (F8) {
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = bf2f(pre[i][e]);
                put_a8(r, c8, f);
            } else {
                *reinterpret_cast<bf16x8*>(As + r * LDA + c8) = pre[i];
            }
        }
    };

    const int arow = lane & 15, ag = lane >> 4;

    // Weight fragments are fetched one group of G k-steps ahead of the MFMAs that use them, across k-chunk and
    // n-chunk boundaries: the group for the next n-chunk is in flight during this chunk's epilogue and the first
    // group overlaps the staging of A.  (Fetched at their use, every k-step exposed an L2 round trip.)
#ifndef HS_G_F8_128
#define HS_G_F8_128 2
#endif
#ifndef HS_G_BF_128
#define HS_G_BF_128 4
#endif
#ifndef HS_G_F8_64
#define HS_G_F8_64 2
#endif
    constexpr int G = F8 ? (BM == 128 ? HS_G_F8_128 : HS_G_F8_64)
#ifndef HS_G_BF_LN
#define HS_G_BF_LN 4      /* k-steps of weight fragments in flight in the bf16 LayerNorm-prologue GEMMs: 2 -> 4 once the staging batch went to 4 rows (153 VGPRs, still 3 waves per SIMD): Large 37.41 -> 37.16 ms */
#endif
                         : (DUAL ? 2 : (AK == A_F32_LN ? HS_G_BF_LN : (BM == 128 ? HS_G_BF_128 : 4)));
    constexpr int KSTEP = F8 ? 128 : 32;                      // K per MFMA
    using Frag = typename std::conditional<F8 != 0, i32x8, bf16x8>::type;
    struct Grp { Frag b[G][2]; Frag b2[DUAL ? G : 1][2]; unsigned sb[2]; unsigned sb2[2]; };
    auto fetch = [&](int nc_, int kc_, int gb_, Grp& gr) {
        const int ks0_ = kc_ * (KC / KSTEP);
        const int nks_ = min(KC / KSTEP, KS_total - ks0_);
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nt = nc_ * 8 + wave * 2 + j, ks = gb_ + i;
                const bool live = nc_ < n_chunks && ks < nks_ && nt < NT_total;
                if constexpr (F8) {
                    i32x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    gr.b[i][j] = z;
                    if constexpr (DUAL) gr.b2[i][j] = z;
                    if (live) {
                        const size_t off = (((size_t)nt * KS_total + ks0_ + ks) * 64 + lane) * 32;
                        const int4 lo = *reinterpret_cast<const int4*>(p.W8 + off), hi = *reinterpret_cast<const int4*>(p.W8 + off + 16);
                        gr.b[i][j] = i32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        if constexpr (DUAL) {
                            const int4 l2 = *reinterpret_cast<const int4*>(p.W8b + off), h2 = *reinterpret_cast<const int4*>(p.W8b + off + 16);
                            gr.b2[i][j] = i32x8{l2.x, l2.y, l2.z, l2.w, h2.x, h2.y, h2.z, h2.w};
                        }
                    }
                } else {
                    if (live) {
                        const size_t off = (((size_t)nt * KS_total + ks0_ + ks) * 64 + lane) * 8;
                        gr.b[i][j] = *reinterpret_cast<const bf16x8*>(p.W + off);
                        if constexpr (DUAL) gr.b2[i][j] = *reinterpret_cast<const bf16x8*>(p.W2 + off);
                    } else {
                        gr.b[i][j] = zero8();
                        if constexpr (DUAL) gr.b2[i][j] = zero8();
                    }
                }
            }
        if constexpr (F8) {       // the scale dword of this (n-tile, 512-deep chunk): byte s = k-step s of the chunk
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nt = nc_ * 8 + wave * 2 + j;
                const bool live = nc_ < n_chunks && kc_ < k_chunks && nt < NT_total;
                const size_t so = (((size_t)nt * k_chunks + kc_) * 64 + lane) * 4;
                gr.sb[j] = live ? *reinterpret_cast<const unsigned*>(p.S8 + so) : 0u;
                gr.sb2[j] = 0u;
                if constexpr (DUAL) gr.sb2[j] = live ? *reinterpret_cast<const unsigned*>(p.S8b + so) : 0u;
            }
        }
    };
    Grp cur;
    fetch(0, 0, 0, cur);
    PH_DECL

    // one group of G k-steps of the current A chunk against the weight fragments in `cur`
    auto mma = [&](f32x4 (&acc)[MT][2], f32x4 (&acc2)[DUAL ? MT : 1][2], const Grp& cur, int gb, int nks, const unsigned (&sa)[F8 ? MT : 1]) {
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    const int ks = gb + i;
                    if (ks < nks) {
                        if constexpr (F8) {
                            const int sh = 8 * ks;                     // this k-step's byte of the scale dwords -> byte 0
                            const int sb0 = (int)(cur.sb[0] >> sh), sb1 = (int)(cur.sb[1] >> sh);
                            const int tb0 = (int)(cur.sb2[0] >> sh), tb1 = (int)(cur.sb2[1] >> sh);
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) {
                                const unsigned char* ap = As8 + (mt * 16 + arow) * LDA + ks * 128 + ag * 16;
                                const int4 lo = *reinterpret_cast<const int4*>(ap), hi = *reinterpret_cast<const int4*>(ap + 64);
                                const i32x8 a = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                                const int sam = (int)(sa[mt] >> sh);
                                acc[mt][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, cur.b[i][0], acc[mt][0], 0, 0, 0, sam, 0, sb0);
                                acc[mt][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, cur.b[i][1], acc[mt][1], 0, 0, 0, sam, 0, sb1);
                                if constexpr (DUAL) {
                                    acc2[mt][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, cur.b2[i][0], acc2[mt][0], 0, 0, 0, sam, 0, tb0);
                                    acc2[mt][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, cur.b2[i][1], acc2[mt][1], 0, 0, 0, sam, 0, tb1);
                                }
                            }
                        } else {
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt) {
                                const bf16x8 a = *reinterpret_cast<const bf16x8*>(As + (mt * 16 + arow) * LDA + ks * 32 + ag * 8);
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    acc[mt][j] = mfma16(a, cur.b[i][j], acc[mt][j]);
                                    if constexpr (DUAL) acc2[mt][j] = mfma16(a, cur.b2[i][j], acc2[mt][j]);
                                }
                            }
                        }
                    }
                }
    };

    // epilogue of one 128-column chunk
    auto epilogue = [&](int nc, f32x4 (&acc)[MT][2], f32x4 (&acc2)[DUAL ? MT : 1][2]) {
        // ---------------------------------------------------------------- epilogue for this 128-column chunk
        const int ccols = min(128, p.N - nc * 128);            // valid columns in this chunk (multiple of 16)
        if constexpr (EPI == E_LN_BWD) {
            // LayerNorm backward on the product (N = 128: a row is the 16 adjacent lanes of one piece row).
            // The row's x and residual-gradient loads are issued before the tile exchange so they overlap it.
            const int c8 = (tid & 15) * 8;
            float gm[8], dgam[8], dbet[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { gm[e] = p.gamma[c8 + e]; dgam[e] = 0.f; dbet[e] = 0.f; }
#pragma unroll
            for (int ps = 0; ps < BM / PR; ++ps) {
                constexpr int NPC = PR * 16 / 256;
                __builtin_amdgcn_sched_barrier(0);         // keep each pass's loads in its pass (hoisted, they cost a wave of occupancy)
                float xr[NPC][8], rs[NPC][8];
#pragma unroll
                for (int i = 0; i < NPC; ++i) {
                    const int row = row0 + ps * PR + ((tid + 256 * i) >> 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { xr[i][e] = 0.f; rs[i][e] = 0.f; }
                    if (row < p.M) {
                        const float* xp = p.lnx + (size_t)row * p.ldr + c8;
                        const float* rp = p.res + (size_t)row * p.ldr + c8;
                        const float4 a0 = *reinterpret_cast<const float4*>(xp), a1 = *reinterpret_cast<const float4*>(xp + 4);
                        const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                        xr[i][0] = a0.x; xr[i][1] = a0.y; xr[i][2] = a0.z; xr[i][3] = a0.w;
                        xr[i][4] = a1.x; xr[i][5] = a1.y; xr[i][6] = a1.z; xr[i][7] = a1.w;
                        rs[i][0] = r0.x; rs[i][1] = r0.y; rs[i][2] = r0.z; rs[i][3] = r0.w;
                        rs[i][4] = r1.x; rs[i][5] = r1.y; rs[i][6] = r1.z; rs[i][7] = r1.w;
                        if (p.accumulate) {
                            const float* op = reinterpret_cast<const float*>(p.out) + (size_t)row * p.ldo + c8;
                            const float4 o0 = *reinterpret_cast<const float4*>(op), o1 = *reinterpret_cast<const float4*>(op + 4);
                            rs[i][0] += o0.x; rs[i][1] += o0.y; rs[i][2] += o0.z; rs[i][3] += o0.w;
                            rs[i][4] += o1.x; rs[i][5] += o1.y; rs[i][6] += o1.z; rs[i][7] += o1.w;
                        }
                    }
                }
                lds_barrier();
#pragma unroll
                for (int mi = 0; mi < PR / 16; ++mi)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            T1[(mi * 16 + ag * 4 + r) * TS + wave * 32 + j * 16 + arow] = acc[ps * (PR / 16) + mi][j][r];
                lds_barrier();
#pragma unroll
                for (int i = 0; i < NPC; ++i) {
                    const int rl = (tid + 256 * i) >> 4;
                    const int row = row0 + ps * PR + rl;
                    const float4 t0 = *reinterpret_cast<const float4*>(T1 + rl * TS + c8);
                    const float4 t1 = *reinterpret_cast<const float4*>(T1 + rl * TS + c8 + 4);
                    const float du[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                    float sm = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) sm += xr[i][e];
                    sm = lanes_sum<16>(sm);
                    const float mean = sm * (1.f / 128.f);
                    float q = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { xr[i][e] -= mean; q += xr[i][e] * xr[i][e]; }
                    q = lanes_sum<16>(q);
                    const float rstd = rsqrtf(q * (1.f / 128.f) + 1e-5f);
                    float a = 0.f, b = 0.f, t[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { xr[i][e] *= rstd; t[e] = du[e] * gm[e]; a += t[e]; b += t[e] * xr[i][e]; }
                    a = lanes_sum<16>(a); b = lanes_sum<16>(b);
                    a *= (1.f / 128.f); b *= (1.f / 128.f);
                    if (row < p.M) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            v[e] = rs[i][e] + rstd * (t[e] - a - xr[i][e] * b);
                            dgam[e] += du[e] * xr[i][e];
                            dbet[e] += du[e];
                        }
                        float* op = reinterpret_cast<float*>(p.out) + (size_t)row * p.ldo + c8;
                        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    }
                }
            }
            // dgamma / dbeta: 16 threads per octet in this workgroup -> LDS, one atomic per column
            lds_barrier();
            float* red = T1;                                   // [2][256][8] floats = 16 KB <= PR * TS * 4
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[tid * 8 + e] = dgam[e]; red[2048 + tid * 8 + e] = dbet[e]; }
            lds_barrier();
            {
                const int which = tid >> 7, c = tid & 127, o8 = c >> 3, e = c & 7;
                float sacc = 0.f;
                for (int t2 = o8; t2 < 256; t2 += 16) sacc += red[which * 2048 + t2 * 8 + e];
                hs_gadd(HsDet{p.det_base, reinterpret_cast<long long*>(p.det_acc)}, (which ? p.dbeta : p.dgamma) + c, sacc);
            }
            PH(2)
            return;
        }
#pragma unroll
        for (int ps = 0; ps < BM / PR; ++ps) {
            lds_barrier();                                   // previous pass fully consumed
#pragma unroll
            for (int mi = 0; mi < PR / 16; ++mi)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = (mi * 16 + ag * 4 + r) * TS + wave * 32 + j * 16 + arow;
                        T1[o] = acc[ps * (PR / 16) + mi][j][r];
                        if constexpr (DUAL) T2[o] = acc2[ps * (PR / 16) + mi][j][r];
                    }
            lds_barrier();
#pragma unroll
            for (int i = 0; i < PR * 16 / 256; ++i) {
                const int piece = tid + 256 * i;
                const int rl = piece >> 4, c8 = (piece & 15) * 8;
                const int row = row0 + ps * PR + rl;
                const int col = nc * 128 + c8;
                if (row >= p.M || c8 >= ccols) continue;
                const float4 t0 = *reinterpret_cast<const float4*>(T1 + rl * TS + c8);
                const float4 t1 = *reinterpret_cast<const float4*>(T1 + rl * TS + c8 + 4);
                float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                const int nv = p.n_valid - col;                // valid columns in this octet (>= 8: all)
                const bool cvalid = nv > 0;
                if (p.bias && cvalid) {
                    if (nv >= 8) {
                        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + col);
                        const float4 b1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
                        v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                        v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (e < nv) v[e] += p.bias[col + e];
                    }
                }
                if constexpr (EPI == E_BF16) {
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)row * p.ldo + col) = cvt8(v);
                } else if constexpr (EPI == E_F32 || EPI == E_RES_F32 || EPI == E_POS_F32) {
                    if (!cvalid) continue;
                    if constexpr (EPI == E_RES_F32) {
                        if (p.out_rowscale) {                     // DropPath: x + scale * branch
                            const float rs = p.out_rowscale[row];
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] *= rs;
                        }
                        const float* rp = p.res + (size_t)row * p.ldr + col;
                        const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                        v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                        if (p.res2) {
                            const float* qp = p.res2 + (size_t)row * p.ldr + col;
                            const float4 q0 = *reinterpret_cast<const float4*>(qp), q1 = *reinterpret_cast<const float4*>(qp + 4);
                            v[0] += q0.x; v[1] += q0.y; v[2] += q0.z; v[3] += q0.w;
                            v[4] += q1.x; v[5] += q1.y; v[6] += q1.z; v[7] += q1.w;
                        }
                    } else if constexpr (EPI == E_POS_F32) {
                        const float* rp = p.pos + (size_t)p.ids[row] * p.ldpos + col;
                        const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w;
                        v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                    }
                    float* op = reinterpret_cast<float*>(p.out) + (size_t)row * p.ldo + col;
                    *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
                } else if constexpr (EPI == E_SWIGLU) {
                    const float4 s0 = *reinterpret_cast<const float4*>(T2 + rl * TS + c8);
                    const float4 s1 = *reinterpret_cast<const float4*>(T2 + rl * TS + c8 + 4);
                    float w[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                    if (p.bias2 && cvalid) {
                        if (nv >= 8) {
                            const float4 b0 = *reinterpret_cast<const float4*>(p.bias2 + col);
                            const float4 b1 = *reinterpret_cast<const float4*>(p.bias2 + col + 4);
                            w[0] += b0.x; w[1] += b0.y; w[2] += b0.z; w[3] += b0.w;
                            w[4] += b1.x; w[5] += b1.y; w[6] += b1.z; w[7] += b1.w;
                        } else {
#pragma unroll
                            for (int e = 0; e < 8; ++e) if (e < nv) w[e] += p.bias2[col + e];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (e >= nv) { v[e] = 0.f; w[e] = 0.f; }
                    const bf16x8 h1 = cvt8(v), h3 = cvt8(w);
                    HS_NT(true, reinterpret_cast<bf16x8*>(p.h13 + (size_t)row * p.ldh + col), h1);             // pre-activations: saved for the backward only
                    HS_NT(true, reinterpret_cast<bf16x8*>(p.h13 + (size_t)row * p.ldh + p.hoff + col), h3);
                    float g[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float a1 = bf2f(h1[e]), a3 = bf2f(h3[e]);       // the values the backward will see
                        g[e] = silu_nr(a1) * a3;
                    }
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)row * p.ldo + col) = cvt8(g);
                } else if constexpr (EPI == E_SWIGLU_BWD) {
                    const bf16x8 h1 = *reinterpret_cast<const bf16x8*>(p.h13 + (size_t)row * p.ldh + col);
                    const bf16x8 h3 = *reinterpret_cast<const bf16x8*>(p.h13 + (size_t)row * p.ldh + p.hoff + col);
                    float d1[8], d3[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float a1 = bf2f(h1[e]), a3 = bf2f(h3[e]);
                        const float s = __builtin_amdgcn_rcpf(1.f + __expf(-a1));
                        d1[e] = v[e] * a3 * s * (1.f + a1 * (1.f - s));
                        d3[e] = v[e] * a1 * s;
                    }
                    bf16_t* o = reinterpret_cast<bf16_t*>(p.out);
                    *reinterpret_cast<bf16x8*>(o + (size_t)row * p.ldo + col) = cvt8(d1);
                    *reinterpret_cast<bf16x8*>(o + (size_t)row * p.ldo + p.hoff + col) = cvt8(d3);
                }
            }
        }
        PH(2)
    };

    // LayerNorm backward over rows wider than one 128-column chunk (k-outer only: all NCH chunks' accumulators are live, so
    // the whole row of du is on chip): out = res (+ out) + LNbwd(du; lnx, gamma), dgamma / dbeta through an LDS table
    // (ds_add per pass) and one commit per column and workgroup.  Replaces a separate du store + ln_bwd pass at d = 256 / 512.
    auto epilogue_ln_ko = [&](f32x4 (&accs)[NCH][MT][2]) {
        if constexpr (EPI == E_LN_BWD && NCH > 1) {
            constexpr int PRL = 16;                               // rows per pass: NCH tiles of [16][TS] floats
            constexpr int NW = 128 * NCH;                         // row width = N
            float* TT = T1;                                       // NCH tiles
            float dgam[NCH][8], dbet[NCH][8];                     // this thread's column octets, summed over its rows
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) { dgam[c][e] = 0.f; dbet[c][e] = 0.f; }
            const int c8 = (tid & 15) * 8, rl = tid >> 4;         // 16 rows x 16 octets per pass and chunk
            float gm[NCH][8];
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) gm[c][e] = p.gamma[c * 128 + c8 + e];
            const float invn = 1.f / (float)NW;
#pragma unroll
            for (int ps = 0; ps < BM / PRL; ++ps) {
                __builtin_amdgcn_sched_barrier(0);
                const int row = row0 + ps * PRL + rl;
                const bool rok = row < p.M;
                float xr[NCH][8];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) xr[c][e] = 0.f;
                    if (rok) {
                        const float* xp = p.lnx + (size_t)row * p.ldr + c * 128 + c8;
                        const float4 a0 = *reinterpret_cast<const float4*>(xp), a1 = *reinterpret_cast<const float4*>(xp + 4);
                        xr[c][0] = a0.x; xr[c][1] = a0.y; xr[c][2] = a0.z; xr[c][3] = a0.w;
                        xr[c][4] = a1.x; xr[c][5] = a1.y; xr[c][6] = a1.z; xr[c][7] = a1.w;
                    }
                }
                lds_barrier();                                    // previous pass's tiles consumed (and GB zeroed)
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            TT[(c * PRL + ag * 4 + r) * TS + wave * 32 + j * 16 + arow] = accs[c][ps][j][r];
                lds_barrier();
                // row statistics of x over all NCH chunks (16 lanes x NCH octets)
                float sm = 0.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e) sm += xr[c][e];
                sm = lanes_sum<16>(sm);
                const float mean = sm * invn;
                float q = 0.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { xr[c][e] -= mean; q += xr[c][e] * xr[c][e]; }
                q = lanes_sum<16>(q);
                const float rstd = rsqrtf(q * invn + 1e-5f);
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const float4 t0 = *reinterpret_cast<const float4*>(TT + (c * PRL + rl) * TS + c8);
                    const float4 t1 = *reinterpret_cast<const float4*>(TT + (c * PRL + rl) * TS + c8 + 4);
                    const float du[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        xr[c][e] *= rstd;                         // x-hat
                        const float t = du[e] * gm[c][e];
                        a += t; b += t * xr[c][e];
                        if (rok) { dgam[c][e] += du[e] * xr[c][e]; dbet[c][e] += du[e]; }
                    }
                }
                a = lanes_sum<16>(a); b = lanes_sum<16>(b);
                a *= invn; b *= invn;
                if (rok) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const float4 t0 = *reinterpret_cast<const float4*>(TT + (c * PRL + rl) * TS + c8);
                        const float4 t1 = *reinterpret_cast<const float4*>(TT + (c * PRL + rl) * TS + c8 + 4);
                        const float du[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                        const float* rp = p.res + (size_t)row * p.ldr + c * 128 + c8;
                        const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                        float v[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
                        float* op = reinterpret_cast<float*>(p.out) + (size_t)row * p.ldo + c * 128 + c8;
                        if (p.accumulate) {
                            const float4 o0 = *reinterpret_cast<const float4*>(op), o1 = *reinterpret_cast<const float4*>(op + 4);
                            v[0] += o0.x; v[1] += o0.y; v[2] += o0.z; v[3] += o0.w; v[4] += o1.x; v[5] += o1.y; v[6] += o1.z; v[7] += o1.w;
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += rstd * (du[e] * gm[c][e] - a - xr[c][e] * b);
                        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
                        // optional bf16 copy of the result (u_out / ldu, round 6): the weight-gradient operand the caller otherwise
                        // makes with a rows_to_bf16 pass over the rows just written (1.18 ms of the Huge step for both copies)
                        if (p.u_out) *reinterpret_cast<bf16x8*>(p.u_out + (size_t)row * p.ldu + c * 128 + c8) = cvt8(v);
                    }
                }
            }
            // dgamma / dbeta: the 16 threads that share an octet -> LDS, fixed-order sum, one commit per column and workgroup
            const HsDet det{p.det_base, reinterpret_cast<long long*>(p.det_acc)};
            float* red = T1;                                      // [2][256][8] floats = 16 KB <= NCH * PRL * TS * 4
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                lds_barrier();
#pragma unroll
                for (int e = 0; e < 8; ++e) { red[tid * 8 + e] = dgam[c][e]; red[2048 + tid * 8 + e] = dbet[c][e]; }
                lds_barrier();
                const int which = tid >> 7, col = tid & 127, o8 = col >> 3, e = col & 7;
                float sacc = 0.f;
                for (int t2 = o8; t2 < 256; t2 += 16) sacc += red[which * 2048 + t2 * 8 + e];
                hs_gadd(det, (which ? p.dbeta : p.dgamma) + c * 128 + col, sacc);
            }
        }
    };

    if constexpr (NCH == 1) {
    for (int nc = 0; nc < n_chunks; ++nc) {
        f32x4 acc[MT][2];
        f32x4 acc2[DUAL ? MT : 1][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (DUAL) acc2[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        for (int kc = 0; kc < k_chunks; ++kc) {
            if (k_chunks > 1 || nc == 0) {
                if (!(nc == 0 && kc == 0)) lds_barrier();
                if constexpr (AK == A_F32_LN) stage_ln(); else stage(kc);
                lds_barrier();
                PH(0)
            }
            const int ks0 = kc * (KC / KSTEP);
            const int nks = min(KC / KSTEP, KS_total - ks0);
            unsigned sa[F8 ? MT : 1];
            if constexpr (F8) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) sa[mt] = *reinterpret_cast<const unsigned*>(Sc + ((mt * 16 + arow) * 4 + ag) * 4);
            }
            for (int gb = 0; gb < nks; gb += G) {
                int nnc = nc, nkc = kc, ngb = gb + G;
                if (ngb >= nks) { ngb = 0; if (++nkc >= k_chunks) { nkc = 0; ++nnc; } }
                Grp nxt;
                fetch(nnc, nkc, ngb, nxt);
                mma(acc, acc2, cur, gb, nks, sa);
                cur = nxt;
            }
        }
        PH(1)
        epilogue(nc, acc, acc2);
    }
    } else {
        // k outer: every A chunk is staged (and, in fp8, quantised) ONCE and multiplied against all n-chunks, whose
        // accumulators are all live (deep-K products with N <= 128 NCH: w2, the W1|W3 and q|k|v data gradients)
        static_assert(NCH == 1 || !DUAL, "the gate pair keeps one accumulator set");
        f32x4 accs[NCH][MT][2];
        f32x4 acc2[1][2];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 2; ++j) accs[c][mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 pre[NPF];
        if constexpr (AK == A_BF16) stage_load(0, pre);
        for (int kc = 0; kc < k_chunks; ++kc) {
            if (kc) lds_barrier();
            if constexpr (AK == A_BF16) stage_store(pre); else stage(kc);
            lds_barrier();
            if constexpr (AK == A_BF16) {
                if (kc + 1 < k_chunks) stage_load(kc + 1, pre);
            }
            PH(0)
            const int ks0 = kc * (KC / KSTEP);
            const int nks = min(KC / KSTEP, KS_total - ks0);
            unsigned sa[F8 ? MT : 1];
            if constexpr (F8) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) sa[mt] = *reinterpret_cast<const unsigned*>(Sc + ((mt * 16 + arow) * 4 + ag) * 4);
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c < n_chunks) {
                    for (int gb = 0; gb < nks; gb += G) {
                        int nnc = c, nkc = kc, ngb = gb + G;
                        if (ngb >= nks) { ngb = 0; if (++nnc >= n_chunks) { nnc = 0; ++nkc; } }
                        Grp nxt;
                        fetch(nnc, nkc, ngb, nxt);
                        mma(accs[c], acc2, cur, gb, nks, sa);
                        cur = nxt;
                    }
                }
            }
        }
        PH(1)
        if constexpr (EPI == E_LN_BWD) {
            epilogue_ln_ko(accs);
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (c < n_chunks) epilogue(c, accs[c], acc2);
        }
    }
    PH_FLUSH(((AK * 3 + (EPI == E_LN_BWD ? 2 : (EPI == E_BF16 ? 0 : 1))) * 4) % 64)
}

template <int AK, int EPI, int KC, int BM, int F8 = 0, int NCH = 1>
int launch(const GemmParams& p, hipStream_t s) {
    const int grid = (p.M + BM - 1) / BM;
    const size_t abytes = F8 ? (size_t)BM * (KC + 16) + BM * 16 : (size_t)BM * (KC + 8) * 2;
    size_t tiles = (EPI == E_SWIGLU ? 2 : 1) * EpiRows<KC, F8, BM>::v * TS * sizeof(float);
    if (EPI == E_LN_BWD && NCH > 1) tiles = std::max(tiles, (size_t)NCH * 16 * TS * sizeof(float));
    const size_t lds = abytes + BM * 2 * sizeof(float) + tiles;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<AK, EPI, KC, BM, F8, NCH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<AK, EPI, KC, BM, F8, NCH>), dim3(grid), dim3(256), lds, s, p);
    return (int)hipGetLastError();
}

// Tiling of the wide layers, from scripts/gemm_sweep.py at the Large / Huge shapes (profiles/r02_gemm_sweep.txt):
// 64-row panels (two to three workgroups per CU: one panel's staging / epilogue overlaps another's MFMA loop) win
// wherever the weight image is small enough that streaming it once per 64 rows instead of once per 128 costs less than
// the overlap gains: q|k|v at d = 256 235 -> 175 us, gate backward 461 -> 216 us; they lose at d = 512 when K is deep
// (w2 303 -> 355 us).  256-deep A chunks pay only together with 64-row panels (alone: 1.3-1.7x slower).
// (The A/B switches of this family — HSIMAE_GEMM_WIDE, _GEMM_KO_BM, _GEMM_KOUTER, _GEMM_F8_BM — went in round 6; hsimae_gemm_tiled
//  still forces a tile shape for sweeps.)
static int wide_mode() { return 1; }
static thread_local int g_force_bm = 0, g_force_kc = 0;       // hsimae_gemm_tiled (tile sweeps): 0 = the shape rule below
static int ko_bm() { return 0; }          // panel height of the k-outer products: the shape rule below
static bool k_outer() { return true; }
static bool small_weights(const GemmParams& p, bool dual) {
    const int64_t nk = (int64_t)p.N * p.K;
    if (p.K < 256) return false;
    if (dual) return nk <= 400 * 1024;
    return p.K <= 512 ? nk <= 800 * 1024 : nk <= 400 * 1024;
}

// fp8 (MX) form: 512-deep e4m3 chunks; 64-row panels by the same weight-size rule (any K)
template <int AK, int EPI>
int launch_f8(const GemmParams& p, hipStream_t s) {
    if constexpr (EPI == E_LN_BWD || EPI == E_POS_F32) {
        return HS_EUNSUPPORTED;
    } else {
        if (!p.W8 || !p.S8 || (EPI == E_SWIGLU && (!p.W8b || !p.S8b))) return HSIMAE_ENULL;
        if (AK == A_F32_LN && p.K > 512) return HS_EUNSUPPORTED;
        const int64_t nk = (int64_t)p.N * p.K * (EPI == E_SWIGLU ? 2 : 1);
        // 64-row panels for every fp8 product: the e4m3 weight stream per panel is half the bf16 one, and two to three resident
        // workgroups hide the fragment fetches that one 128-row workgroup exposes (Huge step 57.7 -> 55.6 ms; 128 rows everywhere:
        // 76.8 ms, and a deeper fragment prefetch there does not help: 87.8 ms)
        bool bm64 = true;
        (void)nk;
        if (g_force_bm) bm64 = g_force_bm == 64;
        if constexpr (AK != A_F32_LN && EPI != E_SWIGLU) {
            // deep K (several 512-chunks) and at most 4 n-chunks: k outer on 64-row panels, every chunk quantised once
            if (k_outer() && p.K > 512 && p.N > 128 && p.N <= 512 && g_force_bm != 128) {
                if constexpr (AK == A_BF16) {         // register-prefetched A chunks: 32-row panels keep them at 32 registers
                    if (ko_bm() != 64) return p.N <= 256 ? launch<AK, EPI, 512, 32, 1, 2>(p, s) : launch<AK, EPI, 512, 32, 1, 4>(p, s);
                }
                return p.N <= 256 ? launch<AK, EPI, 512, 64, 1, 2>(p, s) : launch<AK, EPI, 512, 64, 1, 4>(p, s);
            }
        }
        return bm64 ? launch<AK, EPI, 512, 64, 1>(p, s) : launch<AK, EPI, 512, 128, 1>(p, s);
    }
}

template <int AK, int EPI>
int launch_kc(const GemmParams& p, hipStream_t s) {
    if (p.prec == HSIMAE_PREC_FP8) return launch_f8<AK, EPI>(p, s);
    bool bm64 = wide_mode() && small_weights(p, EPI == E_SWIGLU);
    if (g_force_bm) bm64 = g_force_bm == 64;
    if constexpr (AK == A_F32_LN) {
        if (p.K <= 128) return launch<AK, EPI, 128, 128>(p, s);
        if (p.K <= 256) return bm64 ? launch<AK, EPI, 256, 64>(p, s) : launch<AK, EPI, 256, 128>(p, s);
        if (p.K <= 512) return bm64 ? launch<AK, EPI, 512, 64>(p, s) : launch<AK, EPI, 512, 128>(p, s);
        return HS_EUNSUPPORTED;
    } else if constexpr (EPI == E_LN_BWD) {
        return launch<AK, EPI, 128, 128>(p, s);
    } else {
        const bool kc256 = g_force_kc ? g_force_kc == 256 : (bm64 && p.K >= 256 && p.K <= 1024);
        if constexpr (EPI != E_SWIGLU && EPI != E_SWIGLU_BWD) {
            if (k_outer() && !g_force_bm && p.K > 256 && p.N > 128 && p.N <= 512) {
                if constexpr (AK == A_BF16) {
                    if (ko_bm() == 32) return p.N <= 256 ? launch<AK, EPI, 256, 32, 0, 2>(p, s) : launch<AK, EPI, 256, 32, 0, 4>(p, s);
                }
                return p.N <= 256 ? launch<AK, EPI, 256, 64, 0, 2>(p, s) : launch<AK, EPI, 256, 64, 0, 4>(p, s);
            }
        }
        if (kc256) return bm64 ? launch<AK, EPI, 256, 64>(p, s) : launch<AK, EPI, 256, 128>(p, s);
        return bm64 ? launch<AK, EPI, 128, 64>(p, s) : launch<AK, EPI, 128, 128>(p, s);
    }
}

}  // namespace

int hs_gemm(const GemmParams& p, int akind, int epi, hipStream_t s);
int hs_gemm_tiled(const GemmParams& p, int akind, int epi, int bm, int kc, hipStream_t s) {
    if ((bm != 0 && bm != 64 && bm != 128) || (kc != 0 && kc != 128 && kc != 256)) return HS_EDIMS;
    g_force_bm = bm; g_force_kc = kc;
    const int rc = hs_gemm(p, akind, epi, s);
    g_force_bm = 0; g_force_kc = 0;
    return rc;
}

int hs_gemm(const GemmParams& p, int akind, int epi, hipStream_t s) {
    if (p.M <= 0) return HS_OK;
    if (p.K % 32 || p.N % 16 || p.lda % 8 || p.ldo % 8) return HS_EDIMS;
    if ((epi == E_F32 || epi == E_RES_F32 || epi == E_POS_F32) && p.n_valid % 8) return HS_EDIMS;
#define CASE(AK, EP) \
    if (akind == AK && epi == EP) return launch_kc<AK, EP>(p, s);
    CASE(A_F32_LN, E_BF16)
    CASE(A_F32_LN, E_SWIGLU)
    CASE(A_F32_LN, E_F32)
    CASE(A_BF16, E_RES_F32)
    CASE(A_BF16, E_POS_F32)
    CASE(A_BF16, E_F32)
    CASE(A_BF16, E_BF16)
    if (akind == A_BF16 && epi == E_LN_BWD && p.N == 512) {
        // 512-wide rows: four accumulator sets on 32-row panels (on 64-row panels they spilled 100 registers)
        if (p.n_valid != p.N || !p.lnx || !p.res || !p.gamma || !p.dgamma || !p.dbeta || p.ldr % 4 || p.ldo % 4) return HS_EUNSUPPORTED;
        if (p.prec == HSIMAE_PREC_FP8) {
            if (!p.W8 || !p.S8) return HSIMAE_ENULL;
            return launch<A_BF16, E_LN_BWD, 512, 32, 1, 4>(p, s);
        }
        return launch<A_BF16, E_LN_BWD, 256, 32, 0, 4>(p, s);
    }
    if (akind == A_BF16 && epi == E_LN_BWD && p.N == 256) {
        // 256-wide rows: the k-outer form (both chunks' accumulators live), bf16 or MX fp8 operands.  (At N = 512 the four
        // accumulator sets + the per-thread dgamma / dbeta sums spill 100 registers; an LDS table for those sums with
        // ds_add was 9 ms slower per Huge step than the separate ln_bwd pass.)
        if (p.n_valid != p.N || !p.lnx || !p.res || !p.gamma || !p.dgamma || !p.dbeta || p.ldr % 4 || p.ldo % 4) return HS_EUNSUPPORTED;
        if (p.prec == HSIMAE_PREC_FP8) {
            if (!p.W8 || !p.S8) return HSIMAE_ENULL;
            return ko_bm() != 64 ? launch<A_BF16, E_LN_BWD, 512, 32, 1, 2>(p, s) : launch<A_BF16, E_LN_BWD, 512, 64, 1, 2>(p, s);
        }
        return ko_bm() == 32 ? launch<A_BF16, E_LN_BWD, 256, 32, 0, 2>(p, s) : launch<A_BF16, E_LN_BWD, 256, 64, 0, 2>(p, s);
    }
    if (akind == A_BF16 && epi == E_LN_BWD) {
        if (p.prec == HSIMAE_PREC_FP8) return HS_EUNSUPPORTED;
        if (p.N != 128 || p.n_valid != 128 || !p.lnx || !p.res || !p.gamma || !p.dgamma || !p.dbeta || p.ldr % 4 || p.ldo % 4)
            return HS_EUNSUPPORTED;
        return launch_kc<A_BF16, E_LN_BWD>(p, s);
    }
    CASE(A_F32, E_SWIGLU_BWD)
    CASE(A_F32, E_BF16)
    CASE(A_F32, E_F32)
#undef CASE
    return HS_EUNSUPPORTED;
}

HS_UNIT_VARIANT_BITS(gemm)
