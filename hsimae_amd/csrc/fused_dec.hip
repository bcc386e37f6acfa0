// Fused decoder Block (Models.py:303-306 at d=64, 8 heads of 8, SwiGLU hidden 172) — one workgroup per sample.
//
// The decoder is [8 blocks, width 64] for every HSIMAE size (Model_Pretraining.py:131) and runs on all T*9
// tokens, so it is half of the step.  Layer-at-a-time kernels are HBM-bound here (32 FLOP/B at d=64); this
// kernel keeps one sample's 54/108/216 tokens (4/7/14 MFMA m-tiles) resident in LDS/registers through the
// whole block: LN1 -> q|k|v -> 8-head attention -> proj(+x) -> LN2 -> W1|W3 -> SiLU gate -> W2(+x1).
// HBM traffic per token: read x (256 B), write x1 and x2 (512 B) instead of ~3.2 KB.
//
// Workgroup = 8 waves as 4(M) x 2(N): wave (wm, wn) owns m-tiles [wm*MH, ...) and n-tiles {2wn, 2wn+1} of every
// 64-column chunk.  The residual stream lives in registers in MFMA accumulator layout (proj / W2 accumulate
// straight onto it); LayerNorm is done in a row-contiguous "wide" layout (8 lanes per row) through an fp32
// LDS staging tile, which is also how x / x1 / x2 move to and from HBM with 16-B accesses.
#include "common.h"
#include "kernels.h"
#include <type_traits>
#include <cstdlib>

// Per-phase cycle accounting for scripts/phase_timing.py (compiled only with -DHS_PHASE_TIMING; never in the shipped library)
#ifdef HS_PHASE_TIMING
__device__ unsigned long long hs_phase_cycles[32];
extern "C" __attribute__((visibility("default"))) int hsimae_debug_phases(unsigned long long* out, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hs_phase_cycles), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(hs_phase_cycles), z, sizeof(z)); }
    return rc;
}
#define PH_DECL unsigned long long ph_t0 = __builtin_readcyclecounter(), ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PH(i) { const unsigned long long ph_t = __builtin_readcyclecounter(); ph_acc[i] += ph_t - ph_t0; ph_t0 = ph_t; }
#define PH_FLUSH(base) if (threadIdx.x == 0) { for (int i = 0; i < 8; ++i) atomicAdd(&hs_phase_cycles[(base) + i], ph_acc[i]); }
// a second set of stamps inside one phase (round 5: the du + dWqkv phase of dec_bwd_attn), slots 24..31
#define PH2_DECL unsigned long long ph_t1 = 0, ph_acc2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PH2_START ph_t1 = __builtin_readcyclecounter();
#define PH2(i) { const unsigned long long ph_t = __builtin_readcyclecounter(); ph_acc2[i] += ph_t - ph_t1; ph_t1 = ph_t; }
#define PH2_FLUSH if (threadIdx.x == 0) { for (int i = 0; i < 8; ++i) atomicAdd(&hs_phase_cycles[24 + i], ph_acc2[i]); }
#else
#define PH_DECL
#define PH(i)
#define PH_FLUSH(base)
#define PH2_DECL
#define PH2_START
#define PH2(i)
#define PH2_FLUSH
#endif

#ifndef HS_NT_D
#define HS_NT_D 1      /* x1 / o saved by the decoder forward for the backward as streaming stores (step -0.5 %) */
#endif
/* next-sample L2 / TLB warm-up loads in the persistent decoder kernels (forward / MLP backward / attention backward) */
#ifndef HS_TOUCH_F
#define HS_TOUCH_F 0
#endif
#ifndef HS_TOUCH_M
#define HS_TOUCH_M 0
#endif
// Start stagger of the persistent decoder backward kernels (round 5).  All 256 workgroups start together and walk 16 samples of
// identical work each, so the whole chip is in one phase at a time: every sample's rows (73 KB per workgroup in dec_bwd_attn)
// are requested by all CUs at once — at the CU's 1/256 share of HBM, ~10 B/clk, that burst takes ~3 us during which the issuing
// waves are blocked (phase stamps, profiles/r05_f_phase_timing.txt: ISSUING the next sample's loads is 55 % of the du + dWqkv
// phase of dec_bwd_attn, 33 us per launch; the prologue wait of dec_bwd_mlp is 21 % of that kernel) while HBM idles for the rest
// of the sample.  With HS_DEC_STG_N = n classes, workgroup b sleeps ((b >> 3) % n) x HS_DEC_STG_ATTN / _MLP x 0.85 us once, before
// its first sample, so that the classes' bursts interleave for the whole launch.  Measured (profiles/r05_g_*, r05_h_*, r05_i_*,
// kernel statistics of three boxes): dec_bwd_attn with 2 classes 3.4-5.1 us apart 322 -> 305, 323 -> 309 / 310 us (- 4 %; 1.7 us:
// 321, 6.8 us: 312, 10 us: 310 / 301, 15 us: 328; 3 classes 312, 4 classes 305-308, 8 classes 307); dec_bwd_mlp LOSES with every
// offset tried (213 -> 217-220 us at 3.4-6.8 us, 213 at 1.7 us): its stagger stays off.  The same on blk128_fwd / blk128_bwd: neutral.
#ifndef HS_DEC_STG_N
#define HS_DEC_STG_N 2
#endif
#ifndef HS_DEC_STG_ATTN
#define HS_DEC_STG_ATTN 5
#endif
// (round 6: with the next sample's rows requested early, HS_DEC_MLP_PREFETCH = 2, the stagger pays in dec_bwd_mlp too — 213.8 / 214.1 /
//  214.8 us without, 208.4 / 206.6 / 208.5 at 4, 209.4 / 208.1 / 209.1 at 5, 209.2 / 207.6 / 208.9 at 6; dec_bwd_attn is flat between
//  3 and 8: profiles/r06_k_dec_stagger_resweep.txt, r06_l_dec_stagger_combos.txt)
#ifndef HS_DEC_STG_MLP
#define HS_DEC_STG_MLP 4
#endif
// dec_bwd_mlp: when the next sample's x1 / dY rows are requested.  0 = at the top of its own iteration (rounds 2-5: the prologue wait
// was 21 % of the kernel); 1 = in front of the epilogue (round 5 with spills: 214 -> 220 us; round 6 without: 224 -> 221 us — half a
// round trip of lead time); 2 = at the start of the LAST hidden chunk's weight-gradient phase, a weight-gradient + du2 phase + epilogue
// ahead (round 6, default: 224.8 / 224.7 / 228.4 -> 215.8 / 214.2 / 215.5 us, profiles/r06_i_dec_mlp_prefetch2_ab.txt; no spill in
// the loop); 3 = a chunk earlier still: 31 registers spilled, not measured.
#ifndef HS_DEC_MLP_PREFETCH
#define HS_DEC_MLP_PREFETCH 2
#endif
#ifndef HS_TOUCH_A
#define HS_TOUCH_A 0
#endif
// Round 6: de-phasing the two waves of a SIMD inside the persistent 8-wave backward kernels.
//   HS_DEC_PRIO      s_setprio level of waves 4-7 (the second-dispatched half: MI355X_MICROARCH.md "Two waves per SIMD" item 4),
//                    set once before the sample loop; 0 = off
//   HS_DEC_CORE_NQ   query tiles a wave of dec_bwd_attn's attention core keeps in hand (1 = rounds 2-5; 2 = two independent
//                    score -> softmax -> dS -> transposition chains per wave, the K / V row fragments read once for both)
#ifndef HS_DEC_PRIO
#define HS_DEC_PRIO 0
#endif
#ifndef HS_DEC_CORE_NQ
#define HS_DEC_CORE_NQ 2      /* measured (profiles/r06_d_decoder_ab.txt): 308 -> 299 us per launch with two; the core is then ~75 % issue-bound */
#endif
// Lane geometry (GeoB and what is derived from it) re-derived from an opaque copy of threadIdx.x instead of kept alive across the
// sample loop: 0 = never, 1 = on entry to and exit from the attention core of dec_bwd_attn only, 2 = at every phase.
// Measured (profiles/r06_c_decoder_ab.txt, one box): with every phase re-deriving, dec_bwd_mlp 227 -> 240 us and dec_bwd_attn 312 ->
// 320 us (the ~30 VALU instructions per call are not free in kernels whose waves wait on VALU issue); without any, the
// two-query-tile core spills again.
#ifndef HS_DEC_REGEO
#define HS_DEC_REGEO 1
#endif
#ifndef HS_DEC_REGEO_MLP
#define HS_DEC_REGEO_MLP 0
#endif
#ifndef HS_DEC_PRELOOP_WAIT
#define HS_DEC_PRELOOP_WAIT 1
#endif
#ifndef HS_DEC_CORE_STG
#define HS_DEC_CORE_STG 0   /* waves 4-7 enter the attention core this many x 64 clocks late (half a tile period = 4) */
#endif

namespace {

constexpr int D = 64, HD = 8, HPD = 192;
constexpr int LU = D + 8;        // bf16 image row stride (elements)
constexpr int LG = HPD + 8;
constexpr int LX = D + 4;        // fp32 staging row stride (floats)
constexpr int WRM = HPD * LU;    // one row-major [192][LU] bf16 weight image staged in LDS (elements)
constexpr int NT_ = 512;         // threads per workgroup (8 waves: one attention head per wave)
constexpr int kDwSlotsMlp = 72, kDwSlotsAttn = 32, kDwSlots = kDwSlotsMlp + kDwSlotsAttn;   // in-register dW values per thread
// bias / LayerNorm gradient sums of a workgroup (second part of the slab, [workgroup][kVec]): offsets of the vectors
// (w1b / w3b: one 192-wide partial per wave row wm = 0..3 of the 4 x 2 wave grid — four waves hold sums of the same column)
constexpr int kVN2W = 0, kVN2B = 64, kVW2B = 128, kVW1B = 192, kVW3B = 960, kVN1W = 1728, kVN1B = 1792, kVPB = 1856, kVQB = 1920,
              kVKB = 1984, kVVB = 2048, kVec = 2112;
constexpr size_t kSlabTileFloats = (size_t)256 * kDwSlots * NT_;    // the vector part starts here
static_assert(kDecSlabFloats == 256ll * (kDwSlots * NT_ + kVec), "kernels.h kDecSlabFloats must cover the slab these kernels write");

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int MT>
struct DL {
    static constexpr int R = MT * 16;
    static constexpr int MH = (MT + 3) / 4;                // m-tiles per wave, 8-wave kernels (4 x 2 wave grid)
    static constexpr int MHF = (MT + 1) / 2;               // forward kernel: 4 waves (2 x 2), two workgroups per CU
    static constexpr int VST = R + 8;                      // transposed-V row stride (elements)
    static constexpr int U_BYTES = R * LU * 2;
    static constexpr int QKV_BYTES = 2 * U_BYTES + D * VST * 2;
    static constexpr int REG2 = cmax(cmax(QKV_BYTES, R * LG * 2), R * LX * 4);
    static constexpr int FWD_TOTAL = U_BYTES + REG2;
    // backward kernels (round 4 layouts, see "backward: LDS layouts" below): images are R x 128 B (no pad, XOR-swizzled chunks)
    static constexpr int IMGB = R * 64 * 2;
    // attention backward: 6 images, logsumexp + delta [8][R] fp32, 8 per-wave transposition tile pairs, Wq|Wk|Wv|Wp at pitch
    // D + 16, bq|bk|bv + LayerNorm-1 gamma / beta
    static constexpr int BWD_ATTN_LDS = 6 * IMGB + 2 * 8 * R * 4 + 8 * HS_DEC_CORE_NQ * 2 * 16 * 16 * 2 + 4 * D * (D + 16) * 2 + 5 * D * 4;
    // the fp32 staging tile of du goes over Ob | DXb and the first bytes of the logsumexp table (all dead by then)
    static_assert(2 * IMGB + 8 * R * 4 >= R * LX * 4, "fp32 staging tile must fit over Ob|DXb|lse");
    // the unguarded m-tile products (CHK = false) read one m-tile (16 rows) past an image: every image of the
    // backward kernels is followed by at least that much of the same LDS allocation (logsumexp / delta / tiles / weights)
    static_assert(BWD_ATTN_LDS - 6 * IMGB >= 16 * 64 * 2, "image overrun of the unguarded products must stay inside the allocation");
};
constexpr int TTS = 16;                    // transposition tile row stride (elements): 32-B rows, rotation-swizzled 8-B chunks
constexpr int TT_PAIR = 2 * 16 * TTS;      // one tile pair: [P | dS][16 queries][TTS]
constexpr int TT_WAVE = HS_DEC_CORE_NQ * TT_PAIR;      // per wave: one pair per query tile in hand

struct Geo4 { int lane, c16, g, wave, wm, wn; };

// SCALAR_WAVE: the wave index goes through readfirstlane, so everything derived from it is scalar (uniform branches,
// SALU address math).
template <bool SCALAR_WAVE = true>
__device__ __forceinline__ Geo4 geo() {
    Geo4 q;
    q.lane = threadIdx.x & 63; q.c16 = q.lane & 15; q.g = q.lane >> 4;
    q.wave = SCALAR_WAVE ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x >> 6);
    q.wm = q.wave >> 1; q.wn = q.wave & 1;
    return q;
}

// acc[mi][j] += A[rows of m-tile mt0+mi][k] * W[n-tile nt0+j][k]  over KS k-steps starting at A column `kofs`
// and packed k-step ks0.  All B fragments are fetched up front (one exposed L2 latency per call).
typedef __attribute__((address_space(3))) const bf16x8* lds_cb128;
typedef __attribute__((address_space(3))) const float* lds_cf32;

// WLDS: the packed weight image is resident in LDS (explicit address space: a generic pointer would become flat loads)
template <int MH, int KS, bool WLDS = false>
__device__ __forceinline__ void mm(const bf16_t* A, int lda, int kofs, const bf16_t* W, int KS_total, int nt0,
                                   int ks0, int mt0, int MT, const Geo4& q, f32x4 (&acc)[MH][2]) {
    bf16x8 b[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (WLDS) b[ks][j] = *(lds_cb128)(W + (((nt0 + j) * KS_total + ks0 + ks) * 64 + q.lane) * 8);
            else b[ks][j] = *reinterpret_cast<const bf16x8*>(W + (((size_t)(nt0 + j) * KS_total + ks0 + ks) * 64 + q.lane) * 8);
        }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mi = 0; mi < MH; ++mi) {
            const int mt = mt0 + mi;
            if (mt < MT) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + (mt * 16 + q.c16) * lda + kofs + ks * 32 + q.g * 8);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[mi][j] = mfma16(a, b[ks][j], acc[mi][j]);
            }
        }
}

// Weight fragments of one product, fetched ahead of use (global/L2 latency is 1-2 us under load: every
// exposed fetch costs more than the MFMAs it feeds, so callers issue `load` one stage early).
template <int KS>
struct Fr {
    bf16x8 b[KS][2];
    __device__ __forceinline__ void load(const bf16_t* W, int KS_total, int nt0, int ks0, const Geo4& q) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                b[ks][j] = *reinterpret_cast<const bf16x8*>(W + (((size_t)(nt0 + j) * KS_total + ks0 + ks) * 64 + q.lane) * 8);
    }
};

// SW: operands swapped -> the accumulator holds the TRANSPOSED tile: acc[r] = C[row c16][column 4 g + r], so a lane owns
// four consecutive columns of one row and tiles go to LDS as one 8-byte (bf16) / 16-byte (fp32) access instead of four
template <int MH, int KS, bool SW = false>
__device__ __forceinline__ void mm_f(const bf16_t* A, int lda, int kofs, const Fr<KS>& f, int mt0, int MT, const Geo4& q,
                                     f32x4 (&acc)[MH][2]) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mi = 0; mi < MH; ++mi) {
            const int mt = mt0 + mi;
            if (mt < MT) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + (mt * 16 + q.c16) * lda + kofs + ks * 32 + q.g * 8);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[mi][j] = SW ? mfma16(f.b[ks][j], a, acc[mi][j]) : mfma16(a, f.b[ks][j], acc[mi][j]);
            }
        }
}

// Loads rows of `src` (global fp32 [Ts,64]; rows >= Ts read as zero), optionally mirrors them into the fp32
// staging tile, and writes LayerNorm(row) as bf16 into the LDS image `U`.
template <int MT, bool FROM_LDS, int NTHR = NT_>
__device__ __forceinline__ void ln_rows(const float* src, int Ts, const float* gamma, const float* beta, bf16_t* U,
                                        float* XS, float* copy_out) {
    constexpr int R = MT * 16;
    const int c8 = (threadIdx.x & 7) * 8;
    float gm[8], bt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gm[e] = gamma[c8 + e]; bt[e] = beta[c8 + e]; }
    constexpr int NI = (R * 8 + NTHR - 1) / NTHR;
    float4 ga[FROM_LDS ? 1 : NI], gb[FROM_LDS ? 1 : NI];
    if constexpr (!FROM_LDS) {                       // all global loads of the sample first: one exposed round trip
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = threadIdx.x + NTHR * i, row = p >> 3;
            ga[i] = make_float4(0.f, 0.f, 0.f, 0.f); gb[i] = ga[i];
            if (p < R * 8 && row < Ts) {
                ga[i] = *reinterpret_cast<const float4*>(src + (size_t)row * D + c8);
                gb[i] = *reinterpret_cast<const float4*>(src + (size_t)row * D + c8 + 4);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int p = threadIdx.x + NTHR * i;
        if (p < R * 8) {
            const int row = p >> 3;
            float f[8];
            if constexpr (FROM_LDS) {
                const float4 a = *reinterpret_cast<const float4*>(XS + row * LX + c8);
                const float4 b = *reinterpret_cast<const float4*>(XS + row * LX + c8 + 4);
                f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
                if (copy_out && row < Ts) {
                    HS_NT(HS_NT_D, reinterpret_cast<f32x4*>(copy_out + (size_t)row * D + c8), (f32x4{a.x, a.y, a.z, a.w}));   // x1: read again only by the backward
                    HS_NT(HS_NT_D, reinterpret_cast<f32x4*>(copy_out + (size_t)row * D + c8 + 4), (f32x4{b.x, b.y, b.z, b.w}));
                }
            } else {
                const float4 a = ga[i], b = gb[i];
                f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
                if (XS) {
                    *reinterpret_cast<float4*>(XS + row * LX + c8) = a;
                    *reinterpret_cast<float4*>(XS + row * LX + c8 + 4) = b;
                }
            }
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += f[e];
            s = lanes_sum<8>(s);
            const float mean = s * (1.f / D);
            float v = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dl = f[e] - mean; v += dl * dl; }
            v = lanes_sum<8>(v);
            const float rstd = rsqrtf(v * (1.f / D) + 1e-5f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (f[e] - mean) * rstd * gm[e] + bt[e];
            *reinterpret_cast<bf16x8*>(U + row * LU + c8) = cvt8(f);
        }
    }
}

template <int MH, bool SW = false>
__device__ __forceinline__ void acc_from_xs(const float* XS, int mt0, int MT, const Geo4& q, f32x4 (&x)[MH][2]) {
#pragma unroll
    for (int mi = 0; mi < MH; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if constexpr (SW) {
                const int mt = mt0 + mi;
                x[mi][j] = (mt < MT) ? *reinterpret_cast<const f32x4*>(XS + (mt * 16 + q.c16) * LX + (q.wn * 2 + j) * 16 + q.g * 4)
                                     : f32x4{0.f, 0.f, 0.f, 0.f};
            } else
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mt = mt0 + mi;
                x[mi][j][r] = (mt < MT) ? XS[(mt * 16 + q.g * 4 + r) * LX + (q.wn * 2 + j) * 16 + q.c16] : 0.f;
            }
}

template <int MH, bool SW = false>
__device__ __forceinline__ void acc_to_xs(float* XS, int mt0, int MT, const Geo4& q, const f32x4 (&x)[MH][2]) {
#pragma unroll
    for (int mi = 0; mi < MH; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if constexpr (SW) {
                const int mt = mt0 + mi;
                if (mt < MT) *reinterpret_cast<f32x4*>(XS + (mt * 16 + q.c16) * LX + (q.wn * 2 + j) * 16 + q.g * 4) = x[mi][j];
            } else
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mt = mt0 + mi;
                if (mt < MT) XS[(mt * 16 + q.g * 4 + r) * LX + (q.wn * 2 + j) * 16 + q.c16] = x[mi][j][r];
            }
}

template <int NTHR = NT_>
__device__ __forceinline__ void store_rows(const float* XS, int R, int Ts, float* dst) {
    const int c8 = (threadIdx.x & 7) * 8;
    for (int p = threadIdx.x; p < R * 8; p += NTHR) {
        const int row = p >> 3;
        if (row < Ts) {
            *reinterpret_cast<float4*>(dst + (size_t)row * D + c8) = *reinterpret_cast<const float4*>(XS + row * LX + c8);
            *reinterpret_cast<float4*>(dst + (size_t)row * D + c8 + 4) = *reinterpret_cast<const float4*>(XS + row * LX + c8 + 4);
        }
    }
}

__device__ __forceinline__ bf16x4 cvt4(f32x4 v) {
    bf16x4 r;
    r[0] = (bf16_t)v[0]; r[1] = (bf16_t)v[1]; r[2] = (bf16_t)v[2]; r[3] = (bf16_t)v[3];
    return r;
}

__device__ __forceinline__ bf16x8 pack2(f32x4 a, f32x4 b) {
    bf16x8 r;
    r[0] = (bf16_t)a[0]; r[1] = (bf16_t)a[1]; r[2] = (bf16_t)a[2]; r[3] = (bf16_t)a[3];
    r[4] = (bf16_t)b[0]; r[5] = (bf16_t)b[1]; r[6] = (bf16_t)b[2]; r[7] = (bf16_t)b[3];
    return r;
}

__device__ __forceinline__ bf16x8 rowfrag8(const bf16_t* img, int ld, int row, int coloff, int g) {
    // head dim 8: only k-group 0 carries data
    if (g == 0) return *reinterpret_cast<const bf16x8*>(img + row * ld + coloff);
    return zero8();
}

// One head of full (unmasked) attention over the sample's Ts tokens; writes O (bf16) into `O` columns head*8..
template <int MT>
__device__ __forceinline__ void attn_head_fwd(const bf16_t* Qb, const bf16_t* Kb, const bf16_t* Vt, bf16_t* O, int head,
                                              int Ts, const Geo4& q, float* lse_out) {
    using L = DL<MT>;
    const float sc = 0.35355339059327373f * 1.4426950408889634f;     // 8^-0.5 * log2(e)
    const bf16_t* vrow = Vt + ((head * HD + q.c16) & 63) * L::VST;
    const int kcol = (head * HD + 8 * q.g) & 63;
    // Padding keys are masked through the MFMA's C operand (-inf rows; padded K rows are finite: LN of zero rows).
    // The launchers use MT = 7 for 65..112 tokens, so key tiles 0..3 are always full there.
    constexpr int KMIN = (MT > 4) ? 4 : 0;
    f32x4 cinit[MT - KMIN];
#pragma unroll
    for (int kt = KMIN; kt < MT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) cinit[kt - KMIN][r] = (kt * 16 + q.g * 4 + r >= Ts) ? -INFINITY : 0.f;
    for (int qt = 0; qt * 16 < Ts; ++qt) {
        const int query = qt * 16 + q.c16;
        const bf16x8 bq = rowfrag8(Qb, LU, query, head * HD, q.g);
        f32x4 s[MT];
#pragma unroll
        for (int kt = 0; kt < MT; ++kt)       // K side unmasked: lane groups 1-3 read other heads' (finite) columns against bq's zeros
            s[kt] = mfma16(*reinterpret_cast<const bf16x8*>(Kb + (kt * 16 + q.c16) * LU + kcol), bq,
                           kt < KMIN ? f32x4{0.f, 0.f, 0.f, 0.f} : cinit[kt < KMIN ? 0 : kt - KMIN]);
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < MT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, s[kt][r]);
        m = rows_max(m);
        const float nm = -m * sc;
        const f32x2 sc2 = {sc, sc}, nm2 = {nm, nm};
        f32x2 ls2 = {0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < MT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {          // packed fp32: two scores per v_pk_fma / v_pk_add
                const f32x2 t = __builtin_elementwise_fma(f32x2{s[kt][r], s[kt][r + 1]}, sc2, nm2);
                const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                s[kt][r] = e[0]; s[kt][r + 1] = e[1];
                ls2 += e;
            }
        float lsum = ls2[0] + ls2[1];
        lsum = rows_sum(lsum);
        const float inv = 1.f / lsum;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pp = 0; pp < (MT + 1) / 2; ++pp) {
            const int ta = 2 * pp, tb = 2 * pp + 1;
            const bool tb_ok = tb < MT;
            const bf16x8 bp = pack2(s[ta], tb_ok ? s[tb_ok ? tb : 0] : f32x4{0.f, 0.f, 0.f, 0.f});
            // V^T rows past the head's 8 (c16 >= 8) are other heads' rows: they only reach output rows nobody stores
            const u32x2 vlo = *reinterpret_cast<const u32x2*>(vrow + ta * 16 + q.g * 4);
            const u32x2 vhi = *reinterpret_cast<const u32x2*>(vrow + (tb_ok ? tb : ta) * 16 + q.g * 4);
            const u32x4 vv = {vlo[0], vlo[1], vhi[0], vhi[1]};
            o = mfma16(__builtin_bit_cast(bf16x8, vv), bp, o);
        }
        if (q.g < 2) {
            bf16x4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)(o[r] * inv);
            *reinterpret_cast<bf16x4*>(O + query * LU + head * HD + q.g * 4) = ov;
        }
        if (lse_out && q.g == 0 && query < Ts) lse_out[(size_t)query * 8 + head] = m * sc + __builtin_amdgcn_logf(lsum);   // log2-domain lse, [row][head]
    }
}

struct DecW {               // one decoder block's parameters
    const float *n1w, *n1b, *bqkv, *pb, *n2w, *n2b, *w1b, *w3b, *w2b;
    const bf16_t *qkv, *p, *w1, *w3, *w2;
    int h;
};

// The weight images are invariant across the samples a workgroup walks; left alone, the compiler hoists every
// fragment load out of the sample loop and then spills them.  Laundering the pointers once per iteration keeps
// the loads where they are used.
__device__ __forceinline__ int launder_i(int v) {
    asm volatile("" : "+s"(v));
    return v;
}
template <class T>
__device__ __forceinline__ const T* launder(const T* p) {
    // launder it as a global-memory pointer: through a generic one every load becomes a flat_load, which ties the
    // LDS counter (lgkmcnt) to L2 latency
    const __attribute__((address_space(1))) T* g = (const __attribute__((address_space(1))) T*)p;
    asm volatile("" : "+s"(g));
    return (const T*)g;
}
// Global addressing of the persistent backward kernels (round 6): a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-lane BYTE
// offset, which is the `global_load ... v_off, s[base]` form — one VGPR per access instead of a 64-bit VGPR pair per row group —
// and the lane offsets are re-derived from an opaque copy of threadIdx.x where they are used.  Left to itself hipcc computes every
// row group's 64-bit address once, keeps the pairs alive across the whole sample loop and spills them (11 registers in
// dec_bwd_attn_kernel<7>); the reload of one of them sat in the MIDDLE of the next sample's prefetch burst, and a scratch reload
// is a vector-memory load: its s_waitcnt vmcnt(0) waited for the ten HBM loads just issued — the "issue" time of that burst in
// the round-5 phase stamps (33 us per launch) was this wait.
__device__ __forceinline__ int fresh_tid() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
template <class T>
__device__ __forceinline__ const T* at_bytes(const T* base, unsigned byte_off) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ T* at_bytes(T* base, unsigned byte_off) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off);
}
// One sample's rows as a raw buffer: `bytes` = the sample's extent (0 for a sample past the batch), so lanes whose row is past
// the sequence read ZEROS and their stores are dropped by the bounds check — no exec-masked branch around the accesses (with the
// branches hipcc's wait-count pass fell back to s_waitcnt vmcnt(0) where a counted wait would have left a prefetch in flight).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const void* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void bld8(__amdgpu_buffer_rsrc_t r, unsigned bo, float* o) {
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)bo, 0, 0));
    const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)bo + 16, 0, 0));
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
__device__ __forceinline__ void bst8(__amdgpu_buffer_rsrc_t r, unsigned bo, const float* v) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, f32x4{v[0], v[1], v[2], v[3]}), r, (int)bo, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, f32x4{v[4], v[5], v[6], v[7]}), r, (int)bo + 16, 0, 0);
}
__device__ __forceinline__ DecW launder_w(const DecW& a) {
    DecW w = a;
    w.qkv = launder(a.qkv); w.p = launder(a.p); w.w1 = launder(a.w1); w.w3 = launder(a.w3); w.w2 = launder(a.w2);
    return w;
}

struct DecFwdArgs { const float* x; float* x1; float* x2; bf16_t* o; float* lse; int nsamples, Ts; DecW w; };

// q|k|v for the whole sample from the LN image U:  Qb, Kb row-major bf16; V transposed into Vt.
template <int MT, int MHX>
__device__ __forceinline__ void qkv_stage(const bf16_t* U, const DecW& w, const Fr<2> (&fq)[3], bf16_t* Qb, bf16_t* Kb,
                                          bf16_t* Vt, int mt0, const Geo4& q) {
    using L = DL<MT>;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f32x4 acc[MHX][2];
        if (c < 2) {                        // q, k: transposed accumulators -> one 8-byte LDS write per tile
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(w.bqkv + c * D + (q.wn * 2 + j) * 16 + q.g * 4);
#pragma unroll
                for (int mi = 0; mi < MHX; ++mi) acc[mi][j] = b;
            }
            mm_f<MHX, 2, true>(U, LU, 0, fq[c], mt0, MT, q, acc);
            bf16_t* dst = (c == 0 ? Qb : Kb);
#pragma unroll
            for (int mi = 0; mi < MHX; ++mi) {
                const int mt = mt0 + mi;
                if (mt >= MT) continue;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<bf16x4*>(dst + (mt * 16 + q.c16) * LU + (q.wn * 2 + j) * 16 + q.g * 4) = cvt4(acc[mi][j]);
            }
        } else {                            // v: row-major accumulators are what the V^T image wants
#pragma unroll
            for (int mi = 0; mi < MHX; ++mi)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float b = w.bqkv[c * D + (q.wn * 2 + j) * 16 + q.c16];
                    acc[mi][j] = f32x4{b, b, b, b};
                }
            mm_f<MHX, 2>(U, LU, 0, fq[c], mt0, MT, q, acc);
#pragma unroll
            for (int mi = 0; mi < MHX; ++mi) {
                const int mt = mt0 + mi;
                if (mt >= MT) continue;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = (q.wn * 2 + j) * 16 + q.c16;
                    const int row = mt * 16 + q.g * 4;
                    *reinterpret_cast<bf16x4*>(Vt + col * L::VST + row) = cvt4(acc[mi][j]);
                }
            }
        }
    }
}

__device__ __forceinline__ float silu_f(float a) { return silu_nr(a); }   // forward: Newton-refined reciprocal (loss gate 1e-4)

template <int MT>
__global__ __launch_bounds__(256, 2) void dec_block_fwd_kernel(DecFwdArgs p) {
    using L = DL<MT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* U = reinterpret_cast<bf16_t*>(smem);
    char* reg2 = smem + L::U_BYTES;
    bf16_t* Qb = reinterpret_cast<bf16_t*>(reg2);
    bf16_t* Kb = Qb + L::R * LU;
    bf16_t* Vt = Kb + L::R * LU;
    float* XS = reinterpret_cast<float*>(reg2);
    bf16_t* Gb = reinterpret_cast<bf16_t*>(reg2);
    const Geo4 q = geo();
    const int mt0 = q.wm * L::MHF;

    PH_DECL
    for (int sample = blockIdx.x; sample < p.nsamples; sample += gridDim.x) {
        const size_t rb = (size_t)sample * p.Ts;
        const DecW w = launder_w(p.w);
        // LN1 (wide layout) + residual into accumulator layout; q|k|v weight fragments are already in flight
        Fr<2> fq[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) fq[c].load(w.qkv, 2, c * 4 + q.wn * 2, 0, q);
        ln_rows<MT, false, 256>(p.x + rb * D, p.Ts, w.n1w, w.n1b, U, XS, nullptr);
        lds_barrier();
        PH(0)
        f32x4 xr[L::MHF][2];
        acc_from_xs<L::MHF, true>(XS, mt0, MT, q, xr);
        lds_barrier();
        PH(0)
        qkv_stage<MT, L::MHF>(U, w, fq, Qb, Kb, Vt, mt0, q);
        Fr<2> fp;
        fp.load(w.p, 2, q.wn * 2, 0, q);
        lds_barrier();
        PH(1)
        // persistent grid: warm L2 / TLB with the next sample's rows while the attention runs (its only other
        // global traffic is the already-issued proj fragments)
        float touchx = 0.f;
        {
            const int nxt = sample + gridDim.x;
            if (HS_TOUCH_F && nxt < p.nsamples && threadIdx.x * 32 < p.Ts * D) touchx = p.x[(size_t)nxt * p.Ts * D + threadIdx.x * 32];
        }
#pragma unroll 1
        for (int hh = 0; hh < 2; ++hh) attn_head_fwd<MT>(Qb, Kb, Vt, U, q.wave * 2 + hh, p.Ts, q, p.lse + rb * 8);
        asm volatile("" :: "v"(touchx));
        lds_barrier();
        PH(2)
        // attention output kept for the backward (dWp operand; saves it the softmax recompute), 16-B row pieces
        for (int pc = threadIdx.x; pc < L::R * 8; pc += 256) {
            const int row = pc >> 3, k8 = (pc & 7) * 8;
            if (row < p.Ts) HS_NT(HS_NT_D, reinterpret_cast<bf16x8*>(p.o + (rb + row) * D + k8), *reinterpret_cast<const bf16x8*>(U + row * LU + k8));
        }
        // proj accumulates onto the residual (transposed accumulators: see mm_f)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(w.pb + (q.wn * 2 + j) * 16 + q.g * 4);
#pragma unroll
            for (int mi = 0; mi < L::MHF; ++mi) xr[mi][j] += b;
        }
        mm_f<L::MHF, 2, true>(U, LU, 0, fp, mt0, MT, q, xr);
        Fr<2> f1, f3;
        f1.load(w.w1, 2, q.wn * 2, 0, q);
        f3.load(w.w3, 2, q.wn * 2, 0, q);
        acc_to_xs<L::MHF, true>(XS, mt0, MT, q, xr);
        lds_barrier();
        PH(3)
        ln_rows<MT, true, 256>(nullptr, p.Ts, w.n2w, w.n2b, U, XS, p.x1 + rb * D);      // LN2; x1 saved for the backward
        lds_barrier();
        PH(4)
        Fr<6> f2;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f32x4 h1[L::MHF][2], h3[L::MHF][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = c * 64 + (q.wn * 2 + j) * 16 + q.g * 4;
                f32x4 b1, b3;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    b1[r] = col + r < w.h ? w.w1b[col + r] : 0.f;
                    b3[r] = col + r < w.h ? w.w3b[col + r] : 0.f;
                }
#pragma unroll
                for (int mi = 0; mi < L::MHF; ++mi) { h1[mi][j] = b1; h3[mi][j] = b3; }
            }
            mm_f<L::MHF, 2, true>(U, LU, 0, f1, mt0, MT, q, h1);
            mm_f<L::MHF, 2, true>(U, LU, 0, f3, mt0, MT, q, h3);
            if (c < 2) {
                f1.load(w.w1, 2, (c + 1) * 4 + q.wn * 2, 0, q);
                f3.load(w.w3, 2, (c + 1) * 4 + q.wn * 2, 0, q);
            } else {
                f2.load(w.w2, 6, q.wn * 2, 0, q);
            }
#pragma unroll
            for (int mi = 0; mi < L::MHF; ++mi) {
                const int mt = mt0 + mi;
                if (mt >= MT) continue;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = c * 64 + (q.wn * 2 + j) * 16 + q.g * 4;
                    f32x4 gv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) gv[r] = col + r < w.h ? silu_f(h1[mi][j][r]) * h3[mi][j][r] : 0.f;
                    *reinterpret_cast<bf16x4*>(Gb + (mt * 16 + q.c16) * LG + col) = cvt4(gv);
                }
            }
        }
        lds_barrier();
        PH(5)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(w.w2b + (q.wn * 2 + j) * 16 + q.g * 4);
#pragma unroll
            for (int mi = 0; mi < L::MHF; ++mi) xr[mi][j] += b;
        }
        mm_f<L::MHF, 6, true>(Gb, LG, 0, f2, mt0, MT, q, xr);
        lds_barrier();
        PH(6)
        acc_to_xs<L::MHF, true>(XS, mt0, MT, q, xr);
        lds_barrier();
        store_rows<256>(XS, L::R, p.Ts, p.x2 + rb * D);
        lds_barrier();
        PH(7)
    }
    PH_FLUSH(16)
}

typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ f32x4 mfma16k16(bf16x4 a, bf16x4 b, f32x4 c) {
    // D[16x16] += A[16x16] * B[16x16].  lane l: A[row l&15][k 4(l>>4)+j], B[k 4(l>>4)+j][col l&15]; D as mfma16.
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ bf16x4 zero4() {
    u32x2 z = {0u, 0u};
    return __builtin_bit_cast(bf16x4, z);
}

// ====================================================================== forward, second generation: attention half
// x1 = x + proj(attention(LN1 x)) with the sample's q / k / v and the softmax tiles held in REGISTERS (round 3).
// dec_block_fwd_kernel above keeps q | k | v in three LDS images (64 KB per sample with the LayerNorm image => two
// 4-wave workgroups per CU at 256 registers) and reads every score operand back from LDS.  Here wave w owns heads 2w and
// 2w + 1 end to end:
//   * q^T, k^T come out of operand-swapped MFMAs (D[row = dim 4g+r][col = token c16]): a lane holds 4 head dims of one
//     token — exactly the A / B fragment of the K = 16 MFMA whose contraction runs over the head dims, so
//     S^T[key][query] = mfma(k, q) needs no LDS at all; v comes out in the plain orientation (lane = dim, registers =
//     4 tokens), which is the operand of O^T[dim][query] = sum_key v^T[dim][key] P^T[key][query] next to the S^T
//     accumulators themselves (two key tiles packed into one K = 32 MFMA).
//   * LDS holds only the LayerNorm image (A operand of q | k | v) and the attention output image (A operand of the
//     projection): 32 KB per workgroup, <= 128 registers => four workgroups = 16 waves per CU.
//   * the projection accumulates in the swapped orientation too, so a lane owns 4 consecutive output columns of a row:
//     residual load and x1 store are 16-byte accesses straight from / to HBM, no fp32 staging tile.
// The MLP half of the block then runs as the row-panel kernel of fused_enc.hip (enc_mlp_fwd_kernel<64, 192>); x1 is
// written for the backward anyway, so the split costs one L2-hot re-read of x1.
struct DecAttnFwdArgs { const float* x; float* x1; bf16_t* o; float* lse; int nsamples, Ts; DecW w; };

template <int MT>
__global__ __launch_bounds__(256, 4) void dec_attn_fwd_kernel(DecAttnFwdArgs p) {
    constexpr int R = MT * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* U = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Oi = U + R * LU;
    const Geo4 q = geo();
    const float sc = 0.35355339059327373f * 1.4426950408889634f;     // 8^-0.5 * log2(e)
    constexpr int NPAIR = (MT + 1) / 2;

    {   // one sample per workgroup (not persistent: nothing of a sample's working set is worth keeping, and a sample loop
        // makes hipcc hoist ~50 registers of loop-invariant addresses and masks over the whole body)
        const int sample = blockIdx.x;
        const size_t rb = (size_t)sample * p.Ts;
        const DecW& w = p.w;
        // this wave's 16 columns of Wq | Wk | Wv (packed [3 * 4 n-tiles][2 k-steps]); in flight under the LayerNorm
        bf16x8 fw[3][2];
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                fw[c][ks] = *reinterpret_cast<const bf16x8*>(w.qkv + (((size_t)(c * 4 + q.wave) * 2 + ks) * 64 + q.lane) * 8);
        const f32x4 bq4 = *reinterpret_cast<const f32x4*>(w.bqkv + q.wave * 16 + q.g * 4);
        const f32x4 bk4 = *reinterpret_cast<const f32x4*>(w.bqkv + D + q.wave * 16 + q.g * 4);
        const float bv1 = w.bqkv[2 * D + q.wave * 16 + q.c16];
        ln_rows<MT, false, 256>(p.x + rb * D, p.Ts, w.n1w, w.n1b, U, nullptr, nullptr);
        lds_barrier();

        // q^T, k^T (lane = token, registers = 4 dims of this wave's 16) and v (lane = dim, registers = 4 tokens)
        // k and v stay in registers for the whole sample; q^T (needed one tile at a time) is parked in this wave's own 16
        // columns of the O image — the slot that head's output later overwrites (head A's output only touches head A's
        // 8 columns, so head B's q survives it) — no other wave reads those columns before the barrier below
        bf16x4 kT[MT], vN[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(U + (mt * 16 + q.c16) * LU + q.g * 8);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(U + (mt * 16 + q.c16) * LU + 32 + q.g * 8);
            f32x4 aq = bq4, ak = bk4, av = f32x4{bv1, bv1, bv1, bv1};
            aq = mfma16(fw[0][0], a0, aq); aq = mfma16(fw[0][1], a1, aq);
            ak = mfma16(fw[1][0], a0, ak); ak = mfma16(fw[1][1], a1, ak);
            av = mfma16(a0, fw[2][0], av); av = mfma16(a1, fw[2][1], av);
            *reinterpret_cast<bf16x4*>(Oi + (mt * 16 + q.c16) * LU + q.wave * 16 + q.g * 4) = cvt4(aq);
            kT[mt] = cvt4(ak); vN[mt] = cvt4(av);
        }

        // padding keys (rows >= Ts) are masked through the MFMA's C operand; with MT = 7 (65..112 tokens) tiles 0..3 are full
        constexpr int KMIN = (MT > 4) ? 4 : 0;
        f32x4 cinit[MT - KMIN];
#pragma unroll
        for (int kt = KMIN; kt < MT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) cinit[kt - KMIN][r] = (kt * 16 + q.g * 4 + r >= p.Ts) ? -INFINITY : 0.f;

#pragma unroll 1
        for (int hh = 0; hh < 2; ++hh) {
            const bool mine = (q.g >> 1) == hh;            // this lane group carries head hh's dims
            const int head = q.wave * 2 + hh;
#pragma unroll
            for (int qt = 0; qt < MT; ++qt) {
                if (qt * 16 >= p.Ts) continue;             // (uniform)
                const int query = qt * 16 + q.c16;
                bf16x4 bq = *reinterpret_cast<const bf16x4*>(Oi + query * LU + q.wave * 16 + q.g * 4);
                if (!mine) bq = zero4();
                f32x4 s[MT];
#pragma unroll
                for (int kt = 0; kt < MT; ++kt)
                    s[kt] = mfma16k16(kT[kt], bq, kt < KMIN ? f32x4{0.f, 0.f, 0.f, 0.f} : cinit[kt < KMIN ? 0 : kt - KMIN]);
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < MT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) m = fmaxf(m, s[kt][r]);
                m = rows_max(m);
                const float nm = -m * sc;
                const f32x2 sc2 = {sc, sc}, nm2 = {nm, nm};
                f32x2 ls2 = {0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < MT; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {          // packed fp32: two scores per v_pk_fma / v_pk_add (the kernel is VALU-issue bound)
                        const f32x2 t = __builtin_elementwise_fma(f32x2{s[kt][r], s[kt][r + 1]}, sc2, nm2);
                        const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
                        s[kt][r] = e[0]; s[kt][r + 1] = e[1];
                        ls2 += e;
                    }
                float lsum = ls2[0] + ls2[1];
                lsum = rows_sum(lsum);
                const float inv = __builtin_amdgcn_rcpf(lsum);
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pp = 0; pp < NPAIR; ++pp) {
                    const int ta = 2 * pp, tb = (2 * pp + 1 < MT) ? 2 * pp + 1 : 2 * pp;
                    const bf16x8 pc = pack2(s[ta], (2 * pp + 1 < MT) ? s[tb] : f32x4{0.f, 0.f, 0.f, 0.f});
                    const bf16x8 vc = __builtin_shufflevector(vN[ta], vN[tb], 0, 1, 2, 3, 4, 5, 6, 7);
                    o = mfma16(vc, pc, o);                  // O^T[dim 4g+r][query c16]
                }
                if (mine) {
                    bf16x4 ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = (bf16_t)(o[r] * inv);
                    *reinterpret_cast<bf16x4*>(Oi + query * LU + q.wave * 16 + q.g * 4) = ov;
                    if ((q.g & 1) == 0 && query < p.Ts)
                        p.lse[(rb + query) * 8 + head] = m * sc + __builtin_amdgcn_logf(lsum);   // log2-domain, [row][head]
                }
            }
        }
        // projection fragments of this wave's 16 output columns
        bf16x8 fp[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            fp[ks] = *reinterpret_cast<const bf16x8*>(w.p + (((size_t)q.wave * 2 + ks) * 64 + q.lane) * 8);
        const f32x4 pb4 = *reinterpret_cast<const f32x4*>(w.pb + q.wave * 16 + q.g * 4);
        lds_barrier();
        // residual rows of this wave's 16 output columns, all in flight before the projection's MFMAs
        f32x4 xr[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = mt * 16 + q.c16;
            xr[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < p.Ts) xr[mt] = *reinterpret_cast<const f32x4*>(p.x + (rb + row) * D + q.wave * 16 + q.g * 4);
        }
        // attention output kept for the backward (dWp operand), 16-byte row pieces
        for (int pc = threadIdx.x; pc < R * 8; pc += 256) {
            const int row = pc >> 3, k8 = (pc & 7) * 8;
            if (row < p.Ts) HS_NT(HS_NT_D, reinterpret_cast<bf16x8*>(p.o + (rb + row) * D + k8), *reinterpret_cast<const bf16x8*>(Oi + row * LU + k8));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = mt * 16 + q.c16;
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(Oi + row * LU + q.g * 8);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Oi + row * LU + 32 + q.g * 8);
            f32x4 acc = pb4;
            acc = mfma16(fp[0], b0, acc); acc = mfma16(fp[1], b1, acc);     // D[row = out column 4g+r][col = token c16]
            acc += xr[mt];
            if (row < p.Ts) *reinterpret_cast<f32x4*>(p.x1 + (rb + row) * D + q.wave * 16 + q.g * 4) = acc;
        }
    }
}

// ====================================================================== backward
// The decoder block's backward is two persistent kernels per block (weight gradients stay in registers across
// all the samples a workgroup walks and are committed once with atomics — 198 KB of dW per block fit on chip):
//   dec_bwd_mlp_kernel : reads x1, dY          -> dx1 = dY + LN2/SwiGLU branch grads ; dW1 dW3 dW2 db* dLN2
//   dec_bwd_attn_kernel: reads x, dx1          -> dx  = dx1 + LN1/attention branch grads ; dWq dWk dWv dWp db* dLN1
// Everything else (u, q, k, v, P, o, h1, h3, g) is recomputed from x / x1 inside the tile.
typedef __attribute__((address_space(3))) bf16x4* lds_b64;

// ---------------------------------------------------------------------- backward: LDS layouts (round 4)
// Bank model: MI355X_MICROARCH.md "LDS" / scripts/micro/lds_banks.py; the audit of every access pattern of the two backward
// kernels under the round-3 layout (pitch D + 8) and this one is scripts/micro/lds_audit_dec.py.  Round 3 measured 2.5 / 2.2
// bank-conflict cycles per LDS instruction here: with a 144-byte pitch the 16-byte row fragments and the transpose reads are both
// 2-way conflicted, and no single pad fixes both (row fragments want a pitch of 32 mod 64 bytes... 8-byte row reads of the
// attention core one of 16 mod 32).
//   * activation images ([R tokens][64] bf16): UNPADDED 128-byte rows — even rows own banks 0..31, odd rows banks 32..63 — and
//     the 16-byte chunk c of row r lives at chunk c ^ swz(r).  swz is a bijection of the 8 even (odd) rows of a 16-row tile
//     onto 0..7 chosen (brute force over the 8! candidates, lds_audit_dec.py) so that all of these are conflict-free:
//     16-byte row fragments (MFMA A / B operands), the attention core's 8-byte row reads of one head, transpose reads of 8
//     consecutive rows x 32 bytes (ds_read_b64_tr_b16), 16-byte row-contiguous fills.  The 8-byte tile writes of the
//     swapped-operand accumulators stay 2-way (16 rows x 8 bytes meet in 32 banks x 4 bytes: one row of this layout).
//   * transposed fragments of a weight-gradient product contract over the image ROWS; both operands come from the same
//     function, so the row order inside a 32-row k-step is free: lane group g takes rows 4g..4g+3 and 16+4g..16+4g+3 (a
//     32-lane bank group then reads 8 consecutive rows) instead of 8g..8g+7 (rows 0-3 and 8-11: 2-way at every pitch tried).
//     In the tail step (R % 32 == 16) the second half is simply absent.
//   * weight images (row-major bf16 [n][D], read by rows AND by transpose reads): pitch D + 16, and row k of every 32-row
//     group is stored at wrow(k) so that the transpose read above returns k = 8g..8g+7 — the order of the 16-byte row
//     fragment it meets in the MFMA.
//   * the per-wave transposition tiles of the attention core: 32-byte rows, 8-byte chunk c of row r at (c + (r >> 2)) & 3:
//     the 8-byte writes (16 rows, one chunk) and the transpose reads (8 rows x 4 chunks) are both conflict-free
//     (pitch 48 bytes: 2-way both ways).
constexpr int IR = 64;                     // activation image row pitch (elements)
constexpr int LW = D + 16;                 // weight image row pitch (elements)
constexpr int WRB = HPD * LW;              // one [192][LW] weight image (elements)

__host__ __device__ constexpr int swz(int row) { return (((row >> 1) & 3) << 1) ^ (((row >> 3) & 1) * 5); }
__host__ __device__ constexpr int wrow(int n) { return (n & ~31) + 4 * ((n & 31) >> 3) + (n & 3) + 16 * ((n >> 2) & 1); }

struct GeoB : Geo4 {
    int q4, p4;
    int fr;        // swz of the row c16 (+ 16 k): row fragments, tile writes, 8-byte row reads
    int ft;        // swz of the row 4 g + q4 (+ 16 k): transpose reads
    int pc;        // wrow(c16) - the lane's weight row inside a 16-row n-tile (+ 8 for odd n-tiles, + 32 per tile pair)
};
// tid: threadIdx.x, or an opaque copy of it (fresh_tid()) when the caller wants the geometry RE-DERIVED at this point instead of
// kept alive (and spilled) across the phases of a persistent loop
__device__ __forceinline__ GeoB geob(int tid = -1) {
    GeoB q;
    if (tid < 0) static_cast<Geo4&>(q) = geo();
    else {
        q.lane = tid & 63; q.c16 = q.lane & 15; q.g = q.lane >> 4;
        q.wave = __builtin_amdgcn_readfirstlane(tid >> 6); q.wm = q.wave >> 1; q.wn = q.wave & 1;
    }
    q.q4 = q.c16 >> 2; q.p4 = q.c16 & 3;
    q.fr = swz(q.c16); q.ft = swz(4 * q.g + q.q4);
    q.pc = 4 * (q.c16 >> 3) + (q.c16 & 3) + 16 * ((q.c16 >> 2) & 1);
    return q;
}

// 16-byte row fragment of an activation image: row mt*16 + c16, columns ks*32 + 8g .. + 7
__device__ __forceinline__ bf16x8 rowfrag(const bf16_t* img, int mt, int ks, const GeoB& q) {
    return *reinterpret_cast<const bf16x8*>(img + (mt * 16 + q.c16) * IR + (((ks * 4 + q.g) ^ q.fr) << 3));
}
// element offset of the 4 columns nt*16 + 4g .. + 3 of row mt*16 + c16: where a swapped-operand accumulator tile lives
__device__ __forceinline__ int tile_off(int mt, int nt, const GeoB& q) {
    return (mt * 16 + q.c16) * IR + (((2 * nt + (q.g >> 1)) ^ q.fr) << 3) + (q.g & 1) * 4;
}
typedef __attribute__((address_space(3))) bf16x4* lds_w64;
__device__ __forceinline__ void st4(bf16_t* p, bf16x4 v) { *(lds_w64)(p) = v; }      // LDS address space + 8-byte type: ds_write_b64

// 4 consecutive image rows of one column per lane: element j of lane (c16, g) = img[row0(g) + j][col0 + c16];
// `a` is the lane's own 8-byte piece img[row0(g) + (c16 >> 2)][col0 + 4 (c16 & 3) ..].
__device__ __forceinline__ bf16x4 tr4(const bf16_t* a) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64)(a));
}

// Row (token) fragment pair for one k-step of a weight-gradient product  dO[:, n0..]^T * A[:, k0..]  (contraction over
// the image rows, 32 per MFMA) from an activation image: element j of lane group g = img[kk*32 + 4g + (j & 3) + 16 (j >> 2)][nt*16 + c16].
// TAIL = the k-step that runs past row R (R % 32 == 16): rows kk*32 + 16.. do not exist, the second half is zero.
template <bool TAIL>
__device__ __forceinline__ bf16x8 wg_frag(const bf16_t* img, int nt, int kk, const GeoB& q) {
    const bf16_t* a = img + (kk * 32 + 4 * q.g + q.q4) * IR + (((2 * nt + (q.p4 >> 1)) ^ q.ft) << 3) + (q.p4 & 1) * 4;
    const bf16x4 a0 = tr4(a);
    if constexpr (TAIL) return __builtin_shufflevector(a0, zero4(), 0, 1, 2, 3, 4, 5, 6, 7);
    else return __builtin_shufflevector(a0, tr4(a + 16 * IR), 0, 1, 2, 3, 4, 5, 6, 7);
}
// The same fragment of a weight image (rows = the contraction index, stored at wrow()): element j of lane group g =
// W[k0 + 8g + j][nt*16 + c16] — the k order of a 16-byte row fragment.  `Wk` = the image at (physical = logical) row k0, a multiple of 32.
__device__ __forceinline__ bf16x8 wt_frag(const bf16_t* Wk, int nt, const GeoB& q) {
    const bf16_t* a = Wk + (4 * q.g + q.q4) * LW + nt * 16 + 4 * q.p4;
    return __builtin_shufflevector(tr4(a), tr4(a + 16 * LW), 0, 1, 2, 3, 4, 5, 6, 7);
}

// acc += A * W^T with W row-major bf16 [n][LW] resident in LDS (k contiguous, rows at wrow()): B fragment = 16-byte row pieces.
// Operands swapped (weights as A): acc[r] = C[row c16][column 4 g + r].  m-tiles past MT are multiplied too (image rows past R:
// whatever follows the image in LDS — finite or not, the caller never uses those accumulators).
template <int MH, int KS>
__device__ __forceinline__ void mm_rows(const bf16_t* A, const bf16_t* Wr, int nt0, int mt0, const GeoB& q, f32x4 (&acc)[MH][2]) {
    bf16x8 b[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            b[ks][j] = *(lds_cb128)(Wr + (32 * ((nt0 + j) >> 1) + 8 * ((nt0 + j) & 1) + q.pc) * LW + ks * 32 + q.g * 8);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int mi = 0; mi < MH; ++mi) {
            const bf16x8 a = rowfrag(A, mt0 + mi, ks, q);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[mi][j] = mfma16(b[ks][j], a, acc[mi][j]);
        }
}

// acc += A[:, 0:64] * W with W row-major bf16 [64 k-rows][LW] resident in LDS (n contiguous): the B fragment needs 8 consecutive k
// of one column, i.e. a transpose read of the same image the forward-orientation product reads by rows.  Swapped, unguarded (see mm_rows).
template <int MH>
__device__ __forceinline__ void mm_cols(const bf16_t* A, const bf16_t* Wr, int mt0, const GeoB& q, f32x4 (&acc)[MH][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = wt_frag(Wr + ks * 32 * LW, q.wn * 2 + j, q);
#pragma unroll
        for (int mi = 0; mi < MH; ++mi) {
            const bf16x8 a = rowfrag(A, mt0 + mi, ks, q);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[mi][j] = mfma16(b[j], a, acc[mi][j]);
        }
    }
}

__device__ __forceinline__ void ld8(const float* p, float* o) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void st8(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ float red8(float v) {
    v = lanes_sum<8>(v);
    return v;
}

// Commit per-thread column partials (wide layout: thread owns columns 8*(tid&7)..+7) with one atomic per column.
__device__ __forceinline__ void flush_wide(float* red /* [NT_][8] LDS */, const float* part, float* dst, const HsDet& det,
                                           float* vec = nullptr /* this workgroup's slab vector: plain store instead of an atomic */) {
    lds_barrier();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = part[e];
    lds_barrier();
    if (threadIdx.x < D) {
        const int c = threadIdx.x, c8 = c >> 3, e = c & 7;
        float s = 0.f;
        for (int t = c8; t < NT_; t += 8) s += red[t * 8 + e];
        if (vec) vec[c] = s; else hs_gadd(det, dst + c, s);
    }
}

struct DecBwdMlpArgs {
    const float* x1; const float* dy; float* dx1; int nsamples, Ts; DecW w; const bf16_t *w2T, *w13T; const float *w1f, *w3f;
    float *g_n2w, *g_n2b, *g_w1w, *g_w1b, *g_w3w, *g_w3b, *g_w2w, *g_w2b;
    HsDet det;
    float* slab;                 // NULL: weight-gradient tiles committed with atomics; else [workgroup][kSlots][512] partials (dec_dw_reduce_kernel)
};

template <int MT>
__global__ __launch_bounds__(512, 2) void dec_bwd_mlp_kernel(DecBwdMlpArgs p) {
    using L = DL<MT>;
    constexpr int R = L::R, IMG = R * IR, NPW = (R * 8 + NT_ - 1) / NT_;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* U2 = reinterpret_cast<bf16_t*>(smem);
    bf16_t* DYb = U2 + IMG;
    bf16_t* Gc = DYb + IMG;
    bf16_t* DH1 = Gc + IMG;
    bf16_t* DH3 = DH1 + IMG;
    float* XS = reinterpret_cast<float*>(Gc);            // fp32 staging aliases Gc|DH1|DH3 (dead once the chunk loop is done)
    bf16_t* WL = DH3 + IMG;                              // W1 | W3 as row-major bf16 [192][LW] (rows past h zero, row k at wrow(k)), resident
    float* BL = reinterpret_cast<float*>(WL + 2 * WRB);  // b1 | b3, zero-padded to 192
    GeoB q = geob();                  // re-derived at every phase of the sample loop (HS_DEC_REGEO, see at_bytes)
    const int mt0 = q.wm * L::MH;
    int c8 = (threadIdx.x & 7) * 8;
    int wide = ((threadIdx.x & 7) ^ swz(threadIdx.x >> 3)) << 3;   // this thread's 16-byte chunk in the wide layout (row = tid >> 3 + 64 i)
    auto regeo = [&]() { if (HS_DEC_REGEO_MLP) q = geob(fresh_tid()); };
    auto rewide = [&]() { if (HS_DEC_REGEO_MLP) { const int t = fresh_tid(); wide = ((t & 7) ^ swz(t >> 3)) << 3; c8 = (t & 7) * 8; } };
    // The weights are the same for every sample this workgroup walks: stage W1 and W3 once, row-major.  The gate
    // products read them as 16-byte row pieces and the data gradient (which needs the transposed operand) reads the
    // same image with transpose reads, so the per-sample body fetches only the W2^T fragments from L2.
    for (int i = threadIdx.x; i < 2 * HPD * 8; i += NT_) {
        const int m = i / (HPD * 8), row = (i % (HPD * 8)) >> 3, k8 = (i & 7) * 8;
        bf16x8 v = zero8();
        if (row < p.w.h) {
            float f[8];
            ld8((m == 0 ? p.w1f : p.w3f) + (size_t)row * D + k8, f);
            v = cvt8(f);
        }
        *reinterpret_cast<bf16x8*>(WL + m * WRB + wrow(row) * LW + k8) = v;
    }
    for (int i = threadIdx.x; i < 2 * HPD; i += NT_) {
        const int m = i / HPD, o = i % HPD;
        BL[i] = o < p.w.h ? (m == 0 ? p.w.w1b[o] : p.w.w3b[o]) : 0.f;
    }
    // LayerNorm-2 gamma | beta in LDS too (round 6): read from global per sample, the epilogue's gamma load was a vector-memory load
    // YOUNGER than whatever the iteration had prefetched, and its wait drained the prefetch (one in-order counter)
    float* NL = BL + 2 * HPD;
    if (threadIdx.x < 2 * D) NL[threadIdx.x] = threadIdx.x < D ? p.w.n2w[threadIdx.x] : p.w.n2b[threadIdx.x - D];
    lds_barrier();                                       // the first sample's LayerNorm reads NL before the loop's first barrier

    if (HS_DEC_STG_N > 1) {
        const int n = (int)((blockIdx.x >> 3) % (HS_DEC_STG_N > 1 ? HS_DEC_STG_N : 1)) * HS_DEC_STG_MLP;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(32);
    }
    if (HS_DEC_PRIO > 0 && q.wave >= 4) __builtin_amdgcn_s_setprio(HS_DEC_PRIO);      // static priority for the younger half (round 6)
    f32x4 accW[3][3][2];                 // [hidden chunk][this wave's n-tile][this wave's k-tile]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c) accW[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Bias gradients (b2 = column sums of dY, b1 / b3 = column sums of dh1 / dh3) come out of the weight-gradient phase as one more
    // MFMA of the same dO^T fragment against ones: dbw[c][t] belongs to that phase's n-tile t of chunk c (even waves only; the
    // lane keeps column 4 g + (c16 & 3) of the tile).  Nothing in the gate or the prologue accumulates them any more.
    float dgam[8], dbet[8], dbw[3][3];
#pragma unroll
    for (int e = 0; e < 8; ++e) { dgam[e] = 0.f; dbet[e] = 0.f; }
#pragma unroll
    for (int c = 0; c < 3; ++c) { dbw[c][0] = dbw[c][1] = dbw[c][2] = 0.f; }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
    const bool bias_wave = (q.wave & 1) == 0;

    PH_DECL
    // Round-5 experiment (HS_DEC_MLP_PREFETCH=1, not the default): the sample's x1 / dY rows fetched one sample AHEAD, behind the
    // epilogue's own re-reads in the in-order memory counter, as dec_bwd_attn_kernel has done since round 4.  The phase stamps
    // (profiles/r05_c_phase_timing.txt) have this kernel's prologue — 8 loads per thread at the top of the iteration, then
    // everything waits — at 20.7 % of its time (46 us per launch) against 5.7 % in dec_bwd_attn.  Measured on one box, same
    // spill count (8 B/lane): 214.2 us without, 220.3 us WITH (step 15.51 / 15.61 vs 15.55 / 15.64 ms,
    // profiles/r05_d_prefetch_stagger_ab.txt): 32 more live registers through the epilogue cost more than the round trip
    // they hide — the hardware's own answer (the other wave of the SIMD runs while this one waits) is not worse.
    // Second attempt, same round: the next sample's rows by LDS-DMA (`buffer_load ... lds`: no registers at all) — x1 a whole sample
    // ahead into the 28 KB the arena has left, dY into U2 | DYb once the chunk loop is done — 214.6 -> 251.0 us
    // (profiles/r05_r_dec_mlp_dma_ab.txt).  gfx950 counts every load in ONE in-order counter: a DMA in flight is caught by the
    // next wait for ANY younger load (the W2^T fragments of the chunk loop, the epilogue's L2-hot re-reads), and hipcc puts a
    // full vmcnt(0) in front of the first LDS read that follows an LDS-DMA (it may alias).  Not kept.
    float fa[NPW][8], dya[NPW][8];                    // the sample's loads all in flight at once
    auto fetch_sample = [&](int smp) {
        const bool valid = smp < p.nsamples;
        const size_t nb = (size_t)smp * p.Ts;
        const int rbytes = valid ? p.Ts * D * 4 : 0;     // (rows_rsrc: lanes past the sequence / the batch read zeros)
        const __amdgpu_buffer_rsrc_t xr = rows_rsrc(p.x1 + nb * D, rbytes), yr = rows_rsrc(p.dy + nb * D, rbytes);
        const int tid = fresh_tid();
        const unsigned lo = (unsigned)((tid >> 3) * D + (tid & 7) * 8) * 4u;
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const unsigned bo = lo + (unsigned)(i * (NT_ / 8) * D * 4);
            bld8(xr, bo, fa[i]); bld8(yr, bo, dya[i]);
        }
    };
    if (HS_DEC_MLP_PREFETCH) { fetch_sample(blockIdx.x); if (HS_DEC_PRELOOP_WAIT) __builtin_amdgcn_s_waitcnt(0x0F70); }
    for (int sample = blockIdx.x; sample < p.nsamples; sample += gridDim.x) {
        const size_t rb = (size_t)sample * p.Ts;
        const int wl = launder_i(0);                   // keeps the LDS weight reads inside the sample loop (no LICM + spill)
        const bf16_t* w1L = WL + wl;
        const bf16_t* w3L = WL + WRB + wl;
        const bf16_t* w2T = launder(p.w2T);
        Fr<2> f2;                                      // W2^T fragments of the next hidden chunk (the only global weights)
        f2.load(w2T, 2, q.wn * 2, 0, q);
        const float* n2w = NL + wl;
        const float* n2b = NL + D + wl;
        if (!HS_DEC_MLP_PREFETCH) fetch_sample(sample);
        rewide();
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int pc = threadIdx.x + NT_ * i;
            if (pc < R * 8) {
                const int row = pc >> 3;
                float gm[8], bt[8];
                float (&f)[8] = fa[i];
                float (&dyv)[8] = dya[i];
                ld8(n2w + c8, gm); ld8(n2b + c8, bt);
                const float mean = red8(f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7]) * (1.f / D);
                float v = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { f[e] -= mean; v += f[e] * f[e]; }
                const float rstd = rsqrtf(red8(v) * (1.f / D) + 1e-5f);
                float u[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) u[e] = f[e] * rstd * gm[e] + bt[e];
                *reinterpret_cast<bf16x8*>(U2 + row * IR + wide) = cvt8(u);
                *reinterpret_cast<bf16x8*>(DYb + row * IR + wide) = cvt8(dyv);
            }
        }
        lds_barrier();
        PH(0)
        regeo();
        f32x4 du2[L::MH][2];
#pragma unroll
        for (int mi = 0; mi < L::MH; ++mi) { du2[mi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; du2[mi][1] = du2[mi][0]; }
        // warm L2 / TLB with the next sample's rows (one dword per 64 B); nothing else is fetched until the epilogue
        float touch0 = 0.f, touch1 = 0.f;
        {
            const int nxt = sample + gridDim.x;
            const size_t o = (size_t)nxt * p.Ts * D + threadIdx.x * 16;
            if (HS_TOUCH_M && nxt < p.nsamples && threadIdx.x * 16 < p.Ts * D) { touch0 = p.x1[o]; touch1 = p.dy[o]; }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // one m-tile at a time (h1, h3, dg of two tiles at once pushed the kernel into scratch)
#pragma unroll
            for (int mi = 0; mi < L::MH; ++mi) {
                const int mt = mt0 + mi;
                if (mt >= MT) continue;
                // operands swapped: a lane owns 4 consecutive hidden columns of one token (8-byte image writes, 16-byte bias reads)
                f32x4 h1[1][2], h3[1][2], dg[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = c * 64 + (q.wn * 2 + j) * 16 + q.g * 4;
                    h1[0][j] = *reinterpret_cast<const f32x4*>(BL + wl + col);
                    h3[0][j] = *reinterpret_cast<const f32x4*>(BL + HPD + wl + col);
                    dg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                mm_rows<1, 2>(U2, w1L, c * 4 + q.wn * 2, mt, q, h1);
                mm_rows<1, 2>(U2, w3L, c * 4 + q.wn * 2, mt, q, h3);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 a = rowfrag(DYb, mt, ks, q);
#pragma unroll
                    for (int j = 0; j < 2; ++j) dg[j] = mfma16(f2.b[ks][j], a, dg[j]);
                }
                // columns past the hidden width: W1/W3 rows and biases are zero-padded, so g = d1 = d3 = 0 there
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 gv, d1, d3;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a1 = h1[0][j][r], a3 = h3[0][j][r], dv = dg[j][r];
                        const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-a1));
                        const float sl = a1 * sg;
                        gv[r] = sl * a3;
                        d1[r] = dv * a3 * sg * (1.f + a1 * (1.f - sg));
                        d3[r] = dv * sl;
                    }
                    const int o = tile_off(mt, q.wn * 2 + j, q);
                    st4(Gc + o, cvt4(gv));
                    st4(DH1 + o, cvt4(d1));
                    st4(DH3 + o, cvt4(d3));
                }
            }
            if (c < 2) f2.load(w2T, 2, (c + 1) * 4 + q.wn * 2, 0, q);
            lds_barrier();
        PH(1)
            regeo();
            if ((HS_DEC_MLP_PREFETCH == 2 && c == 2) || (HS_DEC_MLP_PREFETCH == 3 && c == 1)) {     // next sample's rows a whole weight-gradient + du2 phase + epilogue ahead of their use (3: a chunk more)
                __builtin_amdgcn_sched_barrier(0);
                fetch_sample(sample + (int)gridDim.x);
                __builtin_amdgcn_sched_barrier(0);
            }
            // weight gradients of this hidden chunk: 12 (n-tile) x 4 (k-tile) output tiles, 3 x 2 per wave; on the even waves the
            // dO^T fragment also meets a tile of ones: the column sums of dY / dh1 / dh3 = this sample's bias-gradient addends
            f32x4 accb[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) accb[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < R / 32; ++kk) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int nt12 = (q.wave >> 1) * 3 + t, mat = nt12 >> 2, nt = nt12 & 3;   // mat 0: dW2, 1: dW1, 2: dW3
                    const bf16_t* dOi = mat == 0 ? DYb : (mat == 1 ? DH1 : DH3);
                    const bf16_t* Ai = mat == 0 ? Gc : U2;
                    const bf16x8 a = wg_frag<false>(dOi, nt, kk, q);
                    accb[t] = mfma16(a, ones, accb[t]);                 // (every wave: a branch here costs more than the odd waves' unused MFMAs)
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2)
                        accW[c][t][k2] = mfma16(a, wg_frag<false>(Ai, (q.wave & 1) * 2 + k2, kk, q), accW[c][t][k2]);
                }
            }
            if constexpr (R % 32 != 0) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int nt12 = (q.wave >> 1) * 3 + t, mat = nt12 >> 2, nt = nt12 & 3;   // mat 0: dW2, 1: dW1, 2: dW3
                    const bf16_t* dOi = mat == 0 ? DYb : (mat == 1 ? DH1 : DH3);
                    const bf16_t* Ai = mat == 0 ? Gc : U2;
                    const bf16x8 a = wg_frag<true>(dOi, nt, R / 32, q);
                    accb[t] = mfma16(a, ones, accb[t]);
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2)
                        accW[c][t][k2] = mfma16(a, wg_frag<true>(Ai, (q.wave & 1) * 2 + k2, R / 32, q), accW[c][t][k2]);
                }
            }
            if (bias_wave) {
                const int r4 = q.c16 & 3;                  // every lane of a row holds the same 4 sums: keep column 4 g + (c16 & 3)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int mat = ((q.wave >> 1) * 3 + t) >> 2;
                    const float v = r4 == 0 ? accb[t][0] : (r4 == 1 ? accb[t][1] : (r4 == 2 ? accb[t][2] : accb[t][3]));
                    if (mat != 0 || c == 0) dbw[c][t] += v;              // dY is the same image in all three chunks: count it once
                }
            }
            PH(2)
            regeo();
            // data gradient through W1 / W3 (this chunk's 64 hidden rows are the contraction index: transpose reads of the images)
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {          // (spelled out: two helper calls here cost 170 B/lane of scratch)
                const bf16_t* Ai = m2 == 0 ? DH1 : DH3;
                const bf16_t* Wc = (m2 == 0 ? w1L : w3L) + c * 64 * LW;      // this chunk's 64 hidden rows
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 b[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[j] = wt_frag(Wc + ks * 32 * LW, q.wn * 2 + j, q);
#pragma unroll
                    for (int mi = 0; mi < L::MH; ++mi) {
                        const bf16x8 a = rowfrag(Ai, mt0 + mi, ks, q);     // (a tile past MT multiplies image overrun: never stored)
#pragma unroll
                        for (int j = 0; j < 2; ++j) du2[mi][j] = mfma16(b[j], a, du2[mi][j]);     // swapped: see the gate products
                    }
                }
            }
            lds_barrier();
        PH(5)
            regeo();
        }
        rewide();
        acc_to_xs<L::MH, true>(XS, mt0, MT, q, du2);
        asm volatile("" :: "v"(touch0), "v"(touch1));
        float xe[NPW][8], dye[NPW][8];                // L2-hot re-reads for the LayerNorm backward, in flight over the barrier
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int row = (threadIdx.x + NT_ * i) >> 3;
#pragma unroll
            for (int e = 0; e < 8; ++e) { xe[i][e] = 0.f; dye[i][e] = 0.f; }
            {
                const unsigned bo = (unsigned)(row * D + c8) * 4u;
                bld8(rows_rsrc(p.x1 + rb * D, p.Ts * D * 4), bo, xe[i]); bld8(rows_rsrc(p.dy + rb * D, p.Ts * D * 4), bo, dye[i]);
            }
        }
        if (HS_DEC_MLP_PREFETCH == 1) {               // next sample's rows: younger than the re-reads above, so the epilogue does not wait for them
            __builtin_amdgcn_sched_barrier(0);
            fetch_sample(sample + (int)gridDim.x);
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
        PH(3)
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int pc = threadIdx.x + NT_ * i;
            if (pc < R * 8) {
                const int row = pc >> 3;
                float du[8], t[8], gm[8];
                float (&xh)[8] = xe[i];
                float (&dyv)[8] = dye[i];
                ld8(XS + row * LX + c8, du);
                ld8(n2w + c8, gm);
                const float mean = red8(xh[0] + xh[1] + xh[2] + xh[3] + xh[4] + xh[5] + xh[6] + xh[7]) * (1.f / D);
                float v = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { xh[e] -= mean; v += xh[e] * xh[e]; }
                const float rstd = rsqrtf(red8(v) * (1.f / D) + 1e-5f);
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { xh[e] *= rstd; t[e] = du[e] * gm[e]; a += t[e]; b += t[e] * xh[e]; }
                a = red8(a) * (1.f / D); b = red8(b) * (1.f / D);
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o[e] = dyv[e] + rstd * (t[e] - a - xh[e] * b);
                    dgam[e] += du[e] * xh[e];
                    dbet[e] += du[e];
                }
                bst8(rows_rsrc(p.dx1 + rb * D, p.Ts * D * 4), (unsigned)(row * D + c8) * 4u, o);     // (rows past Ts: dropped by the bounds check)
            }
        }
        lds_barrier();
        PH(4)
    }

    PH_FLUSH(8)
    // ---- commit
    float* red = XS;
    float* vec = p.slab ? p.slab + kSlabTileFloats + (size_t)blockIdx.x * kVec : nullptr;
    flush_wide(red, dgam, p.g_n2w, p.det, vec ? vec + kVN2W : nullptr);
    flush_wide(red, dbet, p.g_n2b, p.det, vec ? vec + kVN2B : nullptr);
    if (bias_wave && (q.c16 >> 2) == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int nt12 = (q.wave >> 1) * 3 + t, mat = nt12 >> 2, nt = nt12 & 3;
                const int ci = nt * 16 + q.g * 4 + (q.c16 & 3), col = c * 64 + ci;      // column inside the chunk / of the hidden width
                if (mat == 0) {
                    if (c == 0) { if (vec) vec[kVW2B + ci] = dbw[c][t]; else hs_gadd(p.det, p.g_w2b + ci, dbw[c][t]); }
                } else if (col < p.w.h) {
                    if (vec) vec[(mat == 1 ? kVW1B : kVW3B) + col] = dbw[c][t];
                    else hs_gadd(p.det, (mat == 1 ? p.g_w1b : p.g_w3b) + col, dbw[c][t]);
                }
            }
    }
    if (p.slab) {
        // The 72 accumulator registers of every thread leave as 72 fully coalesced 2-KB rows of this workgroup's slab
        // ([slot][thread]); dec_dw_reduce_kernel sums the workgroups and scatters into dW1 / dW3 / dW2 with the index map below.
        // (Committed with float atomics — 16 x 16 fragments = four 64-byte segments per instruction, 36,864 per workgroup — the
        // commit was 86 us of this kernel's 377: scripts/gpu_nocommit.sh.)
        float* sl = p.slab + (size_t)blockIdx.x * kDwSlots * NT_ + threadIdx.x;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sl[(size_t)(((c * 3 + t) * 2 + k2) * 4 + r) * NT_] = accW[c][t][k2][r];
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int nt12 = (q.wave >> 1) * 3 + t, mat = nt12 >> 2, nt = nt12 & 3;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = nt * 16 + q.g * 4 + r, k = ((q.wave & 1) * 2 + k2) * 16 + q.c16;
                    const float v = accW[c][t][k2][r];
                    if (mat == 0) {                       // dW2[d][h]: row n (model dim), column c*64 + k (hidden)
                        if (c * 64 + k < p.w.h) hs_gadd(p.det, p.g_w2w + (size_t)n * p.w.h + c * 64 + k, v);
                    } else {                              // dW1 / dW3 [h][d]: row c*64 + n (hidden), column k
                        float* dst = mat == 1 ? p.g_w1w : p.g_w3w;
                        if (c * 64 + n < p.w.h) hs_gadd(p.det, dst + (size_t)(c * 64 + n) * D + k, v);
                    }
                }
        }
}


// One head of the decoder attention backward, single pass over the (query tile, key tile) grid.
// Scores are formed key-major (S^T[key 4g+r][query c16], K = 16 MFMAs over the 8 head dims): that accumulator layout
// is directly the B operand of dq^T += K^T dS^T.  dk^T / dv^T contract over queries and need the tile transposed:
// P and dS go through a per-wave LDS tile (8-byte writes, one transpose read each) instead of a second
// score/exp pass.  delta and logsumexp come precomputed, so nothing waits on a full row.
// Masking: lse = 1e30 for rows past Ts (P = 0 for dead queries); dead KEYS are handled by data, not by selects: the K / V image
// rows past Ts are zero on entry and the dk / dv rows past Ts are written as zero (see the loop body).
// LDS addressing: "backward: LDS layouts" above (swizzled 128-byte rows; this head's 8 columns are chunk `head` of a row).
template <int MT>
__device__ __forceinline__ void attn_head_bwd(bf16_t* Qb, bf16_t* Kb, bf16_t* Vb, const bf16_t* dOb, const float* lse_h,
                                              const float* dlt_h, bf16_t* T, int head, int Ts, const GeoB& q) {
    const float sc = 0.35355339059327373f * 1.4426950408889634f, scale = 0.35355339059327373f;
    // 8-byte row reads: lane group g takes head dims 4g .. 4g+3 of row 16 t + c16 (groups 2, 3: the next head's columns — finite filler, x 0)
    const int rcol = q.c16 * IR + ((((head + (q.g >> 1)) & 7) ^ q.fr) << 3) + (q.g & 1) * 4;
    // the V / dO reads get their own copy of the offset: from one base register hipcc merges each K | V (Q | dO) pair into a
    // ds_read2_b64, which is banked like an 8-byte WRITE (16 lanes = 16 rows against 32 banks = one row of this layout): 16 LDS
    // cycles instead of 2 x 2 (round-4 counters: 1.31 conflict cycles per LDS instruction with the merge, the model's 0.29 without)
    int rcol2 = rcol;
    asm volatile("" : "+v"(rcol2));
    // transpose reads: rows 4g + q4 of a 16-row tile, columns head*8 + 4 p4 .. (p4 >= 2: the next head's — output rows nobody stores)
    const int troff = (4 * q.g + q.q4) * IR + ((((head + (q.p4 >> 1)) & 7) ^ q.ft) << 3) + (q.p4 & 1) * 4;
    const int wcol = q.c16 * IR + ((head ^ q.fr) << 3) + (q.g & 1) * 4;       // g < 2: this lane's 4 dims of a dq / dk / dv row
    const int tw = q.c16 * TTS + (((q.g + q.q4) & 3) << 2);                   // tile write: row c16, keys 4g.. at chunk (g + (c16 >> 2)) & 3
    const int trd = (4 * q.g + q.q4) * TTS + (((q.p4 + q.g) & 3) << 2);       // tile transpose read: row 4g + q4, chunk p4
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // key tiles that can reach past the sequence (the launchers use MT = 7 for 65..112 tokens, MT = 4 for 16..64): their
    // exponent is clamped at 0 — P <= 1 holds for every real key anyway, and a padded key (K row = 0 => s = 0) of a query whose
    // logsumexp is very negative (diverging run) would otherwise produce P = inf and poison dq through inf * 0
    constexpr int KCL = MT > 4 ? 4 : 1;
    // (round 5: of those tiles only the ones that really hold a padded key are clamped — a wave-uniform test per tile instead of
    //  4 v_min on the loop's longest unit; at 108 tokens that is the last tile of 7 instead of the last three)
    const int clamp_from = Ts >> 4;                        // first key tile with a row >= Ts
    f32x4 dkT[MT], dvT[MT];
    bf16x4 KT[MT];
#pragma unroll
    for (int kt = 0; kt < MT; ++kt) {
        dkT[kt] = z4; dvT[kt] = z4;
        KT[kt] = tr4(Kb + kt * 16 * IR + troff);
    }
    // NQ query tiles in hand (round 6): with one, a wave's (score MFMA -> exp -> dS -> convert -> LDS transposition -> dk / dv MFMA)
    // chain is serial per tile — 539 cycles per tile against ~180 of issue with two waves per SIMD —; with two, the chains of the two
    // tiles are independent, the K / V row fragments of a key tile are read once for both, and each has its own transposition tile pair.
    auto qtiles = [&](auto nq_tag, int qt0) {
        constexpr int NQ = decltype(nq_tag)::value;
        bf16x4 bq[NQ], bdo[NQ], QT[NQ], dOT[NQ], Bp[NQ], Bds[NQ];
        float lqn[NQ], dl[NQ];
        f32x4 dqT[NQ];
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            const int qt = qt0 + n, query = qt * 16 + q.c16;
            bq[n] = *reinterpret_cast<const bf16x4*>(Qb + qt * 16 * IR + rcol);
            bdo[n] = *reinterpret_cast<const bf16x4*>(dOb + qt * 16 * IR + rcol2);
            if (q.g >= 2) { bq[n] = zero4(); bdo[n] = zero4(); }
            lqn[n] = -lse_h[query]; dl[n] = dlt_h[query];
            // (Round 5 tried to take dP - delta out of the VALU — this loop's longest unit — by starting the dP product's accumulator at
            //  -delta, a per-lane splat: 320.2 us against 315.5 with the 4 v_sub per tile, profiles/r05_e_planar_delta_ab.txt.  A zero
            //  accumulator is an inline constant of the MFMA; a splat has to be copied into four accumulator registers per tile.)
            QT[n] = tr4(Qb + qt * 16 * IR + troff);
            dOT[n] = tr4(dOb + qt * 16 * IR + troff);
            dqT[n] = z4; Bp[n] = zero4(); Bds[n] = zero4();
        }
        // Software pipeline over the key tiles: the transposed operands of tile kt-1 (LDS write -> transpose read, ~200
        // cycles of latency) are consumed after the score MFMAs of tile kt have been issued, and those MFMAs' own
        // latency is covered by the dk/dv MFMAs of tile kt-1.
        // (Round 5 measured two deeper orders — the dk/dv products pinned behind this tile's softmax arithmetic, and the score / dP
        //  products issued one tile ahead: 325.2 -> 325.3 / 321.5 / 322.1 us, profiles/r05_n_core_order_ab.txt.  The loop is not
        //  waiting for the transposition round trip.)
        // K / V row fragments one tile ahead as well: their LDS latency is off the per-tile dependency chain
        bf16x4 Kn = *reinterpret_cast<const bf16x4*>(Kb + rcol);
        bf16x4 Vn = *reinterpret_cast<const bf16x4*>(Vb + rcol2);
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            const bf16x4 Kf = Kn, Vf = Vn;
            if (kt + 1 < MT) {
                Kn = *reinterpret_cast<const bf16x4*>(Kb + (kt + 1) * 16 * IR + rcol);
                Vn = *reinterpret_cast<const bf16x4*>(Vb + (kt + 1) * 16 * IR + rcol2);
            }
            f32x4 s[NQ], dp[NQ], pv[NQ], ds[NQ];
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
                s[n] = mfma16k16(Kf, bq[n], z4);
                dp[n] = mfma16k16(Vf, bdo[n], z4);
            }
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
#pragma unroll
#ifdef HS_EXPERIMENT_NOEXP      /* timing experiment only (scripts/phase_timing.py): what the exps cost */
                for (int r = 0; r < 4; ++r) pv[n][r] = fmaf(s[n][r], sc, lqn[n]);
#else
                for (int r = 0; r < 4; ++r) {
                    const float e = fmaf(s[n][r], sc, lqn[n]);
                    pv[n][r] = __builtin_amdgcn_exp2f((kt >= KCL && clamp_from <= kt) ? fminf(e, 0.f) : e);
                }
#endif
                // No key mask in here (rounds 1-2 masked the last tile only — wrong for 65..96-token sequences, found in round 3 by the
                // 64-band case of test_fused_decoder_matches_layerwise_decoder — and the general per-tile form compiled to 8 selects
                // per tile on ALL tiles, a quarter of this loop's VALU work): the caller zeroes the K and V image rows past Ts, so a
                // padded key has s = 0, dP = 0 and a finite (clamped, see KCL) garbage P / dS that meets K = 0 in dq and is dropped
                // from dk / dv below.
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[n][r] = pv[n][r] * (dp[n][r] - dl[n]);
            }
            if (kt > 0) {                       // previous tile's transposed operands, a VALU phase after their reads were issued
#pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    dkT[kt - 1] = mfma16k16(QT[n], Bds[n], dkT[kt - 1]);
                    dvT[kt - 1] = mfma16k16(dOT[n], Bp[n], dvT[kt - 1]);
                }
            }
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
                const bf16x4 pb = cvt4(pv[n]), dsb = cvt4(ds[n]);
                dqT[n] = mfma16k16(KT[kt], dsb, dqT[n]);
                bf16_t* tp = T + n * TT_PAIR;       // one tile pair per wave and query tile in hand: LDS executes a wave's accesses in order
                bf16_t* td = tp + 16 * TTS;
                st4(tp + tw, pb);
                st4(td + tw, dsb);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int n = 0; n < NQ; ++n) {
                Bp[n] = tr4(T + n * TT_PAIR + trd);
                Bds[n] = tr4(T + n * TT_PAIR + 16 * TTS + trd);
            }
        }
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            dkT[MT - 1] = mfma16k16(QT[n], Bds[n], dkT[MT - 1]);
            dvT[MT - 1] = mfma16k16(dOT[n], Bp[n], dvT[MT - 1]);
            if (q.g < 2) {
                bf16x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (bf16_t)(dqT[n][r] * scale);
                st4(Qb + (qt0 + n) * 16 * IR + wcol, v);
            }
        }
    };
    if constexpr (HS_DEC_CORE_NQ == 2) {
#pragma unroll 1
        for (int qt = 0; qt + 1 < MT; qt += 2) qtiles(std::integral_constant<int, 2>{}, qt);
        if constexpr (MT & 1) qtiles(std::integral_constant<int, 1>{}, MT - 1);
    } else {
#pragma unroll 1
        for (int qt = 0; qt < MT; ++qt) qtiles(std::integral_constant<int, 1>{}, qt);
    }
    if (q.g < 2) {
#pragma unroll
        for (int kt = 0; kt < MT; ++kt) {
            bf16x4 vk, vv;
            const bool live = kt * 16 + q.c16 < Ts;          // padded keys carry garbage (see the loop): their dk / dv rows are zero
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xk = live ? dkT[kt][r] * scale : 0.f, xv = live ? dvT[kt][r] : 0.f;
                vk[r] = (bf16_t)xk; vv[r] = (bf16_t)xv;
            }
            st4(Kb + kt * 16 * IR + wcol, vk);
            st4(Vb + kt * 16 * IR + wcol, vv);
        }
    }
}

struct DecBwdAttnArgs {
    const float* x; const float* dx1; float* dx; const bf16_t* o; const float* lse_g; int nsamples, Ts; DecW w; const bf16_t *qkvT, *pT;
    const float *qf, *kf, *vf, *pf;
    float *g_n1w, *g_n1b, *g_qw, *g_qb, *g_kw, *g_kb, *g_vw, *g_vb, *g_pw, *g_pb;
    HsDet det;
    float* slab;                 // see DecBwdMlpArgs
};

template <int MT>
__global__ __launch_bounds__(512, 2) void dec_bwd_attn_kernel(DecBwdAttnArgs p) {
    using L = DL<MT>;
    constexpr int R = L::R, IMG = R * IR, NPW = (R * 8 + NT_ - 1) / NT_;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* U = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Qb = U + IMG;
    bf16_t* Kb = Qb + IMG;
    bf16_t* Vb = Kb + IMG;
    bf16_t* Ob = Vb + IMG;            // attention output, later the image of dO
    bf16_t* DXb = Ob + IMG;           // bf16 image of dx1
    float* XS = reinterpret_cast<float*>(Ob);                     // fp32 staging tile aliases Ob|DXb and the head of lse (all dead by then)
    float* lse = reinterpret_cast<float*>(DXb + IMG);             // [8][R]
    float* dlt = lse + 8 * R;         // [8][R]
    bf16_t* TT = reinterpret_cast<bf16_t*>(dlt + 8 * R);          // per-wave transposition tiles of the attention core
    bf16_t* WQ = TT + 8 * TT_WAVE;                                // Wq|Wk|Wv row-major bf16 [192][LW] (row k at wrow(k)), resident
    bf16_t* WP = WQ + 3 * D * LW;                                 // Wp row-major bf16 [64][LW], resident
    float* CB = reinterpret_cast<float*>(WP + D * LW);            // bq|bk|bv (192), LN1 gamma (64), beta (64)
    GeoB q = geob();                  // re-derived at every phase of the sample loop (HS_DEC_REGEO, see at_bytes above)
    int wide = ((threadIdx.x & 7) ^ swz(threadIdx.x >> 3)) << 3;   // this thread's 16-byte chunk in the wide layout (row = tid >> 3 + 64 i)
    const int mt0 = q.wm * L::MH;
    int c8 = (threadIdx.x & 7) * 8;
    auto regeo = [&]() { if (HS_DEC_REGEO >= 2) q = geob(fresh_tid()); };
    auto regeo_core = [&]() { if (HS_DEC_REGEO >= 1) q = geob(fresh_tid()); };
    auto rewide = [&]() { if (HS_DEC_REGEO >= 2) { const int t = fresh_tid(); wide = ((t & 7) ^ swz(t >> 3)) << 3; c8 = (t & 7) * 8; } };
    // Weights staged once per workgroup (see dec_bwd_mlp_kernel): q|k|v read them as row pieces, the data gradients
    // (dO = dx1 Wp, du = dq Wq + dk Wk + dv Wv) read the same images with transpose reads.
    for (int i = threadIdx.x; i < 4 * D * 8; i += NT_) {
        const int m = i / (D * 8), row = (i % (D * 8)) >> 3, k8 = (i & 7) * 8;
        float f[8];
        ld8((m == 0 ? p.qf : (m == 1 ? p.kf : (m == 2 ? p.vf : p.pf))) + (size_t)row * D + k8, f);
        *reinterpret_cast<bf16x8*>(WQ + (m * D + wrow(row)) * LW + k8) = cvt8(f);
    }
    for (int i = threadIdx.x; i < 5 * D; i += NT_) CB[i] = i < 3 * D ? p.w.bqkv[i] : (i < 4 * D ? p.w.n1w[i - 3 * D] : p.w.n1b[i - 4 * D]);
    // The first sample's LayerNorm reads gamma / beta from CB before the loop's first barrier: without this one a fast wave
    // could read what a slower wave had not staged yet (stale LDS of the previous kernel) — seen as a 1e-3-level deviation of
    // one decoder block's q / k / v gradients in ~3 % of the runs at batch 64, where every sample is a workgroup's first.
    lds_barrier();

    if (HS_DEC_STG_N > 1) {
        const int n = (int)((blockIdx.x >> 3) % (HS_DEC_STG_N > 1 ? HS_DEC_STG_N : 1)) * HS_DEC_STG_ATTN;
        for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(32);
    }
    if (HS_DEC_PRIO > 0 && q.wave >= 4) __builtin_amdgcn_s_setprio(HS_DEC_PRIO);      // static priority for the younger half (round 6)
    f32x4 accP[2], accQ[3][2];         // dWp: n-tile = wave>>1; dWq|dWk|dWv: 12 n-tiles x 4 k-tiles, 3 x 2 per wave
#pragma unroll
    for (int a = 0; a < 2; ++a) accP[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) accQ[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradients (bp = column sums of dx1; bq / bk / bv = column sums of dq / dk / dv) as in dec_bwd_mlp_kernel: the dO^T
    // fragment of the weight-gradient products against a tile of ones, on the even waves; the lane keeps column 4 g + (c16 & 3)
    float dgam[8], dbet[8], dbpw = 0.f, dbqw[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) { dgam[e] = 0.f; dbet[e] = 0.f; }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;
    const bool bias_wave = (q.wave & 1) == 0;
    const int r4 = q.c16 & 3;

    float fa[NPW][8], d1a[NPW][8];
    f32x4 l4a;
    bf16x8 ova[NPW];
    // (unconditional with a validity flag: under an `if` the old values would stay live through the whole iteration)
    // part 0: the x rows, 1: the dx1 rows, 2: O and logsumexp; -1: everything (the first sample)
    auto fetch_sample = [&](int smp, int part = -1) {               // every load of a sample in flight before the first use
        const bool valid = smp < p.nsamples;
        const size_t nb = (size_t)smp * p.Ts;
        const int rbytes = valid ? p.Ts * D * 4 : 0;                 // (rows_rsrc: lanes past the sequence / the batch read zeros)
        const __amdgpu_buffer_rsrc_t xr = rows_rsrc(p.x + nb * D, rbytes), d1r = rows_rsrc(p.dx1 + nb * D, rbytes);
        const __amdgpu_buffer_rsrc_t orr = rows_rsrc(p.o + nb * D, rbytes >> 1);
        const int tid = fresh_tid();
        const unsigned lo = (unsigned)((tid >> 3) * D + (tid & 7) * 8) * 4u;      // byte offset of this lane's 8 floats in row group 0
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const unsigned bo = lo + (unsigned)(i * (NT_ / 8) * D * 4);
            if (part < 0 || part == 0) bld8(xr, bo, fa[i]);
            if (part < 0 || part == 1) bld8(d1r, bo, d1a[i]);
            if (part < 0 || part == 2) ova[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(orr, (int)(bo >> 1), 0, 0));
        }
        if (part < 0 || part == 2) {
            // (rows past Ts read 0 here; they become 1e30 — exp2(s - 1e30) = 0 — where the table is written to LDS: a select at this
            //  point would be a USE of the load, and hipcc waits for it with vmcnt(0), i.e. for the whole prefetch burst)
            l4a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rows_rsrc(p.lse_g + nb * 8, valid ? p.Ts * 32 : 0), tid * 16, 0, 0));
        }
    };
    fetch_sample(blockIdx.x);
    // (the first sample's rows are wanted at once: waited for HERE, with a builtin the wait-count pass can see, no load is pending on
    //  either way into the loop head — otherwise the merged state makes the loop's first use of a prefetched row an s_waitcnt vmcnt(0),
    //  which on the back edge drains the previous sample's dx stores: attn.hip HS_BB_EARLY_WAIT)
    if (HS_DEC_PRELOOP_WAIT) __builtin_amdgcn_s_waitcnt(0x0F70);
    PH_DECL
    PH2_DECL
    for (int sample = blockIdx.x; sample < p.nsamples; sample += gridDim.x) {
        const size_t rb = (size_t)sample * p.Ts;
        const int wl = launder_i(0);                  // keeps the LDS weight reads inside the sample loop
        const bf16_t* WQl = WQ + wl;
        const bf16_t* WPl = WP + wl;
        const float* CBl = CB + wl;
        // x, dx1, O and logsumexp of this sample are in registers already: fetch_sample() ran in front of the PREVIOUS sample's
        // LayerNorm-backward epilogue, so the HBM round trip is under that epilogue instead of exposed here (17 % of the kernel
        // before).  Works since the kernel no longer spills: scratch reloads share vmcnt with these loads and used to drain them.
        rewide();
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int pc = threadIdx.x + NT_ * i;
            if (pc < R * 8) {
                const int row = pc >> 3;
                float gm[8], bt[8];
                float (&f)[8] = fa[i];
                float (&d1)[8] = d1a[i];
                const bf16x8 ov = ova[i];
                ld8(CBl + 3 * D + c8, gm); ld8(CBl + 4 * D + c8, bt);
                *reinterpret_cast<bf16x8*>(Ob + row * IR + wide) = ov;
                const float mean = red8(f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7]) * (1.f / D);
                float v = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { f[e] -= mean; v += f[e] * f[e]; }
                const float rstd = rsqrtf(red8(v) * (1.f / D) + 1e-5f);
                float u[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) u[e] = f[e] * rstd * gm[e] + bt[e];
                *reinterpret_cast<bf16x8*>(U + row * IR + wide) = cvt8(u);
                *reinterpret_cast<bf16x8*>(DXb + row * IR + wide) = cvt8(d1);
            }
        }
        if (threadIdx.x < 2 * R) {                     // logsumexp [row][8 heads] -> [head][row]
            const int row = threadIdx.x >> 1, h4 = (threadIdx.x & 1) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) lse[(h4 + e) * R + row] = row < p.Ts ? l4a[e] : 1e30f;
        }
        lds_barrier();
        PH(0)
        regeo();
        // q | k | v, all row-major.  Operands swapped (here and in the dO / du products below): a lane owns 4 consecutive columns
        // of one token, so tiles reach the images as 8-byte writes, the bias is one 16-byte LDS read and delta needs no 8-lane sums
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f32x4 acc[L::MH][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(CBl + c * D + (q.wn * 2 + j) * 16 + q.g * 4);
#pragma unroll
                for (int mi = 0; mi < L::MH; ++mi) acc[mi][j] = b;
            }
            mm_rows<L::MH, 2>(U, WQl, c * 4 + q.wn * 2, mt0, q, acc);
            bf16_t* dst = c == 0 ? Qb : (c == 1 ? Kb : Vb);
#pragma unroll
            for (int mi = 0; mi < L::MH; ++mi) {
                const int mt = mt0 + mi;
                if (mt >= MT) continue;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    st4(dst + tile_off(mt, q.wn * 2 + j, q), cvt4(acc[mi][j]));
            }
        }
        PH(1)
        regeo();
        // dO = dx1 * Wp ; dWp += dx1^T * O
        f32x4 dO[L::MH][2];
#pragma unroll
        for (int mi = 0; mi < L::MH; ++mi) { dO[mi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; dO[mi][1] = dO[mi][0]; }
        mm_cols<L::MH>(DXb, WPl, mt0, q, dO);
        f32x4 accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kk = 0; kk < R / 32; ++kk) {
            const bf16x8 a = wg_frag<false>(DXb, q.wave >> 1, kk, q);
            if (bias_wave) accb = mfma16(a, ones, accb);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
                accP[k2] = mfma16(a, wg_frag<false>(Ob, (q.wave & 1) * 2 + k2, kk, q), accP[k2]);
        }
        if constexpr (R % 32 != 0) {
            const bf16x8 a = wg_frag<true>(DXb, q.wave >> 1, R / 32, q);
            if (bias_wave) accb = mfma16(a, ones, accb);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
                accP[k2] = mfma16(a, wg_frag<true>(Ob, (q.wave & 1) * 2 + k2, R / 32, q), accP[k2]);
        }
        dbpw += r4 == 0 ? accb[0] : (r4 == 1 ? accb[1] : (r4 == 2 ? accb[2] : accb[3]));
        PH(2)
        regeo();
        // delta[head][row] = sum_keys P dP = sum_d dO[row][d] O[row][d] over the head's 8 columns (8 adjacent lanes)
        // (a lane holds 4 of a head's 8 columns of one token: 4 products in the lane + the lane group next door, g ^ 1)
        bf16x4 dOb16[L::MH][2];
#pragma unroll
        for (int mi = 0; mi < L::MH; ++mi) {
            const int mt = mt0 + mi;
            if (mt >= MT) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = mt * 16 + q.c16, col = (q.wn * 2 + j) * 16 + q.g * 4;
                dOb16[mi][j] = cvt4(dO[mi][j]);
                const bf16x4 ob = *reinterpret_cast<const bf16x4*>(Ob + tile_off(mt, q.wn * 2 + j, q));
                float v = bf2f(dOb16[mi][j][0]) * bf2f(ob[0]);
#pragma unroll
                for (int r = 1; r < 4; ++r) v = fmaf(bf2f(dOb16[mi][j][r]), bf2f(ob[r]), v);
                float a = v, b = v;
                swap_rows16(a, b);
                if ((q.g & 1) == 0) dlt[(col >> 3) * R + row] = a + b;
            }
        }
        lds_barrier();
        PH(3)
#pragma unroll
        for (int mi = 0; mi < L::MH; ++mi) {
            const int mt = mt0 + mi;
            if (mt >= MT) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                st4(Ob + tile_off(mt, q.wn * 2 + j, q), dOb16[mi][j]);
        }
        // K / V rows of the padding keys -> 0 (every wave's q | k | v columns are in the images since the barrier above):
        // what keeps padded keys out of dq without a mask in the attention loop
        for (int i = threadIdx.x; i < (R - p.Ts) * 16; i += NT_) {
            const int row = p.Ts + (i >> 4), k8 = (i & 7) * 8;
            *reinterpret_cast<bf16x8*>(((i & 8) ? Vb : Kb) + row * IR + k8) = zero8();
        }
        lds_barrier();
        PH(4)
        // warm L2 / TLB with the next sample's rows while the attention core runs (no global loads in there)
        float touch[4] = {0.f, 0.f, 0.f, 0.f};
        {
            const int nxt = sample + gridDim.x;
            const size_t nb = (size_t)nxt * p.Ts;
            const int t16 = threadIdx.x * 16;
            if (HS_TOUCH_A && nxt < p.nsamples) {
                if (t16 < p.Ts * D) { touch[0] = p.x[nb * D + t16]; touch[1] = p.dx1[nb * D + t16]; }
                if (2 * t16 < p.Ts * D) touch[2] = bf2f(p.o[nb * D + 2 * t16]);
                if (t16 < p.Ts * 8) touch[3] = p.lse_g[nb * 8 + t16];
            }
        }
        // attention backward, one head per wave, dq/dk/dv written in place over q/k/v
        // (a half-tile start stagger of waves 4-7, MI355X_MICROARCH.md "two waves per SIMD" item 9, measured neutral in round 1;
        //  HS_DEC_CORE_STG = the same as a knob, in units of 64 clocks, re-measured in round 6: profiles/EXPERIMENTS.md)
        if (HS_DEC_CORE_STG > 0 && q.wave >= 4) __builtin_amdgcn_s_sleep(HS_DEC_CORE_STG);
        regeo_core();
        attn_head_bwd<MT>(Qb, Kb, Vb, Ob, lse + q.wave * R, dlt + q.wave * R, TT + q.wave * TT_WAVE, q.wave, p.Ts, q);
        lds_barrier();
        PH(5)
        PH2_START
        __builtin_amdgcn_sched_barrier(0);
        // (issuing this burst is 55 % of this phase, profiles/r05_f_phase_timing.txt: the waves sit at the issue while every CU of
        //  the chip asks for its sample at once.  Issued in three parts between the products below instead: 323.4 -> 318.3 us alone,
        //  nothing on top of the start stagger (309.0 vs 310.0 us, profiles/r05_i_stagger_sweep2.txt): one burst kept.)
        fetch_sample(sample + (int)gridDim.x);          // ~4 us (du + dWqkv + epilogue) ahead of its use
        __builtin_amdgcn_sched_barrier(0);
        PH2(0)
        regeo_core();
        // du = dq Wq + dk Wk + dv Wv ; dWq|dWk|dWv += d{q,k,v}^T u
        f32x4 du[L::MH][2];
#pragma unroll
        for (int mi = 0; mi < L::MH; ++mi) { du[mi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; du[mi][1] = du[mi][0]; }
        asm volatile("" :: "v"(touch[0]), "v"(touch[1]), "v"(touch[2]), "v"(touch[3]));
        mm_cols<L::MH>(Qb, WQl, mt0, q, du);
        mm_cols<L::MH>(Kb, WQl + D * LW, mt0, q, du);
        mm_cols<L::MH>(Vb, WQl + 2 * D * LW, mt0, q, du);
        PH2(1)
        regeo();
        f32x4 accqb[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) accqb[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kk = 0; kk < R / 32; ++kk) {
            bf16x8 b[2];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) b[k2] = wg_frag<false>(U, (q.wave & 1) * 2 + k2, kk, q);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int nt12 = (q.wave >> 1) * 3 + t, mat = nt12 >> 2, nt = nt12 & 3;
                const bf16_t* dOi = mat == 0 ? Qb : (mat == 1 ? Kb : Vb);
                const bf16x8 a = wg_frag<false>(dOi, nt, kk, q);
                if (bias_wave) accqb[t] = mfma16(a, ones, accqb[t]);
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) accQ[t][k2] = mfma16(a, b[k2], accQ[t][k2]);
            }
        }
        if constexpr (R % 32 != 0) {
            bf16x8 b[2];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) b[k2] = wg_frag<true>(U, (q.wave & 1) * 2 + k2, R / 32, q);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int nt12 = (q.wave >> 1) * 3 + t, mat = nt12 >> 2, nt = nt12 & 3;
                const bf16_t* dOi = mat == 0 ? Qb : (mat == 1 ? Kb : Vb);
                const bf16x8 a = wg_frag<true>(dOi, nt, R / 32, q);
                if (bias_wave) accqb[t] = mfma16(a, ones, accqb[t]);
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) accQ[t][k2] = mfma16(a, b[k2], accQ[t][k2]);
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) dbqw[t] += r4 == 0 ? accqb[t][0] : (r4 == 1 ? accqb[t][1] : (r4 == 2 ? accqb[t][2] : accqb[t][3]));
        PH2(2)
        regeo(); rewide();
        acc_to_xs<L::MH, true>(XS, mt0, MT, q, du);
        float xe[NPW][8], d1e[NPW][8];                // L2-hot re-reads for the LayerNorm backward, in flight over the barrier
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int row = (threadIdx.x + NT_ * i) >> 3;
#pragma unroll
            for (int e = 0; e < 8; ++e) { xe[i][e] = 0.f; d1e[i][e] = 0.f; }
#ifndef HS_ABL_DEC_REREAD   /* timing ablation (variant builds only): what the epilogue's second read of x and dx1 costs */
            {
                const unsigned bo = (unsigned)(row * D + c8) * 4u;
                bld8(rows_rsrc(p.x + rb * D, p.Ts * D * 4), bo, xe[i]); bld8(rows_rsrc(p.dx1 + rb * D, p.Ts * D * 4), bo, d1e[i]);
            }
#endif
        }
        PH2(3)
        lds_barrier();
        // (both row groups' re-reads — and the next sample's rows, issued a phase ago — waited for once, before the first dx store:
        //  a wait for the second group's rows behind the first group's stores would be an s_waitcnt vmcnt(0) that drains them)
        if (HS_DEC_PRELOOP_WAIT) __builtin_amdgcn_s_waitcnt(0x0F70);
        PH2(4)
        PH(6)
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int pc = threadIdx.x + NT_ * i;
            if (pc < R * 8) {
                const int row = pc >> 3;
                float dv[8], t[8], gm[8];
                float (&xh)[8] = xe[i];
                float (&d1)[8] = d1e[i];
                ld8(XS + row * LX + c8, dv);
                ld8(CBl + 3 * D + c8, gm);
                const float mean = red8(xh[0] + xh[1] + xh[2] + xh[3] + xh[4] + xh[5] + xh[6] + xh[7]) * (1.f / D);
                float v = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { xh[e] -= mean; v += xh[e] * xh[e]; }
                const float rstd = rsqrtf(red8(v) * (1.f / D) + 1e-5f);
                float a = 0.f, b = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { xh[e] *= rstd; t[e] = dv[e] * gm[e]; a += t[e]; b += t[e] * xh[e]; }
                a = red8(a) * (1.f / D); b = red8(b) * (1.f / D);
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o[e] = d1[e] + rstd * (t[e] - a - xh[e] * b);
                    dgam[e] += dv[e] * xh[e];
                    dbet[e] += dv[e];
                }
                bst8(rows_rsrc(p.dx + rb * D, p.Ts * D * 4), (unsigned)(row * D + c8) * 4u, o);      // (rows past Ts: dropped by the bounds check)
            }
        }
        lds_barrier();
        PH(7)
    }

    PH_FLUSH(0)
    PH2_FLUSH
    // ---- commit
    float* red = XS;
    float* vec = p.slab ? p.slab + kSlabTileFloats + (size_t)blockIdx.x * kVec : nullptr;
    flush_wide(red, dgam, p.g_n1w, p.det, vec ? vec + kVN1W : nullptr);
    flush_wide(red, dbet, p.g_n1b, p.det, vec ? vec + kVN1B : nullptr);
    if (bias_wave && (q.c16 >> 2) == 0) {
        const int ci = q.g * 4 + r4;
        const int cp = (q.wave >> 1) * 16 + ci;
        if (vec) vec[kVPB + cp] = dbpw; else hs_gadd(p.det, p.g_pb + cp, dbpw);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int nt12 = (q.wave >> 1) * 3 + t, mat = nt12 >> 2, col = (nt12 & 3) * 16 + ci;
            if (vec) vec[(mat == 0 ? kVQB : (mat == 1 ? kVKB : kVVB)) + col] = dbqw[t];
            else hs_gadd(p.det, (mat == 0 ? p.g_qb : (mat == 1 ? p.g_kb : p.g_vb)) + col, dbqw[t]);
        }
    }
    if (p.slab) {                 // slots 72..103 of this workgroup's slab: accP[k2][r] then accQ[t][k2][r]
        float* sl = p.slab + ((size_t)blockIdx.x * kDwSlots + kDwSlotsMlp) * NT_ + threadIdx.x;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int r = 0; r < 4; ++r) sl[(size_t)(k2 * 4 + r) * NT_] = accP[k2][r];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int r = 0; r < 4; ++r) sl[(size_t)(8 + (t * 2 + k2) * 4 + r) * NT_] = accQ[t][k2][r];
        return;
    }
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            hs_gadd(p.det, p.g_pw + (size_t)((q.wave >> 1) * 16 + q.g * 4 + r) * D + ((q.wave & 1) * 2 + k2) * 16 + q.c16, accP[k2][r]);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int nt12 = (q.wave >> 1) * 3 + t, mat = nt12 >> 2, nt = nt12 & 3;
        float* dst = mat == 0 ? p.g_qw : (mat == 1 ? p.g_kw : p.g_vw);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                hs_gadd(p.det, dst + (size_t)(nt * 16 + q.g * 4 + r) * D + ((q.wave & 1) * 2 + k2) * 16 + q.c16, accQ[t][k2][r]);
    }
}

// Sums the workgroups' weight-gradient partials of one decoder block ([workgroup][slot][thread], written by the two
// persistent backward kernels) and adds them into the block's dW tensors: workgroup = slot, thread = the committing thread
// of the backward kernels (same index map as their atomic commits).  Reads are coalesced 2-KB rows; the sum runs in a fixed
// order, so these gradients are bit-reproducible without the fixed-point shadow buffer.
struct DecDwReduceArgs {
    const float* slab; int nwg; int h;
    float *g_w1w, *g_w3w, *g_w2w, *g_qw, *g_kw, *g_vw, *g_pw;
    float *g_n2w, *g_n2b, *g_w2b, *g_w1b, *g_w3b, *g_n1w, *g_n1b, *g_pb, *g_qb, *g_kb, *g_vb;
};
// fixed-order sum of src[w * stride], w < nwg, with 16 loads in flight per thread (the kernel is a pure latency-bound gather)
__device__ __forceinline__ float slab_sum(const float* src, size_t stride, int nwg) {
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    int w = 0;
    for (; w + 16 <= nwg; w += 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += src[(size_t)(w + i) * stride];
    }
    for (; w < nwg; ++w) acc[0] += src[(size_t)w * stride];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1)
#pragma unroll
        for (int i = 0; i < o; ++i) acc[i] += acc[i + o];
    return acc[0];
}
__global__ __launch_bounds__(NT_) void dec_dw_reduce_kernel(DecDwReduceArgs p) {
    const int slot = blockIdx.x, tid = threadIdx.x;
    if (slot >= kDwSlots) {                               // the bias / LayerNorm vectors: [workgroup][kVec]
        // the bias / LayerNorm vectors: one WAVE per output element (9 vectors of 64 and 2 of 192 = 960 elements, 8 per
        // workgroup): its 64 lanes split the workgroups' partials, then a fixed-order
        // lane tree — a thread per element walked up to 1,024 partials alone and was the launch's 35-us tail
        const int e = (slot - kDwSlots) * (NT_ / 64) + (tid >> 6), lane = tid & 63;
        if (e >= 9 * D + 2 * HPD) return;
        int which, c;
        if (e < 3 * D) { which = e / D; c = e % D; }                               // n2w, n2b, w2b
        else if (e < 3 * D + 2 * HPD) { which = 3 + (e - 3 * D) / HPD; c = (e - 3 * D) % HPD; }   // w1b, w3b
        else { which = 5 + (e - 3 * D - 2 * HPD) / D; c = (e - 3 * D - 2 * HPD) % D; }           // n1w, n1b, pb, qb, kb, vb
        const int off[11] = {kVN2W, kVN2B, kVW2B, kVW1B, kVW3B, kVN1W, kVN1B, kVPB, kVQB, kVKB, kVVB};
        float* const dst[11] = {p.g_n2w, p.g_n2b, p.g_w2b, p.g_w1b, p.g_w3b, p.g_n1w, p.g_n1b, p.g_pb, p.g_qb, p.g_kb, p.g_vb};
        int o = off[0];
        float* d = dst[0];
#pragma unroll
        for (int i = 1; i < 11; ++i) if (which == i) { o = off[i]; d = dst[i]; }
        const bool wide = which == 3 || which == 4;
        if (wide && c >= p.h) return;                    // w1b / w3b: padded to 192, valid below the hidden width
        const float* src = p.slab + kSlabTileFloats + o + c;
        float v = 0.f;
        for (int w = lane; w < p.nwg; w += 64) {
            const float* q = src + (size_t)w * kVec;
            v += q[0];
        }
        v = wave_sum(v);
        if (lane == 0) d[c] += v;
        return;
    }
    const float v = slab_sum(p.slab + (size_t)slot * NT_ + tid, (size_t)kDwSlots * NT_, p.nwg);
    const int lane = tid & 63, c16 = lane & 15, g = lane >> 4, wave = tid >> 6;
    if (slot < kDwSlotsMlp) {
        const int r = slot & 3, k2 = (slot >> 2) & 1, ct = slot >> 3, t = ct % 3, c = ct / 3;
        const int nt12 = (wave >> 1) * 3 + t, mat = nt12 >> 2, nt = nt12 & 3;
        const int n = nt * 16 + g * 4 + r, k = ((wave & 1) * 2 + k2) * 16 + c16;
        if (mat == 0) {                                   // dW2[d][h]: row n (model dim), column c*64 + k (hidden)
            if (c * 64 + k < p.h) p.g_w2w[(size_t)n * p.h + c * 64 + k] += v;
        } else {                                          // dW1 / dW3 [h][d]: row c*64 + n (hidden), column k
            float* dst = mat == 1 ? p.g_w1w : p.g_w3w;
            if (c * 64 + n < p.h) dst[(size_t)(c * 64 + n) * D + k] += v;
        }
    } else {
        const int a = slot - kDwSlotsMlp;
        if (a < 8) {                                      // accP[k2][r]
            const int r = a & 3, k2 = a >> 2;
            p.g_pw[(size_t)((wave >> 1) * 16 + g * 4 + r) * D + ((wave & 1) * 2 + k2) * 16 + c16] += v;
        } else {                                          // accQ[t][k2][r]
            const int b = a - 8, r = b & 3, k2 = (b >> 2) & 1, t = b >> 3;
            const int nt12 = (wave >> 1) * 3 + t, mat = nt12 >> 2, nt = nt12 & 3;
            float* dst = mat == 0 ? p.g_qw : (mat == 1 ? p.g_kw : p.g_vw);
            dst[(size_t)(nt * 16 + g * 4 + r) * D + ((wave & 1) * 2 + k2) * 16 + c16] += v;
        }
    }
}

template <int MT>
int launch_fwd(const DecFwdArgs& a, hipStream_t s) {
    using L = DL<MT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_block_fwd_kernel<MT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, L::FWD_TOTAL);
        attr_set = true;
    }
    static int wgs = 0;                       // workgroups: 2 per CU walking the samples (HSIMAE_DEC_FWD_WGS overrides)
    if (!wgs) wgs = 512;
    hipLaunchKernelGGL((dec_block_fwd_kernel<MT>), dim3(a.nsamples < wgs ? a.nsamples : wgs), dim3(256), L::FWD_TOTAL, s, a);
    return (int)hipGetLastError();
}

template <int MT>
int launch_attn_fwd(const DecAttnFwdArgs& a, hipStream_t s) {
    constexpr int LDS = 2 * MT * 16 * LU * 2;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_attn_fwd_kernel<MT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL((dec_attn_fwd_kernel<MT>), dim3(a.nsamples), dim3(256), LDS, s, a);
    return (int)hipGetLastError();
}

}  // namespace

template <int MT>
int launch_bwd(const DecBwdMlpArgs& a, const DecBwdAttnArgs& b, hipStream_t s) {
    using L = DL<MT>;
    constexpr int IMG = L::IMGB;
    constexpr int LDS_A = 5 * IMG + 2 * WRB * 2 + 2 * HPD * 4 + 2 * D * 4;      // images, W1 | W3, b1 | b3, LayerNorm-2 gamma | beta
    static_assert(3 * IMG >= L::R * LX * 4, "fp32 staging tile must fit over Gc|DH1|DH3");
    static_assert(LDS_A - 5 * IMG >= 16 * IR * 2, "image overrun of the unguarded du2 product must stay inside the allocation");
    constexpr int LDS_B = L::BWD_ATTN_LDS;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_bwd_mlp_kernel<MT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_A);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_bwd_attn_kernel<MT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
        attr_set = true;
    }
    const int grid = a.nsamples < 256 ? a.nsamples : 256;      // persistent: one workgroup per CU
    hipLaunchKernelGGL((dec_bwd_mlp_kernel<MT>), dim3(grid), dim3(NT_), LDS_A, s, a);
    hipLaunchKernelGGL((dec_bwd_attn_kernel<MT>), dim3(grid), dim3(NT_), LDS_B, s, b);
    if (a.slab) {
        DecDwReduceArgs r;
        r.slab = a.slab; r.nwg = grid; r.h = a.w.h;
        r.g_w1w = a.g_w1w; r.g_w3w = a.g_w3w; r.g_w2w = a.g_w2w; r.g_qw = b.g_qw; r.g_kw = b.g_kw; r.g_vw = b.g_vw; r.g_pw = b.g_pw;
        r.g_n2w = a.g_n2w; r.g_n2b = a.g_n2b; r.g_w2b = a.g_w2b; r.g_w1b = a.g_w1b; r.g_w3b = a.g_w3b;
        r.g_n1w = b.g_n1w; r.g_n1b = b.g_n1b; r.g_pb = b.g_pb; r.g_qb = b.g_qb; r.g_kb = b.g_kb; r.g_vb = b.g_vb;
        hipLaunchKernelGGL(dec_dw_reduce_kernel, dim3(kDwSlots + (9 * D + 2 * HPD + NT_ / 64 - 1) / (NT_ / 64)), dim3(NT_), 0, s, r);
    }
    return (int)hipGetLastError();
}

bool hs_dec_fused_supported(int d, int heads, int hidden, int Ts) {
    // the backward kernels keep 5-6 bf16 images + an fp32 tile of the sample in LDS: up to 7 m-tiles (112 tokens).
    // Every kernel of this file is compiled for HPD = 192-row hidden images (w1 / w3 / w2 / w2T packed to rup(hidden, 32) = 192
    // rows, 6 k-steps): a narrower hidden width would be read with the wrong k-step layout and past the caller's images
    // (ADVICE r03), so it is refused here — the layer-at-a-time schedule covers it.
    return d == D && heads == 8 && (hidden + 31) / 32 * 32 == HPD && hidden % 4 == 0 && Ts <= 112 && Ts >= 16;
}

int hs_dec_block_bwd(const float* x, const float* x1, const float* dy, float* dx1_tmp, float* dx, const hs_bf16* o,
                     const float* lse, int nsamples, int Ts, const DecBlockPtrs& bp, const DecBlockGrads& g, hipStream_t s,
                     float* slab) {
    DecW w;
    w.n1w = bp.n1w; w.n1b = bp.n1b; w.bqkv = bp.bqkv; w.pb = bp.pb; w.n2w = bp.n2w; w.n2b = bp.n2b;
    w.w1b = bp.w1b; w.w3b = bp.w3b; w.w2b = bp.w2b;
    w.qkv = bp.qkv; w.p = bp.p; w.w1 = bp.w1; w.w3 = bp.w3; w.w2 = bp.w2; w.h = bp.h;
    DecBwdMlpArgs a;
    a.x1 = x1; a.dy = dy; a.dx1 = dx1_tmp; a.nsamples = nsamples; a.Ts = Ts; a.w = w; a.w2T = bp.w2T; a.w13T = bp.w13T;
    a.w1f = bp.w1f; a.w3f = bp.w3f;
    a.g_n2w = g.n2w; a.g_n2b = g.n2b; a.g_w1w = g.w1w; a.g_w1b = g.w1b; a.g_w3w = g.w3w; a.g_w3b = g.w3b;
    a.g_w2w = g.w2w; a.g_w2b = g.w2b; a.det = g.det; a.slab = slab;
    DecBwdAttnArgs b;
    b.x = x; b.dx1 = dx1_tmp; b.dx = dx; b.o = o; b.lse_g = lse; b.nsamples = nsamples; b.Ts = Ts; b.w = w; b.qkvT = bp.qkvT; b.pT = bp.pT;
    b.qf = bp.qf; b.kf = bp.kf; b.vf = bp.vf; b.pf = bp.pf;
    b.g_n1w = g.n1w; b.g_n1b = g.n1b; b.g_qw = g.qw; b.g_qb = g.qb; b.g_kw = g.kw; b.g_kb = g.kb; b.g_vw = g.vw;
    b.g_vb = g.vb; b.g_pw = g.pw; b.g_pb = g.pb; b.det = g.det; b.slab = slab;
    const int mt = (Ts + 15) / 16;
    if (mt <= 4) return launch_bwd<4>(a, b, s);
    if (mt <= 7) return launch_bwd<7>(a, b, s);
    return HS_EUNSUPPORTED;
}

// attention half of the block only (x1 = x + proj(attention(LN1 x)); O and logsumexp kept for the backward)
int hs_dec_attn_fwd(const float* x, float* x1, hs_bf16* o, float* lse, int nsamples, int Ts, const DecBlockPtrs& bp, hipStream_t s) {
    DecAttnFwdArgs a;
    a.x = x; a.x1 = x1; a.o = o; a.lse = lse; a.nsamples = nsamples; a.Ts = Ts;
    a.w.n1w = bp.n1w; a.w.n1b = bp.n1b; a.w.bqkv = bp.bqkv; a.w.pb = bp.pb; a.w.n2w = bp.n2w; a.w.n2b = bp.n2b;
    a.w.w1b = bp.w1b; a.w.w3b = bp.w3b; a.w.w2b = bp.w2b;
    a.w.qkv = bp.qkv; a.w.p = bp.p; a.w.w1 = bp.w1; a.w.w3 = bp.w3; a.w.w2 = bp.w2; a.w.h = bp.h;
    const int mt = (Ts + 15) / 16;
    if (mt <= 4) return launch_attn_fwd<4>(a, s);
    if (mt <= 7) return launch_attn_fwd<7>(a, s);
    return HS_EUNSUPPORTED;
}

int hs_dec_block_fwd(const float* x, float* x1, float* x2, hs_bf16* o, float* lse, int nsamples, int Ts,
                     const DecBlockPtrs& bp, hipStream_t s) {
    DecFwdArgs a;
    a.x = x; a.x1 = x1; a.x2 = x2; a.o = o; a.lse = lse; a.nsamples = nsamples; a.Ts = Ts;
    a.w.n1w = bp.n1w; a.w.n1b = bp.n1b; a.w.bqkv = bp.bqkv; a.w.pb = bp.pb; a.w.n2w = bp.n2w; a.w.n2b = bp.n2b;
    a.w.w1b = bp.w1b; a.w.w3b = bp.w3b; a.w.w2b = bp.w2b;
    a.w.qkv = bp.qkv; a.w.p = bp.p; a.w.w1 = bp.w1; a.w.w3 = bp.w3; a.w.w2 = bp.w2; a.w.h = bp.h;
    const int mt = (Ts + 15) / 16;
    if (mt <= 4) return launch_fwd<4>(a, s);
    if (mt <= 7) return launch_fwd<7>(a, s);
    return HS_EUNSUPPORTED;
}

HS_UNIT_VARIANT_BITS(fused_dec)
