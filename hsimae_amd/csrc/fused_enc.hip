// Fused MLP half of an encoder Block at D = 128 (HSIMAE-Base):  x2 = x1 + W2( silu(W1 LN2(x1)) * (W3 LN2(x1)) )
// (Models.py:231-232, 299, 305) and its backward.  Layer-at-a-time, this half moves ~1.1 KB of bf16
// intermediates per token per direction (u2, h1|h3, g) and is HBM-bound at 64 FLOP/B; here a workgroup keeps a
// 128-row panel in LDS/registers from LayerNorm to the residual add.
//   forward : reads x1 (512 B/row), writes x2 (512 B/row)                — nothing else touches HBM
//   backward: reads x1, dY, recomputes u2/h1/h3/g, writes dx1 and the three weight-gradient operands
//             (u2, dh1|dh3, g) that `wgrad_kernel` consumes; LayerNorm-2 parameter grads via atomics.
// Workgroup = 4 waves as 2(M) x 2(N) on a 64-row panel: wave (wm, wn) owns m-tiles {2wm, 2wm+1} and n-tiles {2wn, 2wn+1} of every
// 64-column chunk.  Weight fragments are prefetched one chunk ahead.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

// Per-phase cycle accounting for scripts/phase_timing.py (compiled only with -DHS_PHASE_TIMING; never in the shipped library)
#ifdef HS_PHASE_TIMING
__device__ unsigned long long hs_phase_cycles_enc[32];
extern "C" __attribute__((visibility("default"))) int hsimae_debug_phases_enc(unsigned long long* out, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hs_phase_cycles_enc), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(hs_phase_cycles_enc), z, sizeof(z)); }
    return rc;
}
#define PH_DECL unsigned long long ph_t0 = __builtin_readcyclecounter(), ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PH(i) { const unsigned long long ph_t = __builtin_readcyclecounter(); ph_acc[i] += ph_t - ph_t0; ph_t0 = ph_t; }
#define PH_FLUSH(base) if (threadIdx.x == 0) { for (int i = 0; i < 8; ++i) atomicAdd(&hs_phase_cycles_enc[(base) + i], ph_acc[i]); }
#else
#define PH_DECL
#define PH(i)
#define PH_FLUSH(base)
#endif

#ifndef HS_NT_A
#define HS_NT_A 1      /* dh1|dh3 / g of enc_mlp_bwd as streaming stores: step -1.3 % */
#endif
#ifndef HS_NT_B
#define HS_NT_B 0      /* u2 / dY copy / dx1 / dx1 copy */
#endif
// enc_mlp_bwd, planar operands (round 6): the chunk's g / dh1 / dh3 pieces leave AFTER the chunk's data-gradient products, as
// unconditional bounds-checked buffer stores.  On gfx950 loads and stores share one in-order counter: with the stores in front of
// the du2 products (rounds 2-5) every wait for a W1^T | W3^T fragment also waited for all but the last few operand stores of the
// chunk — their trip to HBM — and the conditional store loop (1.5 iterations per thread) made hipcc's wait-count pass assume the
// worst at the next chunk's first fragment use (s_waitcnt vmcnt(0)).  Behind the products the stores have a whole chunk of
// arithmetic to drain under, and every wave issues the same six store instructions (lanes with nothing to store address past the
// buffer's extent), so the next chunk's waits are counted ones that leave them in flight.
#ifndef HS_MLPB_LATE_STORES
#define HS_MLPB_LATE_STORES 1
#endif

// Forward kernel occupancy knobs (round 3, profiles/r03_d_variants.txt).  Default: the panel's fp32 copy stays in LDS for the
// residual (XR) and three workgroups share a CU.  Without the copy (residual re-read from L2 in the store loop) a workgroup
// needs 27 KB and four fit: <128, 352> 59.3 -> 57.0 us, <64, 192> 87.4 -> 80.2 us per launch, i.e. 0.1 ms per step, for
// +1.9 GB of fetches per step (the re-reads are counted at the fabric) — not adopted; five per CU (96 registers) spill: 115 us.
#ifndef HS_MLP_XR_128
#define HS_MLP_XR_128 1
#endif
#ifndef HS_MLP_XR_64
#define HS_MLP_XR_64 1
#endif
#ifndef HS_MLP_WPCF_128
#define HS_MLP_WPCF_128 3
#endif
#ifndef HS_MLP_WPCF_64
#define HS_MLP_WPCF_64 3
#endif
#ifndef HS_SWZ256
#define HS_SWZ256 0          /* 1: the swizzled unpadded layout at D = 256 too (experiment) */
#endif
#ifndef HS_MLP_FWD_PERSIST
#define HS_MLP_FWD_PERSIST 0     /* 1: the forward kernel walks panels with the next panel's rows prefetched into registers.  Measured
                                    (r04_l, same box): <128,352> 57.9 -> 62.1 us, <64,192> 88.3 -> 93.6 us — slower: gfx950 counts loads with
                                    one in-order counter, so every weight-fragment wait of the panel in hand first waits for the prefetched
                                    HBM rows in front of it; the round trip moves from the prologue into the first k-step.  With the
                                    prefetch issued behind the panel's last fragment load instead (r04_l2, no spill): 55.2 -> 57.1 us and
                                    84.2 -> 89.1 us — the hardware's own workgroup dispatch (a new panel starts the moment a slot frees)
                                    overlaps better than the loop.  Not the default. */
#endif

// (Round 5 measured a start stagger of the first round of workgroups — blockIdx.x / 256 = 1, 2 sleeping 3.4 / 6.8 us (and 10 / 20 us)
//  so that the three workgroups of a CU run a third of a panel apart, on the theory that a launch of exactly three rounds keeps the
//  whole chip in one phase at a time: step 15.55 / 15.64 ms without, 15.54 / 15.64 and 15.63 / 15.71 with, enc_mlp_bwd 142.2 -> 144.1 us;
//  profiles/r05_d_prefetch_stagger_ab.txt.  Neutral to negative: not kept.)

namespace {

constexpr int MH = 2, NTH = 256;
template <int D> constexpr int LCd = (D <= 128 || HS_SWZ256) ? 64 : 64 + 8;      // 64-column chunk image row stride (D <= 128: unpadded + swizzled, see below)

// LDS layouts (round 4; bank model: scripts/micro/lds_banks.py, audit of the same scheme: scripts/micro/lds_audit_dec.py).  Round 3
// padded the panel rows (D + 16 at D <= 128: conflict-free 16-byte row fragments, 4-way 8-byte tile writes; D + 8 at D = 256:
// both 2-way) and the chunk images (64 + 8: both 2-way): 1.8-3.9 bank-conflict cycles per LDS instruction at the counters.
// Now no pad and an XOR swizzle of the 16-byte chunks:
//   * 128-byte rows (chunk images, the D = 64 panels): swz64, the map of the decoder backward kernels (fused_dec.hip);
//   * 256- / 512-byte rows (D = 128 / 256 panels): chunk c of row r at c ^ (r & 15) — the 16 lanes of a 16-byte row-fragment
//     group ({rows 0-3, 12-15 | chunk c} u {rows 4-11 | chunk c ^ 1}) then hit 16 different 16-byte bank slots.
// Row-contiguous 16-byte fills stay conflict-free, the 8-byte tile writes are 2-way (16 rows x 8 bytes against 32 banks).
// Measured (profiles/r04_j_swizzle_ab.txt, same box): the conflicts were not what these kernels wait for — <128,352> backward
// 139.4 -> 139.4 us, forward 55.2 -> 55.1, <64,192> forward 83.4 -> 82.1 — and at D = 256 the extra address arithmetic costs the
// backward kernel 6 more spilled registers (376 -> 424 us): D = 256 keeps the round-3 padded layout (no swizzle).
__host__ __device__ constexpr int swz64(int row) { return (((row >> 1) & 3) << 1) ^ (((row >> 3) & 1) * 5); }
template <int D> __host__ __device__ constexpr int swzp(int row) { return D == 64 ? swz64(row) : ((D == 128 || HS_SWZ256) ? (row & 15) : 0); }
template <int D> __host__ __device__ constexpr int swzc(int row) { return (D <= 128 || HS_SWZ256) ? swz64(row) : 0; }

// Geometry for model width D, padded hidden width HP (multiples of 64 / 32) and R_-row panels.
//   D = 128: 48-row panels, three 4-wave workgroups per CU (2304 = 3 x 768 workgroups at M = 110,592)
//   D = 256: 48-row panels, two workgroups per CU (the weight fragments of a 64-column hidden chunk are 64 / 96 registers)
//   D =  64: 64-row panels (8 lanes per row: the panel must be a multiple of 32 rows), three workgroups per CU — the MLP half
//            of a decoder block when the decoder runs layer at a time (216-token sequences, Huge)
template <int D, int HP, int R_ = (D == 64 ? 64 : 48)>
struct MG {
    static constexpr int R = R_;
    static constexpr int WPC = D <= 128 ? 3 : 2;            // workgroups per CU the kernels are compiled for
    // bf16 panel row stride (elements)
    static constexpr int LU = (D <= 128 || HS_SWZ256) ? D : D + 8;      // (see "LDS layouts" above; was D + 16 at D <= 128)
    static constexpr int LC = LCd<D>;
    static constexpr int LX = D + 4;            // fp32 staging row stride (floats)
    static constexpr int LG = HP + 8;           // gate image row stride
    static constexpr int NCH = (HP + 63) / 64;  // hidden chunks (the last one may be half full)
    static constexpr int NCC = D / 64;          // 64-column chunks of the model width
    static constexpr int KSD = D / 32;          // k-steps over the model width
    static constexpr int KSH = HP / 32;         // k-steps over the hidden width
    static constexpr int LPR = D / 8;           // lanes per row in the wide layout
    static constexpr int KA = KSH > 6 ? 6 : KSH, KB = KSH - KA;
    // forward: the panel's fp32 copy (residual) is kept in LDS at D = 128 only; wider panels re-read x1 (L2-hot) instead
    static constexpr bool KEEP_XR = (D == 128 && HS_MLP_XR_128) || (D == 64 && HS_MLP_XR_64);
    static constexpr int WPCF = D == 128 ? HS_MLP_WPCF_128 : (D == 64 ? HS_MLP_WPCF_64 : WPC);   // forward kernel
    static constexpr bool PERSIST = D <= 128 && HS_MLP_FWD_PERSIST;   // forward kernel walks panels with the next one's rows prefetched
    static constexpr int LDS_FWD_IMG = R * LU * 2 + 2 * R * LC * 2;
    static constexpr int LDS_FWD = (LDS_FWD_IMG > R * LX * 4 ? LDS_FWD_IMG : R * LX * 4) + (KEEP_XR ? R * LX * 4 : 0);
    static constexpr int LDS_BWD_IMG = 2 * R * LU * 2 + 3 * R * LC * 2;
    static constexpr int LDS_BWD = LDS_BWD_IMG + 2 * NCH * 64 * 4;       // + b1 | b3, zero-padded to whole chunks (round 6)
    // backward: the fp32 tile of du2 goes over the dY panel + chunk images when they are large enough (D = 128), else
    // over the whole arena (the U2 panel is dead by then)
    static constexpr bool XS_AT_DY = R * LU * 2 + 3 * R * LC * 2 >= R * LX * 4;
    static_assert(LDS_BWD_IMG >= R * LX * 4, "the arena must hold the fp32 staging tile");
    static_assert(2 * R * LU * 2 >= 2 * NTH * 8 * 4, "reduction scratch must fit in the two panels");
};

struct G8 { int lane, c16, g, wave, wm, wn; int fp, fc; };     // fp / fc: swizzle of row c16 (+ 16 k) in a panel / a chunk image
template <int D>
__device__ __forceinline__ G8 geo8() {
    G8 q;
    q.lane = threadIdx.x & 63; q.c16 = q.lane & 15; q.g = q.lane >> 4;
    q.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: SGPR, scalar branches
    q.wm = q.wave >> 1; q.wn = q.wave & 1;
    q.fp = swzp<D>(q.c16); q.fc = swzc<D>(q.c16);
    return q;
}
// 16-byte row fragment (MFMA operand) of a panel / of a chunk image: row mt*16 + c16, columns ks*32 + 8g .. + 7
template <int LU>
__device__ __forceinline__ bf16x8 pfrag(const bf16_t* img, int mt, int ks, const G8& q) {
    return *reinterpret_cast<const bf16x8*>(img + (mt * 16 + q.c16) * LU + (((ks * 4 + q.g) ^ q.fp) << 3));
}
template <int LC>
__device__ __forceinline__ bf16x8 cfrag(const bf16_t* img, int mt, int ks, const G8& q) {
    return *reinterpret_cast<const bf16x8*>(img + (mt * 16 + q.c16) * LC + (((ks * 4 + q.g) ^ q.fc) << 3));
}
// element offset in a chunk image of the 4 columns wave*16 + 4g .. of row mt*16 + c16 (a swapped-operand accumulator tile)
template <int LC>
__device__ __forceinline__ int ctile(int mt, const G8& q) {
    return (mt * 16 + q.c16) * LC + (((2 * q.wave + (q.g >> 1)) ^ q.fc) << 3) + (q.g & 1) * 4;
}

template <int KS>
struct Fr {
    bf16x8 b[KS][2];
    __device__ __forceinline__ void load(const bf16_t* W, int KS_total, int nt0, int ks0, int nt_total, const G8& q) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                b[ks][j] = (nt0 + j < nt_total && ks0 + ks < KS_total)
                               ? *reinterpret_cast<const bf16x8*>(W + (((size_t)(nt0 + j) * KS_total + ks0 + ks) * 64 + q.lane) * 8)
                               : zero8();
    }
};

// Column-split decomposition (forward kernel): the 4 waves split the COLUMNS of every product and each covers all
// 4 m-tiles of the 64-row panel, so no weight fragment is fetched by two waves.  (In the 2 x 2 split each wave
// pulled half of every weight matrix: 528 KB of L2 -> register traffic per 64-row panel, 0.9 GB per launch.)
template <int KS, int NJ>
struct FrN {
    bf16x8 b[KS][NJ];
    // CHK_N = false: the caller knows every n-tile is inside the matrix (with the wave index in nt0 the compiler cannot, and
    // guards every fragment with a zero fill and a branch — 500 of the 2,700 instructions of enc_mlp_fwd_kernel<128, 352>)
    template <bool CHK_N = true>
    __device__ __forceinline__ void load(const bf16_t* W, int KS_total, int nt0, int ks0, int nt_total, int lane) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                b[ks][j] = ((!CHK_N || nt0 + j < nt_total) && ks0 + ks < KS_total)
                               ? *reinterpret_cast<const bf16x8*>(W + (((size_t)(nt0 + j) * KS_total + ks0 + ks) * 64 + lane) * 8)
                               : zero8();
    }
};

__device__ __forceinline__ bf16x4 cvt4(f32x4 v) {
    bf16x4 r;
    r[0] = (bf16_t)v[0]; r[1] = (bf16_t)v[1]; r[2] = (bf16_t)v[2]; r[3] = (bf16_t)v[3];
    return r;
}

template <class T>
__device__ __forceinline__ const T* launder(const T* p) { asm volatile("" : "+s"(p)); return p; }

__device__ __forceinline__ void ld8(const float* p, float* o) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void st8(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <int LPR>
__device__ __forceinline__ float redrow(float v) {          // sum over the LPR adjacent lanes that own one row
    v = lanes_sum<LPR>(v);
    return v;
}

// b1 | b3 of the 4 hidden columns a lane owns in the swapped-operand products (zero past the hidden width: the padded W1 / W3
// rows are zero too).  Fetched one chunk ahead with the weight fragments: loaded where they are used, every chunk began with an
// exposed L2 round trip (s_waitcnt vmcnt(0) in front of the first MFMA).
struct Bias13 {
    f32x4 b1, b3;
    __device__ __forceinline__ void load(const float* w1b, const float* w3b, int col, int h) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { b1[r] = col + r < h ? w1b[col + r] : 0.f; b3[r] = col + r < h ? w3b[col + r] : 0.f; }
    }
};

struct EncMlpW {
    const float *n2w, *n2b, *w1b, *w3b, *w2b;
    const bf16_t *w1, *w3, *w2, *w2T, *w13T;
    int h;
};

struct EncMlpFwdArgs { const float* x1; const float* res2; float* x2; int M; EncMlpW w; const float* rowscale; };
// (Round 4 also had a "pair launch" form — blocks i of the two axis stacks as one launch of twice the workgroups, blockIdx.y
//  picking the argument set.  Measured and rejected: a pair launch takes exactly twice a single one, profiles/r04_p_pair_launch.txt;
//  removed from the library in round 5, the code is in the git history at e1f4f12.)

// x2 = x1 + rs * (b2 + (silu(u2 W1^T + b1) * (u2 W3^T + b3)) W2^T)  (+ res2), u2 = LN2(x1), one 64-row panel per workgroup.
// The W2 product is accumulated per 64-column hidden chunk (the gate lives in two 9-KB chunk images instead of a
// 46-KB panel image) and the residual is added from an L2-hot re-read in the store loop, so a workgroup needs 36 KB of
// LDS and ~150 registers: three workgroups per CU instead of two.
template <int D, int HPE>
__global__ __launch_bounds__(NTH, (MG<D, HPE>::WPCF)) void enc_mlp_fwd_kernel(EncMlpFwdArgs p) {
    using G = MG<D, HPE>;
    constexpr int R = G::R;
    constexpr int LU = G::LU, LC = G::LC, LX = G::LX, NCH = G::NCH, KSD = G::KSD, LPR = G::LPR;
    constexpr int MT4 = R / 16;                         // every wave covers all m-tiles of the panel
    constexpr int NJO = D / 64;                         // output n-tiles per wave (D / 16 over 4 waves)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* U2 = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Gc = U2 + R * LU;                           // two chunk images [64][LC]
    float* XS = reinterpret_cast<float*>(smem);         // fp32 store tile over U2 | Gc once the products are done
    float* XR = reinterpret_cast<float*>(smem + (G::LDS_FWD_IMG > R * LX * 4 ? G::LDS_FWD_IMG : R * LX * 4));   // the panel's x1 in fp32: the
                                                // residual (25 KB at D = 128; a re-read from L2 missed: 57 MB of extra fetches per launch)
    const G8 q0 = geo8<D>();
    const int nt_h = HPE / 16;
    const int c8_0 = (threadIdx.x % LPR) * 8;
    // HS_MLP_FWD_PERSIST (round-4 experiment, off): the units of this kernel run one after the other (DESIGN 7 budget table:
    // bytes 19 + MFMA 12 + VALU 18 + LDS 10 us = 59 us against 55 measured) because all 768 resident workgroups of a round
    // start together, wait for their rows together and compute together.  In the persistent form a workgroup walks panels
    // blockIdx.x, + gridDim.x, ... and fetches the NEXT panel's rows into registers (24 at D = 128) as soon as the LayerNorm has
    // consumed the current ones.  It measured 7 % slower (see the knob): the prefetched loads sit in front of the weight-fragment
    // loads in the in-order vmcnt queue.  D = 256 never has the registers (230 of 256 at two workgroups per CU).
    constexpr bool PERSIST = G::PERSIST;
    constexpr int NI = R * LPR / NTH;
    const int npanels = (p.M + R - 1) / R;
    float fa[NI][8];
    auto fetch_panel = [&](int panel) {                   // the whole panel in flight at once
        const int r0 = panel * R;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = (threadIdx.x + NTH * i) / LPR;
#pragma unroll
            for (int e = 0; e < 8; ++e) fa[i][e] = 0.f;
#ifndef HS_ABL_FWD_NOLOAD   /* timing ablation (variant builds only): the panel's rows are not fetched = the kernel without its exposed HBM round trip */
            if (panel < npanels && r0 + row < p.M) ld8(p.x1 + (size_t)(r0 + row) * D + c8_0, fa[i]);
#endif
        }
    };
    fetch_panel(blockIdx.x);

    PH_DECL
  int panel = blockIdx.x;
  do {
    const int row0 = panel * R;
    // (the weight images are the same for every panel: left alone, hipcc hoists their fragment loads out of the panel loop and
    //  spills them — 82 registers; laundering the pointers once per panel keeps the loads where they are used)
    // (... and the per-lane address pieces, which it would otherwise precompute for every access of the body and keep live)
    G8 q = q0;
    int c8 = c8_0;
    if constexpr (PERSIST) asm volatile("" : "+v"(q.lane), "+v"(q.c16), "+v"(q.g), "+v"(q.fp), "+v"(q.fc), "+v"(c8));
    EncMlpW w = p.w;
    if constexpr (PERSIST) { w.w1 = launder(p.w.w1); w.w3 = launder(p.w.w3); w.w2 = launder(p.w.w2); w.w1b = launder(p.w.w1b); w.w3b = launder(p.w.w3b); w.w2b = launder(p.w.w2b); }
    FrN<KSD, 1> f1, f3;                                 // this wave's n-tile of the current hidden chunk
    f1.template load<false>(w.w1, KSD, q.wave, 0, nt_h, q.lane);          // chunk 0: all four n-tiles exist (HPE >= 64)
    f3.template load<false>(w.w3, KSD, q.wave, 0, nt_h, q.lane);
    Bias13 bn;                                          // biases of the NEXT chunk (see Bias13)
    bn.load(w.w1b, w.w3b, q.wave * 16 + q.g * 4, w.h);
    __builtin_amdgcn_sched_barrier(0);           // keep the fetches here: hipcc otherwise sinks them next to the MFMAs
    {   // LayerNorm-2 in the wide layout (16 lanes per row)
        float gm[8], bt[8];
        ld8(w.n2w + c8, gm); ld8(w.n2b + c8, bt);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int pc = threadIdx.x + NTH * i, row = pc / LPR;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fa[i][e];
            if constexpr (G::KEEP_XR) st8(XR + row * LX + c8, f);
            const float mean = redrow<LPR>(f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7]) * (1.f / D);
            float v = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { f[e] -= mean; v += f[e] * f[e]; }
            const float rstd = rsqrtf(redrow<LPR>(v) * (1.f / D) + 1e-5f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = f[e] * rstd * gm[e] + bt[e];
            *reinterpret_cast<bf16x8*>(U2 + row * LU + (((c8 >> 3) ^ swzp<D>(row)) << 3)) = cvt8(f);
        }
    }
    FrN<2, NJO> f2;                                     // W2 fragments of the current chunk (k-steps 2c, 2c+1)
    f2.template load<false>(w.w2, G::KSH, q.wave * NJO, 0, D / 16, q.lane);
    if constexpr (PERSIST) fetch_panel(panel + (int)gridDim.x);          // next panel's rows: in flight under this panel's products
    lds_barrier();
    PH(0)
    // Every product runs with the MFMA operands swapped (weights as A, the panel as B): a lane then owns 4 consecutive
    // COLUMNS of one row (token c16, columns 4g .. 4g+3 of the n-tile), so the gate leaves as one 8-byte LDS write per m-tile
    // instead of four 2-byte ones and the result tile as 16-byte writes — same values, same accumulation order.
    f32x4 xr[MT4][NJO];                            // b2 + the W2 product, [m-tile][this wave's output n-tile]
#pragma unroll
    for (int j = 0; j < NJO; ++j) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(w.w2b + (q.wave * NJO + j) * 16 + q.g * 4);
#pragma unroll
        for (int mt = 0; mt < MT4; ++mt) xr[mt][j] = b;
    }
    PH(1)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int nt = c * 4 + q.wave;                                  // this wave's n-tile of the chunk
        const bool full = c * 4 + 3 < nt_h;                             // (compile-time per unrolled chunk) all four waves have columns
        const bool live = full || nt < nt_h;                            // last chunk: only waves 0, 1 have columns
        bf16_t* Gi = Gc + (c & 1) * R * LC;
        f32x4 h1[MT4], h3[MT4];
#pragma unroll
        for (int mt = 0; mt < MT4; ++mt) { h1[mt] = bn.b1; h3[mt] = bn.b3; }
        if (live) {
#pragma unroll
            for (int ks = 0; ks < KSD; ++ks)
#pragma unroll
                for (int mt = 0; mt < MT4; ++mt) {                      // one A fragment feeds both products
                    const bf16x8 a = pfrag<LU>(U2, mt, ks, q);
                    h1[mt] = mfma16(f1.b[ks][0], a, h1[mt]);
                    h3[mt] = mfma16(f3.b[ks][0], a, h3[mt]);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifndef HS_ABL_WSTREAM      /* timing ablation (variant builds only): every chunk computes with chunk 0's weight fragments = the L2 -> CU weight stream / NCH */
        if (c + 1 < NCH) {
            if ((c + 1) * 4 + 3 < nt_h) {
                f1.template load<false>(w.w1, KSD, (c + 1) * 4 + q.wave, 0, nt_h, q.lane);
                f3.template load<false>(w.w3, KSD, (c + 1) * 4 + q.wave, 0, nt_h, q.lane);
            } else {
                f1.load(w.w1, KSD, (c + 1) * 4 + q.wave, 0, nt_h, q.lane);
                f3.load(w.w3, KSD, (c + 1) * 4 + q.wave, 0, nt_h, q.lane);
            }
            bn.load(w.w1b, w.w3b, ((c + 1) * 4 + q.wave) * 16 + q.g * 4, w.h);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT4; ++mt) {
            f32x4 gv;
            // columns past the hidden width: the packed W1 / W3 rows and the biases read as zero -> h1 = h3 = 0 -> g = 0
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[r] = silu_nr(h1[mt][r]) * h3[mt][r];
            *reinterpret_cast<bf16x4*>(Gi + ctile<LC>(mt, q)) = cvt4(gv);
        }
        lds_barrier();                             // chunk image complete; the other image is free again after this barrier
        // x2 += g_c W2_c^T : this wave's NJO output n-tiles, all 4 m-tiles; the last chunk of 352 is half full
        constexpr bool kHalfLast = (HPE % 64) != 0;
        FrN<2, NJO> f2n;
#ifdef HS_ABL_WSTREAM
        f2n = f2;
#else
        if (c + 1 < NCH) f2n.template load<false>(w.w2, G::KSH, q.wave * NJO, 2 * (c + 1), D / 16, q.lane);
#endif
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (kHalfLast && c == NCH - 1 && ks == 1) continue;
#pragma unroll
            for (int mt = 0; mt < MT4; ++mt) {
                const bf16x8 a = cfrag<LC>(Gi, mt, ks, q);
#pragma unroll
                for (int j = 0; j < NJO; ++j) xr[mt][j] = mfma16(f2.b[ks][j], a, xr[mt][j]);
            }
        }
        if (c + 1 < NCH) f2 = f2n;
    }
    PH(2)
    lds_barrier();                             // all products done: reuse the panel + chunk images as the fp32 store tile
    PH(3)
#pragma unroll
    for (int mt = 0; mt < MT4; ++mt)
#pragma unroll
        for (int j = 0; j < NJO; ++j)
            *reinterpret_cast<f32x4*>(XS + (mt * 16 + q.c16) * LX + (q.wave * NJO + j) * 16 + q.g * 4) = xr[mt][j];
    lds_barrier();
#pragma unroll
    for (int i = 0; i < R * LPR / NTH; ++i) {
        const int pc = threadIdx.x + NTH * i, row = pc / LPR;
        if (row0 + row < p.M) {
            float f[8], t[8];
            ld8(XS + row * LX + c8, f);
            if constexpr (G::KEEP_XR) ld8(XR + row * LX + c8, t);     // residual from the panel's fp32 copy
            else ld8(p.x1 + (size_t)(row0 + row) * D + c8, t);         // ... or from an L2-hot re-read
            const float rs = p.rowscale ? p.rowscale[row0 + row] : 1.f;   // DropPath: x1 + scale * mlp(x1)
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], rs, t[e]);
            if (p.res2) {
                ld8(p.res2 + (size_t)(row0 + row) * D + c8, t);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += t[e];
            }
#ifdef HS_ABL_FWD_NOSTORE
            if (f[0] == 123.456f)
#endif
            st8(p.x2 + (size_t)(row0 + row) * D + c8, f);
        }
    }
    PH(4)
    if constexpr (PERSIST) lds_barrier();               // XS / XR are read: the next panel's LayerNorm may overwrite them
  } while (PERSIST && (panel += (int)gridDim.x) < npanels);
    PH_FLUSH(8)
}

// (Round 5 built a weights-resident persistent form of the D = 128 forward kernel — ONE 8-wave workgroup per CU holding all of
//  W1 | W3 | W2 in registers (96 + 44 per lane, 236 VGPRs, no scratch), 32-row panels, the next panel's rows fetched a whole panel
//  ahead as the only global loads of the loop; bit-identical x2.  Timing ablations had put the panel kernel at 54.8 us as shipped,
//  45.6 without its L2 weight stream, 45.4 without the row fetch, 36.5 without both (profiles/r05_p_mlp_fwd_ablations.txt).  The
//  resident kernel measured 53.0 us against 54.5, and the two-stream step 15.82-16.04 against 15.85-15.88 ms
//  (profiles/r05_q_mlp_fwd_resident_ab.txt): with every wave of the CU in the same phase the gate products re-read the panel from
//  LDS once per n-tile (400 KB of LDS reads per 32-row panel) and nothing overlaps the barriers.  Rejected; the code is in the git
//  history at 1e9414a.)

struct EncMlpBwdArgs {
    const float* x1; const float* dy; float* dx1; bf16_t* u2; bf16_t* dh13; bf16_t* g; int M; EncMlpW w;
    float* g_n2w; float* g_n2b;
    bf16_t* dyb; bf16_t* dx1b;       // bf16 copies of dY and dx1: the dO operands of dW2 / dWproj (and proj's data gradient)
    // DropPath (NULL = none): the MLP branch saw rs_mlp * dY, so dyb and everything derived from it carry the factor;
    // dx1b is what the attention branch sees, rs_attn * dx1.  dx1 itself (the residual path) is unscaled.
    const float* rs_mlp; const float* rs_attn;
    HsDet det;
    int plane_rows;                  // > 0: g / dh1 / dh3 as 64-column planes [plane][plane_rows][64] (include/hsimae_hip.h, hsimae_wgrad_task)
};

// LATE: planar operands stored behind the chunk's data-gradient products (HS_MLPB_LATE_STORES above) — a compile-time choice, so
// that the wait-count pass sees ONE order of loads and stores per instantiation
// The panel's rows as raw buffers (round 6): a wave-uniform base + a 32-bit lane byte offset per access, rows past M read as zeros and
// their stores are dropped by the bounds check — no 64-bit per-row-group addresses (24 registers at D = 256, where the prologue sits at
// the 256-register limit) and no exec-masked branches around the accesses.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t panel_rsrc(const void* base, long long rows, int row_bytes) {
    const long long b = rows > 0 ? (rows < 4096 ? rows : 4096) * row_bytes : 0;      // (a panel is <= 64 rows: the extent never nears 2^31)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)b, 0x00020000);
}
__device__ __forceinline__ void bld8(__amdgpu_buffer_rsrc_t r, unsigned bo, float* o) {
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)bo, 0, 0));
    const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)bo + 16, 0, 0));
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
typedef __attribute__((ext_vector_type(4))) unsigned int u32q;
__device__ __forceinline__ void bst8f(__amdgpu_buffer_rsrc_t r, unsigned bo, const float* v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32q, f32x4{v[0], v[1], v[2], v[3]}), r, (int)bo, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32q, f32x4{v[4], v[5], v[6], v[7]}), r, (int)bo + 16, 0, 0);
}
__device__ __forceinline__ void bst8h(__amdgpu_buffer_rsrc_t r, unsigned bo, bf16x8 v, int aux) {
    if (aux) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32q, v), r, (int)bo, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32q, v), r, (int)bo, 0, 0);
}

template <int D, int HPE, bool LATE_ST = false>
__global__ __launch_bounds__(NTH, (MG<D, HPE>::WPC)) void enc_mlp_bwd_kernel(EncMlpBwdArgs p) {
    using G = MG<D, HPE>;
    constexpr int R = G::R;
    constexpr int LU = G::LU, LC = G::LC, LX = G::LX, NCH = G::NCH, KSD = G::KSD, KSH = G::KSH, LPR = G::LPR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* U2 = reinterpret_cast<bf16_t*>(smem);
    bf16_t* DYb = U2 + R * LU;
    bf16_t* DH1 = DYb + R * LU;
    bf16_t* DH3 = DH1 + R * LC;
    bf16_t* Gc = DH3 + R * LC;
    float* XS = reinterpret_cast<float*>(G::XS_AT_DY ? reinterpret_cast<char*>(DYb) : smem);   // after the chunk loop: fp32 tile of du2
    const G8 q = geo8<D>();
    const int row0 = blockIdx.x * R;
    const EncMlpW& w = p.w;
    const int nt_h = HPE / 16;
    const int c8 = (threadIdx.x % LPR) * 8;
    constexpr bool late_stores = LATE_ST;

    PH_DECL
    // column-split decomposition (see FrN): wave w owns n-tile w of every 64-column hidden chunk and the output
    // n-tiles {2w, 2w+1} of the data gradient, each for all 4 m-tiles of the panel
    constexpr int MT4 = R / 16, NJO = D / 64;
    FrN<KSD, 1> f1, f3, f2;
    f1.template load<false>(w.w1, KSD, q.wave, 0, nt_h, q.lane);
    f3.template load<false>(w.w3, KSD, q.wave, 0, nt_h, q.lane);
    f2.template load<false>(w.w2T, KSD, q.wave, 0, nt_h, q.lane);
    // b1 | b3 staged once per panel in LDS (round 6): fetched from global per chunk they were vector-memory loads used at once — an
    // s_waitcnt vmcnt(0) at the head of every chunk, which on gfx950 also waits for the previous chunk's operand stores
    float* BL = reinterpret_cast<float*>(smem + G::LDS_BWD_IMG);        // [2][NCH * 64]
    {
        float gm[8], bt[8];
        ld8(w.n2w + c8, gm); ld8(w.n2b + c8, bt);
        // all of the panel's loads first: one exposed HBM round trip instead of one per row group
        constexpr int NI = R * LPR / NTH;
        float fa[NI][8], dya[NI][8];
        const long long left = (long long)p.M - row0;                         // rows of this panel inside the matrix
        const __amdgpu_buffer_rsrc_t xr = panel_rsrc(p.x1 + (size_t)row0 * D, left, D * 4), yr = panel_rsrc(p.dy + (size_t)row0 * D, left, D * 4);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const unsigned bo = (unsigned)(((threadIdx.x + NTH * i) / LPR) * D + c8) * 4u;
            bld8(xr, bo, fa[i]); bld8(yr, bo, dya[i]);
        }
        const __amdgpu_buffer_rsrc_t ur = panel_rsrc(p.u2 ? p.u2 + (size_t)row0 * D : nullptr, p.u2 ? left : 0, D * 2);
        const __amdgpu_buffer_rsrc_t ybr = panel_rsrc(p.u2 ? p.dyb + (size_t)row0 * D : nullptr, p.u2 ? left : 0, D * 2);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int pc = threadIdx.x + NTH * i, row = pc / LPR;
            const bool ok = row0 + row < p.M;
            float (&f)[8] = fa[i];
            float (&dyv)[8] = dya[i];
            const float mean = redrow<LPR>(f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7]) * (1.f / D);
            float v = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { f[e] -= mean; v += f[e] * f[e]; }
            const float rstd = rsqrtf(redrow<LPR>(v) * (1.f / D) + 1e-5f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = f[e] * rstd * gm[e] + bt[e];
            const bf16x8 ub = cvt8(f);
            *reinterpret_cast<bf16x8*>(U2 + row * LU + (((c8 >> 3) ^ swzp<D>(row)) << 3)) = ub;
            if (p.rs_mlp && ok) {
                const float rs = p.rs_mlp[row0 + row];
                float sv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) sv[e] = dyv[e] * rs;
#pragma unroll
                for (int e = 0; e < 8; ++e) dyv[e] = sv[e];
            }
            const bf16x8 dyb8 = cvt8(dyv);
            *reinterpret_cast<bf16x8*>(DYb + row * LU + (((c8 >> 3) ^ swzp<D>(row)) << 3)) = dyb8;
            {                                     // wgrad operands (NULL operands = an empty buffer: data path only — the caller takes
                                                  // its weight gradients elsewhere; rows past M: dropped by the bounds check)
                const unsigned bo = (unsigned)(row * D + c8) * 2u;
                bst8h(ur, bo, ub, HS_NT_B);
#ifndef HS_ABL_DW2          /* timing ablation (variant builds only): what "dW2 kept on chip" could save at most — see DESIGN 7.2 */
                bst8h(ybr, bo, dyb8, HS_NT_B);
#endif
            }
        }
    }
    {   // (staged behind the panel's prologue: in front of it the loads competed with the panel's rows for the last registers at D = 256)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        for (int i = tid; i < 2 * NCH * 64; i += NTH) {
            const int m = i / (NCH * 64), o = i % (NCH * 64);
            BL[i] = o < w.h ? (m == 0 ? w.w1b[o] : w.w3b[o]) : 0.f;
        }
    }
    lds_barrier();
    PH(0)
    f32x4 du2[MT4][NJO];
#pragma unroll
    for (int mt = 0; mt < MT4; ++mt)
#pragma unroll
        for (int j = 0; j < NJO; ++j) du2[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int nt = c * 4 + q.wave;
        const bool full = c * 4 + 3 < nt_h;                     // (compile-time per unrolled chunk)
        const bool live = full || nt < nt_h;
        {
            // operands swapped as in the forward kernel: a lane owns 4 consecutive hidden columns of one row
            f32x4 h1[MT4], h3[MT4], dg[MT4];
            {
                const int col = nt * 16 + q.g * 4;
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(BL + col), b3 = *reinterpret_cast<const f32x4*>(BL + NCH * 64 + col);
#pragma unroll
                for (int mt = 0; mt < MT4; ++mt) { h1[mt] = b1; h3[mt] = b3; dg[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            }
            if (live) {
#pragma unroll
                for (int ks = 0; ks < KSD; ++ks)
#pragma unroll
                    for (int mt = 0; mt < MT4; ++mt) {
                        const bf16x8 au = pfrag<LU>(U2, mt, ks, q);
                        const bf16x8 ad = pfrag<LU>(DYb, mt, ks, q);
                        h1[mt] = mfma16(f1.b[ks][0], au, h1[mt]);
                        h3[mt] = mfma16(f3.b[ks][0], au, h3[mt]);
                        dg[mt] = mfma16(f2.b[ks][0], ad, dg[mt]);
                    }
            }
#ifndef HS_ABL_WSTREAM
            if (c + 1 < NCH) {
                if ((c + 1) * 4 + 3 < nt_h) {
                    f1.template load<false>(w.w1, KSD, (c + 1) * 4 + q.wave, 0, nt_h, q.lane);
                    f3.template load<false>(w.w3, KSD, (c + 1) * 4 + q.wave, 0, nt_h, q.lane);
                    f2.template load<false>(w.w2T, KSD, (c + 1) * 4 + q.wave, 0, nt_h, q.lane);
                } else {
                    f1.load(w.w1, KSD, (c + 1) * 4 + q.wave, 0, nt_h, q.lane);
                    f3.load(w.w3, KSD, (c + 1) * 4 + q.wave, 0, nt_h, q.lane);
                    f2.load(w.w2T, KSD, (c + 1) * 4 + q.wave, 0, nt_h, q.lane);
                }
            }
#endif
#pragma unroll
            for (int mt = 0; mt < MT4; ++mt) {
                f32x4 gv, d1, d3;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a1 = h1[mt][r], a3 = h3[mt][r], dv = dg[mt][r];
                    const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-a1));
                    const float sl = a1 * sg;
                    // columns past the hidden width: packed W1 / W3 / W2^T rows and the biases are zero -> a1 = a3 = dv = 0 ->
                    // g = dh1 = dh3 = 0 without a select
                    gv[r] = sl * a3;
                    d1[r] = dv * a3 * sg * (1.f + a1 * (1.f - sg));
                    d3[r] = dv * sl;
                }
                const int o = ctile<LC>(mt, q);
                *reinterpret_cast<bf16x4*>(Gc + o) = cvt4(gv);
                *reinterpret_cast<bf16x4*>(DH1 + o) = cvt4(d1);
                *reinterpret_cast<bf16x4*>(DH3 + o) = cvt4(d3);
            }
        }
        lds_barrier();
        PH(1)
        // W1^T | W3^T fragments of this chunk's data gradient (this wave's output n-tiles): issued before the operand
        // stores so that their L2 round trip overlaps them
        // (D = 256: the W3^T fragments of the second k-step — 16 registers — are fetched in front of the first k-step's products
        //  instead: all 64 at once pushed this instantiation into scratch — 15 spilled registers in round 3, 3 now; moving the
        //  W1^T half as well makes hipcc spill 17)
        constexpr bool LATE = D >= 256;
        FrN<2, NJO> wa;
        FrN<1, NJO> wb0, wb1;
#ifdef HS_ABL_WSTREAM
        constexpr int cw = 0;
#else
        const int cw = c;
#endif
        wa.template load<false>(w.w13T, 2 * KSH, q.wave * NJO, 2 * cw, D / 16, q.lane);
        wb0.template load<false>(w.w13T, 2 * KSH, q.wave * NJO, KSH + 2 * cw, D / 16, q.lane);
        if constexpr (!LATE) wb1.template load<false>(w.w13T, 2 * KSH, q.wave * NJO, KSH + 2 * cw + 1, D / 16, q.lane);
        // weight-gradient operands of this chunk to HBM (row-contiguous 16-B stores), columns < 352 only
        if (!late_stores && p.dh13) {
            const int ncol = (c * 64 + 64 <= HPE) ? 64 : HPE - c * 64;       // 64 or 32
            for (int pc = threadIdx.x; pc < R * 8; pc += NTH) {
                const int row = pc >> 3, k8 = (pc & 7) * 8;
                if (row0 + row < p.M && k8 < ncol) {
                    const size_t gr = (size_t)(row0 + row);
                    if (p.plane_rows) {
                        // 64-column planes (round 5): this chunk IS plane c, so the panel's three pieces are row-contiguous blocks
                        // of 48 x 128 B instead of 128-byte pieces at pitches of 704 / 1408 B: enc_mlp_bwd 149.4 -> 138.7 us
                        // (profiles/r05_c_ablation_ab.txt), the same bytes at 5.2 instead of 4.1 TB/s in scripts/micro/hbm_stride.hip
                        const size_t po = ((size_t)c * p.plane_rows + gr) * 64 + k8;
                        HS_NT(HS_NT_A, reinterpret_cast<bf16x8*>(p.g + po), *reinterpret_cast<const bf16x8*>(Gc + row * LC + (((k8 >> 3) ^ swzc<D>(row)) << 3)));
                        HS_NT(HS_NT_A, reinterpret_cast<bf16x8*>(p.dh13 + po), *reinterpret_cast<const bf16x8*>(DH1 + row * LC + (((k8 >> 3) ^ swzc<D>(row)) << 3)));
                        HS_NT(HS_NT_A, reinterpret_cast<bf16x8*>(p.dh13 + (size_t)NCH * 64 * p.plane_rows + po), *reinterpret_cast<const bf16x8*>(DH3 + row * LC + (((k8 >> 3) ^ swzc<D>(row)) << 3)));
                        continue;
                    }
#ifndef HS_ABL_DW2
                    HS_NT(HS_NT_A, reinterpret_cast<bf16x8*>(p.g + gr * HPE + c * 64 + k8), *reinterpret_cast<const bf16x8*>(Gc + row * LC + (((k8 >> 3) ^ swzc<D>(row)) << 3)));
#endif
                    HS_NT(HS_NT_A, reinterpret_cast<bf16x8*>(p.dh13 + gr * 2 * HPE + c * 64 + k8), *reinterpret_cast<const bf16x8*>(DH1 + row * LC + (((k8 >> 3) ^ swzc<D>(row)) << 3)));
                    HS_NT(HS_NT_A, reinterpret_cast<bf16x8*>(p.dh13 + gr * 2 * HPE + HPE + c * 64 + k8), *reinterpret_cast<const bf16x8*>(DH3 + row * LC + (((k8 >> 3) ^ swzc<D>(row)) << 3)));
                }
            }
        }
        PH(2)
        // data gradient through W1 / W3:  du2 += dh1_c W1[c] + dh3_c W3[c]   (packed [N=128][K=704], W3 at k-step 11);
        // the last chunk of HPE = 352 is half full: its second k-step is past the hidden width
        constexpr bool kHalfLast = (HPE % 64) != 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (kHalfLast && c == NCH - 1 && ks == 1) continue;
            if constexpr (LATE) {
                if (ks == 0 && !(kHalfLast && c == NCH - 1)) wb1.template load<false>(w.w13T, 2 * KSH, q.wave * NJO, KSH + 2 * cw + 1, D / 16, q.lane);
            }
#pragma unroll
            for (int mt = 0; mt < MT4; ++mt) {
                const bf16x8 a1 = cfrag<LC>(DH1, mt, ks, q);
                const bf16x8 a3 = cfrag<LC>(DH3, mt, ks, q);
#pragma unroll
                for (int j = 0; j < NJO; ++j) {
                    du2[mt][j] = mfma16(wa.b[ks][j], a1, du2[mt][j]);
                    du2[mt][j] = mfma16(ks == 0 ? wb0.b[0][j] : wb1.b[0][j], a3, du2[mt][j]);
                }
            }
        }
        if (late_stores && p.dh13) {
            const int ncol = (c * 64 + 64 <= HPE) ? 64 : HPE - c * 64;       // 64 or 32
            const unsigned plane = (unsigned)p.plane_rows * 64u * 2u;          // bytes per 64-column plane
            const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(p.g, 0, (int)(NCH * plane), 0x00020000);
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(p.dh13, 0, (int)(2 * NCH * plane), 0x00020000);
            typedef __attribute__((ext_vector_type(4))) unsigned int u4;
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));              // opaque: the lane offsets below are derived here, per chunk, not hoisted and kept (spilled) across the kernel
#pragma unroll
            for (int it = 0; it < (R * 8 + NTH - 1) / NTH; ++it) {
                const int pc = tid + it * NTH, row = pc >> 3, k8 = (pc & 7) * 8;
                const bool ok = pc < R * 8 && row0 + row < p.M && k8 < ncol;
                const int lrow = ok ? row : 0;
                const int lo = lrow * LC + (((k8 >> 3) ^ swzc<D>(lrow)) << 3);
                const unsigned po = ok ? (unsigned)c * plane + ((unsigned)(row0 + row) * 64u + (unsigned)k8) * 2u : 0xffffff00u;    // past the extent: dropped
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, *reinterpret_cast<const bf16x8*>(Gc + lo)), rg, (int)po, 0, HS_NT_A ? 2 : 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, *reinterpret_cast<const bf16x8*>(DH1 + lo)), rd, (int)po, 0, HS_NT_A ? 2 : 0);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, *reinterpret_cast<const bf16x8*>(DH3 + lo)), rd, (int)(ok ? po + NCH * plane : po), 0, HS_NT_A ? 2 : 0);
            }
        }
        lds_barrier();
        PH(3)
    }
#pragma unroll
    for (int mt = 0; mt < MT4; ++mt)
#pragma unroll
        for (int j = 0; j < NJO; ++j)
            *reinterpret_cast<f32x4*>(XS + (mt * 16 + q.c16) * LX + (q.wave * NJO + j) * 16 + q.g * 4) = du2[mt][j];
    lds_barrier();
    float dgam[8], dbet[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { dgam[e] = 0.f; dbet[e] = 0.f; }
    {
        float gm[8];
        ld8(w.n2w + c8, gm);
        constexpr int NI = R * LPR / NTH;
        float xa[NI][8], dya[NI][8];
        const long long left = (long long)p.M - row0;
        const __amdgpu_buffer_rsrc_t xr = panel_rsrc(p.x1 + (size_t)row0 * D, left, D * 4), yr = panel_rsrc(p.dy + (size_t)row0 * D, left, D * 4);
        const __amdgpu_buffer_rsrc_t dxr = panel_rsrc(p.dx1 + (size_t)row0 * D, left, D * 4);
        const __amdgpu_buffer_rsrc_t dbr = panel_rsrc(p.dx1b ? p.dx1b + (size_t)row0 * D : nullptr, p.dx1b ? left : 0, D * 2);
#pragma unroll
        for (int i = 0; i < NI; ++i) {                 // L2-hot re-reads, all in flight at once
            const unsigned bo = (unsigned)(((threadIdx.x + NTH * i) / LPR) * D + c8) * 4u;
            bld8(xr, bo, xa[i]); bld8(yr, bo, dya[i]);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int pc = threadIdx.x + NTH * i, row = pc / LPR;
            const bool ok = row0 + row < p.M;
            float du[8], t[8];
            float (&xh)[8] = xa[i];
            float (&dyv)[8] = dya[i];
            ld8(XS + row * LX + c8, du);
            const float mean = redrow<LPR>(xh[0] + xh[1] + xh[2] + xh[3] + xh[4] + xh[5] + xh[6] + xh[7]) * (1.f / D);
            float v = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { xh[e] -= mean; v += xh[e] * xh[e]; }
            const float rstd = rsqrtf(redrow<LPR>(v) * (1.f / D) + 1e-5f);
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { xh[e] *= rstd; t[e] = du[e] * gm[e]; a += t[e]; b += t[e] * xh[e]; }
            a = redrow<LPR>(a) * (1.f / D); b = redrow<LPR>(b) * (1.f / D);
            if (ok) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o[e] = dyv[e] + rstd * (t[e] - a - xh[e] * b);
                    dgam[e] += du[e] * xh[e];
                    dbet[e] += du[e];
                }
                bst8f(dxr, (unsigned)(row * D + c8) * 4u, o);
                if (p.rs_attn) {
                    const float rs = p.rs_attn[row0 + row];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] *= rs;
                }
                bst8h(dbr, (unsigned)(row * D + c8) * 2u, cvt8(o), HS_NT_B);
            }
        }
    }
    // LayerNorm-2 parameter grads: reduce the 32 threads that share a column octet, one atomic per column
    lds_barrier();
    PH(4)
    float* red = reinterpret_cast<float*>(smem);            // [512][8] x 2 fits in the U2 + DYb panels
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[threadIdx.x * 8 + e] = dgam[e]; red[NTH * 8 + threadIdx.x * 8 + e] = dbet[e]; }
    lds_barrier();
    for (int t2 = threadIdx.x; t2 < 2 * D; t2 += NTH) {
        const int which = t2 / D, c = t2 % D, o8 = c >> 3, e = c & 7;
        float s = 0.f;
        for (int t = o8; t < NTH; t += LPR) s += red[which * NTH * 8 + t * 8 + e];
        hs_gadd(p.det, (which ? p.g_n2b : p.g_n2w) + c, s);
    }
    PH(5)
    PH_FLUSH(0)
}

}  // namespace

static int hp_of(int hidden) { return (hidden + 31) / 32 * 32; }

bool hs_enc_mlp_fused_supported(int d, int hidden) {
    // the kernels are templates on <D, HP>: Base (128, 344 -> 352), Large (256, 684 -> 704) and the decoder width
    // (64, 172 -> 192) are instantiated
    return (d == 128 && hp_of(hidden) == 352) || (d == 256 && hp_of(hidden) == 704) || (d == 64 && hp_of(hidden) == 192);
}

static EncMlpW mkw(const EncMlpPtrs& b) {
    EncMlpW w;
    w.n2w = b.n2w; w.n2b = b.n2b; w.w1b = b.w1b; w.w3b = b.w3b; w.w2b = b.w2b;
    w.w1 = b.w1; w.w3 = b.w3; w.w2 = b.w2; w.w2T = b.w2T; w.w13T = b.w13T; w.h = b.h;
    return w;
}

template <int D, int HP>
static void set_attrs() {
    static bool done = false;
    if (done) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(enc_mlp_fwd_kernel<D, HP>), hipFuncAttributeMaxDynamicSharedMemorySize, MG<D, HP>::LDS_FWD);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(enc_mlp_bwd_kernel<D, HP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, MG<D, HP>::LDS_BWD);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(enc_mlp_bwd_kernel<D, HP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, MG<D, HP>::LDS_BWD);
    done = true;
}

template <int D, int HP>
static int fwd_grid(int M) {
    const int panels = (M + MG<D, HP>::R - 1) / MG<D, HP>::R;
    if (!MG<D, HP>::PERSIST) return panels;
    static int slots = 0;                     // resident workgroups: WPCF per CU (HSIMAE_MLP_FWD_WGS overrides)
    if (!slots) slots = 256 * MG<D, HP>::WPCF;
    return panels < slots ? panels : slots;
}

static int launch_fwd(const EncMlpFwdArgs& a, int M, int d, hipStream_t s) {
    if (d == 128) {
        set_attrs<128, 352>();
        hipLaunchKernelGGL((enc_mlp_fwd_kernel<128, 352>), dim3(fwd_grid<128, 352>(M)), dim3(NTH), (MG<128, 352>::LDS_FWD), s, a);
    } else if (d == 256) {
        set_attrs<256, 704>();
        hipLaunchKernelGGL((enc_mlp_fwd_kernel<256, 704>), dim3(fwd_grid<256, 704>(M)), dim3(NTH), (MG<256, 704>::LDS_FWD), s, a);
    } else if (d == 64) {
        set_attrs<64, 192>();
        hipLaunchKernelGGL((enc_mlp_fwd_kernel<64, 192>), dim3(fwd_grid<64, 192>(M)), dim3(NTH), (MG<64, 192>::LDS_FWD), s, a);
    } else {
        return HS_EUNSUPPORTED;
    }
    return (int)hipGetLastError();
}

int hs_enc_mlp_fwd(const float* x1, const float* res2, float* x2, int M, int d, const EncMlpPtrs& b, hipStream_t s,
                   const float* rowscale) {
    if (M <= 0) return HS_OK;
    EncMlpFwdArgs a;
    a.x1 = x1; a.res2 = res2; a.x2 = x2; a.M = M; a.w = mkw(b); a.rowscale = rowscale;
    return launch_fwd(a, M, d, s);
}

template <int D, int HP>
static void launch_bwd_t(const EncMlpBwdArgs& a, int M, hipStream_t s) {
    constexpr int R = MG<D, HP>::R;
    set_attrs<D, HP>();
    // planar operands whose extent fits a raw buffer's 32-bit range: the late-store instantiation (HS_MLPB_LATE_STORES)
    // (D >= 256 only: at D = 128 the kernel sits on its HBM bytes either way — 136.4 / 137.2 us late against 136.9 / 135.4 early,
    //  Large 381.1 / 374.9 against 381.7 / 380.6 and 390.2 / 389.1 before this round: profiles/r06_f_enc_mlp_bwd_ab.txt)
    const bool late = HS_MLPB_LATE_STORES && D >= 256 && a.dh13 && a.plane_rows > 0 && (size_t)2 * MG<D, HP>::NCH * a.plane_rows * 128 < 0xffffff00u;
    if (late) hipLaunchKernelGGL((enc_mlp_bwd_kernel<D, HP, true>), dim3((M + R - 1) / R), dim3(NTH), (MG<D, HP>::LDS_BWD), s, a);
    else hipLaunchKernelGGL((enc_mlp_bwd_kernel<D, HP, false>), dim3((M + R - 1) / R), dim3(NTH), (MG<D, HP>::LDS_BWD), s, a);
}

static int launch_bwd(const EncMlpBwdArgs& a, int M, int d, hipStream_t s) {
    if (d == 128) launch_bwd_t<128, 352>(a, M, s);
    else if (d == 256) launch_bwd_t<256, 704>(a, M, s);
    else if (d == 64) launch_bwd_t<64, 192>(a, M, s);
    else return HS_EUNSUPPORTED;
    return (int)hipGetLastError();
}

static EncMlpBwdArgs mk_bwd(const float* x1, const float* dy, float* dx1, hs_bf16* u2, hs_bf16* dh13, hs_bf16* g, hs_bf16* dyb,
                            hs_bf16* dx1b, int M, const EncMlpPtrs& b, float* g_n2w, float* g_n2b, const float* rs_mlp,
                            const float* rs_attn, HsDet det, int plane_rows) {
    EncMlpBwdArgs a; a.x1 = x1; a.dy = dy; a.dx1 = dx1; a.u2 = u2; a.dh13 = dh13; a.g = g; a.M = M; a.w = mkw(b);
    a.g_n2w = g_n2w; a.g_n2b = g_n2b; a.dyb = dyb; a.dx1b = dx1b; a.rs_mlp = rs_mlp; a.rs_attn = rs_attn; a.det = det;
    a.plane_rows = plane_rows;
    return a;
}

int hs_enc_mlp_bwd(const float* x1, const float* dy, float* dx1, hs_bf16* u2, hs_bf16* dh13, hs_bf16* g, hs_bf16* dyb,
                   hs_bf16* dx1b, int M, int d, const EncMlpPtrs& b, float* g_n2w, float* g_n2b, hipStream_t s,
                   const float* rs_mlp, const float* rs_attn, HsDet det, int plane_rows) {
    if (M <= 0) return HS_OK;
    return launch_bwd(mk_bwd(x1, dy, dx1, u2, dh13, g, dyb, dx1b, M, b, g_n2w, g_n2b, rs_mlp, rs_attn, det, plane_rows), M, d, s);
}

HS_UNIT_VARIANT_BITS(fused_enc)
