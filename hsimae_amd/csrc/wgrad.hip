// Weight gradients  dW[N][K] += sum_m dO[m][N] * A[m][K]  (contraction over the token rows) on MFMA.
//
// Both operands live token-major in HBM.  Each 64-row chunk is copied as-is (16-B pieces) into row-major
// LDS tiles and the MFMA fragments, which need 8 consecutive ROWS of one column per lane, are read with
// the gfx950 transpose read `ds_read_b64_tr_b16` (two per fragment) — no software transposition.
// One workgroup owns a 128x128 tile of dW for a slice of the rows (split-M), accumulates it in registers
// (16 accumulators per wave) and commits it with fp32 atomics; bias gradients (column sums of dO) ride
// along.  Up to 8 linears are batched per launch (one transformer block's q, k, v, proj, w1, w3, w2).
// Launches whose matrices are all at least 256 wide (d = 256 / 512) use 256x256 tiles instead (WT<true> below).
#include "common.h"
#include "kernels.h"
#include <cstdlib>
#include <algorithm>

#ifndef HS_NT_W
#define HS_NT_W 0      /* cache policy of the LDS-DMA operand loads: 2 = nt (streaming) */
#endif

namespace {

constexpr int MC = 64;            // rows per chunk (2 k-steps)
constexpr int TST = 128 + 8;      // LDS tile row stride (elements)

typedef __attribute__((address_space(3))) bf16x4* lds_b64_ptr;

__device__ __forceinline__ bf16x8 frag_tr_rows(const bf16_t* tile, int mbase, int col0, int lane) {
    // 16x32 MFMA operand whose k index runs over tile ROWS: element j of lane l = tile[mbase + 8*(l>>4) + j][col0 + (l&15)]
    const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
    const bf16_t* a0 = tile + (mbase + 8 * g + q) * TST + col0 + 4 * pq;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_ptr)(a0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b64_ptr)(a0 + 4 * TST));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// One 64-row chunk of an operand slab, held in registers between its global load and its LDS store so that the
// loads of chunk c+1 are in flight while chunk c is being multiplied (the kernel was bound by one exposed HBM
// round trip per chunk).  fp32 operands (dx1, dy) stay raw and are rounded to bf16 on the way into LDS.
struct RowRegs {
    bf16x8 h[4];
    float4 f0[4], f1[4];
};

__device__ __forceinline__ void load_rows(const void* src, bool f32, int ld, int ncols8, int M, int m0, int c0, int tid,
                                          RowRegs& rr, const float* rowscale = nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx >> 4, c8 = (idx & 15) * 8;
        const int row = m0 + r, col = c0 + c8;
        const bool ok = row < M && col < ncols8;
        if (f32) {
            rr.f0[i] = make_float4(0.f, 0.f, 0.f, 0.f); rr.f1[i] = rr.f0[i];
            if (ok) {
                const float* s = reinterpret_cast<const float*>(src) + (size_t)row * ld + col;
                rr.f0[i] = *reinterpret_cast<const float4*>(s);
                rr.f1[i] = *reinterpret_cast<const float4*>(s + 4);
                if (rowscale) {                          // DropPath: this linear saw scale * dO
                    const float q = rowscale[row];
                    rr.f0[i].x *= q; rr.f0[i].y *= q; rr.f0[i].z *= q; rr.f0[i].w *= q;
                    rr.f1[i].x *= q; rr.f1[i].y *= q; rr.f1[i].z *= q; rr.f1[i].w *= q;
                }
            }
        } else {
            rr.h[i] = zero8();
            if (ok) rr.h[i] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(src) + (size_t)row * ld + col);
        }
    }
}

__device__ __forceinline__ void store_rows(const RowRegs& rr, bool f32, bf16_t* img, int tid) {
    // img[m][f] = slab[m0 + m][c0 + f] for m in [0,64), f in [0,128); zero outside the matrix
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx >> 4, c8 = (idx & 15) * 8;
        bf16x8 v;
        if (f32) {
            const float f[8] = {rr.f0[i].x, rr.f0[i].y, rr.f0[i].z, rr.f0[i].w, rr.f1[i].x, rr.f1[i].y, rr.f1[i].z, rr.f1[i].w};
            v = cvt8(f);
        } else {
            v = rr.h[i];
        }
        *reinterpret_cast<bf16x8*>(img + r * TST + c8) = v;
    }
}

// Workgroup -> (tile w, row slice ms).  Workgroups go round-robin over the 8 XCDs (blockIdx % 8) and each XCD has its
// own L2.  With whole groups of 8 row slices (narrow layers: 13 tiles x 16..40 slices) all tiles of one row slice sit on
// one XCD, next to each other in launch order, so the operand slabs they share (x-hat under q|k|v, u2 under w1|w3, dy
// under w2's k-slabs) are fetched from HBM once and hit in L2 for the rest (measured before: 1.8x the algorithmic bytes).
// With fewer slices (wide layers: 196 tiles x 2 slices at d = 512) that map would leave XCDs idle (slice ms only ever
// ran on XCD ms % 8: 1127 us per launch at d = 512): the tile list, which is ordered (task, n-slab, k-slab), is cut into 8
// contiguous parts instead, one per XCD, so tiles that share a dO slab still share an L2 and every XCD has work.
__device__ __forceinline__ bool decode_wg(const WgradParams& p, int tiles, int& w, int& ms) {
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    if ((p.msplit & 7) == 0) {
        w = li % tiles; ms = xcd + 8 * (li / tiles);
        return ms < p.msplit;
    }
    const int tpx = (tiles + 7) >> 3;
    w = xcd * tpx + li % tpx; ms = li / tpx;
    return w < tiles && ms < p.msplit;
}

__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t dOt[MC * TST];
    __shared__ __attribute__((aligned(16))) bf16_t At[MC * TST];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // decode (task, n-slab, k-slab, m-split)
    int tiles = 0;
    for (int i = 0; i < p.ntasks; ++i) tiles += ((p.t[i].N + 127) / 128) * ((p.t[i].K + 127) / 128);
    int w, ms;
    if (!decode_wg(p, tiles, w, ms)) return;
    const HsDet det{p.det_base, reinterpret_cast<long long*>(p.det_acc)};
    int ti = 0, ns = 0, ks = 0;
    for (; ti < p.ntasks; ++ti) {
        const int nsl = (p.t[ti].N + 127) / 128, ksl = (p.t[ti].K + 127) / 128;
        if (w < nsl * ksl) { ns = w / ksl; ks = w % ksl; break; }
        w -= nsl * ksl;
    }
    if (ti >= p.ntasks) return;
    const WgradTask t = p.t[ti];
    const int n0 = ns * 128, k0 = ks * 128;
    const int nchunks = (p.M + MC - 1) / MC;
    const int cpw = (nchunks + p.msplit - 1) / p.msplit;
    const int cbeg = ms * cpw, cend = min(nchunks, cbeg + cpw);
    const int n8 = (t.N + 7) & ~7, k8 = (t.K + 7) & ~7;
    const bool want_bias = (t.db != nullptr) && (ks == 0);

    f32x4 acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const int c16 = lane & 15, g = lane >> 4;
    const int bcol = tid & 127, bhalf = tid >> 7;

    const bool dof32 = t.dO_f32 != 0;
    RowRegs rd, ra;
    if (cbeg < cend) {
        load_rows(t.dO, dof32, t.ldo, n8, p.M, cbeg * MC, n0, tid, rd, t.dO_rowscale);
        load_rows(t.A, false, t.lda, k8, p.M, cbeg * MC, k0, tid, ra);
    }
    for (int c = cbeg; c < cend; ++c) {
        lds_barrier();                                   // everyone is done reading the previous chunk's tiles
        store_rows(rd, dof32, dOt, tid);
        store_rows(ra, false, At, tid);
        if (c + 1 < cend) {                              // next chunk's loads fly during this chunk's MFMAs
            load_rows(t.dO, dof32, t.ldo, n8, p.M, (c + 1) * MC, n0, tid, rd, t.dO_rowscale);
            load_rows(t.A, false, t.lda, k8, p.M, (c + 1) * MC, k0, tid, ra);
        }
        lds_barrier();
        if (want_bias) {
#pragma unroll 8
            for (int i = 0; i < MC / 2; ++i) bsum += bf2f(dOt[(bhalf * (MC / 2) + i) * TST + bcol]);
        }
#pragma unroll
        for (int kk = 0; kk < MC / 32; ++kk) {
            bf16x8 a[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = frag_tr_rows(dOt, kk * 32, (wave * 2 + i) * 16, lane);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16x8 b = frag_tr_rows(At, kk * 32, j * 16, lane);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = mfma16(a[i], b, acc[i][j]);
            }
        }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + (wave * 2 + i) * 16 + g * 4 + r;
                const int k = k0 + j * 16 + c16;
                if (n < t.N && k < t.K) hs_gadd(det, t.dW + (size_t)n * t.ldw + k, acc[i][j][r]);
            }
    if (want_bias && n0 + bcol < t.N) hs_gadd(det, t.db + n0 + bcol, bsum);
}


// ---------------------------------------------------------------------------------------------------------------
// All-bf16 operands: the same 128x128 tile, fed by buffer_load ... lds (16 B per lane straight into LDS, no VGPR
// staging) through a ring of DS stages of 32 rows.  The register-staged kernel above keeps one chunk in flight per
// workgroup and measured ~2.4 TB/s with HBM traffic already at the algorithmic minimum: each chunk exposed a full
// memory round trip.  Here DS-1 chunks per workgroup are always in flight (3 workgroups per CU x 2 x 16 KB).
//
// LDS stage = dO[32 rows][128 cols] | A[32 rows][128 cols], rows unpadded (256 B: the DMA writes 1 KB = 4 rows per
// wave instruction, lane i -> bytes [16 i, 16 i + 16)).  Bank conflicts of the transpose reads are avoided by
// permuting which 16-byte column chunk each lane FETCHES: slot s of row r holds column chunk s ^ swz(r).
constexpr int DC = 32;                     // rows per stage (one MFMA k-step)

typedef __attribute__((address_space(3))) void* lds_vptr;

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); }

// Tile geometry of the DMA kernel.
//   BIG = false: 128 x 128 dW tile, 4 waves (wave = 32 n x 128 k), stage = 2 x 32 rows x 256 B = 16 KB, up to 3 workgroups per CU.
//   BIG = true : 256 x 256 dW tile, 8 waves as 4 (n) x 2 (k) (wave = 64 n x 128 k, 128 accumulator registers), stage = 32 KB,
//                one workgroup per CU.  Every workgroup streams (TN + TK) operand columns per row: at d >= 256 the 128-tiles
//                fetched 2.8 GB (Large) / 5.5 GB (Huge) per launch through L2 / Infinity Cache at ~10 TB/s — that, not the MFMA
//                (26 % busy) or the LDS, was the bound — and the 256-tiles fetch half of it.
template <bool BIG>
struct WT {
    static constexpr int T = BIG ? 256 : 128;          // tile edge (columns of dO and of A per workgroup)
    static constexpr int NTHR = BIG ? 512 : 256;
    static constexpr int NF = BIG ? 4 : 2;             // n-fragments (16 dW rows each) per wave
    static constexpr int ROWB = T * 2;                 // bytes per LDS row
    static constexpr int STAGE_ELEMS = 2 * DC * T;     // bf16 elements per stage: dO[32][T] | A[32][T]
    static constexpr int RPI = 1024 / ROWB;            // rows per DMA wave-instruction (1 KB)
};

// The transpose reads of the DMA kernel are issued as inline asm: for a ds_read the compiler can see, it inserts
// `s_waitcnt vmcnt(0)` first (the LDS-DMA loads might alias it), which would drain the whole prefetch ring every
// chunk.  Arrival of a stage is tracked by hand instead (wait_vm + s_barrier), and so is lgkmcnt (wait_lds).
template <bool BIG>
__device__ __forceinline__ uint32_t frag_sw_addr(uint32_t tile_bytes, int col0, int lane) {
    // element j of lane l = tile[8 (l>>4) + j][col0 + (l&15)] of a swizzled [32][T] stage tile (see frag_tr_rows)
    const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
    const int row = 8 * g + q;
    const int chunk = ((col0 >> 3) + (pq >> 1)) ^ swz(row);       // the swizzle permutes 16-B chunks inside a 256-B group
    return tile_bytes + (uint32_t)(row * WT<BIG>::ROWB + chunk * 16 + (pq & 1) * 8);       // rows +4: same swizzle
}
struct Frag2 { bf16x4 lo, hi; };
template <bool BIG>
__device__ __forceinline__ void tr_read2(uint32_t addr, Frag2& f) {
    if constexpr (BIG)
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048"
                     : "=&v"(f.lo), "=&v"(f.hi) : "v"(addr) : "memory");
    else
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:1024"
                     : "=&v"(f.lo), "=&v"(f.hi) : "v"(addr) : "memory");
}
__device__ __forceinline__ bf16x8 join(const Frag2& f) { return __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7); }
// lgkmcnt(0), tied to the fragments so that nothing that uses them is scheduled above it
template <int NA>
__device__ __forceinline__ void wait_lds(Frag2 (&x)[NA], Frag2 (&y)[4]) {
    if constexpr (NA == 2)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(x[0].lo), "+v"(x[0].hi), "+v"(x[1].lo), "+v"(x[1].hi), "+v"(y[0].lo), "+v"(y[0].hi), "+v"(y[1].lo),
                       "+v"(y[1].hi), "+v"(y[2].lo), "+v"(y[2].hi), "+v"(y[3].lo), "+v"(y[3].hi) :: "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(x[0].lo), "+v"(x[0].hi), "+v"(x[1].lo), "+v"(x[1].hi), "+v"(x[2].lo), "+v"(x[2].hi), "+v"(x[3].lo),
                       "+v"(x[3].hi), "+v"(y[0].lo), "+v"(y[0].hi), "+v"(y[1].lo),
                       "+v"(y[1].hi), "+v"(y[2].lo), "+v"(y[2].hi), "+v"(y[3].lo), "+v"(y[3].hi) :: "memory");
}
__device__ __forceinline__ void wait_lds(Frag2 (&y)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(y[0].lo), "+v"(y[0].hi), "+v"(y[1].lo), "+v"(y[1].hi), "+v"(y[2].lo), "+v"(y[2].hi), "+v"(y[3].lo),
                   "+v"(y[3].hi) :: "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__host__ __device__ inline int wg_tiles(const WgradParams& p, int T) {
    int tiles = 0;
    for (int i = 0; i < p.ntasks; ++i) tiles += ((p.t[i].N + T - 1) / T) * ((p.t[i].K + T - 1) / T);
    return tiles;
}

template <int DS, bool BIG>                        // ring stages, tile geometry
__global__ __launch_bounds__(WT<BIG>::NTHR) void wgrad_dma_kernel(WgradParams p) {
    using G = WT<BIG>;
    constexpr int T = G::T, NF = G::NF, STAGE_ELEMS = G::STAGE_ELEMS;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = BIG ? wave >> 1 : wave, wk = BIG ? wave & 1 : 0;      // this wave's n-group (NF fragments) and 128-wide k half

    const int tiles = wg_tiles(p, T);
    int w, ms;
    if constexpr (BIG) {
        // one workgroup per CU: the launch is cut into 8 equal contiguous parts of the (row slice, tile) list, one per XCD,
        // so the tiles of a row slice (which share its operand slabs) sit on one L2 and every XCD has the same load
        const int total = tiles * p.msplit, per = (total + 7) >> 3;
        const int li = blockIdx.x >> 3;
        const int f = (blockIdx.x & 7) * per + li;
        if (li >= per || f >= total) return;
        ms = f / tiles; w = f - ms * tiles;
    } else {
        if (!decode_wg(p, tiles, w, ms)) return;
    }
    const HsDet det{p.det_base, reinterpret_cast<long long*>(p.det_acc)};
    const int w0 = w;                          // tile index of the launch (slab addressing)
    int ti = 0, ns = 0, ks = 0;
    for (; ti < p.ntasks; ++ti) {
        const int nsl = (p.t[ti].N + T - 1) / T, ksl = (p.t[ti].K + T - 1) / T;
        if (w < nsl * ksl) { ns = w / ksl; ks = w % ksl; break; }
        w -= nsl * ksl;
    }
    if (ti >= p.ntasks) return;
    const WgradTask t = p.t[ti];
    const int n0 = ns * T, k0 = ks * T;
    const int nchunks = (p.M + DC - 1) / DC;
    const int cpw = (nchunks + p.msplit - 1) / p.msplit;
    const int cbeg = ms * cpw, cend = min(nchunks, cbeg + cpw);
    const int nch = cend - cbeg;
    const bool want_bias = (t.db != nullptr) && (ks == 0) && (wk == 0);

    // Out-of-range rows (last chunk) come back as zeros from the buffer bounds check.  Columns past N / K of a slab
    // are whatever follows in the row: they only reach dW rows / columns that are never committed.
    // (Round 5 masked those lanes off instead of fetching them — a quarter of the last tile of dh1 / dh3 / g at hidden width 344,
    //  192 B of the 6.6 KB a row costs this launch: 126.4 -> 127.5 us at D = 128, 195.4 -> 203.1 us with the 256-tiles
    //  (profiles/r05_s_wgrad_edge_skip_ab.txt): the exec-mask bookkeeping around four DMA instructions per chunk costs more than the
    //  3 % of bytes it saves.  Not kept.)
    const uint32_t bytes_d = (uint32_t)min((int64_t)(t.dO_plane_rows ? t.dO_plane_rows : p.M) * t.ldo * 2, (int64_t)0xffffffffu);
    const uint32_t bytes_a = (uint32_t)min((int64_t)(t.A_plane_rows ? t.A_plane_rows : p.M) * t.lda * 2, (int64_t)0xffffffffu);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(t.dO), 0, bytes_d, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)t.A, 0, bytes_a, 0x00020000);
    // this wave's two DMA instructions per operand and chunk (1 KB = RPI rows each): rows RPI (2 wave + i) + lane / (64 / RPI),
    // 16-byte slot lane % (64 / RPI)
    // Planar operands (hsimae_wgrad_task: [ld / 64][M][64]): the lane's 16-byte piece of row r is at plane (col / 64), row pitch 64
    // elements — a per-lane constant plus the same per-chunk scalar offset for every lane, as in the row-major form.
    uint32_t vd[2], va[2];
    const int pd = t.dO_plane_rows ? 64 : t.ldo, pa = t.A_plane_rows ? 64 : t.lda;      // row pitch in elements
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        constexpr int LPR = 64 / G::RPI;                     // lanes per row
        const int row = G::RPI * (2 * wave + i) + lane / LPR;
        const int chunk = (lane % LPR) ^ swz(row);
        const int cd = n0 + 8 * chunk, ca = k0 + 8 * chunk;
        vd[i] = t.dO_plane_rows ? ((uint32_t)(cd >> 6) * (uint32_t)t.dO_plane_rows * 64u + (uint32_t)(row * 64 + (cd & 63))) * 2u
                                : (uint32_t)(row * t.ldo + cd) * 2u;
        va[i] = t.A_plane_rows ? ((uint32_t)(ca >> 6) * (uint32_t)t.A_plane_rows * 64u + (uint32_t)(row * 64 + (ca & 63))) * 2u
                               : (uint32_t)(row * t.lda + ca) * 2u;
    }
    auto issue = [&](int c, int stage) {
        bf16_t* st = smem + stage * STAGE_ELEMS;
        const uint32_t od = (uint32_t)c * DC * pd * 2u, oa = (uint32_t)c * DC * pa * 2u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_vptr)(st + (2 * wave + i) * 512), 16, vd[i], od, 0, HS_NT_W);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vptr)(st + DC * T + (2 * wave + i) * 512), 16, va[i], oa, 0, HS_NT_W);
    };

    f32x4 acc[NF][8], accb[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16_t)1.0f;

#pragma unroll
    for (int i = 0; i < DS - 1; ++i)
        if (i < nch) issue(cbeg + i, i);
    int stage = 0;
    for (int i = 0; i < nch; ++i) {
        // chunk i has landed once at most the DS-2 younger chunks (4 instructions each) are still outstanding
        if (i + DS - 2 < nch) wait_vm<(DS - 2) * 4>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();          // every wave's part of chunk i is in LDS; everyone is done with stage i-1
        if (i + DS - 1 < nch) issue(cbeg + i + DS - 1, stage == 0 ? DS - 1 : stage - 1);
        const uint32_t dOt = lds_base + (uint32_t)stage * (STAGE_ELEMS * 2), At = dOt + DC * G::ROWB;
        Frag2 fa[NF], fb0[4], fb1[4];
#pragma unroll
        for (int ii = 0; ii < NF; ++ii) tr_read2<BIG>(frag_sw_addr<BIG>(dOt, (wn * NF + ii) * 16, lane), fa[ii]);
#pragma unroll
        for (int j = 0; j < 4; ++j) tr_read2<BIG>(frag_sw_addr<BIG>(At, wk * 128 + j * 16, lane), fb0[j]);
        wait_lds<NF>(fa, fb0);
#pragma unroll
        for (int j = 0; j < 4; ++j) tr_read2<BIG>(frag_sw_addr<BIG>(At, wk * 128 + (4 + j) * 16, lane), fb1[j]);
        bf16x8 a[NF];
#pragma unroll
        for (int ii = 0; ii < NF; ++ii) a[ii] = join(fa[ii]);
        // column sums of dO as one more MFMA against ones: no extra LDS traffic.  On every workgroup, wanted or not: under a branch
        // hipcc moves the bias accumulators between AGPRs and VGPRs around every iteration (16 moves + waits per chunk)
#pragma unroll
        for (int ii = 0; ii < NF; ++ii) accb[ii] = mfma16(a[ii], ones, accb[ii]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8 b = join(fb0[j]);
#pragma unroll
            for (int ii = 0; ii < NF; ++ii) acc[ii][j] = mfma16(a[ii], b, acc[ii][j]);
        }
        wait_lds(fb1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bf16x8 b = join(fb1[j]);
#pragma unroll
            for (int ii = 0; ii < NF; ++ii) acc[ii][4 + j] = mfma16(a[ii], b, acc[ii][4 + j]);
        }
        stage = stage == DS - 1 ? 0 : stage + 1;
    }

    const int c16 = lane & 15, g = lane >> 4;
    if (BIG && p.slab) {
        // this workgroup's partial tile as NF * 8 * 4 coalesced 2-KB rows of its slab: [row slice ms][tile][slot][thread];
        // wgrad_slab_reduce_kernel sums the row slices and scatters with the index map of the atomic commit below
        float* sl = p.slab + ((size_t)(ms * tiles + (w0)) * (NF * 32)) * G::NTHR + tid;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) sl[(size_t)((i * 8 + j) * 4 + r) * G::NTHR] = acc[i][j][r];
    } else {
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + (wn * NF + i) * 16 + g * 4 + r;
                const int k = k0 + wk * 128 + j * 16 + c16;
                if (n < t.N && k < t.K) hs_gadd(det, t.dW + (size_t)n * t.ldw + k, acc[i][j][r]);
            }
    }
    if (want_bias && c16 == 0) {
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + (wn * NF + i) * 16 + g * 4 + r;
                if (n < t.N) hs_gadd(det, t.db + n, accb[i][r]);
            }
    }
}

// Sums the row slices' partial 256 x 256 tiles ([row slice][tile][slot][thread], wgrad_dma_kernel<DS, true> with p.slab) in a
// fixed order and adds them into dW: workgroup = (tile, slot), thread = the committing thread of the main kernel.
__global__ __launch_bounds__(512) void wgrad_slab_reduce_kernel(WgradParams p) {
    constexpr int T = 256, NF = 4, NTHR = 512;
    const int tiles = wg_tiles(p, T);
    const int slot = blockIdx.x % (NF * 32);
    int w = blockIdx.x / (NF * 32);
    const int w0 = w;
    int ti = 0, ns = 0, ks = 0;
    for (; ti < p.ntasks; ++ti) {
        const int nsl = (p.t[ti].N + T - 1) / T, ksl = (p.t[ti].K + T - 1) / T;
        if (w < nsl * ksl) { ns = w / ksl; ks = w % ksl; break; }
        w -= nsl * ksl;
    }
    if (ti >= p.ntasks) return;
    const WgradTask& t = p.t[ti];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, g = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;
    const int r = slot & 3, j = (slot >> 2) & 7, i = slot >> 5;
    const int n = ns * T + (wn * NF + i) * 16 + g * 4 + r, k = ks * T + wk * 128 + j * 16 + c16;
    if (n >= t.N || k >= t.K) return;
    const float* src = p.slab + ((size_t)w0 * (NF * 32) + slot) * NTHR + tid;
    const size_t stride = (size_t)tiles * (NF * 32) * NTHR;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int m = 0;
    for (; m + 4 <= p.msplit; m += 4) {
        a0 += src[(size_t)m * stride]; a1 += src[(size_t)(m + 1) * stride];
        a2 += src[(size_t)(m + 2) * stride]; a3 += src[(size_t)(m + 3) * stride];
    }
    for (; m < p.msplit; ++m) a0 += src[(size_t)m * stride];
    t.dW[(size_t)n * t.ldw + k] += (a0 + a1) + (a2 + a3);
}


// (Round 4 also built rectangular one-workgroup-per-linear tiles here — q | k | v as one 384 x 128 tile, 8 waves, one workgroup per
//  CU: a third fewer bytes through the CUs, measured 135-140 us per launch against 122 for the 128 x 128 tiles below, step + 0.25 ms,
//  profiles/r04_x_wgrad_rect.txt: three independent 4-wave workgroups per CU keep the memory system busier than one 8-wave
//  workgroup in barrier lock-step.  Rejected; removed from the library in round 5, the code is in the git history at e1f4f12.)

}  // namespace

int hs_wgrad(const WgradParams& p, hipStream_t s) {
    if (p.ntasks <= 0 || p.M <= 0) return HS_OK;
    if (p.ntasks > 16 || p.msplit < 1) return HS_EDIMS;
    bool any_planar = false;
    for (int i = 0; i < p.ntasks; ++i) {
        if (p.t[i].ldo % 8 || p.t[i].lda % 8) return HS_EDIMS;
        // planar operands: padded width a multiple of 64, whole 32-row DMA chunks (a row past M would be the next plane's row 0)
        if (p.t[i].dO_plane_rows && (p.t[i].ldo % 64 || p.t[i].dO_f32 || p.M % DC || p.t[i].dO_plane_rows < p.M)) return HS_EDIMS;
        if (p.t[i].A_plane_rows && (p.t[i].lda % 64 || p.M % DC || p.t[i].A_plane_rows < p.M)) return HS_EDIMS;
        // 32-bit byte offsets: an edge tile's lanes address plane index (n0 + 255) / 64, i.e. up to 256 columns past ld, and rely on
        // the buffer bounds check to return zeros there — which holds only while that offset does not wrap
        if ((int64_t)std::max(p.t[i].dO_plane_rows, p.t[i].A_plane_rows) * (std::max(p.t[i].ldo, p.t[i].lda) + 256) * 2 >= (1ll << 32)) return HS_EDIMS;
        any_planar = any_planar || p.t[i].dO_plane_rows || p.t[i].A_plane_rows;
    }
    const int tiles = wg_tiles(p, 128);
    // all operands bf16 and addressable with 32-bit buffer offsets -> LDS-DMA kernels; fp32 dO (the layer-at-a-time schedule) or larger
    // extents -> the register-staged wgrad_kernel.  (The switches that forced the older choice — HSIMAE_WGRAD_DMA, _WGRAD_BIG, _WGRAD_DS —
    // and the ring depths nobody ran went in round 6.)
    static bool attrs = false;
    if (!attrs) {
        attrs = true;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_dma_kernel<6, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * WT<false>::STAGE_ELEMS * 2);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_dma_kernel<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WT<true>::STAGE_ELEMS * 2);
    }
    bool dma = true, big = true;
    for (int i = 0; i < p.ntasks; ++i) {
        const WgradTask& t = p.t[i];
        if (t.dO_f32 || (int64_t)(p.M + DC) * t.ldo * 2 >= (1ll << 32) || (int64_t)(p.M + DC) * t.lda * 2 >= (1ll << 32)) dma = false;
        if ((reinterpret_cast<uintptr_t>(t.dO) | reinterpret_cast<uintptr_t>(t.A)) & 15) dma = false;
        if (t.N < 256 || t.K < 256) big = false;            // wide layers only: a 256-tile of a 128-wide matrix is half padding
    }
    if (any_planar && !dma) return HS_EUNSUPPORTED;     // only the LDS-DMA kernels address planes
    if (dma && big) {
        // 256 x 256 tiles, one workgroup per CU: the row split that fills the chip once (p.msplit is sized for 128-tiles)
        WgradParams q = p;
        const int t256 = wg_tiles(p, 256), nchunks = (p.M + DC - 1) / DC;
        static int wgs = 0;
        if (!wgs) wgs = 256;
        q.msplit = std::max(1, std::min(wgs / std::max(1, t256), nchunks));
        const int total = t256 * q.msplit;
        const dim3 grid(8 * ((total + 7) / 8));
        // slab + reduce pays where many row slices meet in one tile (Large: 13 tiles x 19 slices, 264 -> 196 us of atomics out,
        // 35.70 -> 35.27 ms per step); with few slices per tile (Huge: 52 tiles x 4) the atomics are uncontended and the extra
        // 64 MB pass costs more than it saves (48.86 -> 49.06 ms): atomics there
        if (total > 256 || q.msplit < 8) q.slab = nullptr;
        hipLaunchKernelGGL((wgrad_dma_kernel<4, true>), grid, dim3(512), 4 * WT<true>::STAGE_ELEMS * 2, s, q);
        if (q.slab) hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3(t256 * 128), dim3(512), 0, s, q);
        return (int)hipGetLastError();
    }
    const dim3 grid((p.msplit & 7) == 0 ? 8 * tiles * (p.msplit / 8) : 8 * ((tiles + 7) / 8) * p.msplit);      // decode_wg
    if (dma) {
        // ring depth: a launch of at most one workgroup per CU has the LDS to itself (6 stages = 96 KB); larger
        // launches keep 3 stages so that three workgroups fit a CU
        constexpr int SB = WT<false>::STAGE_ELEMS * 2;
        if (grid.x > 256) hipLaunchKernelGGL((wgrad_dma_kernel<3, false>), grid, dim3(256), 3 * SB, s, p);
        else hipLaunchKernelGGL((wgrad_dma_kernel<6, false>), grid, dim3(256), 6 * SB, s, p);
    } else {
        hipLaunchKernelGGL(wgrad_kernel, grid, dim3(256), 0, s, p);
    }
    return (int)hipGetLastError();
}

HS_UNIT_VARIANT_BITS(wgrad)
