// Weight gradients  dW[N][K] += sum_m dO[m][N] * A[m][K]  (contraction over the token rows) on MFMA.
//
// Both operands live token-major in HBM.  Each 64-row chunk is copied as-is (16-B pieces) into row-major
// LDS tiles and the MFMA fragments, which need 8 consecutive ROWS of one column per lane, are read with
// the gfx950 transpose read `ds_read_b64_tr_b16` (two per fragment) — no software transposition.
// One workgroup owns a 128x128 tile of dW for a slice of the rows (split-M), accumulates it in registers
// (16 accumulators per wave) and commits it with fp32 atomics; bias gradients (column sums of dO) ride
// along.  Up to 8 linears are batched per launch (one transformer block's q, k, v, proj, w1, w3, w2).
#include "common.h"
#include "kernels.h"

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

__device__ __forceinline__ void stage_rows(const void* src, bool f32, int ld, int ncols8, int M, int m0, int c0,
                                           bf16_t* img, int tid) {
    // img[m][f] = src[m0 + m][c0 + f] for m in [0,64), f in [0,128); zero outside the matrix
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx >> 4, c8 = (idx & 15) * 8;
        const int row = m0 + r, col = c0 + c8;
        bf16x8 v = zero8();
        if (row < M && col < ncols8) {
            if (f32) {
                const float* s = reinterpret_cast<const float*>(src) + (size_t)row * ld + col;
                const float4 x0 = *reinterpret_cast<const float4*>(s);
                const float4 x1 = *reinterpret_cast<const float4*>(s + 4);
                const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                v = cvt8(f);
            } else {
                v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(src) + (size_t)row * ld + col);
            }
        }
        *reinterpret_cast<bf16x8*>(img + r * TST + c8) = v;
    }
}

__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t dOt[MC * TST];
    __shared__ __attribute__((aligned(16))) bf16_t At[MC * TST];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // decode (task, n-slab, k-slab, m-split).  Workgroups go round-robin over the 8 XCDs (blockIdx % 8) and each XCD
    // has its own L2: all tiles of one row slice are placed on the same XCD, next to each other in launch order, so
    // the operand slabs they share (x-hat under q|k|v, u2 under w1|w3, dy under w2's k-slabs) are fetched from HBM
    // once and hit in L2 for the rest (measured before: 1.8x the algorithmic bytes).
    int tiles = 0;
    for (int i = 0; i < p.ntasks; ++i) tiles += ((p.t[i].N + 127) / 128) * ((p.t[i].K + 127) / 128);
    const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
    int w = li % tiles;
    const int ms = xcd + 8 * (li / tiles);
    if (ms >= p.msplit) return;
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

    for (int c = cbeg; c < cend; ++c) {
        const int m0 = c * MC;
        lds_barrier();
        stage_rows(t.dO, t.dO_f32 != 0, t.ldo, n8, p.M, m0, n0, dOt, tid);
        stage_rows(t.A, false, t.lda, k8, p.M, m0, k0, At, tid);
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
                if (n < t.N && k < t.K) atomicAdd(t.dW + (size_t)n * t.ldw + k, acc[i][j][r]);
            }
    if (want_bias && n0 + bcol < t.N) atomicAdd(t.db + n0 + bcol, bsum);
}

}  // namespace

int hs_wgrad(const WgradParams& p, hipStream_t s) {
    if (p.ntasks <= 0 || p.M <= 0) return HS_OK;
    if (p.ntasks > 8 || p.msplit < 1) return HS_EDIMS;
    int tiles = 0;
    for (int i = 0; i < p.ntasks; ++i) {
        if (p.t[i].ldo % 8 || p.t[i].lda % 8) return HS_EDIMS;
        tiles += ((p.t[i].N + 127) / 128) * ((p.t[i].K + 127) / 128);
    }
    hipLaunchKernelGGL(wgrad_kernel, dim3(8 * tiles * ((p.msplit + 7) / 8)), dim3(256), 0, s, p);
    return (int)hipGetLastError();
}
