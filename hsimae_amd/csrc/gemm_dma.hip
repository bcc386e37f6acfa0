// Persistent, LDS-DMA-fed row-panel kernel for the largest encoder GEMM of the backward:
//     dx = dres + LayerNormBackward( dqkv[M,384] * [Wq;Wk;Wv] ; x, gamma ),  dgamma / dbeta     (d = 128)
// (autograd of Models.py:304 `x + attn(norm1(x))` w.r.t. x; same arithmetic as gemm_kernel<A_BF16, E_LN_BWD>).
//
// The generic row-panel GEMM spends 49 % of a workgroup's life loading its A panel and 48 % in the epilogue, 3 % in
// MFMAs (scripts/phase_timing.py), at 3.3 TB/s.  Here 512 persistent workgroups walk the rows in 32-row chunks:
//   * the chunk's 32 x 384 bf16 operand comes in through `buffer_load ... lds` into one of two LDS stages while the
//     previous chunk is being multiplied and normalised (the pattern that took the weight-gradient kernel from 2.3 to
//     4.5 TB/s); x and dres rows of the next chunk are prefetched into registers at the same time;
//   * the packed weight fragments (24 per wave) are loaded ONCE per workgroup and stay in registers;
//   * dgamma / dbeta accumulate in registers over all chunks: one atomic flush per workgroup.
// The two stages and the fp32 exchange tile are separate __shared__ objects on purpose: hipcc then knows that reading
// one does not alias the DMA in flight into another and does not drain vmcnt in front of every LDS access.
#include "common.h"
#include "kernels.h"
#include <cstdlib>

namespace {

constexpr int DC = 32;                 // rows per chunk (two m-tiles)
constexpr int KA = 384, KSA = KA / 32; // contraction length (3 d) and its k-steps
constexpr int TSX = 128;               // fp32 exchange tile row stride (unpadded: the three LDS objects are exactly 64 KB static)

typedef __attribute__((address_space(3))) void* lds_vptr;

// 16-byte slot `s` of row r holds global column chunk s ^ (r & 15): the 16 rows of an m-tile then hit 16 different
// bank groups when a wave reads one k-group of all of them (row pitch 768 B = 0 mod 256 B)
__device__ __forceinline__ int swz(int row) { return row & 15; }

__global__ __launch_bounds__(256, 2) void lnbwd_dma_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t stage0[DC * KA];
    __shared__ __attribute__((aligned(16))) bf16_t stage1[DC * KA];
    __shared__ __attribute__((aligned(16))) float T1[DC * TSX];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int nchunks = (p.M + DC - 1) / DC;

    // weight fragments of this wave's two n-tiles, all 12 k-steps: resident for the whole kernel
    bf16x8 bw[KSA][2];
#pragma unroll
    for (int ks = 0; ks < KSA; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            bw[ks][j] = *reinterpret_cast<const bf16x8*>(p.W + (((size_t)(wave * 2 + j) * KSA + ks) * 64 + lane) * 8);

    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.A), 0, (uint32_t)min((int64_t)p.M * p.lda * 2, (int64_t)0xffffffffu), 0x00020000);
    // this wave's 6 DMA instructions per chunk: LDS bytes [1024 q, 1024 q + 1024), q = 6 wave + i
    uint32_t voff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int off = (wave * 6 + i) * 1024 + lane * 16;
        const int row = off / (KA * 2), slot = (off % (KA * 2)) >> 4;
        voff[i] = (uint32_t)(row * p.lda * 2 + ((slot ^ swz(row)) << 4));
    }
    auto issue = [&](int chunk, bf16_t* st) {
        const uint32_t so = (uint32_t)chunk * DC * p.lda * 2u;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_vptr)(st + (wave * 6 + i) * 512), 16, voff[i], so, 0, 0);
    };

    // wide layout of the epilogue: thread -> (row = tid >> 4 (+16), columns 8 (tid & 15) ..)
    const int c8 = (tid & 15) * 8;
    float gm[8], dgam[8], dbet[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gm[e] = p.gamma[c8 + e]; dgam[e] = 0.f; dbet[e] = 0.f; }
    float xr[2][8], rs[2][8];           // x and dres rows of the chunk being processed
    auto fetch_rows = [&](int chunk, float (&xo)[2][8], float (&ro)[2][8]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = chunk * DC + (tid >> 4) + 16 * i;
#pragma unroll
            for (int e = 0; e < 8; ++e) { xo[i][e] = 0.f; ro[i][e] = 0.f; }
            if (row < p.M) {
                const float* xp = p.lnx + (size_t)row * p.ldr + c8;
                const float* rp = p.res + (size_t)row * p.ldr + c8;
                const float4 a0 = *reinterpret_cast<const float4*>(xp), a1 = *reinterpret_cast<const float4*>(xp + 4);
                const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                xo[i][0] = a0.x; xo[i][1] = a0.y; xo[i][2] = a0.z; xo[i][3] = a0.w; xo[i][4] = a1.x; xo[i][5] = a1.y; xo[i][6] = a1.z; xo[i][7] = a1.w;
                ro[i][0] = r0.x; ro[i][1] = r0.y; ro[i][2] = r0.z; ro[i][3] = r0.w; ro[i][4] = r1.x; ro[i][5] = r1.y; ro[i][6] = r1.z; ro[i][7] = r1.w;
            }
        }
    };

    // one chunk: `cur` holds it (DMA complete and visible), `nxt` receives the following one meanwhile
    auto process = [&](int chunk, int next_chunk, const bf16_t* cur, bf16_t* nxt) {
        float xn[2][8], rn[2][8];
        if (next_chunk < nchunks) { issue(next_chunk, nxt); fetch_rows(next_chunk, xn, rn); }
        f32x4 acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) { acc[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mt][1] = acc[mt][0]; }
#pragma unroll
        for (int ks = 0; ks < KSA; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = mt * 16 + c16;
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(cur + row * KA + (((ks * 4 + g) ^ swz(row)) << 3));
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[mt][j] = mfma16(a, bw[ks][j], acc[mt][j]);
            }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) T1[(mt * 16 + 4 * g + r) * TSX + (wave * 2 + j) * 16 + c16] = acc[mt][j][r];
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rl = (tid >> 4) + 16 * i, row = chunk * DC + rl;
            const float4 t0 = *reinterpret_cast<const float4*>(T1 + rl * TSX + c8);
            const float4 t1 = *reinterpret_cast<const float4*>(T1 + rl * TSX + c8 + 4);
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
        if (next_chunk < nchunks) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) { xr[i][e] = xn[i][e]; rs[i][e] = rn[i][e]; }
        }
        // this wave's part of the next chunk has landed; the barrier publishes every wave's part and retires T1 / `cur`
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
    };

    int chunk = blockIdx.x;
    if (chunk < nchunks) {
        issue(chunk, stage0);
        fetch_rows(chunk, xr, rs);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
    }
    while (chunk < nchunks) {
        process(chunk, chunk + (int)gridDim.x, stage0, stage1);
        chunk += gridDim.x;
        if (chunk >= nchunks) break;
        process(chunk, chunk + (int)gridDim.x, stage1, stage0);
        chunk += gridDim.x;
    }

    // dgamma / dbeta: 16 threads per column octet -> LDS, one atomic per column and workgroup
    float* red = T1;                                   // [2][256][8] floats = 16 KB = DC * TSX * 4
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 8 + e] = dgam[e]; red[2048 + tid * 8 + e] = dbet[e]; }
    lds_barrier();
    {
        const int which = tid >> 7, c = tid & 127, o8 = c >> 3, e = c & 7;
        float sacc = 0.f;
        for (int t2 = o8; t2 < 256; t2 += 16) sacc += red[which * 2048 + t2 * 8 + e];
        hs_gadd(HsDet{p.det_base, reinterpret_cast<long long*>(p.det_acc)}, (which ? p.dbeta : p.dgamma) + c, sacc);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent LayerNorm-1 + q|k|v kernel of the encoder forward (d = 128):
//     u = LN(x) (bf16, kept for the weight gradients),  qkv = u * [Wq;Wk;Wv]^T + b   (Models.py:194-208, 304)
// Same arithmetic as gemm_kernel<A_F32_LN, E_BF16>, which spends 39 % of a workgroup's life on the panel load and
// 40 % in the epilogue.  Here 512 workgroups walk 32-row chunks: the next chunks' x rows are already in flight in
// registers, the 24 weight fragments per wave stay in registers, and the 32 x 384 result leaves through an LDS tile as
// whole 768-byte rows.
constexpr int LQU = 128 + 8;           // LN-output tile row stride (elements)
constexpr int LQO = 384 + 8;           // result tile row stride (elements)

__global__ __launch_bounds__(256, 2) void lnqkv_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t Ut[DC * LQU];
    __shared__ __attribute__((aligned(16))) bf16_t Ot[DC * LQO];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int nchunks = (p.M + DC - 1) / DC;
    const float* X = reinterpret_cast<const float*>(p.A);
    bf16_t* out = reinterpret_cast<bf16_t*>(p.out);

    bf16x8 bw[4][6];                   // this wave's 6 n-tiles (96 columns), 4 k-steps
    float bias[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        bias[j] = p.bias[(wave * 6 + j) * 16 + c16];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            bw[ks][j] = *reinterpret_cast<const bf16x8*>(p.W + (((size_t)(wave * 6 + j) * 4 + ks) * 64 + lane) * 8);
    }
    const int c8 = (tid & 15) * 8;
    float gm[8], bt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gm[e] = p.gamma[c8 + e]; bt[e] = p.beta[c8 + e]; }

    auto fetch = [&](int chunk, float (&xo)[2][8]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = chunk * DC + (tid >> 4) + 16 * i;
#pragma unroll
            for (int e = 0; e < 8; ++e) xo[i][e] = 0.f;
            if (chunk < nchunks && row < p.M) {
                const float* xp = X + (size_t)row * p.lda + c8;
                const float4 a0 = *reinterpret_cast<const float4*>(xp), a1 = *reinterpret_cast<const float4*>(xp + 4);
                xo[i][0] = a0.x; xo[i][1] = a0.y; xo[i][2] = a0.z; xo[i][3] = a0.w; xo[i][4] = a1.x; xo[i][5] = a1.y; xo[i][6] = a1.z; xo[i][7] = a1.w;
            }
        }
    };
    float xa[2][8], xb[2][8];          // two chunks ahead
    fetch(blockIdx.x, xa);
    fetch(blockIdx.x + gridDim.x, xb);
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        // LayerNorm of the chunk's rows (16 lanes per row) -> bf16 tile + HBM copy
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rl = (tid >> 4) + 16 * i, row = chunk * DC + rl;
            float sm = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sm += xa[i][e];
            sm = lanes_sum<16>(sm);
            const float mean = sm * (1.f / 128.f);
            float q = 0.f, f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { f[e] = xa[i][e] - mean; q += f[e] * f[e]; }
            q = lanes_sum<16>(q);
            const float rstd = rsqrtf(q * (1.f / 128.f) + 1e-5f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = f[e] * rstd * gm[e] + bt[e];
            const bf16x8 u8 = cvt8(f);
            *reinterpret_cast<bf16x8*>(Ut + rl * LQU + c8) = u8;
            if (row < p.M && p.u_out) *reinterpret_cast<bf16x8*>(p.u_out + (size_t)row * p.ldu + c8) = u8;
        }
        // rotate the prefetch registers and put the chunk after next in flight
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) xa[i][e] = xb[i][e];
        fetch(chunk + 2 * (int)gridDim.x, xb);
        lds_barrier();
        f32x4 acc[2][6];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[mt][j] = f32x4{bias[j], bias[j], bias[j], bias[j]};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ut + (mt * 16 + c16) * LQU + ks * 32 + g * 8);
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[mt][j] = mfma16(a, bw[ks][j], acc[mt][j]);
            }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) Ot[(mt * 16 + 4 * g + r) * LQO + (wave * 6 + j) * 16 + c16] = (bf16_t)acc[mt][j][r];
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int idx = tid + 256 * i, rl = idx / 48, pc = idx - rl * 48;
            const int row = chunk * DC + rl;
            if (row < p.M) *reinterpret_cast<bf16x8*>(out + (size_t)row * p.ldo + pc * 8) = *reinterpret_cast<const bf16x8*>(Ot + rl * LQO + pc * 8);
        }
        lds_barrier();
    }
}

}  // namespace

bool hs_lnqkv_supported(const GemmParams& p) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("HSIMAE_LNQKV_PERSISTENT"); on = !(e && e[0] == '0'); }
    return on && p.K == 128 && p.N == 384 && p.n_valid == 384 && p.bias && p.gamma && p.beta && p.lda % 4 == 0 && p.ldo % 8 == 0 &&
           (!p.u_out || p.ldu % 8 == 0);
}

int hs_lnqkv(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0) return HS_OK;
    if (!hs_lnqkv_supported(p)) return HS_EUNSUPPORTED;
    const int nchunks = (p.M + DC - 1) / DC;
    hipLaunchKernelGGL(lnqkv_kernel, dim3(nchunks < 512 ? nchunks : 512), dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

bool hs_lnbwd_dma_supported(const GemmParams& p) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("HSIMAE_LNBWD_DMA"); on = !(e && e[0] == '0'); }
    return on && p.N == 128 && p.n_valid == 128 && p.K == KA && p.lda % 8 == 0 && !p.accumulate && p.lnx && p.res && p.gamma &&
           p.dgamma && p.dbeta && p.ldr % 4 == 0 && p.ldo % 4 == 0 && (int64_t)(p.M + DC) * p.lda * 2 < (1ll << 32) &&
           !(reinterpret_cast<uintptr_t>(p.A) & 15) && p.out != (void*)p.lnx;
}

int hs_lnbwd_dma(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0) return HS_OK;
    if (!hs_lnbwd_dma_supported(p)) return HS_EUNSUPPORTED;
    const int nchunks = (p.M + DC - 1) / DC;
    hipLaunchKernelGGL(lnbwd_dma_kernel, dim3(nchunks < 512 ? nchunks : 512), dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

HS_UNIT_VARIANT_BITS(gemm_dma)
