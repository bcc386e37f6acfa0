// HBM-bound glue kernels of the HSIMAE pretraining path: structured masking, patch gather,
// LayerNorm backward, decoder sequence assembly (fwd/bwd), normalised-pixel MSE loss + recons.
#include "common.h"
#include "kernels.h"
#include "plan.h"
#include <algorithm>
#include <cmath>

namespace {

// ------------------------------------------------------------------ masking (Models.py:495-535, closed form)
// One thread per sample: rank-select the len_t smallest of noise_1[n,:T] and the len_l smallest of
// noise_2[n,:L] (ties: lower index wins), then emit ids_keep (ascending), ids_restore and mask.
__global__ void mask_kernel(MaskParams p) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= p.N) return;
    const float* n1 = p.noise1 + (size_t)n * p.T;
    const float* n2 = p.noise2 + (size_t)n * p.L;
    unsigned long long kt = 0ull, kl = 0ull;
    for (int a = 0; a < p.T; ++a) {
        const float v = n1[a];
        int rank = 0;
        for (int b = 0; b < p.T; ++b) {
            const float w = n1[b];
            rank += (w < v) || (w == v && b < a);
        }
        if (rank < p.len_t) kt |= 1ull << a;
    }
    for (int a = 0; a < p.L; ++a) {
        const float v = n2[a];
        int rank = 0;
        for (int b = 0; b < p.L; ++b) {
            const float w = n2[b];
            rank += (w < v) || (w == v && b < a);
        }
        if (rank < p.len_l) kl |= 1ull << a;
    }
    const int TL = p.T * p.L, K = p.len_t * p.len_l;
    const int c1 = p.len_t * (p.L - p.len_l) + (p.T - p.len_t) * p.len_l;
    int cnt[3] = {0, K, K + c1};
    int32_t* keep = p.ids_keep + (size_t)n * K;
    int32_t* rest = p.ids_restore + (size_t)n * TL;
    float* mask = p.mask + (size_t)n * TL;
    for (int a = 0; a < p.T; ++a) {
        const int ca = ((kt >> a) & 1ull) ? 0 : 1;
        for (int b = 0; b < p.L; ++b) {
            const int c = ca + (((kl >> b) & 1ull) ? 0 : 1);
            const int i = a * p.L + b;
            const int pos = cnt[c]++;
            rest[i] = pos;
            mask[i] = c ? 1.f : 0.f;
            if (c == 0) keep[pos] = i;
        }
    }
}

// ------------------------------------------------------------------ patch gather (Models.py:151-160 as a GEMM operand)
// out[n*K + k][f] = x[n, 0, 8*tau + u, 3*i + p, 3*j + q],  f = 9u + 3p + q, token ids_keep[n,k] = 9*tau + 3*i + j
__global__ __launch_bounds__(256) void patch_gather_kernel(PatchParams p) {
    const int64_t total = (int64_t)p.N * p.K * 12;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t row = idx / 12;
        const int oct = (int)(idx % 12);
        const int n = (int)(row / p.K);
        bf16x8 v = zero8();
        if (oct < 9) {
            const int tok = p.ids_keep[row];
            const int tau = tok / 9, sp = tok % 9, gi = sp / 3, gj = sp % 3;
            const float* base = p.x + (size_t)n * p.sn;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ft = oct * 8 + e;
                const int u = ft / 9, pq = ft % 9, pp = pq / 3, q = pq % 3;
                f[e] = base[(int64_t)(8 * tau + u) * p.sb + (int64_t)(3 * gi + pp) * p.sh + (int64_t)(3 * gj + q) * p.sw];
            }
            v = cvt8(f);
        }
        *reinterpret_cast<bf16x8*>(p.out + row * 96 + oct * 8) = v;
    }
}

// ------------------------------------------------------------------ LayerNorm backward
// dx = dres + rstd * (g*du - mean(g*du) - xhat * mean(g*du*xhat));  dgamma += du * xhat, dbeta += du.
// A row is owned by TPR adjacent lanes (8 columns each); each thread keeps 4 rows of loads in flight so the
// kernel streams instead of paying one memory latency per row.  Column sums: per-thread partials over the
// workgroup's rows, reduced across row groups in LDS, one atomic per column per workgroup.
template <int TPR>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdParams p) {
    constexpr int G = 256 / TPR, NI = 8, UB = 4, W = 8 * TPR;
    __shared__ float red[2][G][W];
    const int tid = threadIdx.x, grp = tid / TPR, c8 = (tid % TPR) * 8;
    const bool cok = c8 < p.d;
    float gam[8], dg[8], db[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gam[e] = cok ? p.gamma[c8 + e] : 0.f; dg[e] = 0.f; db[e] = 0.f; }
    const float invd = 1.f / (float)p.d;
    const int base = blockIdx.x * (G * NI);
    auto ld8 = [](const float* q, float* o) {
        const float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    };
    for (int it = 0; it < NI; it += UB) {
        float x[UB][8], du[UB][8], rs[UB][8];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int r = base + (it + u) * G + grp;
            const bool ok = cok && r < p.M;
#pragma unroll
            for (int e = 0; e < 8; ++e) { x[u][e] = 0.f; du[u][e] = 0.f; rs[u][e] = 0.f; }
            if (ok) {
                const size_t o = (size_t)r * (p.ld ? p.ld : p.d) + c8;
                ld8(p.x + o, x[u]);
                ld8(p.du + o, du[u]);
                if (p.dres) ld8(p.dres + o, rs[u]);
                if (p.accumulate) {
                    float t[8];
                    ld8(p.dx + o, t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) rs[u][e] += t[e];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int r = base + (it + u) * G + grp;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += x[u][e];
            s = lanes_sum<TPR>(s);
            const float mean = s * invd;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float dl = cok ? x[u][e] - mean : 0.f; x[u][e] = dl; q += dl * dl; }
            q = lanes_sum<TPR>(q);
            const float rstd = rsqrtf(q * invd + 1e-5f);
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x[u][e] *= rstd;                                   // xhat
                const float t = du[u][e] * gam[e];
                a += t;
                b += t * x[u][e];
            }
            a = lanes_sum<TPR>(a); b = lanes_sum<TPR>(b);
            a *= invd; b *= invd;
            if (cok && r < p.M) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = rstd * (du[u][e] * gam[e] - a - x[u][e] * b) + rs[u][e];
                    dg[e] += du[u][e] * x[u][e];
                    db[e] += du[u][e];
                }
                float* op = p.dx + (size_t)r * (p.ld ? p.ld : p.d) + c8;
                *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][grp][c8 + e] = dg[e]; red[1][grp][c8 + e] = db[e]; }
    __syncthreads();
    for (int c = tid; c < p.d; c += 256) {
        float g = 0.f, b = 0.f;
#pragma unroll 4
        for (int k = 0; k < G; ++k) { g += red[0][k][c]; b += red[1][k][c]; }
        const HsDet det{p.det_base, reinterpret_cast<long long*>(p.det_acc)};
        if (p.dgamma) hs_gadd(det, p.dgamma + c, g);
        if (p.dbeta) hs_gadd(det, p.dbeta + c, b);
    }
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* x, const float* gamma, const float* beta, float* out,
                                                      int M, int d, int ldx, int ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= M) return;
    float v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < d ? x[(size_t)r * ldx + c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        const float dl = c < d ? v[i] - mean : 0.f;
        q += dl * dl;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        if (c < d) out[(size_t)r * ldo + c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
    }
}

// ------------------------------------------------------------------ decoder sequence assembly (Models.py:579-592)
// yfull[n,i,:] = (ids_restore[n,i] < K ? y[n, ids_restore[n,i], :] : mean_k y[n,k,:]) + decoder_pos_embed[i,:]
// Fast forms for decoder widths whose row (Dd / 4 float4 lanes) divides the workgroup: a thread owns one float4
// column group and walks rows, so every access is a coalesced 16-byte piece and the mean token / its gradient is a
// two-level sum (row groups in registers, then LDS) instead of one serial chain per column.
template <int LPR>                                        // lanes per row = Dd / 4
__global__ __launch_bounds__(256) void assemble_fwd_fast_kernel(AssembleParams p) {
    constexpr int RG = 256 / LPR, Dd = LPR * 4;
    __shared__ float4 part[RG][LPR];
    __shared__ int rest[512];
    const int n = blockIdx.x, c4 = threadIdx.x % LPR, rg = threadIdx.x / LPR;
    const float4* y = reinterpret_cast<const float4*>(p.y + (size_t)n * p.K * Dd);
    for (int i = threadIdx.x; i < p.TL; i += 256) rest[i] = p.ids_restore[(size_t)n * p.TL + i];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = rg; k < p.K; k += RG) {
        const float4 v = y[(size_t)k * LPR + c4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    part[rg][c4] = s;
    __syncthreads();
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < RG; ++g) { const float4 v = part[g][c4]; m.x += v.x; m.y += v.y; m.z += v.z; m.w += v.w; }
    const float ik = 1.f / (float)p.K;
    m.x *= ik; m.y *= ik; m.z *= ik; m.w *= ik;
    const float4* pos = reinterpret_cast<const float4*>(p.pos);
    float4* out = reinterpret_cast<float4*>(p.yfull + (size_t)n * p.TL * Dd);
    for (int i = rg; i < p.TL; i += RG) {
        const int r = rest[i];
        const float4 v = (r < p.K) ? y[(size_t)r * LPR + c4] : m;
        const float4 q = pos[(size_t)i * LPR + c4];
        out[(size_t)i * LPR + c4] = make_float4(v.x + q.x, v.y + q.y, v.z + q.z, v.w + q.w);
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void assemble_bwd_fast_kernel(AssembleParams p) {
    constexpr int RG = 256 / LPR, Dd = LPR * 4;
    __shared__ float4 part[RG][LPR];
    __shared__ int rest[512];
    const int n = blockIdx.x, c4 = threadIdx.x % LPR, rg = threadIdx.x / LPR;
    const float4* dyf = reinterpret_cast<const float4*>(p.dyfull + (size_t)n * p.TL * Dd);
    for (int i = threadIdx.x; i < p.TL; i += 256) rest[i] = p.ids_restore[(size_t)n * p.TL + i];
    __syncthreads();
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = rg; i < p.TL; i += RG)
        if (rest[i] >= p.K) {
            const float4 v = dyf[(size_t)i * LPR + c4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    part[rg][c4] = s;
    __syncthreads();
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < RG; ++g) { const float4 v = part[g][c4]; m.x += v.x; m.y += v.y; m.z += v.z; m.w += v.w; }
    const float ik = 1.f / (float)p.K;
    m.x *= ik; m.y *= ik; m.z *= ik; m.w *= ik;
    for (int i = rg; i < p.TL; i += RG) {
        const int r = rest[i];
        if (r < p.K) {
            const float4 v = dyf[(size_t)i * LPR + c4];
            bf16x4 o;
            o[0] = (bf16_t)(v.x + m.x); o[1] = (bf16_t)(v.y + m.y); o[2] = (bf16_t)(v.z + m.z); o[3] = (bf16_t)(v.w + m.w);
            *reinterpret_cast<bf16x4*>(p.dy + ((size_t)n * p.K + r) * Dd + c4 * 4) = o;
        }
    }
}

__global__ __launch_bounds__(256) void assemble_fwd_kernel(AssembleParams p) {
    const int LD = p.ld ? p.ld : p.Dd;           // row stride of y / yfull (storage width)
    __shared__ float meanv[512];
    const int n = blockIdx.x;
    const float* y = p.y + (size_t)n * p.K * LD;
    for (int c = threadIdx.x; c < p.Dd; c += 256) {
        float s = 0.f;
        for (int k = 0; k < p.K; ++k) s += y[(size_t)k * LD + c];
        meanv[c] = s / (float)p.K;
    }
    __syncthreads();
    const int total = p.TL * p.Dd;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int i = e / p.Dd, c = e - i * p.Dd;
        const int r = p.ids_restore[(size_t)n * p.TL + i];
        const float v = (r < p.K) ? y[(size_t)r * LD + c] : meanv[c];
        p.yfull[((size_t)n * p.TL + i) * LD + c] = v + p.pos[(size_t)i * p.Dd + c];
    }
}

// dy[n,k,:] = dyfull[n, slot(k), :] + (1/K) * sum_{masked i} dyfull[n,i,:]      (bf16 out: GEMM / wgrad operand)
__global__ __launch_bounds__(256) void assemble_bwd_kernel(AssembleParams p) {
    const int LD = p.ld ? p.ld : p.Dd;
    __shared__ float msum[512];
    const int n = blockIdx.x;
    const float* dyf = p.dyfull + (size_t)n * p.TL * LD;
    const int32_t* rest = p.ids_restore + (size_t)n * p.TL;
    for (int c = threadIdx.x; c < p.Dd; c += 256) {
        float s = 0.f;
        for (int i = 0; i < p.TL; ++i)
            if (rest[i] >= p.K) s += dyf[(size_t)i * LD + c];
        msum[c] = s / (float)p.K;
    }
    __syncthreads();
    const int total = p.TL * p.Dd;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int i = e / p.Dd, c = e - i * p.Dd;
        const int r = rest[i];
        if (r < p.K) p.dy[((size_t)n * p.K + r) * LD + c] = (bf16_t)(dyf[(size_t)i * LD + c] + msum[c]);
    }
}

// ------------------------------------------------------------------ loss + recons (Models.py:603-625)
// One wave per token row; lanes own features f = lane and lane + 64 (< 72).
using hsplan::LOSS_ROWS_PER_WG;       // plan.h (the workspace carve sizes the partial-sum buffer with it)

__global__ __launch_bounds__(256) void loss_kernel(LossParams p) {
    __shared__ float wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int TL = p.T * 9;
    const int64_t M = (int64_t)p.N * TL;
    const int64_t r0 = (int64_t)blockIdx.x * LOSS_ROWS_PER_WG;
    float lacc = 0.f;
    for (int64_t row = r0 + wave; row < min(r0 + (int64_t)LOSS_ROWS_PER_WG, M); row += 4) {
        const int n = (int)(row / TL), tok = (int)(row % TL);
        const int tau = tok / 9, sp = tok % 9, gi = sp / 3, gj = sp % 3;
        const float* base = p.x + (size_t)n * p.sn;
        float t[2], pr[2];
        int64_t off[2];
        float s = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ft = lane + 64 * h;
            const bool ok = ft < 72;
            const int u = ft / 9, pq = ft % 9, pp = pq / 3, q = pq % 3;
            off[h] = (int64_t)(8 * tau + u) * p.sb + (int64_t)(3 * gi + pp) * p.sh + (int64_t)(3 * gj + q) * p.sw;
            t[h] = ok ? base[off[h]] : 0.f;
            pr[h] = ok ? p.pred[row * 72 + ft] : 0.f;
            s += t[h];
        }
        float mean = 0.f, std = 1.f;
        if (p.norm_pix) {
            mean = wave_sum(s) * (1.f / 72.f);
            float q2 = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float dlt = (lane + 64 * h < 72) ? t[h] - mean : 0.f;
                q2 += dlt * dlt;
            }
            std = sqrtf(wave_sum(q2) * (1.f / 71.f) + 1.0e-6f);
        }
        const float mk = p.mask[row];
        float e2 = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ft = lane + 64 * h;
            if (ft < 72) {
                const float tg = p.norm_pix ? (t[h] - mean) / std : t[h];
                const float diff = pr[h] - tg;
                e2 += diff * diff;
                if (p.dpred) p.dpred[row * 96 + ft] = (bf16_t)(2.f * mk * diff * p.inv_scale);
                if (p.pred_img) {
                    // contiguous [N,1,B,9,9] output image
                    const int u = ft / 9, pq = ft % 9, pp = pq / 3, q = pq % 3;
                    const size_t o = ((size_t)n * p.T * 8 + (8 * tau + u)) * 81 + (3 * gi + pp) * 9 + (3 * gj + q);
                    p.pred_img[o] = p.norm_pix ? pr[h] * std + mean : pr[h];
                    p.mask_img[o] = mk;
                }
            } else if (ft < 96 && p.dpred) {
                p.dpred[row * 96 + ft] = (bf16_t)0.f;
            }
        }
        e2 = wave_sum(e2);
        lacc += e2 * (1.f / 72.f) * mk;
    }
    if (lane == 0) wsum[wave] = lacc;
    __syncthreads();
    if (threadIdx.x == 0) p.partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// One workgroup per sample: the cube is read once, coalesced, into an LDS image (the token view is 24 runs of 3 floats
// per token: as global accesses that is 4-byte gathers / scatters), the reconstruction image is assembled in LDS and
// written back as whole rows.  Same arithmetic per token as loss_kernel.
constexpr int LSN = 1024;                     // threads per sample: 16 waves walk the tokens
__global__ __launch_bounds__(LSN) void loss_sample_kernel(LossParams p) {
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    __shared__ float wsum[LSN / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int B = p.T * 8, E = B * 81, TL = p.T * 9;
    float* X = lsm;
    float* P = lsm + E;
    float* MK = P + (p.pred_img ? E : 0);
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < TL; i += LSN) MK[i] = p.mask[(size_t)n * TL + i];
    const float* base = p.x + (size_t)n * p.sn;
    if (p.sw == 1 && p.sh == 9 && p.sb == 81 && (p.sn & 3) == 0 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0) {
        for (int e = threadIdx.x; e < E / 4; e += LSN) reinterpret_cast<float4*>(X)[e] = reinterpret_cast<const float4*>(base)[e];
    } else if (p.sb == 1) {                         // band-fastest cubes (the loader's layout): bands are the contiguous axis
        for (int e = threadIdx.x; e < E; e += LSN) {
            // (round 6 tried (e / B, e % B) by increments instead of a division by the run-time B per element: 162.5 -> 169 us, the
            //  carry branch costs more than the division; reverted)
            const int ij = e / B, b = e - ij * B;
            X[b * 81 + ij] = base[(int64_t)b + (int64_t)(ij / 9) * p.sh + (int64_t)(ij % 9) * p.sw];
        }
    } else {
        for (int e = threadIdx.x; e < E; e += LSN) {
            const int b = e / 81, ij = e - b * 81;
            X[e] = base[(int64_t)b * p.sb + (int64_t)(ij / 9) * p.sh + (int64_t)(ij % 9) * p.sw];
        }
    }
    __syncthreads();
    float lacc = 0.f;
    for (int tok = wave; tok < TL; tok += LSN / 64) {
        const int64_t row = (int64_t)n * TL + tok;
        const int tau = tok / 9, sp = tok % 9, gi = sp / 3, gj = sp % 3;
        float t[2], pr[2];
        int off[2];
        float s = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ft = lane + 64 * h;
            const bool ok = ft < 72;
            const int u = ft / 9, pq = ft % 9, pp = pq / 3, q = pq % 3;
            off[h] = (8 * tau + u) * 81 + (3 * gi + pp) * 9 + (3 * gj + q);
            t[h] = ok ? X[off[h]] : 0.f;
            pr[h] = ok ? p.pred[row * 72 + ft] : 0.f;
            s += t[h];
        }
        float mean = 0.f, std = 1.f;
        if (p.norm_pix) {
            mean = wave_sum(s) * (1.f / 72.f);
            float q2 = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float dlt = (lane + 64 * h < 72) ? t[h] - mean : 0.f;
                q2 += dlt * dlt;
            }
            std = sqrtf(wave_sum(q2) * (1.f / 71.f) + 1.0e-6f);
        }
        const float mk = MK[tok];
        float e2 = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ft = lane + 64 * h;
            if (ft < 72) {
                const float tg = p.norm_pix ? (t[h] - mean) / std : t[h];
                const float diff = pr[h] - tg;
                e2 += diff * diff;
                if (p.dpred) p.dpred[row * 96 + ft] = (bf16_t)(2.f * mk * diff * p.inv_scale);
                if (p.pred_img) P[off[h]] = p.norm_pix ? pr[h] * std + mean : pr[h];
            } else if (ft < 96 && p.dpred) {
                p.dpred[row * 96 + ft] = (bf16_t)0.f;
            }
        }
        e2 = wave_sum(e2);
        lacc += e2 * (1.f / 72.f) * mk;
    }
    if (lane == 0) wsum[wave] = lacc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < LSN / 64; ++i) t += wsum[i];
        p.partial[blockIdx.x] = t;
    }
    if (p.pred_img) {                              // contiguous [N,1,B,9,9] images, 16 bytes per lane
        float4* po = reinterpret_cast<float4*>(p.pred_img + (size_t)n * E);
        float4* mo = reinterpret_cast<float4*>(p.mask_img + (size_t)n * E);
        for (int e4 = threadIdx.x; e4 < E / 4; e4 += LSN) {
            {                                       // the two images are outputs nobody on the GPU reads back: streaming stores
                const float4 pv = reinterpret_cast<const float4*>(P)[e4];
                __builtin_nontemporal_store((f32x4{pv.x, pv.y, pv.z, pv.w}), reinterpret_cast<f32x4*>(po) + e4);
            }
            float m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e4 * 4 + k, b = e / 81, ij = e - b * 81;
                m[k] = MK[(b >> 3) * 9 + (ij / 27) * 3 + (ij % 9) / 3];
            }
            __builtin_nontemporal_store((f32x4{m[0], m[1], m[2], m[3]}), reinterpret_cast<f32x4*>(mo) + e4);
        }
    }
}

__global__ __launch_bounds__(256) void loss_final_kernel(const float* partial, int n, float sum_mask, float* loss) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(red[0] / (double)sum_mask);
}

// ------------------------------------------------------------------ fused AdamW over the flat parameter buffer
// torch.optim.AdamW semantics (decoupled decay, bias correction), one launch for all 535 tensors
// (Model_Pretraining.py:80-86,102).  group[i]: 0 = weight decay, 1 = no decay ('bias' / 'norm' names), 2 = frozen.
__global__ __launch_bounds__(256) void adamw_kernel(float4* p, const float4* g, float4* m, float4* v, const uchar4* group,
                                                    int64_t n4, float lr, float b1, float b2, float eps, float wd,
                                                    float inv_bc1, float inv_sqrt_bc2) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const uchar4 gr = group[i];
        if (gr.x == 2 && gr.y == 2 && gr.z == 2 && gr.w == 2) continue;
        float4 P = p[i], M = m[i], V = v[i];
        const float4 G = g[i];
        float* pp = &P.x; float* mm = &M.x; float* vv = &V.x;
        const float* gg = &G.x;
        const unsigned char grp[4] = {gr.x, gr.y, gr.z, gr.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (grp[e] == 2) continue;
            float x = pp[e];
            if (grp[e] == 0) x *= 1.f - lr * wd;
            const float mn = mm[e] + (gg[e] - mm[e]) * (1.f - b1);          // lerp, as torch does it
            const float vn = vv[e] * b2 + gg[e] * gg[e] * (1.f - b2);
            const float denom = sqrtf(vn) * inv_sqrt_bc2 + eps;
            pp[e] = x - lr * inv_bc1 * (mn / denom);
            mm[e] = mn; vv[e] = vn;
        }
        p[i] = P; m[i] = M; v[i] = V;
    }
}

__global__ __launch_bounds__(256) void add2_kernel(const float4* a, const float4* b, float4* o, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        o[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}

}  // namespace

int hs_mask(const MaskParams& p, hipStream_t s) {
    if (p.N <= 0) return HS_OK;
    if (p.T > 64 || p.L > 64 || p.len_t > p.T || p.len_l > p.L || p.len_t < 1 || p.len_l < 1) return HS_EDIMS;
    hipLaunchKernelGGL(mask_kernel, dim3((p.N + 63) / 64), dim3(64), 0, s, p);
    return (int)hipGetLastError();
}

int hs_patch_gather(const PatchParams& p, hipStream_t s) {
    if (p.N <= 0) return HS_OK;
    const int64_t total = (int64_t)p.N * p.K * 12;
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(patch_gather_kernel, dim3(grid), dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

template <int TPR>
int launch_ln_bwd(const LnBwdParams& p, hipStream_t s) {
    const int rows_per_wg = (256 / TPR) * 8;
    hipLaunchKernelGGL(ln_bwd_kernel<TPR>, dim3((p.M + rows_per_wg - 1) / rows_per_wg), dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

int hs_ln_bwd(const LnBwdParams& p, hipStream_t s) {
    if (p.M <= 0) return HS_OK;
    if (p.d > 512 || p.d % 8) return HS_EUNSUPPORTED;
    if (p.d <= 64) return launch_ln_bwd<8>(p, s);
    if (p.d <= 128) return launch_ln_bwd<16>(p, s);
    if (p.d <= 256) return launch_ln_bwd<32>(p, s);
    return launch_ln_bwd<64>(p, s);
}

int hs_ln_fwd(const float* x, const float* gamma, const float* beta, float* out, int M, int d, hipStream_t s, int ldx, int ldo) {
    if (M <= 0) return HS_OK;
    if (d > 512) return HS_EUNSUPPORTED;
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, gamma, beta, out, M, d, ldx ? ldx : d, ldo ? ldo : d);
    return (int)hipGetLastError();
}

int hs_assemble_fwd(const AssembleParams& p, hipStream_t s) {
    if (p.N <= 0) return HS_OK;
    if (p.Dd > 512) return HS_EUNSUPPORTED;
    if (p.TL <= 512 && p.Dd == 64 && (!p.ld || p.ld == 64)) hipLaunchKernelGGL(assemble_fwd_fast_kernel<16>, dim3(p.N), dim3(256), 0, s, p);
    else if (p.TL <= 512 && p.Dd == 32 && (!p.ld || p.ld == 32)) hipLaunchKernelGGL(assemble_fwd_fast_kernel<8>, dim3(p.N), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(assemble_fwd_kernel, dim3(p.N), dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

int hs_assemble_bwd(const AssembleParams& p, hipStream_t s) {
    if (p.N <= 0) return HS_OK;
    if (p.Dd > 512) return HS_EUNSUPPORTED;
    if (p.TL <= 512 && p.Dd == 64 && (!p.ld || p.ld == 64)) hipLaunchKernelGGL(assemble_bwd_fast_kernel<16>, dim3(p.N), dim3(256), 0, s, p);
    else if (p.TL <= 512 && p.Dd == 32 && (!p.ld || p.ld == 32)) hipLaunchKernelGGL(assemble_bwd_fast_kernel<8>, dim3(p.N), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(assemble_bwd_kernel, dim3(p.N), dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

int hs_loss(const LossParams& p, hipStream_t s) {
    const int64_t M = (int64_t)p.N * p.T * 9;
    if (M <= 0) return HS_OK;
    // per-sample form while two cube images + the mask row fit in LDS (T <= 24 with images, T <= 48 without)
    const size_t lds = ((size_t)p.T * 8 * 81 * (p.pred_img ? 2 : 1) + (size_t)p.T * 9) * 4;
    int grid;
    if (lds <= 150 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(loss_sample_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
            attr_set = true;
        }
        grid = p.N;
        hipLaunchKernelGGL(loss_sample_kernel, dim3(grid), dim3(LSN), lds, s, p);
    } else {
        grid = (int)((M + LOSS_ROWS_PER_WG - 1) / LOSS_ROWS_PER_WG);
        hipLaunchKernelGGL(loss_kernel, dim3(grid), dim3(256), 0, s, p);
    }
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, p.partial, grid, p.sum_mask, p.loss);
    return (int)hipGetLastError();
}
// 'AGG' pooling of the fine-tuning head (Models.py:962-970, 1150-1156): latent [N, T*L, D] -> [N, T*D], mean over the
// L spatial tokens of each spectral group (x.reshape(N,T,L,C).permute(0,2,1,3).reshape(N,L,T*C).mean(1)).
static __global__ __launch_bounds__(256) void agg_pool_kernel(const float* __restrict__ latent, float* __restrict__ pooled, int T, int L,
                                                        int D) {
    const int n = blockIdx.x;
    for (int e = threadIdx.x; e < T * D; e += 256) {
        const int t = e / D, c = e - t * D;
        const float* src = latent + ((size_t)n * T * L + (size_t)t * L) * D + c;
        float s = 0.f;
        for (int l = 0; l < L; ++l) s += src[(size_t)l * D];
        pooled[(size_t)n * T * D + e] = s / (float)L;
    }
}

int hs_agg_pool(const float* latent, float* pooled, int N, int T, int L, int D, hipStream_t s) {
    hipLaunchKernelGGL(agg_pool_kernel, dim3(N), dim3(256), 0, s, latent, pooled, T, L, D);
    return (int)hipGetLastError();
}

// Backward of the 'AGG' classification head (Models.py:962-970: class_pred = pooled W^T + b, pooled[n][t*D + c] = mean_l latent[n][t][l][c])
// from dL/d(class_pred): dW = g^T pooled, db = column sums of g, dL/d(latent)[n][t][l][c] = (g W)[n][t*D + c] / L for every l.
// fp32 throughout (N is a fine-tuning batch of tens of cubes, C <= 32 classes: three tiny reductions, one launch).
// Workgroups 0 .. ceil(C*TD/256)-1 own 256 elements of dW (+ db), the rest 256 elements of g W each and broadcast them over L.
static __global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ g, const float* __restrict__ pooled,
                                                              const float* __restrict__ w, float* __restrict__ gw, float* __restrict__ gb,
                                                              float* __restrict__ dlat, int N, int C, int T, int L, int D, int wblocks) {
    const int TD = T * D;
    if ((int)blockIdx.x < wblocks) {
        const int idx = blockIdx.x * 256 + threadIdx.x;
        if (idx < C * TD) {
            const int c = idx / TD, k = idx - c * TD;
            float acc = 0.f;
            for (int n = 0; n < N; ++n) acc = fmaf(g[n * C + c], pooled[(size_t)n * TD + k], acc);
            gw[idx] = acc;
        }
        if (blockIdx.x == 0 && (int)threadIdx.x < C) {
            float acc = 0.f;
            for (int n = 0; n < N; ++n) acc += g[n * C + threadIdx.x];
            gb[threadIdx.x] = acc;
        }
        return;
    }
    const int idx = (blockIdx.x - wblocks) * 256 + threadIdx.x;
    if (idx >= N * TD) return;
    const int n = idx / TD, k = idx - n * TD, t = k / D, c = k - t * D;
    float acc = 0.f;
    for (int j = 0; j < C; ++j) acc = fmaf(g[n * C + j], w[(size_t)j * TD + k], acc);
    acc /= (float)L;
    float* dst = dlat + (((size_t)n * T + t) * L) * D + c;
    for (int l = 0; l < L; ++l) dst[(size_t)l * D] = acc;
}

int hs_head_bwd(const float* g, const float* pooled, const float* w, float* gw, float* gb, float* dlat, int N, int C, int T, int L,
                int D, hipStream_t s) {
    if (N <= 0) return HS_OK;
    if (C < 1 || C > 256 || T < 1 || L < 1 || D < 1) return HS_EDIMS;
    const int TD = T * D, wblocks = (C * TD + 255) / 256, lblocks = (N * TD + 255) / 256;
    hipLaunchKernelGGL(head_bwd_kernel, dim3(wblocks + lblocks), dim3(256), 0, s, g, pooled, w, gw, gb, dlat, N, C, T, L, D, wblocks);
    return (int)hipGetLastError();
}

// fp32 rows -> bf16 copy (optionally times a per-row factor): the layer-at-a-time backward's dY / dx1 as operands of the
// LDS-DMA weight-gradient kernel, which takes bf16 only (the register-staged fp32 path runs at half its bandwidth).
static __global__ __launch_bounds__(256) void rows_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n8, int d8,
                                                           const float* __restrict__ rowscale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4 a = *reinterpret_cast<const float4*>(src + i * 8), b = *reinterpret_cast<const float4*>(src + i * 8 + 4);
        float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if (rowscale) {
            const float q = rowscale[i / d8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= q;
        }
        *reinterpret_cast<bf16x8*>(dst + i * 8) = cvt8(f);
    }
}

int hs_rows_to_bf16(const float* src, hs_bf16* dst, int64_t rows, int d, const float* rowscale, hipStream_t s) {
    if (rows <= 0) return HS_OK;
    if (d % 8) return HS_EDIMS;
    const int64_t n8 = rows * (d / 8);
    const int grid = (int)std::min<int64_t>((n8 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(rows_to_bf16_kernel, dim3(grid), dim3(256), 0, s, src, dst, n8, d / 8, rowscale);
    return (int)hipGetLastError();
}

// fp32 rows [rows][cols] -> bf16 rows [rows][ldd] zero-padded (dL/dpred of a stand-alone decode backward: 72 -> 96)
static __global__ __launch_bounds__(256) void rows_pad_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t rows, int cols,
                                                            int ldd) {
    const int64_t n = rows * ldd;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ldd;
        const int c = (int)(i - r * ldd);
        dst[i] = c < cols ? (bf16_t)src[r * cols + c] : (bf16_t)0.f;
    }
}

int hs_rows_pad_bf16(const float* src, hs_bf16* dst, int64_t rows, int cols, int ldd, hipStream_t s) {
    if (rows <= 0) return HS_OK;
    if (cols > ldd) return HS_EDIMS;
    const int grid = (int)std::min<int64_t>((rows * ldd + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(rows_pad_bf16_kernel, dim3(grid), dim3(256), 0, s, src, dst, rows, cols, ldd);
    return (int)hipGetLastError();
}

// deterministic mode: fixed-point shadow sums of grads[off, off + n) -> fp32 (added onto what the plain stores left there)
static __global__ __launch_bounds__(256) void det_convert_kernel(const long long* __restrict__ acc, float* __restrict__ g, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        g[i] += (float)((double)acc[i] * (1.0 / (double)HS_DET_SCALE));
}
int hs_det_convert(const int64_t* acc, float* g, int64_t n, hipStream_t s) {
    if (n <= 0) return HS_OK;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(det_convert_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const long long*>(acc), g, n);
    return (int)hipGetLastError();
}

int hs_loss_partials(int N, int T) { return hsplan::loss_partials(N, T); }

int hs_add2(const float* a, const float* b, float* out, int64_t n, hipStream_t s) {
    if (n <= 0) return HS_OK;
    if (n % 4) return HS_EDIMS;
    const int64_t n4 = n / 4;
    const int grid = (int)std::min<int64_t>((n4 + 255) / 256, 4096);
    hipLaunchKernelGGL(add2_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4*>(a),
                       reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(out), n4);
    return (int)hipGetLastError();
}

int hs_adamw(float* p, const float* g, float* m, float* v, const unsigned char* group, int64_t n, float lr, float b1, float b2,
             float eps, float wd, int step, hipStream_t s) {
    if (n <= 0) return HS_OK;
    if (n % 4 || step < 1) return HS_EDIMS;
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    const int64_t n4 = n / 4;
    const int grid = (int)std::min<int64_t>((n4 + 255) / 256, 2048);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<float4*>(p), reinterpret_cast<const float4*>(g),
                       reinterpret_cast<float4*>(m), reinterpret_cast<float4*>(v), reinterpret_cast<const uchar4*>(group), n4, lr, b1,
                       b2, eps, wd, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)));
    return (int)hipGetLastError();
}

HS_UNIT_VARIANT_BITS(elem)
