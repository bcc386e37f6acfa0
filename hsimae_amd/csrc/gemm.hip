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
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128;

template <int AK, int EPI, int KC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDA = KC + 8;                       // LDS row stride (elements): 16-B pad => conflict-free b128 reads
    bf16_t* As = reinterpret_cast<bf16_t*>(smem);     // [128][KC+8]
    float* rstat = reinterpret_cast<float*>(smem + BM * LDA * 2);   // [128][2] mean, rstd
    constexpr bool DUAL = (EPI == E_SWIGLU);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * BM;
    const int KS_total = p.K / 32;
    const int NT_total = p.N / 16;
    const int n_chunks = (NT_total + 7) / 8;
    const int k_chunks = (p.K + KC - 1) / KC;

    if constexpr (AK == A_F32_LN) {
        // LayerNorm statistics for the panel's rows: one wave per row, two-pass in registers (K <= 512).
        const float* A = reinterpret_cast<const float*>(p.A);
        for (int r = wave; r < BM; r += 4) {
            const int row = min(row0 + r, p.M - 1);
            float v[8];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane + 64 * i;
                v[i] = (c < p.K) ? A[(size_t)row * p.lda + c] : 0.f;
                s += v[i];
            }
            const float mean = wave_sum(s) / (float)p.K;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane + 64 * i;
                const float d = (c < p.K) ? v[i] - mean : 0.f;
                q += d * d;
            }
            const float rstd = rsqrtf(wave_sum(q) / (float)p.K + 1e-5f);
            if (lane == 0) {
                rstat[2 * r] = mean;
                rstat[2 * r + 1] = rstd;
                if (p.stats && row0 + r < p.M) {
                    p.stats[2 * (size_t)(row0 + r)] = mean;
                    p.stats[2 * (size_t)(row0 + r) + 1] = rstd;
                }
            }
        }
        __syncthreads();
    }

    auto stage = [&](int kc) {
        constexpr int TPR = KC / 8;                   // threads per row (8 elements each)
        constexpr int RPP = 256 / TPR;                // rows per pass
        const int c8 = (tid % TPR) * 8;
        const int kcol = kc * KC + c8;
#pragma unroll 4
        for (int r = tid / TPR; r < BM; r += RPP) {
            const int row = min(row0 + r, p.M - 1);
            bf16x8 val = zero8();
            if (kcol < p.K) {
                if constexpr (AK == A_BF16) {
                    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
                    val = *reinterpret_cast<const bf16x8*>(A + (size_t)row * p.lda + kcol);
                } else {
                    const float* A = reinterpret_cast<const float*>(p.A);
                    const float4 x0 = *reinterpret_cast<const float4*>(A + (size_t)row * p.lda + kcol);
                    const float4 x1 = *reinterpret_cast<const float4*>(A + (size_t)row * p.lda + kcol + 4);
                    float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
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
                            *reinterpret_cast<bf16x8*>(p.u_out + (size_t)(row0 + r) * p.ldu + kcol) = val;
                    }
                }
            }
            *reinterpret_cast<bf16x8*>(As + r * LDA + c8) = val;
        }
    };

    const int arow = lane & 15, ag = lane >> 4;

    for (int nc = 0; nc < n_chunks; ++nc) {
        f32x4 acc[8][2];
        f32x4 acc2[DUAL ? 8 : 1][2];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (DUAL) acc2[mt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        const int nt0 = nc * 8 + wave * 2;

        for (int kc = 0; kc < k_chunks; ++kc) {
            if (k_chunks > 1 || nc == 0) {
                if (!(nc == 0 && kc == 0)) __syncthreads();
                stage(kc);
                __syncthreads();
            }
            const int ks0 = kc * (KC / 32);
            const int nks = min(KC / 32, KS_total - ks0);
            for (int ks = 0; ks < nks; ++ks) {
                bf16x8 b[2], b2[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nt = nt0 + j;
                    if (nt < NT_total) {
                        const size_t off = (((size_t)nt * KS_total + ks0 + ks) * 64 + lane) * 8;
                        b[j] = *reinterpret_cast<const bf16x8*>(p.W + off);
                        if constexpr (DUAL) b2[j] = *reinterpret_cast<const bf16x8*>(p.W2 + off);
                    } else {
                        b[j] = zero8();
                        if constexpr (DUAL) b2[j] = zero8();
                    }
                }
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(As + (mt * 16 + arow) * LDA + ks * 32 + ag * 8);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[mt][j] = mfma16(a, b[j], acc[mt][j]);
                        if constexpr (DUAL) acc2[mt][j] = mfma16(a, b2[j], acc2[mt][j]);
                    }
                }
            }
        }

        // ---------------------------------------------------------------- epilogue for this column chunk
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nt = nt0 + j;
            if (nt >= NT_total) continue;
            const int col = nt * 16 + (lane & 15);
            const bool cvalid = col < p.n_valid;
            const float bias = (p.bias && cvalid) ? p.bias[col] : 0.f;
            float bias2 = 0.f;
            if constexpr (DUAL) bias2 = (p.bias2 && cvalid) ? p.bias2[col] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + mt * 16 + ag * 4 + r;
                    if (row >= p.M) continue;
                    const float v = acc[mt][j][r] + bias;
                    if constexpr (EPI == E_BF16) {
                        reinterpret_cast<bf16_t*>(p.out)[(size_t)row * p.ldo + col] = (bf16_t)v;
                    } else if constexpr (EPI == E_F32) {
                        if (cvalid) reinterpret_cast<float*>(p.out)[(size_t)row * p.ldo + col] = v;
                    } else if constexpr (EPI == E_RES_F32) {
                        if (cvalid) {
                            float o = v + p.res[(size_t)row * p.ldr + col];
                            if (p.res2) o += p.res2[(size_t)row * p.ldr + col];
                            reinterpret_cast<float*>(p.out)[(size_t)row * p.ldo + col] = o;
                        }
                    } else if constexpr (EPI == E_POS_F32) {
                        if (cvalid) {
                            const float o = v + p.pos[(size_t)p.ids[row] * p.ldpos + col];
                            reinterpret_cast<float*>(p.out)[(size_t)row * p.ldo + col] = o;
                        }
                    } else if constexpr (EPI == E_SWIGLU) {
                        const float h1 = cvalid ? v : 0.f;
                        const float h3 = cvalid ? acc2[mt][j][r] + bias2 : 0.f;
                        const bf16_t h1b = (bf16_t)h1, h3b = (bf16_t)h3;
                        p.h13[(size_t)row * p.ldh + col] = h1b;
                        p.h13[(size_t)row * p.ldh + p.hoff + col] = h3b;
                        const float a1 = bf2f(h1b), a3 = bf2f(h3b);     // same values the backward will see
                        const float g = a1 / (1.f + __expf(-a1)) * a3;
                        reinterpret_cast<bf16_t*>(p.out)[(size_t)row * p.ldo + col] = (bf16_t)g;
                    } else if constexpr (EPI == E_SWIGLU_BWD) {
                        const float h1 = bf2f(p.h13[(size_t)row * p.ldh + col]);
                        const float h3 = bf2f(p.h13[(size_t)row * p.ldh + p.hoff + col]);
                        const float s = 1.f / (1.f + __expf(-h1));
                        const float dg = acc[mt][j][r];
                        const float dh1 = dg * h3 * s * (1.f + h1 * (1.f - s));
                        const float dh3 = dg * h1 * s;
                        bf16_t* o = reinterpret_cast<bf16_t*>(p.out);
                        o[(size_t)row * p.ldo + col] = (bf16_t)dh1;
                        o[(size_t)row * p.ldo + p.hoff + col] = (bf16_t)dh3;
                    }
                }
            }
        }
    }
}

template <int AK, int EPI, int KC>
int launch(const GemmParams& p, hipStream_t s) {
    const int grid = (p.M + BM - 1) / BM;
    const size_t lds = (size_t)BM * (KC + 8) * 2 + BM * 2 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<AK, EPI, KC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<AK, EPI, KC>), dim3(grid), dim3(256), lds, s, p);
    return (int)hipGetLastError();
}

template <int AK, int EPI>
int launch_kc(const GemmParams& p, hipStream_t s) {
    if constexpr (AK == A_F32_LN) {
        if (p.K <= 128) return launch<AK, EPI, 128>(p, s);
        if (p.K <= 256) return launch<AK, EPI, 256>(p, s);
        if (p.K <= 512) return launch<AK, EPI, 512>(p, s);
        return HS_EUNSUPPORTED;
    } else {
        return launch<AK, EPI, 128>(p, s);
    }
}

}  // namespace

int hs_gemm(const GemmParams& p, int akind, int epi, hipStream_t s) {
    if (p.M <= 0) return HS_OK;
    if (p.K % 32 || p.N % 16 || p.lda % 8) return HS_EDIMS;
#define CASE(AK, EP) \
    if (akind == AK && epi == EP) return launch_kc<AK, EP>(p, s);
    CASE(A_F32_LN, E_BF16)
    CASE(A_F32_LN, E_SWIGLU)
    CASE(A_F32_LN, E_F32)
    CASE(A_BF16, E_RES_F32)
    CASE(A_BF16, E_POS_F32)
    CASE(A_BF16, E_F32)
    CASE(A_BF16, E_BF16)
    CASE(A_F32, E_SWIGLU_BWD)
    CASE(A_F32, E_BF16)
    CASE(A_F32, E_F32)
#undef CASE
    return HS_EUNSUPPORTED;
}
