// Pins the operand / scale lane maps of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3) with exact integer data.
//   hipcc --offload-arch=gfx950 -O2 scripts/micro/mx_layout.hip -o /tmp/mx_layout && /tmp/mx_layout
// Hypothesis: lane l holds A[row l&15][k = 32 (l>>4) + j] and B[k = 32 (l>>4) + j][col l&15], j = 0..31 (byte j of the
// 8-dword operand), and its scale byte applies to exactly those 32 elements; C/D: col = l&15, row = 4 (l>>4) + r.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ int kmap(int mode, int g, int j) {
    if (mode == 0) return 32 * g + j;
    if (mode == 1) return 16 * g + (j & 15) + 64 * (j >> 4);
    if (mode == 2) return 8 * g + (j & 7) + 32 * (j >> 3);
    return 4 * g + (j & 3) + 16 * (j >> 2);
}
__global__ void k(const uint8_t* A, const uint8_t* B, const uint8_t* sa, const uint8_t* sb, float* C, float* cvt_out, const float* cvt_in,
                  int mode, int smode) {
    const int l = threadIdx.x;
    i32x8 a, b;
    uint8_t ba[32], bb[32];
    for (int j = 0; j < 32; ++j) { ba[j] = A[(l & 15) * 128 + kmap(mode, l >> 4, j)]; bb[j] = B[(l & 15) * 128 + kmap(mode, l >> 4, j)]; }
    for (int i = 0; i < 8; ++i) {
        a[i] = ba[4 * i] | (ba[4 * i + 1] << 8) | (ba[4 * i + 2] << 16) | (ba[4 * i + 3] << 24);
        b[i] = bb[4 * i] | (bb[4 * i + 1] << 8) | (bb[4 * i + 2] << 16) | (bb[4 * i + 3] << 24);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    int sca = 127, scb = 127;
    if (smode == 1) { sca = sa[(l & 15) * 4 + (l >> 4)]; scb = sb[(l & 15) * 4 + (l >> 4)]; }
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sca, 0, scb);
    for (int r = 0; r < 4; ++r) C[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
    // conversion check: v_cvt_pk_fp8_f32 (OCP e4m3 on gfx950): saturation and rounding behaviour
    if (l < 16) {
        int packed = __builtin_amdgcn_cvt_pk_fp8_f32(cvt_in[2 * l], cvt_in[2 * l + 1], 0, false);
        cvt_out[2 * l] = (float)(packed & 0xff);
        cvt_out[2 * l + 1] = (float)((packed >> 8) & 0xff);
    }
}

static float e4m3_to_f(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x;
    if (e == 0) x = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) x = NAN;
    else x = ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}
static uint8_t f_to_e4m3(float x) {       // exact inputs only
    for (int v = 0; v < 256; ++v) if (e4m3_to_f((uint8_t)v) == x && !(v == 0x80)) return (uint8_t)v;
    return 0x7f;
}

int main() {
    std::vector<uint8_t> A(16 * 128), B(16 * 128), sa(64), sb(64);
    std::vector<float> Af(16 * 128), Bf(16 * 128);
    uint32_t st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (int)((st >> 16) % 9) - 4; };
    for (int i = 0; i < 16 * 128; ++i) { Af[i] = (float)rnd(); A[i] = f_to_e4m3(Af[i]); Bf[i] = (float)rnd() * 0.5f; B[i] = f_to_e4m3(Bf[i]); }
    for (int r = 0; r < 16; ++r) for (int g = 0; g < 4; ++g) { sa[r * 4 + g] = 127 + ((r + g) % 3) - 1; sb[r * 4 + g] = 127 + ((r * 2 + g) % 4) - 2; }
    std::vector<float> ref(256, 0.f), ref0(256, 0.f);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double acc = 0, acc0 = 0;
        for (int kk = 0; kk < 128; ++kk) {
            acc += (double)Af[i * 128 + kk] * ldexp(1.0, sa[i * 4 + kk / 32] - 127) * (double)Bf[j * 128 + kk] * ldexp(1.0, sb[j * 4 + kk / 32] - 127);
            acc0 += (double)Af[i * 128 + kk] * (double)Bf[j * 128 + kk];
        }
        ref[i * 16 + j] = (float)acc; ref0[i * 16 + j] = (float)acc0;
    }
    float cin[32] = {0.f, 1.f, -1.f, 448.f, 449.f, 480.f, 500.f, 1000.f, 1e9f, -500.f, 0.0625f, 0.001f, 0.0019f, 0.00097f, 17.f, 18.f,
                     19.f, 20.f, 21.f, 22.f, 23.f, 0.3f, 0.33f, 100.f, 104.f, 108.f, 112.f, 116.f, 120.f, 300.f, 464.f, 465.f};
    uint8_t *dA, *dB, *dsa, *dsb; float *dC, *dco, *dci;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dC, 1024); hipMalloc(&dco, 128); hipMalloc(&dci, 128);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
    hipMemcpy(dci, cin, 128, hipMemcpyHostToDevice);
    std::vector<float> C(256); float co[32];
    int bad = 0;
    for (int smode = 0; smode < 2; ++smode)
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC, dco, dci, mode, smode);
            hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost); hipMemcpy(co, dco, 128, hipMemcpyDeviceToHost);
            const std::vector<float>& rf = smode ? ref : ref0;
            bad = 0; double maxd = 0;
            for (int i = 0; i < 256; ++i) { double d = fabs(C[i] - rf[i]); if (d > maxd) maxd = d; if (d > 1e-3) ++bad; }
            printf("scales %s, k-map %d: %d mismatches of 256, max |diff| %.4g  (C[0]=%g ref %g, C[17]=%g ref %g)\n", smode ? "per lane" : "uniform",
                   mode, bad, maxd, C[0], rf[0], C[17], rf[17]);
        }
    printf("cvt_pk_fp8_f32:");
    for (int i = 0; i < 32; ++i) printf(" %g->%g", cin[i], e4m3_to_f((uint8_t)co[i]));
    printf("\n");
    return bad != 0;
}
