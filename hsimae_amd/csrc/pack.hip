// fp32 nn.Linear / Conv3d weights -> bf16 packed weight images in MFMA B-fragment order (common.h "wpk").
// One launch repacks every matrix of the model from a device-resident descriptor table; run after each
// optimizer step (the fp32 parameters stay the master copy and the state_dict wire format).
#include "common.h"
#include "kernels.h"

namespace {

// MX e4m3 image (gemm.hip, F8): one thread per (output row n, 32-block of K) of the source.  Image bytes of k-step ks
// and lane (n & 15) + 16 g': k = 128 ks + 16 g' + j (bytes 0..15) and 128 ks + 64 + 16 g' + j (bytes 16..31); the
// block's e8m0 byte goes to lane (n & 15) + 16 b of the scale dword of (n-tile, ks / 4), byte ks % 4 (b = block in step).
__device__ void pack8(const PackDesc& d) {
    const int Nsrc = d.transpose ? d.cols : d.rows, Ksrc = d.transpose ? d.rows : d.cols;
    const int kblocks = (Ksrc + 31) / 32;
    const int KCH = (d.KS + 3) / 4;
    unsigned char* img = reinterpret_cast<unsigned char*>(d.dst);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Nsrc * kblocks; i += gridDim.x * 256) {
        // adjacent threads take adjacent SOURCE columns: a transposed source ([k][n], n contiguous) is walked with n fastest (every
        // load of the wave is one run of 256 B; with k-blocks fastest each lane touched its own 128-byte line per element:
        // 8.9 ms per repack at embed_dim 512), a plain one ([n][k]) with the k-block fastest and 16-byte loads
        int nl, kb;
        if (d.transpose) { kb = i / Nsrc; nl = i - kb * Nsrc; } else { nl = i / kblocks; kb = i - nl * kblocks; }
        float v[32];
        float am = 0.f;
        if (!d.transpose && !(d.cols & 3) && !(reinterpret_cast<uintptr_t>(d.src) & 15) && kb * 32 + 32 <= Ksrc) {
            const float4* s4 = reinterpret_cast<const float4*>(d.src + (size_t)nl * d.cols + kb * 32);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float4 t = s4[j]; v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w; }
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int kl = kb * 32 + j;
                v[j] = kl < Ksrc ? (d.transpose ? d.src[(size_t)kl * d.cols + nl] : d.src[(size_t)nl * d.cols + kl]) : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) am = fmaxf(am, fabsf(v[j]));
        int eb = (int)((__float_as_uint(am) >> 23) & 0xffu) - 8;
        eb = min(max(eb, 1), 254);
        const float inv = __uint_as_float((unsigned)(254 - eb) << 23);
        int w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float a[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = fminf(fmaxf(v[4 * q + e] * inv, -448.f), 448.f);
            int t = __builtin_amdgcn_cvt_pk_fp8_f32(a[0], a[1], 0, false);
            w[q] = __builtin_amdgcn_cvt_pk_fp8_f32(a[2], a[3], t, true);
        }
        const int n = nl + d.n_off, k = kb * 32 + d.k_off;
        const int nt = n >> 4, ks = k >> 7, kin = k & 127, b = kin >> 5;
        const int gp = (kin & 63) >> 4, half = kin >> 6;            // first 16 values: lane group gp, next 16: gp + 1
        const size_t frag = ((size_t)nt * d.KS + ks) * 64;
        unsigned char* p0 = img + (frag + (n & 15) + 16 * gp) * 32 + 16 * half;
        unsigned char* p1 = img + (frag + (n & 15) + 16 * (gp + 1)) * 32 + 16 * half;
        *reinterpret_cast<int4*>(p0) = make_int4(w[0], w[1], w[2], w[3]);
        *reinterpret_cast<int4*>(p1) = make_int4(w[4], w[5], w[6], w[7]);
        d.scales[((((size_t)nt * KCH + (ks >> 2)) * 64) + (n & 15) + 16 * b) * 4 + (ks & 3)] = (unsigned char)eb;
    }
}

__global__ __launch_bounds__(256) void pack_kernel(const PackDesc* descs, int ndesc) {
    const PackDesc d = descs[blockIdx.y];
    if (d.fp8) { pack8(d); return; }
    const int total = d.rows * d.cols;
    if (d.KS == 0) {      // plain fp32 copy (concatenated bias packs)
        float* dst = reinterpret_cast<float*>(d.dst) + d.n_off;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) dst[i] = d.src[i];
        return;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / d.cols, c = i - r * d.cols;
        const int n = (d.transpose ? c : r) + d.n_off;
        const int k = (d.transpose ? r : c) + d.k_off;
        const int nt = n >> 4, ks = k >> 5;
        const int lane = (n & 15) + 16 * ((k & 31) >> 3);
        const size_t off = (((size_t)nt * d.KS + ks) * 64 + lane) * 8 + (k & 7);
        d.dst[off] = (bf16_t)d.src[i];
    }
}

}  // namespace

int hs_pack(const PackDesc* descs_dev, int ndesc, int max_elems, hipStream_t s) {
    if (ndesc <= 0) return HS_OK;
    int gx = (max_elems + 256 * 8 - 1) / (256 * 8);
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(pack_kernel, dim3(gx, ndesc), dim3(256), 0, s, descs_dev, ndesc);
    return (int)hipGetLastError();
}

HS_UNIT_VARIANT_BITS(pack)
