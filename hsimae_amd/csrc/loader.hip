// Next row N2: the pretraining input pipeline on the device (Model_Pretraining.py:21-54 `HSIdataset4PT`).
//
// The reference keeps the scenes ([h, w, Bands] arrays) in host memory and, per sample, slices a 9x9xBands window,
// normalises it with the scene's (max, min), flips it along w and/or h with probability 1/2 each and permutes it
// to [1, Bands, 9, 9]; a python loop delivers ~13.5 k samples/s.  Here the scenes stay resident in HBM
// (band-fastest, exactly the reference's array layout) and one launch assembles a whole batch from the cut table:
// a pure gather, HBM-bound (read 324*Bands B, write 324*Bands B per cube), bit-exact with the numpy arithmetic.
#include "common.h"
#include "kernels.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void cube_gather_kernel(CubeParams p) {
    const int n = blockIdx.x;
    const int16_t* cu = p.cut + (size_t)p.index[n] * 6;          // (c, h, w, scene, max, min); c is not used by the reference
    const int h0 = cu[1], w0 = cu[2], sc = cu[3];
    const T mx = (T)cu[4], mn = (T)cu[5];
    const T den = (T)(int16_t)(cu[4] - cu[5]);                  // numpy: int16 - int16 stays int16
    const int fl = p.flips ? p.flips[n] : 0;
    const bool fh = fl & 1, fv = fl & 2;
    const int B = p.bands, W = p.scene_w[sc];
    const T* src = reinterpret_cast<const T*>(p.scenes) + p.scene_off[sc];
    float* dst = p.out + (size_t)n * p.sn;
    (void)mx;
    if (sizeof(T) == 4 && p.sb == 1 && (B & 3) == 0 && (p.sw & 3) == 0 && (p.sh & 3) == 0 && (p.sn & 3) == 0) {
        const int B4 = B >> 2;
        for (int e = threadIdx.x; e < 81 * B4; e += 256) {
            const int px = e / B4, b4 = e - px * B4;
            const int i = px / 9, j = px - i * 9;
            const int si = fv ? 8 - i : i, sj = fh ? 8 - j : j;
            const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + ((size_t)(h0 + si) * W + (w0 + sj)) * B + 4 * b4);
            float4 o;
            o.x = (float)(((T)v.x - mn) / den); o.y = (float)(((T)v.y - mn) / den);
            o.z = (float)(((T)v.z - mn) / den); o.w = (float)(((T)v.w - mn) / den);
            *reinterpret_cast<float4*>(dst + (size_t)i * p.sh + (size_t)j * p.sw + 4 * b4) = o;
        }
    } else {
        for (int e = threadIdx.x; e < 81 * B; e += 256) {
            const int px = e / B, b = e - px * B;
            const int i = px / 9, j = px - i * 9;
            const int si = fv ? 8 - i : i, sj = fh ? 8 - j : j;
            const T v = src[((size_t)(h0 + si) * W + (w0 + sj)) * B + b];
            dst[(size_t)b * p.sb + (size_t)i * p.sh + (size_t)j * p.sw] = (float)((v - mn) / den);
        }
    }
}

}  // namespace

int hs_cube_gather(const CubeParams& p, hipStream_t s) {
    if (p.N <= 0) return HS_OK;
    if (p.bands <= 0) return HS_EDIMS;
    if (p.scene_f64) hipLaunchKernelGGL(cube_gather_kernel<double>, dim3(p.N), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(cube_gather_kernel<float>, dim3(p.N), dim3(256), 0, s, p);
    return (int)hipGetLastError();
}

HS_UNIT_VARIANT_BITS(loader)
