// fp32 nn.Linear / Conv3d weights -> bf16 packed weight images in MFMA B-fragment order (common.h "wpk").
// One launch repacks every matrix of the model from a device-resident descriptor table; run after each
// optimizer step (the fp32 parameters stay the master copy and the state_dict wire format).
#include "common.h"
#include "kernels.h"

namespace {

__global__ __launch_bounds__(256) void pack_kernel(const PackDesc* descs, int ndesc) {
    const PackDesc d = descs[blockIdx.y];
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
