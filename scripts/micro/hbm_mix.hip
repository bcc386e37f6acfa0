// HBM rates for the traffic mixes the fused kernels produce: read only, write only, copy (1 read : 1 write), 2 reads : 1 write,
// 1 read : 2 writes, with N independent streams per kind (the kernels of this repo touch 5-9 arrays at once).  16 bytes per lane,
// rows of `rowb` bytes walked by workgroups in marching order (workgroup b takes chunk b, b + grid, ...), as the persistent kernels do.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/hbm_mix.hip -o /tmp/hbm_mix && /tmp/hbm_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int NR, int NW, bool NT>
__global__ __launch_bounds__(256) void mix_kernel(const uint4* const* rd, uint4* const* wr, size_t n16, uint4* sink) {
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        uint4 v[NR > 0 ? NR : 1];
#pragma unroll
        for (int r = 0; r < NR; ++r) v[r] = rd[r][i];
#pragma unroll
        for (int r = 0; r < NR; ++r) { acc.x ^= v[r].x; acc.y += v[r].y; acc.z ^= v[r].z; acc.w += v[r].w; }
        uint4 o = acc; o.x += (unsigned)i;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            if constexpr (NT) __builtin_nontemporal_store((u32x4{o.x, o.y, o.z, o.w}), reinterpret_cast<u32x4*>(wr[w]) + i); else wr[w][i] = o;
        }
    }
    if (acc.x == 0x12345678u && acc.y == 42u) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int NR, int NW, bool NT>
void run(const char* name, const uint4* const* rd, uint4* const* wr, size_t n16, uint4* sink, int wgs) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((mix_kernel<NR, NW, NT>), dim3(wgs), dim3(256), 0, 0, rd, wr, n16, sink);
    hipEventRecord(a, 0);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((mix_kernel<NR, NW, NT>), dim3(wgs), dim3(256), 0, 0, rd, wr, n16, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double bytes = (double)n16 * 16 * (NR + NW);
    printf("%-44s wgs %5d  %7.3f ms  %6.2f TB/s\n", name, wgs, ms, bytes / ms * 1e-9);
}

int main() {
    const size_t bytes = (size_t)768 << 20;                 // per array: far beyond the 256 MB Infinity Cache
    const size_t n16 = bytes / 16;
    uint4* arr[6];
    for (int i = 0; i < 6; ++i) { hipMalloc(&arr[i], bytes); hipMemset(arr[i], i + 1, bytes); }
    uint4* sink; hipMalloc(&sink, 8192 * 256 * 16);
    const uint4** rd; uint4** wr;
    hipMalloc(&rd, 6 * sizeof(void*)); hipMalloc(&wr, 6 * sizeof(void*));
    const uint4* hr[6] = {arr[0], arr[1], arr[2], arr[3], arr[4], arr[5]};
    uint4* hw[6] = {arr[3], arr[4], arr[5], arr[0], arr[1], arr[2]};
    hipMemcpy(rd, hr, sizeof(hr), hipMemcpyHostToDevice); hipMemcpy(wr, hw, sizeof(hw), hipMemcpyHostToDevice);
    for (int wgs : {2048, 8192}) {
        run<1, 0, false>("read 1 stream", rd, wr, n16, sink, wgs);
        run<3, 0, false>("read 3 streams", rd, wr, n16, sink, wgs);
        run<0, 1, false>("write 1 stream", rd, wr, n16, sink, wgs);
        run<0, 3, false>("write 3 streams", rd, wr, n16, sink, wgs);
        run<0, 3, true>("write 3 streams, nt", rd, wr, n16, sink, wgs);
        run<1, 1, false>("copy: 1 read + 1 write", rd, wr, n16, sink, wgs);
        run<2, 1, false>("2 reads + 1 write", rd, wr, n16, sink, wgs);
        run<3, 3, false>("3 reads + 3 writes", rd, wr, n16, sink, wgs);
        run<3, 3, true>("3 reads + 3 writes, nt stores", rd, wr, n16, sink, wgs);
        run<1, 2, false>("1 read + 2 writes", rd, wr, n16, sink, wgs);
    }
    return 0;
}
