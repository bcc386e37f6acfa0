// Does the LAYOUT of the weight-gradient operands cost HBM efficiency?  (round 5)
// enc_mlp_bwd stores g / dh1 / dh3 per 64-column hidden chunk as 128-byte pieces of rows whose pitch is 704 B (g) or 1408 B
// (dh1|dh3); wgrad_dma reads 256-byte pieces of those rows (a 128-column tile slice).  This micro-benchmark moves the same bytes
//   (a) in that row-major piece order, and
//   (b) "planar": one [M][64]-column plane per chunk (128-byte rows, so a panel's piece block is 6 KB contiguous and a tile
//       slice is two sequential streams),
// with nothing else in the kernel.  Arrays are far larger than the 256 MB Infinity Cache.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/hbm_stride.hip -o /tmp/hbm_stride && /tmp/hbm_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int R = 48, HPB = 704, NCH = 6;        // panel rows, hidden row bytes (352 bf16), chunks (the last one half full)

// producer: one workgroup per 48-row panel, chunk loop, 3 arrays (g, dh1, dh3), 16 B per lane
template <bool PLANAR, bool NT>
__global__ __launch_bounds__(256) void store_kernel(char* g, char* dh13, size_t M) {
    const size_t row0 = (size_t)blockIdx.x * R;
    for (int c = 0; c < NCH; ++c) {
        const int ncolb = (c * 128 + 128 <= HPB) ? 128 : HPB - c * 128;     // 128 or 64 bytes
        for (int pc = threadIdx.x; pc < R * 8; pc += 256) {
            const int row = pc >> 3, k = (pc & 7) * 16;
            if (row0 + row < M && k < ncolb) {
                const size_t gr = row0 + row;
                const u32x4 v = {(unsigned)gr, (unsigned)c, (unsigned)k, 7u};
                char *p0, *p1, *p2;
                if (PLANAR) {
                    p0 = g + ((size_t)c * M + gr) * 128 + k;
                    p1 = dh13 + ((size_t)c * M + gr) * 128 + k;
                    p2 = dh13 + ((size_t)(NCH + c) * M + gr) * 128 + k;
                } else {
                    p0 = g + gr * HPB + c * 128 + k;
                    p1 = dh13 + gr * 2 * HPB + c * 128 + k;
                    p2 = dh13 + gr * 2 * HPB + HPB + c * 128 + k;
                }
                if (NT) {
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p0));
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p1));
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p2));
                } else {
                    *reinterpret_cast<u32x4*>(p0) = v; *reinterpret_cast<u32x4*>(p1) = v; *reinterpret_cast<u32x4*>(p2) = v;
                }
            }
        }
        __syncthreads();
    }
}

// consumer: workgroup (tile t of 3, row slice ms) reads 256 B of dO (row pitch `pitch`, column offset t * 256) and 256 B of A
// (contiguous 256-byte rows) per row; PLANAR: dO tile slice = two 128-byte-row planes.  16 lanes per 256-B piece, 8 rows per
// wave-instruction pair, 4 instructions in flight per lane.
template <bool PLANAR>
__global__ __launch_bounds__(256) void load_kernel(const char* dO, const char* A, size_t M, int pitch, int msplit, uint4* sink) {
    const int t = blockIdx.x % 3, ms = blockIdx.x / 3;
    const size_t rows = (M + msplit - 1) / msplit, r0 = (size_t)ms * rows, r1 = r0 + rows < M ? r0 + rows : M;
    const int l16 = threadIdx.x & 15, rsub = threadIdx.x >> 4;      // 16 rows per pass of the workgroup
    uint4 acc = {0, 0, 0, 0};
    for (size_t r = r0 + rsub; r < r1; r += 64) {
        uint4 v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t rr = r + 16 * i < r1 ? r + 16 * i : r;
            const char* pd;
            if (PLANAR) pd = dO + ((size_t)(2 * t + (l16 >> 3)) * M + rr) * 128 + (l16 & 7) * 16;
            else pd = dO + rr * pitch + t * 256 + l16 * 16;
            v[2 * i] = *reinterpret_cast<const uint4*>(pd);
            v[2 * i + 1] = *reinterpret_cast<const uint4*>(A + rr * 256 + l16 * 16);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc.x ^= v[i].x; acc.y += v[i].y; acc.z ^= v[i].z; acc.w += v[i].w; }
    }
    if (acc.x == 0x12345678u && acc.y == 42u) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <class F>
float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipEventRecord(a, 0);
    for (int it = 0; it < 5; ++it) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    const size_t M = (size_t)110592 * 8;            // 8 x the C2 encoder rows: g = 623 MB, dh1|dh3 = 1.25 GB
    char *g, *dh13, *A; uint4* sink;
    // (sized for the planar form: 6 planes of 128 B per row)
    hipMalloc(&g, M * NCH * 128 + 4096); hipMalloc(&dh13, M * 2 * NCH * 128 + 4096); hipMalloc(&A, M * 256); hipMalloc(&sink, 8192 * 256 * 16);
    hipMemset(g, 1, M * NCH * 128); hipMemset(dh13, 2, M * 2 * NCH * 128); hipMemset(A, 3, M * 256);
    const int panels = (int)((M + R - 1) / R);
    const double wbytes = (double)M * HPB * 3;
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL((store_kernel<false, false>), dim3(panels), dim3(256), 0, 0, g, dh13, M); });
    printf("store g|dh1|dh3 row-major pieces (as enc_mlp_bwd)      %7.3f ms  %5.2f TB/s\n", ms, wbytes / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL((store_kernel<false, true>), dim3(panels), dim3(256), 0, 0, g, dh13, M); });
    printf("store g|dh1|dh3 row-major pieces, nt                    %7.3f ms  %5.2f TB/s\n", ms, wbytes / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL((store_kernel<true, false>), dim3(panels), dim3(256), 0, 0, g, dh13, M); });
    printf("store g|dh1|dh3 planar (one [M][64] plane per chunk)    %7.3f ms  %5.2f TB/s\n", ms, wbytes / ms * 1e-9);
    ms = timeit([&] { hipLaunchKernelGGL((store_kernel<true, true>), dim3(panels), dim3(256), 0, 0, g, dh13, M); });
    printf("store g|dh1|dh3 planar, nt                              %7.3f ms  %5.2f TB/s\n", ms, wbytes / ms * 1e-9);
    for (int wgs : {768, 1536}) {
        const int msplit = wgs / 3;
        const double rbytes = (double)M * 512 * 3;       // 3 tiles x (256 B of dO + 256 B of A) per row (A is re-read by the 3 tiles: L2 / MALL)
        ms = timeit([&] { hipLaunchKernelGGL((load_kernel<false>), dim3(wgs), dim3(256), 0, 0, dh13, A, M, 2 * HPB, msplit, sink); });
        printf("load 256-B tile slices, pitch 1408 (dh1 of dh1|dh3)  wgs %4d  %7.3f ms  %5.2f TB/s (fetch)\n", wgs, ms, rbytes / ms * 1e-9);
        ms = timeit([&] { hipLaunchKernelGGL((load_kernel<false>), dim3(wgs), dim3(256), 0, 0, g, A, M, HPB, msplit, sink); });
        printf("load 256-B tile slices, pitch 704 (g)                wgs %4d  %7.3f ms  %5.2f TB/s (fetch)\n", wgs, ms, rbytes / ms * 1e-9);
        ms = timeit([&] { hipLaunchKernelGGL((load_kernel<true>), dim3(wgs), dim3(256), 0, 0, dh13, A, M, 0, msplit, sink); });
        printf("load planar (two 128-B-row planes per tile slice)    wgs %4d  %7.3f ms  %5.2f TB/s (fetch)\n", wgs, ms, rbytes / ms * 1e-9);
    }
    return 0;
}
