// L2 -> register streaming rate per CU: every workgroup reads the same `bytes`-sized buffer (L2 resident) `reps` times with
// fully coalesced 16-byte loads (1 KB per wave instruction), `depth` loads in flight per wave.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/l2_stream.hip -o /tmp/l2_stream && /tmp/l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* buf, int n16, int reps, uint4* sink) {
    const int tid = threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (int base = 0; base + DEPTH * 256 <= n16; base += DEPTH * 256) {
            uint4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = buf[base + d * 256 + tid];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y += v[d].y; acc.z ^= v[d].z; acc.w += v[d].w; }
        }
    }
    if (acc.x == 0x12345678u && acc.y == 42u) sink[blockIdx.x * 256 + tid] = acc;
}

template <int DEPTH>
float run(const uint4* buf, int n16, int reps, uint4* sink, int wgs) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(wgs), dim3(256), 0, 0, buf, n16, 2, sink);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(wgs), dim3(256), 0, 0, buf, n16, reps, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    const int sizes_kb[] = {256, 1024, 2048};
    uint4 *buf, *sink;
    hipMalloc(&buf, 8 << 20); hipMemset(buf, 1, 8 << 20);
    hipMalloc(&sink, 4096 * 256 * 16);
    for (int kb : sizes_kb) {
        const int n16 = kb * 1024 / 16;
        for (int wpc : {1, 2, 3, 4}) {
            const int wgs = 256 * wpc, reps = 64;
            const double bytes = (double)wgs * reps * kb * 1024.0;
            const float m4 = run<4>(buf, n16, reps, sink, wgs), m8 = run<8>(buf, n16, reps, sink, wgs), m16 = run<16>(buf, n16, reps, sink, wgs);
            printf("buffer %4d KB, %d workgroups/CU (%2d waves/CU): depth 4: %6.1f GB/s/CU  depth 8: %6.1f  depth 16: %6.1f   (aggregate %.1f TB/s at depth 8)\n",
                   kb, wpc, 4 * wpc, bytes / m4 / 1e6 / 256, bytes / m8 / 1e6 / 256, bytes / m16 / 1e6 / 256, bytes / m8 / 1e9);
        }
    }
    return 0;
}
