// Do waves of one CU that read the SAME 1-KB pieces of an L2-resident stream at (nearly) the same time pay for one L2 -> L1 transfer
// or for one each?  (The row-panel kernels stream their weight fragments from L2 once per panel and workgroup; two panels that
// walked the stream together could share it if the vector L1 merges the requests.)
//   private: wave w of a workgroup reads pieces w, w + W, ... of the buffer (every piece once per workgroup)
//   shared : every wave reads every piece (W x the register-side bytes, the same L2-side bytes if L1 merges)
// Register-side rate per CU is reported for both.  hipcc --offload-arch=gfx950 -O3 scripts/micro/l1_share.hip -o /tmp/l1_share
#include <hip/hip_runtime.h>
#include <cstdio>

template <int DEPTH, bool SHARED>
__global__ __launch_bounds__(512) void k(const uint4* buf, int pieces, int reps, uint4* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
    uint4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        const int step = SHARED ? 1 : W, first = SHARED ? 0 : wave;
        for (int p = first; p + (DEPTH - 1) * step < pieces; p += DEPTH * step) {
            uint4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = buf[(size_t)(p + d * step) * 64 + lane];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y += v[d].y; acc.z ^= v[d].z; acc.w += v[d].w; }
        }
    }
    if (acc.x == 0x12345678u && acc.y == 42u) sink[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int DEPTH, bool SHARED>
float run(const uint4* buf, int pieces, int reps, uint4* sink, int wgs, int threads) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<DEPTH, SHARED>), dim3(wgs), dim3(threads), 0, 0, buf, pieces, 2, sink);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<DEPTH, SHARED>), dim3(wgs), dim3(threads), 0, 0, buf, pieces, reps, sink);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    uint4 *buf, *sink;
    hipMalloc(&buf, 8 << 20); hipMemset(buf, 1, 8 << 20);
    hipMalloc(&sink, 4096 * 512 * 16);
    for (int kb : {512, 1536}) {
        const int pieces = kb;                                   // 1-KB pieces
        for (int cfg = 0; cfg < 4; ++cfg) {
            const int threads = cfg == 0 ? 256 : (cfg == 1 ? 512 : (cfg == 2 ? 256 : 256));
            const int wpc = cfg == 0 ? 1 : (cfg == 1 ? 1 : (cfg == 2 ? 2 : 3));
            const int wgs = 256 * wpc, reps = 32, W = threads / 64;
            const float mp = run<4, false>(buf, pieces, reps, sink, wgs, threads);
            const float ms = run<4, true>(buf, pieces, reps, sink, wgs, threads);
            const double bp = (double)wgs * reps * kb * 1024.0, bs = bp * W;       // register-side bytes
            printf("stream %4d KB, %d workgroup(s)/CU x %d waves: private %6.1f GB/s/CU (%5.1f TB/s)   shared %6.1f GB/s/CU register-side (%5.1f TB/s), %5.2fx the time for %dx the bytes\n",
                   kb, wpc, W, bp / mp / 1e6 / 256, bp / mp / 1e9, bs / ms / 1e6 / 256, bs / ms / 1e9, ms / mp, W);
        }
    }
    return 0;
}
