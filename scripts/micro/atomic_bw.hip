// Microbenchmark: throughput of fp32 global atomic adds when many workgroups reduce into one small buffer
// (the pattern an in-kernel weight-gradient reduction would produce).  hipcc --offload-arch=gfx950 -O3 atomic_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* buf0, int nfloats, int rot, int nbuf) {
    float* buf = buf0 + (size_t)(blockIdx.x % nbuf) * nfloats;       // nbuf partial buffers: contention per address / nbuf
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // start at a per-workgroup rotated offset so that workgroups do not all hit the same line at the same time
    const int per = nfloats / 4;                 // per wave
    const int start = rot ? (int)(((long)blockIdx.x * 4099) % (per / 64)) * 64 : 0;
    for (int i = 0; i < per; i += 64) {
        int o = (i + start) % per;
        int idx;
        if (MODE == 0) idx = wave * per + o + lane;                                   // 256 B contiguous per wave instr
        else idx = wave * per + (o / 64) * 64 + (lane >> 4) * 16 + (lane & 15);       // same thing (placeholder)
        atomicAdd(buf + idx, 1.0f);
    }
}

int main() {
    const int nfloats = 3 * 352 * 128;   // dW1, dW3, dW2 of one encoder block
    float* buf; hipMalloc(&buf, (size_t)nfloats * 4 * 16); hipMemset(buf, 0, (size_t)nfloats * 4 * 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int wgs : {512}) for (int rot : {1}) for (int nbuf : {1, 2, 4, 8, 16}) {
        k<0><<<wgs, 256>>>(buf, nfloats, rot, nbuf); hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 10; ++r) k<0><<<wgs, 256>>>(buf, nfloats, rot, nbuf);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
        printf("wgs=%d rot=%d nbuf=%d: %.1f us, %.1f M atomics, %.2f T atomics/s, %.1f GB/s equivalent\n", wgs, rot, nbuf, ms * 1e3,
               (double)wgs * nfloats / 1e6, (double)wgs * nfloats / ms / 1e9, (double)wgs * nfloats * 4 / ms / 1e6);
    }
    std::vector<float> h(nfloats); hipMemcpy(h.data(), buf, nfloats * 4, hipMemcpyDeviceToHost);
    printf("check: buf[0]=%g buf[last]=%g\n", h[0], h[nfloats - 1]);
    return 0;
}
