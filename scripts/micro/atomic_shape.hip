// Microbenchmark: fp32 global atomic adds by ACCESS SHAPE of one wave instruction, the weight-gradient commit pattern:
//   mode 0: 256 contiguous bytes per instruction
//   mode 1: a 16 x 16 MFMA accumulator register as it stands: 4 rows x 64 bytes (row stride = ld floats)
//   mode 2: plain stores, 256 contiguous bytes (the slab form)
// 256 workgroups x 8 waves, each thread 72 values (the fused decoder MLP backward's commit), all workgroups onto the same
// 147-KB matrix (mode 0/1) or each onto its own slab (mode 2).   hipcc --offload-arch=gfx950 -O3 atomic_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(512) void k(float* buf, int ld) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c16 = lane & 15, g = lane >> 4;
    for (int s = 0; s < 72; ++s) {
        const int tile = wave * 18 + s / 4, r = s & 3;              // 144 tiles of 16 x 16 over a [576][64] matrix (ld = 64)
        const int trow = (tile / 4) * 16, tcol = (tile % 4) * 16;
        if (MODE == 0) atomicAdd(buf + (size_t)(wave * 72 + s) * 64 + lane, 1.0f);
        else if (MODE == 1) atomicAdd(buf + (size_t)(trow + g * 4 + r) * ld + tcol + c16, 1.0f);
        else buf[((size_t)blockIdx.x * 72 + s) * 512 + tid] = 1.0f;
    }
}

int main() {
    float* buf; hipMalloc(&buf, (size_t)256 * 72 * 512 * 4); hipMemset(buf, 0, (size_t)256 * 72 * 512 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](int mode) {
        for (int it = 0; it < 12; ++it) {
            if (it == 2) hipEventRecord(a);
            if (mode == 0) k<0><<<256, 512>>>(buf, 64); else if (mode == 1) k<1><<<256, 512>>>(buf, 64); else k<2><<<256, 512>>>(buf, 64);
        }
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
        printf("mode %d: %.1f us per launch (37.7 MB of adds / stores) = %.2f TB/s\n", mode, ms * 1e3, 37.7e6 / (ms * 1e-3) / 1e12);
    };
    run(0); run(1); run(2);
    return 0;
}
