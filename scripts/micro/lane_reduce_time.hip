// Cost of one cross-row exchange step on gfx950: v_permlane16_swap (+ s_nop 1) against ds_bpermute (__shfl_xor 16) and a DPP row step,
// as dependent chains (latency) and as 4 independent chains per wave (throughput), 1 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float x16s(float v) { float a = v, b = v; asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); return a + b; }
__device__ __forceinline__ float x16b(float v) { return v + __shfl_xor(v, 16, 64); }
__device__ __forceinline__ float x8d(float v) { return v + dpp_mov<0x140>(v); }
template <int MODE> __global__ void k(float* o, long long* cyc, int iters) {
    float v0 = threadIdx.x * 0.001f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { v0 = x16s(v0) * 0.5f; v1 = x16s(v1) * 0.5f; v2 = x16s(v2) * 0.5f; v3 = x16s(v3) * 0.5f; }
        if (MODE == 1) { v0 = x16b(v0) * 0.5f; v1 = x16b(v1) * 0.5f; v2 = x16b(v2) * 0.5f; v3 = x16b(v3) * 0.5f; }
        if (MODE == 2) { v0 = x8d(v0) * 0.5f; v1 = x8d(v1) * 0.5f; v2 = x8d(v2) * 0.5f; v3 = x8d(v3) * 0.5f; }
    }
    const long long t1 = __builtin_readcyclecounter();
    o[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* o; long long* c; (void)hipMalloc(&o, 1 << 22); (void)hipMalloc(&c, 8);
    const char* names[3] = {"v_permlane16_swap + add", "ds_bpermute (__shfl_xor) + add", "DPP row_mirror add"};
    for (int waves = 1; waves <= 8; waves *= 8)
        for (int m = 0; m < 3; ++m) {
            const int iters = 2000; long long h = 0;
            const dim3 grid(256), blk(256 * waves);          // 4 SIMDs x `waves` waves per CU (one workgroup per CU)
            for (int rep = 0; rep < 2; ++rep) {
                if (m == 0) hipLaunchKernelGGL(k<0>, grid, blk, 0, 0, o, c, iters);
                if (m == 1) hipLaunchKernelGGL(k<1>, grid, blk, 0, 0, o, c, iters);
                if (m == 2) hipLaunchKernelGGL(k<2>, grid, blk, 0, 0, o, c, iters);
                (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
            }
            printf("%d wave(s)/SIMD  %-32s %6.1f cycles per step (4 independent chains per wave: %.1f per chain step)\n", waves, names[m], (double)h / iters / 4, (double)h / iters / 4);
        }
}
