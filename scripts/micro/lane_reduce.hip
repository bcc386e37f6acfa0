// Cross-lane reductions without the LDS crossbar: __shfl_xor compiles to ds_bpermute_b32 on gfx950 (an LDS instruction per step);
// DPP row operations (xor 1 / 2 / 4 / 8 inside a 16-lane row) and v_permlane16_swap / v_permlane32_swap (across rows) are plain
// VALU instructions.  Checks the DPP / permlane forms bit for bit against __shfl_xor.   hipcc --offload-arch=gfx950 -O3 lane_reduce.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// (the __builtin_amdgcn_permlane16_swap / 32_swap builtins with both operands derived from one value come back as r[0] + r[0] from
//  hipcc 7.2: the second result is lost — inline asm with two tied registers instead; s_nop covers the VALU-write -> permlane hazard)
__device__ __forceinline__ float x16(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float x32(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__global__ void k(const float* a, float* o) {
    const float v = a[threadIdx.x];
    float d = v; d += dpp_mov<0xB1>(d); o[threadIdx.x] = d; d += dpp_mov<0x4E>(d); o[64 + threadIdx.x] = d;
    d += dpp_mov<0x141>(d); o[128 + threadIdx.x] = d; d += dpp_mov<0x140>(d); o[192 + threadIdx.x] = d;
    d = x16(d); o[256 + threadIdx.x] = d; d = x32(d); o[320 + threadIdx.x] = d;
    float s = v; s += __shfl_xor(s, 1, 64); o[384 + threadIdx.x] = s; s += __shfl_xor(s, 2, 64); o[448 + threadIdx.x] = s;
    s += __shfl_xor(s, 4, 64); o[512 + threadIdx.x] = s; s += __shfl_xor(s, 8, 64); o[576 + threadIdx.x] = s;
    s += __shfl_xor(s, 16, 64); o[640 + threadIdx.x] = s; s += __shfl_xor(s, 32, 64); o[704 + threadIdx.x] = s;
}
int main() {
    float h[64], r[768]; float *a, *o;
    for (int i = 0; i < 64; ++i) h[i] = 1.0f / (1 + i) + (i % 7) * 0.37f;
    (void)hipMalloc(&a, sizeof h); (void)hipMalloc(&o, sizeof r); (void)hipMemcpy(a, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, o); (void)hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int s = 0; s < 6; ++s) { int b = 0; for (int i = 0; i < 64; ++i) b += memcmp(&r[s * 64 + i], &r[384 + s * 64 + i], 4) != 0; printf("step xor %2d: %d lanes differ\n", 1 << s, b); bad += b; }
    printf(bad ? "MISMATCH\n" : "all six steps bit-identical to __shfl_xor\n");
    return bad != 0;
}
