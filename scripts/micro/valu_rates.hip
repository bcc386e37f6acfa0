// Issue rates of a few VALU instructions on gfx950, cycles per wave-instruction with 8 independent chains per wave and 1 wave per SIMD:
// v_mul_f32, v_pk_mul_f32 (2 results), v_exp_f32, v_rcp_f32, v_cvt_pk_bf16_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __global__ void k(float* o, long long* cyc, int iters) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(0.999f));
            if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            if (MODE == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 a = {v[i], v[i + 1]}, b = {0.999f, 0.999f};
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
                v[i] = a[0]; v[i + 1] = a[1];
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float* o; long long* c; (void)hipMalloc(&o, 1 << 22); (void)hipMalloc(&c, 8);
    const char* names[4] = {"v_mul_f32", "v_exp_f32", "v_rcp_f32", "v_pk_mul_f32 (per instruction = 2 results)"};
    const int per[4] = {8, 8, 8, 4};
    for (int m = 0; m < 4; ++m) {
        const int iters = 4000; long long h = 0;
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, o, c, iters);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, o, c, iters);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, o, c, iters);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, o, c, iters);
            (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        }
        printf("%-48s %5.2f cycles per wave-instruction\n", names[m], (double)h / iters / per[m]);
    }
}
