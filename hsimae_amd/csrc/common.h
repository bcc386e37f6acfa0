// Shared device helpers for the HSIMAE gfx950 kernels (wave = 64, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define HS_WAVE 64

// Variant bits of hsimae_build_info() (include/hsimae_hip.h): every compile-time switch that makes a kernel compute wrong results
// on purpose (timing ablations) or instruments it.  Each translation unit reports the bits IT was compiled with
// (HS_UNIT_VARIANT_BITS at its end), api.hip ORs them; tests/test_host_cpu.py greps the sources for HS_ABL_* / HS_EXP_* /
// HS_EXPERIMENT_* / HS_PHASE_TIMING tokens and fails if one is missing from this table.
#define HS_VARIANT_TABLE(X) \
    X(0, HS_ABL_DW2) X(1, HS_ABL_DEC_REREAD) X(2, HS_ABL_FWD_NOLOAD) X(3, HS_ABL_WSTREAM) X(4, HS_ABL_FWD_NOSTORE) \
    X(5, HS_EXP_NO_COMMIT) X(6, HS_EXPERIMENT_NOEXP) X(7, HS_PHASE_TIMING)
static inline unsigned hs_variant_bits() {
    unsigned b = 0;
#ifdef HS_ABL_DW2
    b |= 1u << 0;
#endif
#ifdef HS_ABL_DEC_REREAD
    b |= 1u << 1;
#endif
#ifdef HS_ABL_FWD_NOLOAD
    b |= 1u << 2;
#endif
#ifdef HS_ABL_WSTREAM
    b |= 1u << 3;
#endif
#ifdef HS_ABL_FWD_NOSTORE
    b |= 1u << 4;
#endif
#ifdef HS_EXP_NO_COMMIT
    b |= 1u << 5;
#endif
#ifdef HS_EXPERIMENT_NOEXP
    b |= 1u << 6;
#endif
#ifdef HS_PHASE_TIMING
    b |= 1u << 7;
#endif
    return b;
}
#define HS_UNIT_VARIANT_BITS(unit) extern "C" unsigned hs_variant_bits_##unit() { return hs_variant_bits(); }

// Error codes of the C ABI (include/hsimae_hip.h)
#define HS_OK 0
#define HS_EDIMS (-1)
#define HS_EUNSUPPORTED (-2)
#define HS_EALIGN (-3)

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    // D[16x16] += A[16x32] * B[32x16].  lane l: A[row l&15][k 8(l>>4)+j], B[k 8(l>>4)+j][col l&15];
    // D: col = l&15, row = 4*(l>>4)+r.
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Workgroup barrier that orders LDS traffic only.  `__syncthreads()` also emits `s_waitcnt vmcnt(0)`, which on
// gfx950 counts stores as well as loads: every barrier then drains the prefetched weight fragments and the
// epilogue's HBM stores (measured: the fused kernels spent 50-70 % of their wave-cycles parked, profiles/).
// Use this wherever the barrier only publishes LDS data to the other waves of the workgroup.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ bf16x8 zero8() {
    u32x4 z = {0u, 0u, 0u, 0u};
    return __builtin_bit_cast(bf16x8, z);
}

__device__ __forceinline__ bf16x8 cvt8(const float* v) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (bf16_t)v[i];
    return r;
}

__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }

// a * sigmoid(a) with the reciprocal taken as v_rcp_f32 + one Newton step (<= 1 ulp): the forward kernels' SiLU.
// A plain v_rcp (1 ulp, biased) moved the loss by 1e-4 at stress weights; IEEE division costs ~10 VALU slots per
// element (scale / fmas / fixup), which made the gate the largest VALU item of the fused forward kernels.
__device__ __forceinline__ float silu_nr(float a) {
    const float d = 1.f + __expf(-a);
    float r = __builtin_amdgcn_rcpf(d);
    r = fmaf(fmaf(-d, r, 1.f), r, r);
    return a * r;
}

// Cross-lane reductions without the LDS crossbar.  `__shfl_xor` compiles to ds_bpermute_b32 on gfx950: an LDS instruction (and an
// lgkmcnt round trip) per step, in kernels whose busiest unit is the LDS pipe (132 of them in dec_bwd_attn_kernel alone).  Inside a
// 16-lane row the same sums are DPP modifiers on the v_add / v_max itself; across rows gfx950 has v_permlane16_swap / 32_swap.
// scripts/micro/lane_reduce.hip checks every step bit for bit against __shfl_xor on the GPU.
//   quad_perm [1,0,3,2] = lane ^ 1, quad_perm [2,3,0,1] = lane ^ 2; row_half_mirror (lane 7 - i of the half row) and row_mirror
//   (lane 15 - i) stand in for ^ 4 and ^ 8 once the lanes of a quad / half row hold the same partial sum — so the steps run ascending.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// rows r and r ^ 1 (lanes ^ 16) / halves (lanes ^ 32) exchanged in place: a + b afterwards is v + partner in every lane.
// (The __builtin_amdgcn_permlane16_swap / 32_swap builtins lose their second result in hipcc 7.2 when both operands derive
//  from one value — inline asm with two tied registers; s_nop covers the VALU-write -> permlane read hazard.)
__device__ __forceinline__ void swap_rows16(float& a, float& b) { asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap_rows32(float& a, float& b) { asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }

struct OpSum { static __device__ __forceinline__ float f(float a, float b) { return a + b; } };
struct OpMax { static __device__ __forceinline__ float f(float a, float b) { return fmaxf(a, b); } };

// reduction over the aligned group of N adjacent lanes (N = 2 .. 64), result in every lane of the group
template <int N, class Op>
__device__ __forceinline__ float lanes_reduce(float v) {
    static_assert(N == 2 || N == 4 || N == 8 || N == 16 || N == 32 || N == 64, "group size");
    v = Op::f(v, dpp_mov<0xB1>(v));
    if constexpr (N >= 4) v = Op::f(v, dpp_mov<0x4E>(v));
    if constexpr (N >= 8) v = Op::f(v, dpp_mov<0x141>(v));
    if constexpr (N >= 16) v = Op::f(v, dpp_mov<0x140>(v));
    if constexpr (N >= 32) { float a = v, b = v; swap_rows16(a, b); v = Op::f(a, b); }
    if constexpr (N >= 64) { float a = v, b = v; swap_rows32(a, b); v = Op::f(a, b); }
    return v;
}
template <int N> __device__ __forceinline__ float lanes_sum(float v) { return lanes_reduce<N, OpSum>(v); }
template <int N> __device__ __forceinline__ float lanes_max(float v) { return lanes_reduce<N, OpMax>(v); }
// reduction over the 4 lanes that share lane & 15 (the four 16-lane rows of the wave)
template <class Op>
__device__ __forceinline__ float rows_reduce(float v) {
    float a = v, b = v; swap_rows16(a, b); v = Op::f(a, b);
    a = v; b = v; swap_rows32(a, b); return Op::f(a, b);
}
__device__ __forceinline__ float rows_sum(float v) { return rows_reduce<OpSum>(v); }
__device__ __forceinline__ float rows_max(float v) { return rows_reduce<OpMax>(v); }

__device__ __forceinline__ float wave_sum(float v) { return lanes_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return lanes_max<64>(v); }

// Gradient commits.  Default: fp32 global atomics (summation order varies from run to run in the last bits).  Deterministic
// mode (hsimae_io.det_acc): every addend is converted to 64-bit fixed point (scale 2^44: 5.7e-14 resolution, +-5.2e5 range)
// and added with an INTEGER atomic into a shadow buffer indexed like the gradient buffer — integer addition is
// associative, so the sum does not depend on the order in which the workgroups arrive; the shadow is converted back to
// fp32 once per reported range (det_convert_kernel).  `base` = the flat gradient buffer the pointers point into.
struct HsDet { const float* base; long long* acc; };
#define HS_DET_SCALE 17592186044416.0f         /* 2^44 */
__device__ __forceinline__ void hs_gadd(const HsDet& d, float* ptr, float v) {
#ifdef HS_EXP_NO_COMMIT        /* timing experiment: what the gradient commits (global float atomics) cost each kernel */
    if (v != 12345.678f) return;
#endif
    if (d.acc) {
        // NaN / Inf / out-of-range addends must not vanish in the integer sum (__float2ll_rn(NaN) = 0, large values
        // saturate): poison the fp32 slot instead — it was zero-filled before the backward and det_convert_kernel ADDS the
        // converted sum onto it, so the gradient comes out NaN as it does on the fp32-atomics path
        if (!(fabsf(v) < 2.6e5f)) { *ptr = __builtin_nanf(""); return; }
        const long long q = __float2ll_rn(v * HS_DET_SCALE);
        atomicAdd(reinterpret_cast<unsigned long long*>(d.acc + (ptr - d.base)), static_cast<unsigned long long>(q));
    } else {
        atomicAdd(ptr, v);
    }
}

// Packed weight image ("wpk"): W[N][K] (row-major, K contiguous) zero-padded to NT=ceil(N/16)
// n-tiles and KS=ceil(K/32) k-steps, stored in MFMA B-fragment order:
//   element index ((nt*KS + ks)*64 + lane)*8 + j  holds  W[nt*16 + (lane&15)][ks*32 + 8*(lane>>4) + j]
// so one wave-instruction of 16 B per lane reads one 1-KiB fragment fully coalesced.
__host__ __device__ inline int64_t wpk_elems(int N, int K) {
    return (int64_t)((N + 15) / 16) * ((K + 31) / 32) * 512;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Streaming stores (write-once tensors that are read much later, by another kernel): `nt` stores do not allocate in
// L2 / Infinity Cache, which keeps the lines the next kernels re-read resident.  HS_NT(enabled, ptr, value).
#define HS_NT(on, ptr, val) do { if constexpr (on) __builtin_nontemporal_store((val), (ptr)); else *(ptr) = (val); } while (0)

