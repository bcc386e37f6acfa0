// Internal launcher interface between the kernel translation units and the C ABI (api.hip).
// Parameter blocks are the public C structs of include/hsimae_hip.h.
#pragma once
#include "common.h"
#include "../../include/hsimae_hip.h"

enum AKind { A_BF16 = HSIMAE_A_BF16, A_F32 = HSIMAE_A_F32, A_F32_LN = HSIMAE_A_F32_LN };
enum Epi { E_BF16 = HSIMAE_E_BF16, E_F32 = HSIMAE_E_F32, E_RES_F32 = HSIMAE_E_RES_F32,
           E_POS_F32 = HSIMAE_E_POS_F32, E_SWIGLU = HSIMAE_E_SWIGLU, E_SWIGLU_BWD = HSIMAE_E_SWIGLU_BWD,
           E_LN_BWD = HSIMAE_E_LN_BWD };

typedef hsimae_gemm_params GemmParams;
typedef hsimae_pack_desc PackDesc;
typedef hsimae_attn_params AttnParams;
typedef hsimae_wgrad_task WgradTask;
typedef hsimae_wgrad_params WgradParams;
typedef hsimae_lnbwd_params LnBwdParams;
typedef hsimae_mask_params MaskParams;
typedef hsimae_patch_params PatchParams;
typedef hsimae_assemble_params AssembleParams;
typedef hsimae_loss_params LossParams;
typedef hsimae_cube_params CubeParams;

int hs_gemm(const GemmParams& p, int akind, int epi, hipStream_t s);
int hs_gemm_tiled(const GemmParams& p, int akind, int epi, int bm, int kc, hipStream_t s);   // tile sweep hook
int hs_pack(const PackDesc* descs_dev, int ndesc, int max_elems, hipStream_t s);
int hs_attn_fwd(const AttnParams& p, hipStream_t s);
int hs_attn_bwd(const AttnParams& p, hipStream_t s);
bool hs_attn_proj_fusable(const AttnParams& p);      // shape predicate of the fused attention half (d = 128, 8 heads of 16, Ts <= 32, unpadded rows)
int hs_wgrad(const WgradParams& p, hipStream_t s);
int hs_ln_bwd(const LnBwdParams& p, hipStream_t s);
int hs_ln_fwd(const float* x, const float* gamma, const float* beta, float* out, int M, int d, hipStream_t s, int ldx = 0, int ldo = 0);
int hs_mask(const MaskParams& p, hipStream_t s);
int hs_patch_gather(const PatchParams& p, hipStream_t s);
int hs_assemble_fwd(const AssembleParams& p, hipStream_t s);
int hs_assemble_bwd(const AssembleParams& p, hipStream_t s);
int hs_loss(const LossParams& p, hipStream_t s);
int hs_loss_partials(int N, int T);
int hs_cube_gather(const CubeParams& p, hipStream_t s);
int hs_agg_pool(const float* latent, float* pooled, int N, int T, int L, int D, hipStream_t s);
int hs_head_bwd(const float* g, const float* pooled, const float* w, float* gw, float* gb, float* dlat, int N, int C, int T, int L,
                int D, hipStream_t s);
int hs_rows_to_bf16(const float* src, hs_bf16* dst, int64_t rows, int d, const float* rowscale, hipStream_t s);
int hs_rows_pad_bf16(const float* src, hs_bf16* dst, int64_t rows, int cols, int ldd, hipStream_t s);
int hs_det_convert(const int64_t* acc, float* g, int64_t n, hipStream_t s);
int hs_add2(const float* a, const float* b, float* out, int64_t n, hipStream_t s);

// ------------------------------------------------------------------ fused_dec.hip (decoder Block, one workgroup per sample)
struct DecBlockPtrs {
    const float *n1w, *n1b, *bqkv, *pb, *n2w, *n2b, *w1b, *w3b, *w2b;
    const bf16_t *qkv, *p, *w1, *w3, *w2, *qkvT, *pT, *w13T, *w2T;
    const float *qf, *kf, *vf, *pf, *w1f, *w3f;    // fp32 row-major weights [out][in] (the backward stages them in LDS as bf16)
    int h;
};
struct DecBlockGrads {
    float *n1w, *n1b, *qw, *qb, *kw, *kb, *vw, *vb, *pw, *pb, *n2w, *n2b, *w1w, *w1b, *w2w, *w2b, *w3w, *w3b;
    HsDet det;                    // deterministic commits (common.h); {nullptr, nullptr} = fp32 atomics
};
bool hs_dec_fused_supported(int d, int heads, int hidden, int Ts);
int hs_dec_block_fwd(const float* x, float* x1, float* x2, hs_bf16* o, float* lse, int nsamples, int Ts,
                     const DecBlockPtrs& bp, hipStream_t s);
int hs_dec_attn_fwd(const float* x, float* x1, hs_bf16* o, float* lse, int nsamples, int Ts, const DecBlockPtrs& bp, hipStream_t s);
// slab: NULL = the block's gradients are committed with float atomics; else >= kDecSlabFloats floats of scratch
// (plan.h slab_bytes(); C ABI: hsimae_dec_block_slab_floats()): per-workgroup partials, summed into the gradients by one
// reduce launch per block.  256 workgroups x (104 in-register dW values x 512 threads + 2112 bias / LayerNorm sums).
constexpr long long kDecSlabFloats = HSIMAE_DEC_BLOCK_SLAB_FLOATS;
int hs_dec_block_bwd(const float* x, const float* x1, const float* dy, float* dx1_tmp, float* dx, const hs_bf16* o,
                     const float* lse, int nsamples, int Ts, const DecBlockPtrs& bp, const DecBlockGrads& g, hipStream_t s,
                     float* slab = nullptr);

// ------------------------------------------------------------------ fused_enc.hip (MLP half of an encoder Block, D = 128)
struct EncMlpPtrs {
    const float *n2w, *n2b, *w1b, *w3b, *w2b;
    const bf16_t *w1, *w3, *w2, *w2T, *w13T;
    int h;
};
bool hs_attn_block_fusable(int d, int heads, int Ts);
bool hs_attn_block_bwd_fusable(int d, int heads, int Ts);
int hs_attn_block_bwd(const hs_bf16* qkv, const hs_bf16* u, const hs_bf16* wqkv, const float* bqkv, const hs_bf16* o, const float* lse,
                      const hs_bf16* dx1b, const float* dx1, const float* x, const float* gamma, const hs_bf16* wpT, const hs_bf16* wqkvT,
                      hs_bf16* dqkv, float* dx, float* dgamma, float* dbeta, const float* det_base, long long* det_acc, int Ts,
                      int nsamples, int mode, int len_l, int accumulate, hipStream_t s);
int hs_attn_block_fwd(const float* x, const float* n1w, const float* n1b, const hs_bf16* wqkv, const float* bqkv, const hs_bf16* wp,
                      const float* pb, hs_bf16* u, hs_bf16* qkv, hs_bf16* o, float* lse, float* x1, const float* rowscale, int Ts,
                      int nsamples, int mode, int len_l, hipStream_t s);
// the same half at D = 256 (16 heads of 16, <= 32 tokens): attn_wide.hip.  HSIMAE_FUSED_ATTN_BLOCK256=0 disables (api.hip SC_*).
bool hs_attn_block256_fusable(int d, int heads, int Ts, int nsamples);
int hs_attn_block256_fwd(const float* x, const float* n1w, const float* n1b, const hs_bf16* wqkv, const float* bqkv, const hs_bf16* wp,
                         const float* pb, hs_bf16* u, hs_bf16* qkv, hs_bf16* o, float* lse, float* x1, const float* rowscale, int Ts,
                         int nsamples, int mode, int len_l, hipStream_t s);
// ... and its backward (dO + attention backward + du + LayerNorm-1 backward, round 5).  HSIMAE_FUSED_ATTN_BLOCK256_BWD=0 disables.
bool hs_attn_block256_bwd_fusable(int d, int heads, int Ts, int nsamples);
int hs_attn_block256_bwd(const hs_bf16* qkv, const float* lse, const hs_bf16* dx1b, const float* dx1, const float* x, const float* gamma,
                         const hs_bf16* wpT, const hs_bf16* wqkvT, hs_bf16* dqkv, float* dx, float* dgamma, float* dbeta,
                         const float* det_base, long long* det_acc, int Ts, int nsamples, int mode, int len_l, int accumulate,
                         hipStream_t s);
bool hs_enc_mlp_fused_supported(int d, int hidden);
int hs_enc_mlp_fwd(const float* x1, const float* res2, float* x2, int M, int d, const EncMlpPtrs& b, hipStream_t s,
                   const float* rowscale = nullptr);
int hs_enc_mlp_bwd(const float* x1, const float* dy, float* dx1, hs_bf16* u2, hs_bf16* dh13, hs_bf16* g, hs_bf16* dyb,
                   hs_bf16* dx1b, int M, int d, const EncMlpPtrs& b, float* g_n2w, float* g_n2b, hipStream_t s,
                   const float* rs_mlp = nullptr, const float* rs_attn = nullptr, HsDet det = HsDet{nullptr, nullptr}, int plane_rows = 0);

int hs_adamw(float* p, const float* g, float* m, float* v, const unsigned char* group, int64_t n, float lr, float b1, float b2,
             float eps, float wd, int step, hipStream_t s);
