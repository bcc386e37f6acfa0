// Internal launcher interface between the kernel translation units and the C ABI (api.hip).
// Parameter blocks are the public C structs of include/hsimae_hip.h.
#pragma once
#include "common.h"
#include "../../include/hsimae_hip.h"

enum AKind { A_BF16 = HSIMAE_A_BF16, A_F32 = HSIMAE_A_F32, A_F32_LN = HSIMAE_A_F32_LN };
enum Epi { E_BF16 = HSIMAE_E_BF16, E_F32 = HSIMAE_E_F32, E_RES_F32 = HSIMAE_E_RES_F32,
           E_POS_F32 = HSIMAE_E_POS_F32, E_SWIGLU = HSIMAE_E_SWIGLU, E_SWIGLU_BWD = HSIMAE_E_SWIGLU_BWD };

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

int hs_gemm(const GemmParams& p, int akind, int epi, hipStream_t s);
int hs_pack(const PackDesc* descs_dev, int ndesc, int max_elems, hipStream_t s);
int hs_attn_fwd(const AttnParams& p, hipStream_t s);
int hs_attn_bwd(const AttnParams& p, hipStream_t s);
int hs_wgrad(const WgradParams& p, hipStream_t s);
int hs_ln_bwd(const LnBwdParams& p, hipStream_t s);
int hs_ln_fwd(const float* x, const float* gamma, const float* beta, float* out, int M, int d, hipStream_t s);
int hs_mask(const MaskParams& p, hipStream_t s);
int hs_patch_gather(const PatchParams& p, hipStream_t s);
int hs_assemble_fwd(const AssembleParams& p, hipStream_t s);
int hs_assemble_bwd(const AssembleParams& p, hipStream_t s);
int hs_loss(const LossParams& p, hipStream_t s);
int hs_loss_partials(int N, int T);
int hs_add2(const float* a, const float* b, float* out, int64_t n, hipStream_t s);
