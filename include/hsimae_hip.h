/* libhsimae_hip.so — C ABI of the MI355X (gfx950) kernels behind the HSIMAE pretraining path.
 *
 * The reference (Ryan21wy/HSIMAE) has no native code: its "operator API" for this path is the
 * `HSIMAE` nn.Module (Models.py:309-634) whose forward dispatches ~700 ATen kernels and whose
 * backward is autograd.  A reference-side integration binds the entry points below with ctypes
 *  in place of those ATen calls.  Every entry point names the reference
 * lines it replaces.
 *
 * Contract
 *  - every pointer is device memory owned by the caller (PyTorch tensors); the library never
 *    allocates, frees or retains device memory;
 *  - process-wide state is limited to (a) per DEVICE, created on first use and kept for the life of the process:
 *    one non-blocking side stream that runs the spectral axis stack next to the spatial one, plus a small pool of
 *    timing-disabled events (fork / join, one per gradient range handed to the bucket callback) — the whole-pass
 *    entry points look them up by the device that is current when they are called, so the caller must make the
 *    device of its pointers current (HSIMAE_TWO_STREAMS=0 keeps everything on the caller's stream); (b) the cached
 *    values of the HSIMAE_* environment switches (kernel-generation A/B switches for tests, read once);
 *  - all work is enqueued on the caller's `stream` (hipStream_t passed as void*) or on the side stream, which is
 *    forked from and joined back into it with events inside the same call: no implicit sync, nothing outlives the
 *    caller's stream order;
 *  - return 0 on success; <0 argument errors (HSIMAE_E*); >0 a hipError_t from the launch.
 *    Nothing throws or exits across the boundary.  hsimae_strerror() names a code.
 *  - bf16 buffers are raw 16-bit words (`hs_bf16`); activation buffers that feed a GEMM as its
 *    K operand are zero-padded to a multiple of 32 columns by the kernel that produces them.
 */
#ifndef HSIMAE_HIP_H
#define HSIMAE_HIP_H
#include <stdint.h>

#ifdef __HIPCC__
typedef __bf16 hs_bf16;
#else
typedef uint16_t hs_bf16;
#endif

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: only what this header declares is exported. */
#pragma GCC visibility push(default)

#define HSIMAE_OK 0
#define HSIMAE_EDIMS (-1)        /* bad dimensions / strides */
#define HSIMAE_EUNSUPPORTED (-2) /* configuration outside what the kernels are built for */
#define HSIMAE_EALIGN (-3)       /* misaligned pointer */
#define HSIMAE_ENULL (-4)        /* required pointer is NULL */
#define HSIMAE_ENOFORWARD (-5)   /* a backward entry point was given a workspace that no forward entry point of this process has
                                    filled: the backward follows the schedule its forward recorded (which intermediates exist,
                                    q|k|v saved or recomputed, ...) and never re-derives it from the environment */

/* ABI version of this header; hsimae_version() returns the value the loaded library was built with, and a binder must
 * refuse a library that answers differently (hsimae_amd/_lib.py does).  104 (round 5): HSIMAE_ENOFORWARD, schedule recorded
 * by the forward; 103 -> 104 also covers round 4's incompatible change of the weight-gradient parameter block (t[8] -> t[16]), which had
 * shipped without a bump.  105 (round 6): hsimae_build_info; hsimae_attn_params lost the fused-projection fields
 * (proj_w .. rowscale) with the whole-sample d = 128 attention kernels they drove (superseded by the fused attention half). */
#define HSIMAE_VERSION 105
int hsimae_version(void);
/* What the loaded library was built from (round 6).  `variant_bits` has one bit per compile-time switch that makes a kernel
 * compute WRONG results on purpose (timing ablations: HS_ABL_*, HS_EXP_*, HS_EXPERIMENT_*) or adds instrumentation
 * (HS_PHASE_TIMING), OR-ed over every translation unit of the library; hsimae_variant_name(bit) names bit 0..31 (NULL: unused).
 * A binder must refuse a library with any bit set unless the caller asked for a variant (hsimae_amd/_lib.py:
 * HSIMAE_ALLOW_VARIANT=1).  `kernel_source_hash` = the first 64 bits of hsimae_amd.build.kernel_source_hash() at build time,
 * `flags_hash` = the same of the hipcc flag list (0: built by hand, outside hsimae_amd/build.py); `default_flags` = 1 when that
 * list is build.py's own, i.e. no tuning knob (-DHS_*) was overridden either. */
typedef struct {
    int32_t abi_version; uint32_t variant_bits; uint64_t kernel_source_hash; uint64_t flags_hash; int32_t default_flags; int32_t reserved;
} hsimae_build_info_t;
int hsimae_build_info(hsimae_build_info_t* out);
const char* hsimae_variant_name(int bit);
const char* hsimae_strerror(int code);
/* 1 if the library runs the two axis stacks of the encoder on two streams on the CURRENT device (the side stream is created
 * on first use; HSIMAE_TWO_STREAMS=0 or a failed hipStreamCreate give 0).  The order in which hsimae_backward reports gradient
 * ranges depends on it, so data-parallel callers check that every rank answers the same (hsimae_amd/model.py
 * enable_data_parallel) before they issue collectives in callback order. */
int hsimae_two_streams_active(void);

/* ------------------------------------------------------------------ model geometry */
/* Mirrors the HSIMAE constructor arguments that shape the tensors (Models.py:312-332) for the
 * supported family: img_size 9, patch_size 3, b_patch_size 8, in_chans 1, qkv bias on; embed_dim / dec_dim multiples
 * of 8 up to 512 with head dim 8 or 16 (widths that are not multiples of 32 run zero-padded to the next one). */
typedef struct {
    int32_t bands;          /* B, multiple of 8; T = B/8 */
    int32_t embed_dim;      /* D */
    int32_t depth;          /* total encoder depth */
    int32_t s_depth;        /* depth of each axis stack (blocks_1 / blocks_2) */
    int32_t num_heads;
    int32_t dec_dim;        /* Dd */
    int32_t dec_depth;
    int32_t dec_heads;
    int32_t hidden;         /* SwiGLU hidden of the encoder blocks (Models.py:225) */
    int32_t dec_hidden;
    int32_t norm_pix_loss;
    int32_t precision;      /* HSIMAE_PREC_BF16 (0) or HSIMAE_PREC_FP8: operand type of the encoder linears (see below) */
} hsimae_config;
/* Operand type the encoder blocks' stand-alone linears actually run in for this configuration: HSIMAE_PREC_FP8 only when
   cfg->precision is FP8 AND the width is one where the MX GEMMs are used (embed_dim >= 512, or HSIMAE_FP8_UNFUSED=1 in the
   environment); otherwise HSIMAE_PREC_BF16 — precision = FP8 at embed_dim 128 / 256 computes exactly what BF16 computes.
   < 0: invalid configuration.  (bench.py prices its roofline against the peak of THIS type.) */
int hsimae_effective_precision(const hsimae_config* cfg);
/* precision: BF16 = bf16 MFMA operands everywhere.  FP8 = the encoder blocks' linears (q|k|v, proj, w1|w3, w2 and their
 * data gradients) run on the MX block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4) with OCP e4m3 operands: weights are
 * quantised at pack time, activations as they are staged, both with one e8m0 scale per 32 consecutive K elements (MX),
 * fp32 accumulate; weight gradients, attention, LayerNorm, the decoder and the loss stay as in BF16.  fp32 masters and the
 * state_dict are untouched. */
enum { HSIMAE_PREC_BF16 = 0, HSIMAE_PREC_FP8 = 1 };

/* Flat fp32 parameter buffer: parameters in `named_parameters()` registration order
 * (Models.py:342-424), each contiguous, no padding.  Returns the number of parameter tensors;
 * offsets[i]/sizes[i] (in floats) are filled when non-NULL. Total floats = offsets[n-1]+sizes[n-1]. */
int hsimae_param_layout(const hsimae_config* cfg, int64_t* offsets, int64_t* sizes, int max_entries);

/* bf16 packed weight images (MFMA B-fragment order; W and W^T of every linear). */
int64_t hsimae_wpk_elems(const hsimae_config* cfg);
/* Host-side descriptor table for hsimae_pack_params: fill `table_host` (size from
 * hsimae_pack_table_bytes) for the given device pointers; the caller uploads it to the device. */
int64_t hsimae_pack_table_bytes(const hsimae_config* cfg);
int hsimae_build_pack_table(const hsimae_config* cfg, const float* params_dev, hs_bf16* wpk_dev, void* table_host);
/* fp32 master weights -> packed bf16 images; run after every optimizer step. */
int hsimae_pack_params(const hsimae_config* cfg, const void* table_dev, void* stream);

/* ------------------------------------------------------------------ whole-pass entry points */
/* Replaces HSIMAE.forward (Models.py:627-634): forward_encoder 537-571, forward_decoder 573-601,
 * forward_loss 603-616, recons 618-625; and, for hsimae_backward, the autograd backward of all
 * of it (Model_Pretraining.py:101). */
typedef struct {
    const float* x;               /* [N,1,B,9,9] fp32 cube, arbitrary strides (elements) */
    int64_t sn, sb, sh, sw;
    int32_t N, len_t, len_l;      /* grid drawn on the host (Models.py:484-493) */
    const float* noise1;          /* [N,T]  (Models.py:506) */
    const float* noise2;          /* [N,9]  (Models.py:513) */
    const float* params;          /* flat fp32 parameters */
    const hs_bf16* wpk;           /* packed images */
    void* workspace;              /* hsimae_workspace_bytes() bytes, 256-B aligned.  When embed_dim or dec_dim is not a multiple
                                     of 32, rows are stored at the width rounded up to 32 and the columns past the true
                                     width must be zeros: zero-fill the workspace once before its first use with a given
                                     (N, len_t * len_l) — the kernels never write anything but zeros there. */
    int64_t workspace_bytes;
    float grad_scale;             /* folded into dLoss/dpred (1/world_size for data parallel) */
    int32_t want_recons;          /* write pred_img / mask_img */
    /* outputs */
    float* loss;                  /* [1] */
    float* pred_img;              /* [N,1,B,9,9] contiguous */
    float* mask_img;              /* [N,1,B,9,9] contiguous */
    float* mask;                  /* [N,T*9] 0 keep / 1 remove */
    int32_t* ids_keep;            /* [N,K] ascending */
    int32_t* ids_restore;         /* [N,T*9] */
    float* latent;                /* optional [N*K, D] fp32 copy of the encoder output (may be NULL) */
    float* pred;                  /* optional [N*T*9, 72] fp32 copy of decoder_pred output (may be NULL) */
    /* optional stochastic depth of the encoder blocks (DropPath, Models.py:235-263, 304-305; fine-tuning only):
       per-ROW factors (0 or 1/keep_prob, equal within a sequence) for the attention branch and the MLP branch of
       every encoder block, [n_enc_blocks][2][N*K] fp32 in execution order blocks_1[0..], blocks_2[0..], blocks[0..];
       x += scale * branch(x).  NULL = no DropPath.  Must be the same array in forward and backward. */
    const float* drop_scale;
    /* optional (hsimae_backward / hsimae_encode_backward with a bucket callback): the stream the callback's consumer
       launches its collectives from.  Before each callback the library makes this stream wait (event) for the kernels that
       complete the reported range — including the ones on the side stream — so the ranges of the two axis stacks are
       reported block by block instead of after the join.  NULL: ranges are reported once they are complete on `stream`. */
    void* bucket_stream;
    /* optional deterministic gradient reduction (hsimae_backward / _encode_backward / _decode_backward): an int64 buffer
       with one element per element of `grads` (contents irrelevant on entry).  Weight-gradient partial sums are then
       accumulated in 64-bit fixed point with integer atomics (order-independent) and converted to fp32 at the end, so two
       runs on the same inputs produce bit-identical gradients.  NULL: fp32 atomics (faster; last-bit run-to-run noise). */
    int64_t* det_acc;
} hsimae_io;

int64_t hsimae_workspace_bytes(const hsimae_config* cfg, int32_t N, int32_t len_t, int32_t len_l);
int hsimae_forward(const hsimae_config* cfg, const hsimae_io* io, void* stream);

/* Called on the host thread right after the kernels that complete the gradients of
 * grads[off, off+len) have been enqueued (reverse registration order, so buckets are contiguous
 * suffixes of the flat buffer).  Used to start the RCCL all-reduce of that bucket on a side stream. */
typedef void (*hsimae_bucket_cb)(int32_t stage, int64_t off, int64_t len, void* user);
/* grads: flat fp32 buffer, same layout as params, zeroed by the caller (weight grads are
 * accumulated with atomics). */
int hsimae_backward(const hsimae_config* cfg, const hsimae_io* io, float* grads, hsimae_bucket_cb cb, void* user,
                    void* stream);

/* ------------------------------------------------------------------ per-kernel entry points */
/* Structured random masking, closed form of Models.py:495-535 (bit-exact given the noise). */
typedef struct {
    const float* noise1; const float* noise2; int32_t N, T, L, len_t, len_l;
    int32_t* ids_keep; int32_t* ids_restore; float* mask;
} hsimae_mask_params;
int hsimae_mask_from_noise(const hsimae_mask_params* p, void* stream);

/* Kept-token patch rows of the cube as a bf16 GEMM operand [N*K, 96] (Conv3d k=s=(8,3,3) == GEMM
 * with K=72, Models.py:147-158; selection Models.py:527-528 fused in). */
typedef struct {
    const float* x; int64_t sn, sb, sh, sw; int32_t N, T, K; const int32_t* ids_keep;
    hs_bf16* out; int32_t* pos_ids;
} hsimae_patch_params;
int hsimae_patch_gather(const hsimae_patch_params* p, void* stream);

/* Row-panel MFMA GEMM out = epi(pro(A) * W^T) — every nn.Linear of Attention/SwiGLU/Block
 * (Models.py:180-184, 226-228, 402, 420) with LayerNorm (Models.py:288, 299, 399, 419) fused
 * as prologue and bias / residual / SiLU-gate (Models.py:232, 304-305) as epilogue. */
enum { HSIMAE_A_BF16 = 0, HSIMAE_A_F32 = 1, HSIMAE_A_F32_LN = 2 };
enum { HSIMAE_E_BF16 = 0, HSIMAE_E_F32 = 1, HSIMAE_E_RES_F32 = 2, HSIMAE_E_POS_F32 = 3, HSIMAE_E_SWIGLU = 4,
       HSIMAE_E_SWIGLU_BWD = 5,
       /* E_LN_BWD: the product is dL/d(LayerNorm output); the epilogue applies the LayerNorm backward (autograd of
          Models.py:304 `x + attn(norm1(x))` w.r.t. x) in place of a separate pass: out = res + LNbwd(acc; lnx, gamma)
          (+ out when `accumulate`), dgamma / dbeta accumulated with atomics.  Needs N == n_valid == 128 (one chunk). */
       HSIMAE_E_LN_BWD = 6 };
typedef struct {
    const void* A; int32_t lda;
    int32_t M, N, K;
    int32_t n_valid;
    const hs_bf16* W; const hs_bf16* W2;
    const float* bias; const float* bias2;
    const float* gamma; const float* beta;
    float* stats;
    hs_bf16* u_out; int32_t ldu;
    void* out; int32_t ldo;
    const float* res; const float* res2; int32_t ldr;
    const float* pos; const int32_t* ids; int32_t ldpos;
    hs_bf16* h13; int32_t ldh; int32_t hoff;
    const float* lnx; float* dgamma; float* dbeta; int32_t accumulate;    /* E_LN_BWD only (lnx: the LayerNorm's input, ld = ldr) */
    /* optional per-row factors [M] (DropPath): a_rowscale multiplies the rows of an A_F32 operand as it is staged;
       out_rowscale multiplies (product + bias) before the residual is added in E_RES_F32.  NULL = 1. */
    const float* a_rowscale; const float* out_rowscale;
    /* prec = HSIMAE_PREC_FP8: the MX block-scaled e4m3 form of the same product (fp32 accumulate, same epilogues except
       E_LN_BWD / E_POS_F32).  A is quantised on the fly; W8 / S8 (and W8b / S8b for the second matrix of E_SWIGLU) are the
       e4m3 image and its e8m0 scale image of the weight as produced by hsimae_pack_matrix with desc.fp8 = 1
       (K padded to a multiple of 128).  W / W2 are ignored.  prec = 0: bf16, the fields below are ignored. */
    int32_t prec;
    const uint8_t* W8; const uint8_t* S8; const uint8_t* W8b; const uint8_t* S8b;
    /* A_F32_LN: number of leading columns the LayerNorm runs over when the rows are stored wider than the model width
       (K = storage width, a multiple of 32; the columns past ln_width are read as zeros and stay zeros); 0 = K. */
    int32_t ln_width;
    const float* det_base; int64_t* det_acc;      /* E_LN_BWD: deterministic dgamma / dbeta commits, as in hsimae_lnbwd_params */
} hsimae_gemm_params;
int hsimae_gemm(const hsimae_gemm_params* p, int32_t a_kind, int32_t epilogue, void* stream);
/* The same kernel with the row-panel height (bm: 64 or 128) and the depth of an A chunk (kc: 128 or 256; ignored by the
 * LayerNorm prologue, which stages all of K) forced instead of chosen by shape; 0 = the shape rule.  Measurement hook
 * (scripts/gemm_sweep.py): results are identical for every tiling. */
int hsimae_gemm_tiled(const hsimae_gemm_params* p, int32_t a_kind, int32_t epilogue, int32_t bm, int32_t kc, void* stream);

/* One fp32 matrix -> packed image (placement n_off/k_off lets q|k|v or w1|w3 share an image). */
typedef struct {
    const float* src; int32_t rows, cols;
    int32_t transpose;
    int32_t n_off, k_off;
    int32_t KS;
    hs_bf16* dst;
    /* fp8 = 1: quantise to the MX e4m3 image instead (hsimae_gemm_params.prec): KS = 128-deep k-steps of the image
       (K padded to a multiple of 128), dst = the e4m3 image (bytes), scales = its e8m0 image, one dword per (n-tile,
       512-deep chunk, lane).  k_off and the source's K extent must be multiples of 32 (a scale block never straddles
       two sources).  Both images must be zero-filled once before the first pack. */
    int32_t fp8;
    uint8_t* scales;
} hsimae_pack_desc;
int hsimae_pack_matrix(const hsimae_pack_desc* desc_dev, int32_t ndesc, int32_t max_elems, void* stream);

/* MLP half of an encoder Block in one kernel each way (Models.py:305 `x + mlp(norm2(x))`, SwiGLU :231-232), d = 128 with
 * hidden width padded to 352 (Base): forward  x2 = x1 + rowscale * (b2 + (silu(u2 W1^T + b1) * (u2 W3^T + b3)) W2^T) (+ res2),
 * u2 = LN2(x1); backward recomputes u2 / h1 / h3 and writes dx1 (fp32) plus the bf16 operands of the block's weight
 * gradients: u2, dh1|dh3 [M, 2*352], g [M, 352], dY and dx1 copies; LayerNorm-2 parameter grads are accumulated.
 * Weight images as produced by hsimae_pack_params (w13T = W1^T | W3^T along K).  rs_mlp / rs_attn: DropPath row factors. */
typedef struct {
    const float* n2w; const float* n2b; const float* w1b; const float* w3b; const float* w2b;
    const hs_bf16* w1; const hs_bf16* w3; const hs_bf16* w2; const hs_bf16* w2T; const hs_bf16* w13T;
    int32_t hidden;
} hsimae_mlp_weights;
int hsimae_enc_mlp_fwd(const float* x1, const float* res2, float* x2, int32_t M, int32_t d, const hsimae_mlp_weights* w,
                       const float* rowscale, void* stream);
/* plane_rows = 0: dh13 [M][2 * hp], g [M][hp] row-major (hp = hidden rounded up to 32).  plane_rows = R >= M: 64-column planes
 * (see hsimae_wgrad_task): g = [P][R][64], dh13 = dh1 [P][R][64] followed by dh3 [P][R][64], P = ceil(hp / 64); rows past M
 * and the columns past hp of the last plane are not written. */
int hsimae_enc_mlp_bwd(const float* x1, const float* dy, float* dx1, hs_bf16* u2, hs_bf16* dh13, hs_bf16* g, hs_bf16* dyb,
                       hs_bf16* dx1b, int32_t M, int32_t d, const hsimae_mlp_weights* w, float* g_n2w, float* g_n2b,
                       const float* rs_mlp, const float* rs_attn, int32_t plane_rows, void* stream);

/* Masked multi-head attention over one sample's tokens (Models.py:192-215) and its backward. */
typedef struct {
    const hs_bf16* qkv; int32_t ld;
    int32_t d, heads, hd;
    int32_t Ts;
    int32_t nsamples;
    int32_t mode, len_l;
    hs_bf16* o; int32_t ldo;
    float* lse;
    const hs_bf16* dout; int32_t lddo;
    hs_bf16* dqkv;
    /* column offset of k (and 2x: of v) inside a qkv / dqkv row when rows are stored wider than d (storage width); 0 = d */
    int32_t kv_off;
} hsimae_attn_params;
int hsimae_attn_fwd(const hsimae_attn_params* p, void* stream);
int hsimae_attn_bwd(const hsimae_attn_params* p, void* stream);

/* One fused decoder Block (Models.py:303-306 at width 64, 8 heads of 8, SwiGLU hidden 164..192 — hp = 192, the image height
   the kernels are compiled for; any other hidden width returns HSIMAE_EUNSUPPORTED; sequences of 16..112 tokens),
   as hsimae_forward / hsimae_backward run it for every decoder block — exposed per kernel for the parity tests and the bench's
   kernel replays.  Weights: fp32 vectors, packed bf16 images (hsimae_pack_matrix: qkv = q | k | v fused [192][64], p [64][64],
   w1 / w3 [hp][64], w2 [64][hp], w2T [hp][64] = W2 transposed, hp = hidden rounded up to 32) and the fp32 row-major masters of
   q, k, v, proj, w1, w3 (the backward stages those in LDS itself).  bqkv = bq | bk | bv.
   forward : x [nsamples * Ts][64] fp32 -> x1 = x + proj(attention(LN1 x)), x2 = x1 + mlp(LN2 x1); keeps o (attention output,
             bf16 [rows][64]) and lse (log2-domain logsumexp, [rows][8]) for the backward.  split != 0: the attention half with
             q / k / v in registers + the row-panel MLP kernel (two launches, the default schedule); 0: one kernel.
   backward: from dy = dL/dx2 -> dx = dL/dx (dx1_tmp: scratch [rows][64] fp32, dL/dx1) and the block's 18 parameter gradients,
             ACCUMULATED into the given tensors.  slab: NULL = float atomics; else >= hsimae_dec_block_slab_floats() floats of
             scratch: per-workgroup partials + one fixed-order reduce launch (bit-reproducible). */
typedef struct {
    const float *n1w, *n1b, *bqkv, *pb, *n2w, *n2b, *w1b, *w3b, *w2b;
    const hs_bf16 *qkv, *p, *w1, *w3, *w2, *w2T;
    const float *qf, *kf, *vf, *pf, *w1f, *w3f;
    int32_t hidden;
} hsimae_dec_block_weights;
typedef struct {
    float *n1w, *n1b, *qw, *qb, *kw, *kb, *vw, *vb, *pw, *pb, *n2w, *n2b, *w1w, *w1b, *w2w, *w2b, *w3w, *w3b;
} hsimae_dec_block_grads;
int hsimae_dec_block_fwd(const hsimae_dec_block_weights* w, const float* x, float* x1, float* x2, hs_bf16* o, float* lse,
                         int32_t nsamples, int32_t Ts, int32_t split, void* stream);
#define HSIMAE_DEC_BLOCK_SLAB_FLOATS (256ll * (104 * 512 + 2112))   /* workgroups x (in-register dW values x threads + bias / LayerNorm sums) */
int64_t hsimae_dec_block_slab_floats(void);     /* = HSIMAE_DEC_BLOCK_SLAB_FLOATS of the library that is loaded: size of `slab` above */
/* Bytes of weight-gradient slab that hsimae_workspace_bytes() reserves for the caller's stream (side_stream = 0) or for the
   side stream of the forked spectral stack (1; 0 bytes = that stream commits with float atomics).  Whenever a stream can run
   a 256 x 256-tile weight-gradient launch (an encoder OR — caller's stream, layer-at-a-time decoder — a decoder of storage
   width >= 256) the answer is >= 64 MiB = 256 workgroups x 256 KB, what such a launch may write (csrc/plan.h slab_bytes). */
int64_t hsimae_wgrad_slab_bytes(const hsimae_config* cfg, int32_t side_stream);
int hsimae_dec_block_bwd(const hsimae_dec_block_weights* w, const hsimae_dec_block_grads* g, const float* x, const float* x1,
                         const float* dy, float* dx1_tmp, float* dx, const hs_bf16* o, const float* lse, int32_t nsamples,
                         int32_t Ts, float* slab, void* stream);

/* Weight / bias gradients of up to 16 linears in one launch (autograd of F.linear).  msplit = number of row slices the
   128 x 128 dW tiles are split into (partial sums meet in dW through atomics).  When every task of a launch has N >= 256 and
   K >= 256 and bf16 operands, the launch runs on 256 x 256 tiles, one workgroup per CU, and sizes its own row split (msplit is
   then only validated); the result is the same sum. */
typedef struct {
    const void* dO; int32_t dO_f32; int32_t ldo;
    const hs_bf16* A; int32_t lda;
    int32_t N, K;
    float* dW; int32_t ldw;
    float* db;
    const float* dO_rowscale;     /* optional per-row factor [M] on an fp32 dO (DropPath); NULL = 1 */
    /* Operand layout (round 5).  0 = row-major [M][ldo] / [M][lda].  R > 0 = 64-column PLANES of R >= M rows each: the operand
       is [ldo / 64][R][64] (lda likewise), i.e. column c of row r lives at (c / 64) * R * 64 + r * 64 + c % 64 and ldo / lda is
       the padded width, a multiple of 64.  A producer that emits its operand 64 columns at a time (hsimae_enc_mlp_bwd: one
       hidden chunk per step) then writes row-contiguous blocks instead of 128-byte pieces at the row pitch — 5.2 against
       4.1 TB/s for the same bytes (scripts/micro/hbm_stride.hip) — and a dW tile's operand slice is two sequential streams.
       R > M (hsimae_backward uses M + 48) keeps the planes from starting a large power of two apart.  Planar operands need
       bf16, M % 32 == 0 and the LDS-DMA kernel (any launch with bf16 operands below 4 GB); else HSIMAE_EDIMS / _EUNSUPPORTED. */
    int32_t dO_plane_rows, A_plane_rows;
} hsimae_wgrad_task;
typedef struct {
    hsimae_wgrad_task t[16]; int32_t ntasks; int32_t M; int32_t msplit;      /* (16 since round 4: the 2 x 7 linears of a pair of axis-stack blocks) */
    const float* det_base; int64_t* det_acc;      /* deterministic dW / db commits, as in hsimae_lnbwd_params; NULL = fp32 atomics */
    /* optional scratch of >= 64 MiB (256 workgroups x 256 x 256 floats), private to the calling stream: a launch on 256 x 256
       tiles then writes every workgroup's partial dW tile there as coalesced rows and a second launch sums the row slices in
       a fixed order and adds them into dW — no float atomics on dW (they are ~6x slower per byte than stores, and with every
       workgroup committing at once were 26 % of the launch at d = 256), bit-reproducible.  NULL = atomics. */
    float* slab;
} hsimae_wgrad_params;
int hsimae_wgrad(const hsimae_wgrad_params* p, void* stream);
/* The row split (msplit) hsimae_backward uses for a launch of `tiles` 128x128 dW tiles over M rows: one resident
   wave of workgroups, whole groups of 8 (row slice ms runs on XCD ms % 8). */
int32_t hsimae_wgrad_msplit(int32_t tiles, int64_t M);

/* LayerNorm backward (autograd of Models.py:288/299/399/419), residual grad fused. */
typedef struct {
    const float* du; const float* x; const float* stats; const float* gamma; const float* dres;
    float* dx; int32_t accumulate; float* dgamma; float* dbeta; int32_t M, d;
    int32_t ld;                   /* row stride of du / x / dres / dx in floats (storage width); 0 = d */
    /* deterministic commits (hsimae_io.det_acc): dgamma / dbeta point into the flat gradient buffer starting at det_base and
       the sums go to det_acc[ptr - det_base] in 64-bit fixed point instead; both NULL = fp32 atomics */
    const float* det_base; int64_t* det_acc;
} hsimae_lnbwd_params;
int hsimae_ln_bwd(const hsimae_lnbwd_params* p, void* stream);
/* Plain LayerNorm forward, fp32 in/out (Models.py:570 when a caller wants the fp32 latent). */
int hsimae_ln_fwd(const float* x, const float* gamma, const float* beta, float* out, int32_t M, int32_t d, void* stream);

/* Decoder sequence assembly: mean-token fill + unshuffle + pos (Models.py:583-592) and backward. */
typedef struct {
    const float* y; int32_t N, K, TL, Dd; const int32_t* ids_restore; const float* pos;
    float* yfull; const float* dyfull; hs_bf16* dy;
    int32_t ld;                   /* row stride of y / yfull / dyfull / dy (storage width); 0 = Dd */
} hsimae_assemble_params;
int hsimae_assemble_fwd(const hsimae_assemble_params* p, void* stream);
int hsimae_assemble_bwd(const hsimae_assemble_params* p, void* stream);

/* patchify + norm_pix target + masked MSE + dLoss/dpred + recons images (Models.py:603-625). */
typedef struct {
    const float* x; int64_t sn, sb, sh, sw; int32_t N, T; const float* pred; const float* mask;
    int32_t norm_pix; float inv_scale;
    float* partial; float* loss; float sum_mask;
    hs_bf16* dpred; float* pred_img; float* mask_img;
} hsimae_loss_params;
int hsimae_loss_partials(int32_t N, int32_t T);   /* floats needed in `partial` */
int hsimae_loss(const hsimae_loss_params* p, void* stream);

/* ------------------------------------------------------------------ next row N1: optimizer step */
/* AdamW over the flat parameter / gradient buffers in ONE launch, torch.optim.AdamW semantics
 * (Model_Pretraining.py:80-86,102: betas (.9,.95), decay only on names without 'bias'/'norm').
 * group[i] per element: 0 = decayed, 1 = not decayed, 2 = frozen / unused (left untouched). `step` is 1-based. */
int hsimae_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* group, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream);

/* ------------------------------------------------------------------ next row N3: fine-tuning (DualViT) */
/* Encoder only, through `norm` (DualViT / HSIViT `forward_encoder`, Models.py:869-894, 1119-1146): same io as
 * hsimae_forward; io->latent [N, len_t*len_l, embed_dim] is required, loss / pred / recons outputs are not touched.
 * For the unmasked encoder pass len_t = T, len_l = 9 and increasing noise (ids_keep = identity). */
int hsimae_encode(const hsimae_config* cfg, const hsimae_io* io, void* stream);
/* Backward of hsimae_encode (fine-tuning, Model_Finetuning.py:156 `loss.backward()` through the classification
 * branch): dlatent [N*K, D] fp32 is dL/d(latent); the activations hsimae_encode left in io->workspace are consumed.
 * Accumulates into `grads` like hsimae_backward (the decoder's entries are not touched), same bucket callback. */
int hsimae_encode_backward(const hsimae_config* cfg, const hsimae_io* io, const float* dlatent, float* grads,
                           hsimae_bucket_cb cb, void* user, void* stream);
/* 'AGG' pooling of the classification head (Models.py:962-970): [N, T*L, D] -> [N, T*D], mean over the L tokens of
 * each spectral group; the Linear that follows is hsimae_gemm(A_F32, E_F32). */
int hsimae_agg_pool(const float* latent, float* pooled, int32_t N, int32_t T, int32_t L, int32_t D, void* stream);
/* Backward of that head (Models.py:962-970) from g = dL/d(class_pred) [N][C], fp32: gw [C][T*D] = g^T pooled, gb [C] = column
   sums of g, dlatent [N][T][L][D] = (g W)[n][t*D + c] / L for every l (W [C][T*D] = cls_head.weight).  One launch; results are
   ASSIGNED.  C <= 256. */
int hsimae_head_bwd(const float* g, const float* pooled, const float* w, float* gw, float* gb, float* dlatent, int32_t N, int32_t C,
                    int32_t T, int32_t L, int32_t D, void* stream);
/* Stand-alone decoder (HSIMAE / DualViT `forward_decoder`, Models.py:573-601 / 923-945), inference: latent [N*K, D] fp32
 * (the output of `norm`) -> pred [N*T*9, 72] fp32.  io supplies N, len_t, len_l (K = len_t*len_l), ids_restore, params,
 * wpk and the workspace, as for hsimae_forward. */
int hsimae_decode(const hsimae_config* cfg, const hsimae_io* io, const float* latent, float* pred, void* stream);
/* Backward of hsimae_decode (autograd of Models.py:573-601 for callers that compose forward_encoder / forward_decoder /
 * forward_loss themselves, e.g. Models.py:975-993): dpred [N*T*9, 72] fp32 is dL/d(pred); the activations hsimae_decode
 * left in io->workspace are consumed.  Writes dlatent [N*K, D] fp32 (dL/d(latent), the input of hsimae_decode) and
 * accumulates the decoder's parameter gradients into `grads` (flat layout; encoder entries are not touched). */
int hsimae_decode_backward(const hsimae_config* cfg, const hsimae_io* io, const float* dpred, float* dlatent, float* grads,
                           hsimae_bucket_cb cb, void* user, void* stream);

/* ------------------------------------------------------------------ next row N2: input pipeline */
/* One batch of training cubes assembled on the device from HBM-resident scenes (Model_Pretraining.py:40-51
 * `HSIdataset4PT.__getitem__`): window [h:h+9, w:w+9, :] of scene `num`, (x - min) / (max - min) in the scenes'
 * dtype, optional flips along w (bit 0, np.flip(data, 1)) and h (bit 1, np.flip(data, 0)), written as
 * out[n, 0, b, i, j] through the given element strides (band-fastest: sb = 1, sw = bands, sh = 9 * bands).
 * scenes: all scenes concatenated, each [h][w][bands] row-major; scene_off[s] element offset, scene_w[s] = w.
 * cut: the reference's int16 table, rows (c, h, w, scene, max, min) (Utils/Preprocessing.py:69-117); like the
 * reference, `c` is ignored and all bands are taken.  Bit-exact with the numpy arithmetic for fp32 / fp64 scenes. */
typedef struct {
    const void* scenes; int32_t scene_f64; const int64_t* scene_off; const int32_t* scene_w; int32_t bands;
    const int16_t* cut; const int64_t* index; const uint8_t* flips; int32_t N;
    float* out; int64_t sn, sb, sh, sw;
} hsimae_cube_params;
int hsimae_cube_gather(const hsimae_cube_params* p, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
