// C ABI of libhsimae_hip.so (include/hsimae_hip.h): geometry/layout helpers, per-kernel entry points and
// the whole-pass forward / backward schedules of the HSIMAE pretraining path.
#include "common.h"
#include "kernels.h"
#include "plan.h"
#include <vector>
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

using namespace hsplan;

namespace {

#define CK(expr) do { int _e = (expr); if (_e) return _e; } while (0)
#define CK0(expr) do { int _e0 = (expr); if (_e0) return _e0; } while (0)

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
struct BlkP {            // resolved pointers of one block
    const float *n1w, *n1b, *bqkv, *pb, *n2w, *n2b, *w1b, *w3b, *w2b;
    const hs_bf16 *qkv, *p, *w1, *w3, *w2, *qkvT, *pT, *w13T, *w2T;
    const float *qf, *kf, *vf, *pf, *w1f, *w3f;     // fp32 master weights (row-major), staged as bf16 by the fused decoder backward
    int prec;                                       // HSIMAE_PREC_FP8: the block's linears run on the MX e4m3 images below
    struct I8 { const uint8_t* w; const uint8_t* s; } qkv8, p8, w1_8, w3_8, w2_8, qkvT8, pT8, w13T8, w2T8;
};

BlkP resolve(const BlkOff& o, const BlkW& w, const float* P, const hs_bf16* wpk, const WLayout& WL) {
    const float* fbase = reinterpret_cast<const float*>(wpk + WL.bf16_elems);
    BlkP b;
    b.n1w = P + o.n1w; b.n1b = P + o.n1b; b.bqkv = fbase + w.bqkv; b.pb = P + o.pb; b.n2w = P + o.n2w; b.n2b = P + o.n2b;
    b.w1b = P + o.w1b; b.w3b = P + o.w3b; b.w2b = P + o.w2b;
    b.qkv = wpk + w.qkv; b.p = wpk + w.p; b.w1 = wpk + w.w1; b.w3 = wpk + w.w3; b.w2 = wpk + w.w2;
    b.qkvT = wpk + w.qkvT; b.pT = wpk + w.pT; b.w13T = wpk + w.w13T; b.w2T = wpk + w.w2T;
    b.qf = P + o.qw; b.kf = P + o.kw; b.vf = P + o.vw; b.pf = P + o.pw; b.w1f = P + o.w1w; b.w3f = P + o.w3w;
    const uint8_t* base8 = reinterpret_cast<const uint8_t*>(fbase + WL.f32_elems);
    auto i8 = [&](const Img8& im) { BlkP::I8 r; r.w = base8 + im.w; r.s = base8 + im.s; return r; };
    b.prec = HSIMAE_PREC_BF16;
    b.qkv8 = i8(w.qkv8); b.p8 = i8(w.p8); b.w1_8 = i8(w.w1_8); b.w3_8 = i8(w.w3_8); b.w2_8 = i8(w.w2_8);
    b.qkvT8 = i8(w.qkvT8); b.pT8 = i8(w.pT8); b.w13T8 = i8(w.w13T8); b.w2T8 = i8(w.w2T8);
    return b;
}
// ------------------------------------------------------------------ the schedule of a pass, decided ONCE, by its forward
// Every choice between two kernel generations that changes what the forward leaves in the workspace for the backward — fused
// or layer-at-a-time halves (which intermediates exist), q|k|v saved or recomputed, fp8 GEMMs or fused bf16 kernels — is read from
// the environment exactly once per pass: by the forward entry point (sched_from_env), which records it for its workspace arena
// (record_sched).  The backward entry points look the record up and follow it; they never read the environment.  A switch that
// is flipped between a forward and its backward therefore has no effect on that pass (rounds 1-4 re-derived each decision where
// it was needed, partly per call and partly latched in function statics: a forward that skipped the q|k|v store followed by a
// backward that decided not to recompute read an unwritten buffer and returned HSIMAE_OK — VERDICT r04 item 5).
// A backward on an arena no forward of this process has filled returns HSIMAE_ENOFORWARD.
enum : uint32_t {
    SC_FUSED_DEC = 1u << 0,        // HSIMAE_FUSED_DEC=0 clears: layer-at-a-time decoder
    SC_DEC_SPLIT = 1u << 1,        // HSIMAE_DEC_SPLIT=0 clears: one-kernel decoder block forward
    SC_FP8_UNFUSED = 1u << 2,      // HSIMAE_FP8_UNFUSED=1 sets: every linear of an fp8 encoder block on the MX GEMMs, no fused kernel
    SC_FUSED_MLP = 1u << 3,        // HSIMAE_FUSED_MLP=0 clears: layer-at-a-time MLP half of the encoder blocks
    SC_ATTN_BLOCK = 1u << 4,       // HSIMAE_FUSED_ATTN_BLOCK=0 clears: blk128_fwd
    SC_ATTN_BLOCK256 = 1u << 5,    // HSIMAE_FUSED_ATTN_BLOCK256=0 clears: blk256_fwd
    SC_ATTN_BLOCK_BWD = 1u << 6,   // HSIMAE_FUSED_ATTN_BLOCK_BWD=0 clears: blk128_bwd
    SC_PROJ_BWD = 1u << 7,         // HSIMAE_FUSED_PROJ_BWD=0 clears: the projection's data gradient as its own GEMM
    SC_LNBWD = 1u << 8,            // HSIMAE_FUSED_LNBWD=0 clears: du store + separate ln_bwd pass
    SC_LNBWD_512 = 1u << 9,        // HSIMAE_FUSED_LNBWD_512=0 clears it at d = 512 only
    SC_RECOMPUTE = 1u << 10,       // HSIMAE_ATTN_BWD_RECOMPUTE=0 clears: the forward saves q|k|v
    SC_WGRAD_SLAB = 1u << 11,      // HSIMAE_WGRAD_SLAB=0 clears: float atomics in the 256 x 256-tile weight-gradient launches
    SC_DEC_SLAB = 1u << 12,        // HSIMAE_DEC_SLAB=0 clears: float atomics in the fused decoder backward
    SC_PLANAR = 1u << 13,          // HSIMAE_WGRAD_PLANAR=0 clears: g / dh1|dh3 of the fused MLP backward row-major instead of 64-column planes
    SC_ATTN_BLOCK256_BWD = 1u << 14,   // HSIMAE_FUSED_ATTN_BLOCK256_BWD=0 clears: blk256_bwd (three launches instead)
};
uint32_t sched_from_env() {
    auto off = [](const char* name) { const char* e = getenv(name); return e && e[0] == '0'; };
    auto one = [](const char* name) { const char* e = getenv(name); return e && e[0] == '1'; };
    uint32_t b = 0;
    if (!off("HSIMAE_FUSED_DEC")) b |= SC_FUSED_DEC;
    if (!off("HSIMAE_DEC_SPLIT")) b |= SC_DEC_SPLIT;
    if (one("HSIMAE_FP8_UNFUSED")) b |= SC_FP8_UNFUSED;
    if (!off("HSIMAE_FUSED_MLP")) b |= SC_FUSED_MLP;
    if (!off("HSIMAE_FUSED_ATTN_BLOCK")) b |= SC_ATTN_BLOCK;
    if (!off("HSIMAE_FUSED_ATTN_BLOCK256")) b |= SC_ATTN_BLOCK256;
    if (!off("HSIMAE_FUSED_ATTN_BLOCK_BWD")) b |= SC_ATTN_BLOCK_BWD;
    if (!off("HSIMAE_FUSED_PROJ_BWD")) b |= SC_PROJ_BWD;
    if (!off("HSIMAE_FUSED_LNBWD")) b |= SC_LNBWD;
    if (!off("HSIMAE_FUSED_LNBWD_512")) b |= SC_LNBWD_512;
    if (!off("HSIMAE_ATTN_BWD_RECOMPUTE")) b |= SC_RECOMPUTE;
    if (!off("HSIMAE_WGRAD_SLAB")) b |= SC_WGRAD_SLAB;
    if (!off("HSIMAE_DEC_SLAB")) b |= SC_DEC_SLAB;
    if (!off("HSIMAE_WGRAD_PLANAR")) b |= SC_PLANAR;
    if (!off("HSIMAE_FUSED_ATTN_BLOCK256_BWD")) b |= SC_ATTN_BLOCK256_BWD;
    return b;
}
struct SchedRec { uint32_t enc = 0, dec = 0; bool has_enc = false, has_dec = false; uint64_t gen = 0; };
std::mutex g_sched_mu;
std::unordered_map<const void*, SchedRec> g_sched;          // keyed by the workspace arena the forward filled
uint64_t g_sched_gen = 0;
void record_sched(const void* ws, bool enc, bool dec, uint32_t bits) {
    std::lock_guard<std::mutex> lk(g_sched_mu);
    // A process that keeps handing new arena addresses (caching allocator churn): drop the OLDEST half — never the entry being
    // recorded, and not the recent ones, whose backward may still be pending (clearing the whole map made a live
    // arena's backward fail with HSIMAE_ENOFORWARD)
    if (g_sched.size() > 4096 && g_sched.find(ws) == g_sched.end()) {
        std::vector<uint64_t> gens;
        gens.reserve(g_sched.size());
        for (auto& kv : g_sched) gens.push_back(kv.second.gen);
        std::nth_element(gens.begin(), gens.begin() + gens.size() / 2, gens.end());
        const uint64_t cut = gens[gens.size() / 2];
        for (auto it = g_sched.begin(); it != g_sched.end();) it = it->second.gen < cut ? g_sched.erase(it) : std::next(it);
    }
    SchedRec& r = g_sched[ws];
    r.gen = ++g_sched_gen;
    if (enc) {
        r.enc = bits; r.has_enc = true;
        // an encoder-only forward re-fills the arena: a decoder record of an EARLIER pass no longer describes what is in it
        if (!dec) r.has_dec = false;
    }
    if (dec) { r.dec = bits; r.has_dec = true; }
}
int lookup_sched(const void* ws, bool need_enc, bool need_dec, uint32_t& enc, uint32_t& dec) {
    std::lock_guard<std::mutex> lk(g_sched_mu);
    auto it = g_sched.find(ws);
    if (it == g_sched.end() || (need_enc && !it->second.has_enc) || (need_dec && !it->second.has_dec)) return HSIMAE_ENOFORWARD;
    enc = it->second.enc; dec = it->second.dec;
    return HSIMAE_OK;
}

// encoder blocks under precision = FP8 (the decoder's blocks never are)
BlkP resolve_enc(const Geo& g, uint32_t sc, const BlkOff& o, const BlkW& w, const float* P, const hs_bf16* wpk, const WLayout& WL) {
    BlkP b = resolve(o, w, P, wpk, WL);
    // fp8 where it pays: K >= 512.  At d = 256 the MX GEMMs (A quantised while it is staged) are no faster than the bf16 ones
    // (Large: 37.8 vs 37.2 ms per step with the attention-half linears in fp8, r03_k), at d = 128 every linear sits inside a
    // fused bf16 kernel; HSIMAE_FP8_UNFUSED=1 forces the MX GEMMs at every width (tests of the generic path).
    b.prec = (g.prec == HSIMAE_PREC_FP8 && (g.D >= 512 || (sc & SC_FP8_UNFUSED))) ? HSIMAE_PREC_FP8 : HSIMAE_PREC_BF16;
    return b;
}


// The two axis stacks (blocks_1 / blocks_2, Models.py:556-560) are independent until x1 + x2: the spectral
// stack runs on a side stream so two kernels are resident at once (a single 864-workgroup launch leaves a
// ~45 % tail on 256 CUs).  The stream and its two events are created once per process; HSIMAE_TWO_STREAMS=0
// keeps everything on the caller's stream.
struct Side {
    hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool ok = false, init = false;
    std::vector<hipEvent_t> pool; size_t next = 0;        // events handed to the bucket stream, one per reported range
};
// One Side per device, looked up by the device that is current at the call (include/hsimae_hip.h, Contract): a model
// on cuda:1 gets a side stream on cuda:1.  One host thread per device is assumed (fork / join events are shared).
Side& side() {
    static Side sides[32];
    static Side none;
    static std::mutex mu;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return none;
    std::lock_guard<std::mutex> lk(mu);
    Side& x = sides[dev];
    if (!x.init) {
        x.init = true;
        const char* e = getenv("HSIMAE_TWO_STREAMS");
        if (!(e && e[0] == '0'))
            x.ok = hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking) == hipSuccess &&
                   hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) == hipSuccess &&
                   hipEventCreateWithFlags(&x.join, hipEventDisableTiming) == hipSuccess;
    }
    return x;
}
// next event of the per-device pool (grown on demand, reused round-robin: a wait captures the record that precedes it)
hipEvent_t pool_event(Side& sd) {
    if (sd.pool.size() < 96) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess) { sd.pool.push_back(e); return e; }
    }
    if (sd.pool.empty()) return nullptr;
    return sd.pool[sd.next++ % sd.pool.size()];
}

// Reports gradient ranges to the caller's bucket callback.  With io->bucket_stream the consumer's launch stream is
// made to wait for the stream the range's kernels were enqueued on (so side-stream ranges can be reported at once);
// without it a range may only be reported once it is complete on the caller's stream.
struct Emitter {
    hsimae_bucket_cb cb; void* user; hipStream_t bucket; int stage;
    float* grads; const int64_t* det_acc;              // deterministic mode: the range's fixed-point sums become fp32 first
    int operator()(int64_t off, int64_t end, hipStream_t done_on) {
        if (det_acc) CK0(hs_det_convert(det_acc + off, grads + off, end - off, done_on));
        if (cb) {
            if (bucket) {
                hipEvent_t e = pool_event(side());
                if (!e) return (int)hipErrorOutOfMemory;
                CK0((int)hipEventRecord(e, done_on));
                CK0((int)hipStreamWaitEvent(bucket, e, 0));
            }
            cb(stage, off, end - off, user);
        }
        ++stage;
        return HSIMAE_OK;
    }
};

// SC_FUSED_DEC clear: the layer-at-a-time decoder (A/B testing of the fused decoder kernels)
bool fused_dec_enabled(const Geo& g, uint32_t sc) {
    return (sc & SC_FUSED_DEC) && hs_dec_fused_supported(g.Dd, g.Hd, g.hdec, g.TL);
}

// Forward of one fused decoder block.  Default (round 3): the attention half with q / k / v in registers
// (dec_attn_fwd_kernel, 16 waves per CU) + the MLP half as the row-panel kernel enc_mlp_fwd_kernel<64, 192>;
// SC_DEC_SPLIT clear selects the one-kernel form (dec_block_fwd_kernel) for A/B tests.

// precision = FP8 puts the MX e4m3 images on the linears that run as stand-alone GEMMs.  Where a fused bf16 kernel covers
// the shape (the attention half at d = 128, the MLP half at d = 128 / 256) it is kept: the fused bf16 form beats the
// layer-at-a-time fp8 form (round 2: Base 26.9 vs 19.4 ms, Large 44.1 vs 39.5 ms with every fused kernel switched off).
// SC_FP8_UNFUSED restores that all-fp8 layer-at-a-time schedule (tests of the generic fp8 path at small widths).

// SC_FUSED_MLP clear: the layer-at-a-time MLP half of the encoder blocks
bool fused_mlp_enabled(int d, int h, uint32_t sc) {
    return (sc & SC_FUSED_MLP) && hs_enc_mlp_fused_supported(d, h);
}

EncMlpPtrs mlp_ptrs(const BlkP& b, int h) {
    EncMlpPtrs m;
    m.n2w = b.n2w; m.n2b = b.n2b; m.w1b = b.w1b; m.w3b = b.w3b; m.w2b = b.w2b;
    m.w1 = b.w1; m.w3 = b.w3; m.w2 = b.w2; m.w2T = b.w2T; m.w13T = b.w13T; m.h = h;
    return m;
}

DecBlockPtrs dec_ptrs(const BlkP& b, int h) {
    DecBlockPtrs d;
    d.n1w = b.n1w; d.n1b = b.n1b; d.bqkv = b.bqkv; d.pb = b.pb; d.n2w = b.n2w; d.n2b = b.n2b;
    d.w1b = b.w1b; d.w3b = b.w3b; d.w2b = b.w2b;
    d.qkv = b.qkv; d.p = b.p; d.w1 = b.w1; d.w3 = b.w3; d.w2 = b.w2;
    d.qkvT = b.qkvT; d.pT = b.pT; d.w13T = b.w13T; d.w2T = b.w2T; d.h = h;
    d.qf = b.qf; d.kf = b.kf; d.vf = b.vf; d.pf = b.pf; d.w1f = b.w1f; d.w3f = b.w3f;
    return d;
}

GemmParams gp() { GemmParams p; std::memset(&p, 0, sizeof(p)); return p; }

int dec_block_fwd_fused(const BlkP& bp, const float* z, const BlkBuf& b, int N, int64_t Md, int TL, int Dd, int hdec, uint32_t sc, hipStream_t s) {
    if ((sc & SC_DEC_SPLIT) && hs_enc_mlp_fused_supported(Dd, hdec)) {
        CK(hs_dec_attn_fwd(z, b.x1, b.o, b.lse, N, TL, dec_ptrs(bp, hdec), s));
        return hs_enc_mlp_fwd(b.x1, nullptr, b.x2, (int)Md, Dd, mlp_ptrs(bp, hdec), s);
    }
    return hs_dec_block_fwd(z, b.x1, b.x2, b.o, b.lse, N, TL, dec_ptrs(bp, hdec), s);
}

// The attention half's backward as ONE launch that recomputes q|k|v from u (attn.hip blk128_bwd_kernel<RC>): the same predicate
// of the same recorded schedule word in the forward (which then does not store q|k|v) and in the backward.
static bool attn_bwd_recompute(int d, int dp, int heads, int h, int Ts, bool f8u, uint32_t sc) {
    if (!(sc & SC_PROJ_BWD) || !(sc & SC_LNBWD) || !(sc & SC_RECOMPUTE)) return false;
    return !f8u && d == 128 && dp == d && fused_mlp_enabled(d, h, sc) && (sc & SC_ATTN_BLOCK) && (sc & SC_ATTN_BLOCK_BWD) &&
           hs_attn_block_bwd_fusable(d, heads, Ts);
}

// One transformer Block forward (Models.py:303-306): 5 launches.
// rs_a / rs_m: optional per-row DropPath factors of the attention / MLP branch (Models.py:304-305), NULL = none.
int block_fwd(const BlkP& P, uint32_t sc, const float* x_in, const BlkBuf& b, int64_t M, int d, int heads, int h, int hp, int Ts,
              int nsamples, int mode, int len_l, const float* res2, hipStream_t s, const float* rs_a = nullptr,
              const float* rs_m = nullptr) {
    GemmParams p = gp();
    const int dp = rup(d, 32);                        // storage width of the rows (plan.h Geo::Dp); the fused kernels need dp == d
    const bool f8 = P.prec == HSIMAE_PREC_FP8;        // the stand-alone linears on MX e4m3 images
    const bool f8u = f8 && (sc & SC_FP8_UNFUSED);     // ... and no fused kernel at all (HSIMAE_FP8_UNFUSED=1)
    auto w8 = [&](GemmParams& q, const BlkP::I8& a) { if (f8) { q.prec = HSIMAE_PREC_FP8; q.W8 = a.w; q.S8 = a.s; } };
    if (!f8u && (sc & SC_ATTN_BLOCK) && hs_attn_block_fusable(d, heads, Ts)) {
        // LN1 + q|k|v + attention + projection + residual in one persistent kernel (attn.hip blk128_fwd_kernel)
        CK(hs_attn_block_fwd(x_in, P.n1w, P.n1b, P.qkv, P.bqkv, P.p, P.pb, b.u, attn_bwd_recompute(d, dp, heads, h, Ts, f8u, sc) ? nullptr : b.qkv,
                             b.o, b.lse, b.x1, rs_a, Ts, nsamples, mode, len_l, s));
        if (fused_mlp_enabled(d, h, sc)) return hs_enc_mlp_fwd(b.x1, res2, b.x2, (int)M, d, mlp_ptrs(P, h), s, rs_m);
    } else if (!f8 && dp == d && (sc & SC_ATTN_BLOCK256) && hs_attn_block256_fusable(d, heads, Ts, nsamples)) {
        // the same half at D = 256 (attn_wide.hip blk256_fwd_kernel: 16 waves = 16 heads, weights streamed from L2)
        CK(hs_attn_block256_fwd(x_in, P.n1w, P.n1b, P.qkv, P.bqkv, P.p, P.pb, b.u, b.qkv, b.o, b.lse, b.x1, rs_a, Ts, nsamples, mode,
                                len_l, s));
    } else {
    p.A = x_in; p.lda = dp; p.M = (int)M; p.N = 3 * dp; p.K = dp; p.n_valid = 3 * dp; p.W = P.qkv; p.bias = P.bqkv;
    p.gamma = P.n1w; p.beta = P.n1b; p.u_out = b.u; p.ldu = dp; p.out = b.qkv; p.ldo = 3 * dp; p.ln_width = d;
    w8(p, P.qkv8);
    CK(hs_gemm(p, A_F32_LN, E_BF16, s));
    AttnParams a; std::memset(&a, 0, sizeof(a));
    a.qkv = b.qkv; a.ld = 3 * dp; a.d = d; a.heads = heads; a.hd = d / heads; a.Ts = Ts; a.nsamples = nsamples;
    a.mode = mode; a.len_l = len_l; a.o = b.o; a.ldo = dp; a.lse = b.lse; a.kv_off = dp;
    CK(hs_attn_fwd(a, s));
    p = gp();
    p.A = b.o; p.lda = dp; p.M = (int)M; p.N = dp; p.K = dp; p.n_valid = d; p.W = P.p; p.bias = P.pb;
    p.res = x_in; p.ldr = dp; p.out = b.x1; p.ldo = dp; p.out_rowscale = rs_a;
    w8(p, P.p8);
    CK(hs_gemm(p, A_BF16, E_RES_F32, s));
    }
    if (!f8u && fused_mlp_enabled(d, h, sc)) return hs_enc_mlp_fwd(b.x1, res2, b.x2, (int)M, d, mlp_ptrs(P, h), s, rs_m);
    p = gp();
    p.A = b.x1; p.lda = dp; p.M = (int)M; p.N = hp; p.K = dp; p.n_valid = h; p.W = P.w1; p.W2 = P.w3; p.bias = P.w1b;
    p.bias2 = P.w3b; p.gamma = P.n2w; p.beta = P.n2b; p.u_out = b.u2; p.ldu = dp; p.out = b.g; p.ldo = hp;
    p.h13 = b.h13; p.ldh = 2 * hp; p.hoff = hp; p.ln_width = d;
    if (f8) { w8(p, P.w1_8); p.W8b = P.w3_8.w; p.S8b = P.w3_8.s; }
    CK(hs_gemm(p, A_F32_LN, E_SWIGLU, s));
    p = gp();
    p.A = b.g; p.lda = hp; p.M = (int)M; p.N = dp; p.K = hp; p.n_valid = d; p.W = P.w2; p.bias = P.w2b;
    p.res = b.x1; p.res2 = res2; p.ldr = dp; p.out = b.x2; p.ldo = dp; p.out_rowscale = rs_m;
    w8(p, P.w2_8);
    CK(hs_gemm(p, A_BF16, E_RES_F32, s));
    return HSIMAE_OK;
}


// d = 256 / 512 (rows of exactly two / four 128-column chunks): LayerNorm backward as the epilogue of the k-outer GEMM
// (gemm.hip epilogue_ln_ko; d = 512 on 32-row panels, fp8 only: Huge fp8 49.9 -> 48.9 ms per step, but bf16 60.2 -> 61.5).
// SC_LNBWD clear keeps the separate du store + ln_bwd pass, SC_LNBWD_512 clear keeps it at d = 512 only.
static bool wide_ln_fused(int d, int dp, bool f8, uint32_t sc) {
    return (sc & SC_LNBWD) && dp == d && (d == 256 || (d == 512 && f8 && (sc & SC_LNBWD_512)));
}

// One Block backward: data grads (7 launches) + all weight/bias grads of the block (1 launch).
int block_bwd(const BlkP& P, uint32_t sc, const BlkOff& o, float* grads, const float* x_in, const BlkBuf& b, int64_t M, int d,
              int heads, int h, int hp, int Ts, int nsamples, int mode, int len_l, float* G0, const Scr& w,
              float* dx_out, int accumulate, hipStream_t s, int concurrent = 1, const float* rs_a = nullptr,
              const float* rs_m = nullptr, int64_t* det_acc = nullptr) {
    float* G1 = w.G1;
    GemmParams p = gp();
    const int dp = rup(d, 32);                        // storage width (see block_fwd)
    LnBwdParams l; std::memset(&l, 0, sizeof(l));
    l.M = (int)M; l.d = d; l.ld = dp; l.det_base = grads; l.det_acc = det_acc;
    const bool f8 = P.prec == HSIMAE_PREC_FP8;
    const bool f8u = f8 && (sc & SC_FP8_UNFUSED);
    auto w8 = [&](GemmParams& q, const BlkP::I8& a) { if (f8) { q.prec = HSIMAE_PREC_FP8; q.W8 = a.w; q.S8 = a.s; } };
    const bool fmlp = !f8u && fused_mlp_enabled(d, h, sc);
    bool g1b_done = false;
    // g / dh1 / dh3 as 64-column planes (include/hsimae_hip.h, hsimae_wgrad_task): needs whole 32-row DMA chunks and 32-bit offsets
    const int hp64 = rup(hp, 64);
    const bool planar = fmlp && (sc & SC_PLANAR) && M % 32 == 0 && (M + kPlanePadRows) * 2 * (int64_t)(hp64 + 256) * 2 < (1ll << 32);
    const int prow = planar ? (int)M + HS_PLANE_PAD_ROWS : 0;       // rows per plane (plan.h: the arena reserves kPlanePadRows)
    if (fmlp) {
        // recompute u2 / h1 / h3 / g inside the tile; emits dx1 and the wgrad operands u2, dh1|dh3, g, bf16 dY and dx1
        CK(hs_enc_mlp_bwd(b.x1, G0, G1, b.u2, w.dh13, b.g, w.g0b, w.g1b, (int)M, d, mlp_ptrs(P, h), grads + o.n2w,
                          grads + o.n2b, s, rs_m, rs_a, HsDet{grads, reinterpret_cast<long long*>(det_acc)}, prow));
    } else {
        p.A = G0; p.lda = dp; p.M = (int)M; p.N = hp; p.K = dp; p.n_valid = hp; p.W = P.w2T; p.out = w.dh13; p.ldo = 2 * hp;
        p.h13 = b.h13; p.ldh = 2 * hp; p.hoff = hp; p.a_rowscale = rs_m;        // DropPath: the branch saw rs_m * dY
        w8(p, P.w2T8);
        CK(hs_gemm(p, A_F32, E_SWIGLU_BWD, s));
        p = gp();
        p.A = w.dh13; p.lda = 2 * hp; p.M = (int)M; p.N = dp; p.K = 2 * hp; p.n_valid = d; p.W = P.w13T;
        w8(p, P.w13T8);
        if (wide_ln_fused(d, dp, f8, sc)) {   // LayerNorm-2 backward as the epilogue of the k-outer GEMM (the whole row is on chip)
            p.out = G1; p.ldo = dp; p.res = G0; p.ldr = dp; p.lnx = b.x1; p.gamma = P.n2w; p.accumulate = 0;
            p.dgamma = grads + o.n2w; p.dbeta = grads + o.n2b; p.det_base = grads; p.det_acc = det_acc;
            if (!rs_a) { p.u_out = w.g1b; p.ldu = dp; g1b_done = true; }      // bf16 copy of dx1 from the epilogue (no DropPath factor to fold in)
            CK(hs_gemm(p, A_BF16, E_LN_BWD, s));
        } else {
            p.out = w.du; p.ldo = dp;
            CK(hs_gemm(p, A_BF16, E_F32, s));
            l.du = w.du; l.x = b.x1; l.gamma = P.n2w; l.dres = G0; l.dx = G1; l.accumulate = 0;
            l.dgamma = grads + o.n2w; l.dbeta = grads + o.n2b;
            CK(hs_ln_bwd(l, s));
        }
        // bf16 copies of dY / dx1 (with the DropPath factors folded in) so that the weight gradients below take the
        // LDS-DMA kernel: 666 -> ~300 us per block at D = 256
        CK(hs_rows_to_bf16(G0, w.g0b, M, dp, rs_m, s));
        if (!g1b_done) CK(hs_rows_to_bf16(G1, w.g1b, M, dp, rs_a, s));
    }
    // All weight / bias gradients of the block: 7 tasks for the batched weight-gradient kernel (every operand is a bf16 buffer on the
    // fused path, so no launch below reads G0 / G1, which the LayerNorm-1 backward overwrites with dx when the caller runs in place).
    // (Round 4 measured the same work as two launches, each right behind the kernel that produced its operands: + 0.3 ms per step,
    //  removed in round 5.)
    auto run_wgrad = [&]() -> int {
        WgradParams g; std::memset(&g, 0, sizeof(g));
        auto task = [&](int, const void* dO, int ldo, const hs_bf16* A, int lda, int N, int K, int64_t dW, int64_t db) {
            WgradTask& t = g.t[g.ntasks++];
            t.dO = dO; t.dO_f32 = 0; t.ldo = ldo; t.A = A; t.lda = lda; t.N = N; t.K = K; t.dW = grads + dW; t.ldw = K;
            t.db = grads + db; t.dO_rowscale = nullptr;
        };
        task(0, w.dqkv, 3 * dp, b.u, dp, d, d, o.qw, o.qb);
        task(1, w.dqkv + dp, 3 * dp, b.u, dp, d, d, o.kw, o.kb);
        task(2, w.dqkv + 2 * dp, 3 * dp, b.u, dp, d, d, o.vw, o.vb);
        task(3, w.g1b, dp, b.o, dp, d, d, o.pw, o.pb);              // all-bf16 operands: wgrad takes its LDS-DMA path
        if (planar) {
            task(4, w.dh13, hp64, b.u2, dp, h, d, o.w1w, o.w1b); g.t[g.ntasks - 1].dO_plane_rows = prow;
            task(5, w.dh13 + (int64_t)hp64 * prow, hp64, b.u2, dp, h, d, o.w3w, o.w3b); g.t[g.ntasks - 1].dO_plane_rows = prow;
        } else {
            task(4, w.dh13, 2 * hp, b.u2, dp, h, d, o.w1w, o.w1b);
            task(5, w.dh13 + hp, 2 * hp, b.u2, dp, h, d, o.w3w, o.w3b);
        }
#ifdef HS_ABL_DW2           /* timing ablation (variant builds only, see fused_enc.hip): the W2 task leaves the launch on the fused path */
        if (!fmlp)
#endif
        { task(6, w.g0b, dp, b.g, planar ? hp64 : hp, d, h, o.w2w, o.w2b); g.t[g.ntasks - 1].A_plane_rows = prow; }
        g.M = (int)M; g.det_base = grads; g.det_acc = det_acc;
        g.slab = (sc & SC_WGRAD_SLAB) ? w.slab : nullptr;        // this stream's slab (clear: float atomics on dW also in the 256 x 256-tile launches)
        int tiles = 0;
        for (int i = 0; i < g.ntasks; ++i) tiles += ((g.t[i].N + 127) / 128) * ((g.t[i].K + 127) / 128);
        g.msplit = wgrad_msplit(tiles, M, concurrent);
        return hs_wgrad(g, s);
    };
    AttnParams a; std::memset(&a, 0, sizeof(a));
    a.qkv = b.qkv; a.ld = 3 * dp; a.d = d; a.heads = heads; a.hd = d / heads; a.Ts = Ts; a.nsamples = nsamples;
    a.mode = mode; a.len_l = len_l; a.o = b.o; a.ldo = dp; a.lse = b.lse; a.dout = w.dob; a.lddo = dp; a.dqkv = w.dqkv; a.kv_off = dp;
    const bool fuse_pb = (sc & SC_PROJ_BWD) != 0;   // clear: keep the projection's data gradient a separate GEMM
    const bool fuse_ln = (sc & SC_LNBWD) != 0;
    // round 4: dO, the attention backward, du and the LayerNorm-1 backward as ONE persistent launch (attn.hip blk128_bwd_kernel);
    // with q|k|v recomputed from u when the forward did not save them (attn_bwd_recompute: the same predicate of the same word)
    const bool rc = attn_bwd_recompute(d, dp, heads, h, Ts, f8u, sc);
    const bool blk_bwd = rc || (fuse_pb && fuse_ln && fmlp && !f8u && d == 128 && dp == d && hs_attn_proj_fusable(a) &&
                                (sc & SC_ATTN_BLOCK_BWD) && hs_attn_block_bwd_fusable(d, heads, Ts));
    // round 5, D = 256 (Large): the same four steps as one launch (attn_wide.hip blk256_bwd_kernel), from the saved q|k|v
    const bool blk256_bwd = !blk_bwd && !f8 && dp == d && (sc & SC_ATTN_BLOCK256_BWD) && (sc & SC_LNBWD) && (sc & SC_PROJ_BWD) &&
                            hs_attn_block256_bwd_fusable(d, heads, Ts, nsamples);
    if (blk256_bwd) {
        CK(hs_attn_block256_bwd(b.qkv, b.lse, w.g1b, G1, x_in, P.n1w, P.pT, P.qkvT, w.dqkv, dx_out, grads + o.n1w, grads + o.n1b, grads,
                                reinterpret_cast<long long*>(det_acc), Ts, nsamples, mode, len_l, accumulate, s));
        CK(run_wgrad());
        return HSIMAE_OK;
    }
    if (blk_bwd) {
        CK(hs_attn_block_bwd(rc ? nullptr : b.qkv, b.u, P.qkv, P.bqkv, b.o, b.lse, w.g1b, G1, x_in, P.n1w, P.pT, P.qkvT, w.dqkv, dx_out,
                             grads + o.n1w, grads + o.n1b, grads, reinterpret_cast<long long*>(det_acc), Ts, nsamples, mode, len_l,
                             accumulate, s));
    } else {
        p = gp();
        p.A = w.g1b; p.lda = dp; p.M = (int)M; p.N = dp; p.K = dp; p.n_valid = dp;  // (the bf16 copy carries the DropPath factor)
        p.W = P.pT; p.out = w.dob; p.ldo = dp;
        w8(p, P.pT8);
        CK(hs_gemm(p, A_BF16, E_BF16, s));
    }
    if (!blk_bwd) CK(hs_attn_bwd(a, s));
    CK(run_wgrad());

    if (blk_bwd) return HSIMAE_OK;
    // du = dqkv * Wqkv and the LayerNorm-1 backward: one kernel at d = 128 (LN backward as the GEMM's epilogue,
    // du never goes to HBM), two otherwise.  HSIMAE_FUSED_LNBWD=0 forces the two-kernel form.
    const bool ln_fused = fuse_ln && ((d == 128 && !f8u) || wide_ln_fused(d, dp, f8, sc));
    p = gp();
    p.A = w.dqkv; p.lda = 3 * dp; p.M = (int)M; p.N = dp; p.K = 3 * dp; p.n_valid = d; p.W = P.qkvT;
    if (ln_fused) {
        p.out = dx_out; p.ldo = d; p.res = G1; p.ldr = d; p.lnx = x_in; p.gamma = P.n1w; p.accumulate = accumulate;
        p.dgamma = grads + o.n1w; p.dbeta = grads + o.n1b; p.det_base = grads; p.det_acc = det_acc;
        if (d != 128) w8(p, P.qkvT8);
        CK(hs_gemm(p, A_BF16, E_LN_BWD, s));       // d = 128: row-panel kernel with the LN backward as its epilogue; d = 256 / 512: k-outer
    } else {
        p.out = w.du; p.ldo = dp;
        w8(p, P.qkvT8);
        CK(hs_gemm(p, A_BF16, E_F32, s));
    }

    if (!ln_fused) {
        l.du = w.du; l.x = x_in; l.gamma = P.n1w; l.dres = G1; l.dx = dx_out; l.accumulate = accumulate;
        l.dgamma = grads + o.n1w; l.dbeta = grads + o.n1b;
        CK(hs_ln_bwd(l, s));
    }
    return HSIMAE_OK;
}

struct Ctx {
    Geo g; PLayout L; WLayout W; Ws w;
    int N, K, len_t, len_l;
    int64_t Me, Md;
};

int make_ctx(const hsimae_config* cfg, const hsimae_io* io, Ctx& c, bool need_ws) {
    CK(make_geo(cfg, c.g));
    if (!io) return HSIMAE_ENULL;
    c.N = io->N; c.len_t = io->len_t; c.len_l = io->len_l;
    if (c.N <= 0 || c.len_t < 1 || c.len_t > c.g.T || c.len_l < 1 || c.len_l > 9) return HSIMAE_EDIMS;
    c.K = c.len_t * c.len_l;
    c.Me = (int64_t)c.N * c.K; c.Md = (int64_t)c.N * c.g.TL;
    if (c.Md * 96 >= (1ll << 31) || c.Me * 3 * c.g.Dp >= (1ll << 31)) return HSIMAE_EUNSUPPORTED;   // 32-bit row math in kernels
    make_playout(c.g, c.L);
    make_wlayout(c.g, c.W);
    if (need_ws) {
        if (!io->workspace || !io->params || !io->wpk) return HSIMAE_ENULL;
        if (reinterpret_cast<uintptr_t>(io->workspace) % 256) return HSIMAE_EALIGN;
        carve(c.g, c.N, c.K, reinterpret_cast<char*>(io->workspace), c.w);
        if (c.w.bytes > io->workspace_bytes) return HSIMAE_EDIMS;
    }
    return HSIMAE_OK;
}

}  // namespace

// ====================================================================== C ABI
extern "C" unsigned hs_variant_bits_gemm();
extern "C" unsigned hs_variant_bits_attn();
extern "C" unsigned hs_variant_bits_attn_wide();
extern "C" unsigned hs_variant_bits_wgrad();
extern "C" unsigned hs_variant_bits_elem();
extern "C" unsigned hs_variant_bits_pack();
extern "C" unsigned hs_variant_bits_fused_dec();
extern "C" unsigned hs_variant_bits_fused_enc();
extern "C" unsigned hs_variant_bits_loader();
#ifndef HS_KERNEL_SOURCE_HASH
#define HS_KERNEL_SOURCE_HASH 0ULL
#endif
#ifndef HS_BUILD_FLAGS_HASH
#define HS_BUILD_FLAGS_HASH 0ULL
#endif
#ifndef HS_BUILD_DEFAULT_FLAGS
#define HS_BUILD_DEFAULT_FLAGS 0
#endif

extern "C" {

int hsimae_version(void) { return HSIMAE_VERSION; }

int hsimae_build_info(hsimae_build_info_t* out) {
    if (!out) return HSIMAE_ENULL;
    out->abi_version = HSIMAE_VERSION;
    out->variant_bits = hs_variant_bits() | hs_variant_bits_gemm() | hs_variant_bits_attn() | hs_variant_bits_attn_wide() | hs_variant_bits_wgrad() | hs_variant_bits_elem() | hs_variant_bits_pack() | hs_variant_bits_fused_dec() | hs_variant_bits_fused_enc() | hs_variant_bits_loader();
    out->kernel_source_hash = HS_KERNEL_SOURCE_HASH;
    out->flags_hash = HS_BUILD_FLAGS_HASH;
    out->default_flags = HS_BUILD_DEFAULT_FLAGS;
    out->reserved = 0;
    return HSIMAE_OK;
}
const char* hsimae_variant_name(int bit) {
    switch (bit) {
#define HS_X(b, name) case b: return #name;
        HS_VARIANT_TABLE(HS_X)
#undef HS_X
        default: return nullptr;
    }
}

int hsimae_two_streams_active(void) { return side().ok ? 1 : 0; }
int hsimae_effective_precision(const hsimae_config* cfg) {
    if (!cfg) return HSIMAE_ENULL;
    Geo g; CK(make_geo(cfg, g));
    return (g.prec == HSIMAE_PREC_FP8 && (g.D >= 512 || (sched_from_env() & SC_FP8_UNFUSED))) ? HSIMAE_PREC_FP8 : HSIMAE_PREC_BF16;
}

const char* hsimae_strerror(int code) {
    switch (code) {
        case HSIMAE_OK: return "ok";
        case HSIMAE_EDIMS: return "bad dimensions or strides";
        case HSIMAE_EUNSUPPORTED: return "configuration not supported by the gfx950 kernels";
        case HSIMAE_EALIGN: return "misaligned pointer";
        case HSIMAE_ENULL: return "required pointer is NULL";
        case HSIMAE_ENOFORWARD: return "no forward pass of this process has filled this workspace (the backward follows the forward's recorded schedule)";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int hsimae_param_layout(const hsimae_config* cfg, int64_t* offsets, int64_t* sizes, int max_entries) {
    return param_layout(cfg, offsets, sizes, max_entries);
}

int64_t hsimae_wpk_elems(const hsimae_config* cfg) { return wpk_elems(cfg); }

int64_t hsimae_pack_table_bytes(const hsimae_config* cfg) { return pack_table_bytes(cfg); }

int hsimae_build_pack_table(const hsimae_config* cfg, const float* params_dev, hs_bf16* wpk_dev, void* table_host) {
    return build_pack_table(cfg, params_dev, wpk_dev, table_host);
}

int hsimae_pack_params(const hsimae_config* cfg, const void* table_dev, void* stream) {
    Geo g; CK(make_geo(cfg, g));
    if (!table_dev) return HSIMAE_ENULL;
    const int n = (int)(hsimae_pack_table_bytes(cfg) / (int64_t)sizeof(PackDesc));
    const int maxe = std::max({g.D * g.D, g.h * g.D, g.Dd * g.Dd, g.hdec * g.Dd, g.D * g.Dd, g.D * 72});
    return hs_pack(reinterpret_cast<const PackDesc*>(table_dev), n, maxe, S(stream));
}

int64_t hsimae_workspace_bytes(const hsimae_config* cfg, int32_t N, int32_t len_t, int32_t len_l) {
    return workspace_bytes(cfg, N, len_t, len_l);
}

// DropPath factors of encoder block `e` (execution order blocks_1, blocks_2, blocks): {attention, MLP} row vectors
struct DropRs { const float* a; const float* m; };
static DropRs drop_rs(const hsimae_io* io, int e, int64_t Me) {
    if (!io->drop_scale) return DropRs{nullptr, nullptr};
    const float* base = io->drop_scale + (int64_t)e * 2 * Me;
    return DropRs{base, base + Me};
}

static int forward_impl(const hsimae_config* cfg, const hsimae_io* io, void* stream, bool encoder_only) {
    Ctx c; CK(make_ctx(cfg, io, c, true));
    if (!io->x || !io->noise1 || !io->noise2 || !io->mask || !io->ids_keep || !io->ids_restore) return HSIMAE_ENULL;
    if (encoder_only ? !io->latent : !io->loss) return HSIMAE_ENULL;
    if (!encoder_only && io->want_recons && (!io->pred_img || !io->mask_img)) return HSIMAE_ENULL;
    hipStream_t s = S(stream);
    const Geo& g = c.g; const float* P = io->params; const Ws& w = c.w;
    const uint32_t sc = sched_from_env();             // the pass's schedule: read here, recorded for the backward (see SC_*)
    record_sched(io->workspace, true, !encoder_only, sc);

    MaskParams m; m.noise1 = io->noise1; m.noise2 = io->noise2; m.N = c.N; m.T = g.T; m.L = 9; m.len_t = c.len_t;
    m.len_l = c.len_l; m.ids_keep = io->ids_keep; m.ids_restore = io->ids_restore; m.mask = io->mask;
    CK(hs_mask(m, s));
    PatchParams pp; pp.x = io->x; pp.sn = io->sn; pp.sb = io->sb; pp.sh = io->sh; pp.sw = io->sw; pp.N = c.N; pp.T = g.T;
    pp.K = c.K; pp.ids_keep = io->ids_keep; pp.out = w.a_pe; pp.pos_ids = nullptr;
    CK(hs_patch_gather(pp, s));
    GemmParams p = gp();
    p.A = w.a_pe; p.lda = 96; p.M = (int)c.Me; p.N = g.Dp; p.K = 96; p.n_valid = g.D; p.W = io->wpk + c.W.pe; p.bias = P + c.L.peb;
    p.pos = P + c.L.pos; p.ids = io->ids_keep; p.ldpos = g.D; p.out = w.x0; p.ldo = g.Dp;
    CK(hs_gemm(p, A_BF16, E_POS_F32, s));

    const float* x = w.x0;
    if (g.has_axis) {
        Side& sd = side();
        const bool forked = sd.ok && g.sdepth > 1;
        if (forked) {
            CK((int)hipEventRecord(sd.fork, s));
            CK((int)hipStreamWaitEvent(sd.s, sd.fork, 0));
        }
        const float* xa = w.x0;
        const float* xb = w.x0;
        for (int i = 0; i < g.sdepth; ++i) {
            // spatial stack: attend within one kept band group (Models.py:553,556)
            BlkP b1 = resolve_enc(g, sc, c.L.b1[i], c.W.b1[i], P, io->wpk, c.W);
            const DropRs r1 = drop_rs(io, i, c.Me), r2d = drop_rs(io, g.sdepth + i, c.Me);
            CK(block_fwd(b1, sc, xa, w.b1[i], c.Me, g.D, g.H, g.h, g.hp, c.K, c.N, 1, c.len_l, nullptr, s, r1.a, r1.m));
            xa = w.b1[i].x2;
            // spectral stack: attend within one kept position (Models.py:554,559)
            BlkP b2 = resolve_enc(g, sc, c.L.b2[i], c.W.b2[i], P, io->wpk, c.W);
            const bool last = (i == g.sdepth - 1);
            const float* r2 = last ? xa : nullptr;                      // x1 + x2 fused into the last epilogue (Models.py:564)
            if (last && forked) {                                       // needs the spatial stack's result: rejoin first
                CK((int)hipEventRecord(sd.join, sd.s));
                CK((int)hipStreamWaitEvent(s, sd.join, 0));
            }
            CK(block_fwd(b2, sc, xb, w.b2[i], c.Me, g.D, g.H, g.h, g.hp, c.K, c.N, 2, c.len_l, r2, (forked && !last) ? sd.s : s,
                         r2d.a, r2d.m));
            xb = w.b2[i].x2;
        }
        x = xb;
    }
    for (int i = 0; i < g.nfus; ++i) {
        BlkP bp = resolve_enc(g, sc, c.L.bf[i], c.W.bf[i], P, io->wpk, c.W);
        const DropRs rf = drop_rs(io, (g.has_axis ? 2 * g.sdepth : 0) + i, c.Me);
        CK(block_fwd(bp, sc, x, w.bf[i], c.Me, g.D, g.H, g.h, g.hp, c.K, c.N, 0, c.len_l, nullptr, s, rf.a, rf.m));
        x = w.bf[i].x2;
    }
    if (encoder_only) {                       // `norm` only (Models.py:570 / 892): the latent the fine-tuning head reads
        CK(hs_ln_fwd(x, P + c.L.nw, P + c.L.nb, io->latent, (int)c.Me, g.D, s, g.Dp, g.D));
        return HSIMAE_OK;
    }
    // norm + decoder_embed (Models.py:570, 579)
    p = gp();
    p.A = x; p.lda = g.Dp; p.M = (int)c.Me; p.N = g.Ddp; p.K = g.Dp; p.n_valid = g.Dd; p.W = io->wpk + c.W.de; p.bias = P + c.L.deb;
    p.gamma = P + c.L.nw; p.beta = P + c.L.nb; p.u_out = w.lat; p.ldu = g.Dp; p.out = w.y; p.ldo = g.Ddp; p.ln_width = g.D;
    CK(hs_gemm(p, A_F32_LN, E_F32, s));
    AssembleParams as; std::memset(&as, 0, sizeof(as));
    as.y = w.y; as.N = c.N; as.K = c.K; as.TL = g.TL; as.Dd = g.Dd; as.ld = g.Ddp; as.ids_restore = io->ids_restore; as.pos = P + c.L.dpos;
    as.yfull = w.yfull;
    CK(hs_assemble_fwd(as, s));
    const float* z = w.yfull;
    const bool fdec = fused_dec_enabled(g, sc);
    for (int i = 0; i < g.ddepth; ++i) {
        BlkP bp = resolve(c.L.bd[i], c.W.bd[i], P, io->wpk, c.W);
        if (fdec) CK(dec_block_fwd_fused(bp, z, w.bd[i], c.N, c.Md, g.TL, g.Dd, g.hdec, sc, s));
        else CK(block_fwd(bp, sc, z, w.bd[i], c.Md, g.Dd, g.Hd, g.hdec, g.hpd, g.TL, c.N, 0, 9, nullptr, s));
        z = w.bd[i].x2;
    }
    // decoder_norm + decoder_pred (Models.py:597-600)
    p = gp();
    p.A = z; p.lda = g.Ddp; p.M = (int)c.Md; p.N = 80; p.K = g.Ddp; p.n_valid = 72; p.W = io->wpk + c.W.dp; p.bias = P + c.L.dpb;
    p.gamma = P + c.L.dnw; p.beta = P + c.L.dnb; p.u_out = w.zn; p.ldu = g.Ddp; p.out = w.pred; p.ldo = 72; p.ln_width = g.Dd;
    CK(hs_gemm(p, A_F32_LN, E_F32, s));
    LossParams lp; std::memset(&lp, 0, sizeof(lp));
    const float sum_mask = (float)((int64_t)c.N * (g.TL - c.K));
    lp.x = io->x; lp.sn = io->sn; lp.sb = io->sb; lp.sh = io->sh; lp.sw = io->sw; lp.N = c.N; lp.T = g.T; lp.pred = w.pred;
    lp.mask = io->mask; lp.norm_pix = g.norm_pix; lp.inv_scale = io->grad_scale / (72.f * sum_mask); lp.partial = w.partial;
    lp.loss = io->loss; lp.sum_mask = sum_mask; lp.dpred = w.dpred;
    lp.pred_img = io->want_recons ? io->pred_img : nullptr; lp.mask_img = io->want_recons ? io->mask_img : nullptr;
    CK(hs_loss(lp, s));
    if (io->pred) CK((int)hipMemcpyAsync(io->pred, w.pred, c.Md * 72 * 4, hipMemcpyDeviceToDevice, s));
    if (io->latent) CK(hs_ln_fwd(x, P + c.L.nw, P + c.L.nb, io->latent, (int)c.Me, g.D, s, g.Dp, g.D));
    return HSIMAE_OK;
}

int hsimae_forward(const hsimae_config* cfg, const hsimae_io* io, void* stream) { return forward_impl(cfg, io, stream, false); }
int hsimae_encode(const hsimae_config* cfg, const hsimae_io* io, void* stream) { return forward_impl(cfg, io, stream, true); }

int hsimae_decode(const hsimae_config* cfg, const hsimae_io* io, const float* latent, float* pred, void* stream) {
    Ctx c; CK(make_ctx(cfg, io, c, true));
    if (!latent || !pred || !io->ids_restore) return HSIMAE_ENULL;
    hipStream_t s = S(stream);
    const Geo& g = c.g; const float* P = io->params; const Ws& w = c.w;
    const uint32_t sc = sched_from_env();
    record_sched(io->workspace, false, true, sc);     // (the decoder's part only: an encoder pass in the same arena keeps its own record)
    GemmParams p = gp();                      // decoder_embed (Models.py:579)
    const float* lat_in = latent;
    if (g.Dp != g.D) {      // rows stored wider than the model: stage the caller's [Me][D] latent into a padded buffer (pad columns are zeros)
        CK((int)hipMemcpy2DAsync(w.G2, (size_t)g.Dp * 4, latent, (size_t)g.D * 4, (size_t)g.D * 4, (size_t)c.Me, hipMemcpyDeviceToDevice, s));
        lat_in = w.G2;
    }
    p.A = lat_in; p.lda = g.Dp; p.M = (int)c.Me; p.N = g.Ddp; p.K = g.Dp; p.n_valid = g.Dd; p.W = io->wpk + c.W.de; p.bias = P + c.L.deb;
    p.out = w.y; p.ldo = g.Ddp;
    CK(hs_gemm(p, A_F32, E_F32, s));
    CK(hs_rows_to_bf16(lat_in, w.lat, c.Me, g.Dp, nullptr, s));     // decoder_embed's wgrad operand (hsimae_decode_backward)
    AssembleParams as; std::memset(&as, 0, sizeof(as));
    as.y = w.y; as.N = c.N; as.K = c.K; as.TL = g.TL; as.Dd = g.Dd; as.ld = g.Ddp; as.ids_restore = io->ids_restore; as.pos = P + c.L.dpos;
    as.yfull = w.yfull;
    CK(hs_assemble_fwd(as, s));
    const float* z = w.yfull;
    const bool fdec = fused_dec_enabled(g, sc);
    for (int i = 0; i < g.ddepth; ++i) {
        BlkP bp = resolve(c.L.bd[i], c.W.bd[i], P, io->wpk, c.W);
        if (fdec) CK(dec_block_fwd_fused(bp, z, w.bd[i], c.N, c.Md, g.TL, g.Dd, g.hdec, sc, s));
        else CK(block_fwd(bp, sc, z, w.bd[i], c.Md, g.Dd, g.Hd, g.hdec, g.hpd, g.TL, c.N, 0, 9, nullptr, s));
        z = w.bd[i].x2;
    }
    p = gp();                                 // decoder_norm + decoder_pred (Models.py:597-600)
    p.A = z; p.lda = g.Ddp; p.M = (int)c.Md; p.N = 80; p.K = g.Ddp; p.n_valid = 72; p.W = io->wpk + c.W.dp; p.bias = P + c.L.dpb;
    p.gamma = P + c.L.dnw; p.beta = P + c.L.dnb; p.u_out = w.zn; p.ldu = g.Ddp; p.out = pred; p.ldo = 72; p.ln_width = g.Dd;
    return hs_gemm(p, A_F32_LN, E_F32, s);
}

// Backward of the encoder stacks + patch embedding, from d(x of the last encoder block) in w.G0.  Shared by
// hsimae_backward (after the decoder) and hsimae_encode_backward (after `norm`).
static int encoder_backward(const Ctx& c, const hsimae_io* io, float* grads, hipStream_t s, Emitter& emit, uint32_t sc) {
    const Geo& g = c.g; const float* P = io->params; const Ws& w = c.w; const PLayout& L = c.L;
    for (int i = g.nfus - 1; i >= 0; --i) {
        BlkP bp = resolve_enc(g, sc, L.bf[i], c.W.bf[i], P, io->wpk, c.W);
        const float* xin = (i > 0) ? w.bf[i - 1].x2 : (g.has_axis ? w.b2[g.sdepth - 1].x2 : w.x0);
        const DropRs rf = drop_rs(io, (g.has_axis ? 2 * g.sdepth : 0) + i, c.Me);
        CK(block_bwd(bp, sc, L.bf[i], grads, xin, w.bf[i], c.Me, g.D, g.H, g.h, g.hp, c.K, c.N, 0, c.len_l, w.G0, w.sc, w.G0, 0, s, 1,
                     rf.a, rf.m, io->det_acc));
        CK(emit(L.bf[i].n1w, L.bf[i].end, s));
    }
    if (g.has_axis) {
        // d(x1 + x2) feeds both stacks (Models.py:564); the spectral stack's backward runs on the side stream
        Side& sd = side();
        const bool forked = sd.ok && g.sdepth > 1;
        CK((int)hipMemcpyAsync(w.G2, w.G0, c.Me * g.Dp * 4, hipMemcpyDeviceToDevice, s));
        if (forked) {
            CK((int)hipEventRecord(sd.fork, s));
            CK((int)hipStreamWaitEvent(sd.s, sd.fork, 0));
        }
        hipStream_t s2 = forked ? sd.s : s;
        const Scr& scr2 = forked ? w.sc2 : w.sc;
        // with a bucket stream every block's range is reported as soon as its kernels are enqueued (the consumer waits
        // for the right stream through an event); without one the side stream's ranges wait for the join
        const bool per_block = emit.bucket != nullptr || !forked;
        for (int i = g.sdepth - 1; i >= 0; --i) {
            BlkP b2 = resolve_enc(g, sc, L.b2[i], c.W.b2[i], P, io->wpk, c.W);
            const DropRs r1 = drop_rs(io, i, c.Me), r2d = drop_rs(io, g.sdepth + i, c.Me);
            const float* xin2 = (i > 0) ? w.b2[i - 1].x2 : w.x0;
            CK(block_bwd(b2, sc, L.b2[i], grads, xin2, w.b2[i], c.Me, g.D, g.H, g.h, g.hp, c.K, c.N, 2, c.len_l, w.G0, scr2, w.G0, 0, s2, forked ? 2 : 1,
                         r2d.a, r2d.m, io->det_acc));
            if (per_block) CK(emit(L.b2[i].n1w, L.b2[i].end, s2));
            if (i == 0 && forked) {                 // the spatial stack's last step accumulates onto the spectral dX
                CK((int)hipEventRecord(sd.join, sd.s));
                CK((int)hipStreamWaitEvent(s, sd.join, 0));
            }
            BlkP b1 = resolve_enc(g, sc, L.b1[i], c.W.b1[i], P, io->wpk, c.W);
            const float* xin1 = (i > 0) ? w.b1[i - 1].x2 : w.x0;
            float* out = (i == 0) ? w.G0 : w.G2;
            CK(block_bwd(b1, sc, L.b1[i], grads, xin1, w.b1[i], c.Me, g.D, g.H, g.h, g.hp, c.K, c.N, 1, c.len_l, w.G2, w.sc, out, i == 0, s, forked ? 2 : 1,
                         r1.a, r1.m, io->det_acc));
            if (per_block) CK(emit(L.b1[i].n1w, L.b1[i].end, s));
        }
        if (!per_block) {     // back-to-front, all complete on the caller's stream by now
            for (int i = g.sdepth - 1; i >= 0; --i) CK(emit(L.b2[i].n1w, L.b2[i].end, s));
            for (int i = g.sdepth - 1; i >= 0; --i) CK(emit(L.b1[i].n1w, L.b1[i].end, s));
        }
    }
    {   // patch_embed.proj: only the kept tokens carry gradient (Models.py:528); no input gradient
        WgradParams wg; std::memset(&wg, 0, sizeof(wg));
        WgradTask& t = wg.t[0]; t.dO = w.G0; t.dO_f32 = 1; t.ldo = g.Dp; t.A = w.a_pe; t.lda = 96; t.N = g.D; t.K = 72;
        t.dW = grads + L.pew; t.ldw = 72; t.db = grads + L.peb;
        wg.det_base = grads; wg.det_acc = io->det_acc;
        wg.ntasks = 1; wg.M = (int)c.Me; wg.msplit = wgrad_msplit((g.D + 127) / 128, c.Me);
        CK(hs_wgrad(wg, s));
    }
    CK(emit(0, L.peb + g.D, s));
    return HSIMAE_OK;
}

// Backward of the decoder from dL/dpred (bf16 [Md][96] in w.dpred) down to dL/d(latent) in w.du [Me][D]
// (autograd of Models.py:573-601): decoder_pred / decoder_norm, the decoder blocks, the sequence assembly and
// decoder_embed (its weight gradient included).  Shared by hsimae_backward and hsimae_decode_backward.
static int decoder_backward(const Ctx& c, const hsimae_io* io, float* grads, hipStream_t s, Emitter& emit, uint32_t sc) {
    const Geo& g = c.g; const float* P = io->params; const Ws& w = c.w; const PLayout& L = c.L;
    const float* zlast = w.bd[g.ddepth - 1].x2;
    GemmParams p = gp();
    p.A = w.dpred; p.lda = 96; p.M = (int)c.Md; p.N = g.Ddp; p.K = 96; p.n_valid = g.Dd; p.W = io->wpk + c.W.dpT; p.out = w.du; p.ldo = g.Ddp;
    CK(hs_gemm(p, A_BF16, E_F32, s));
    {
        WgradParams wg; std::memset(&wg, 0, sizeof(wg));
        WgradTask& t = wg.t[0]; t.dO = w.dpred; t.dO_f32 = 0; t.ldo = 96; t.A = w.zn; t.lda = g.Ddp; t.N = 72; t.K = g.Dd;
        t.dW = grads + L.dpw; t.ldw = g.Dd; t.db = grads + L.dpb;
        wg.det_base = grads; wg.det_acc = io->det_acc;
        wg.ntasks = 1; wg.M = (int)c.Md; wg.msplit = wgrad_msplit(1, c.Md);
        CK(hs_wgrad(wg, s));
    }
    LnBwdParams l; std::memset(&l, 0, sizeof(l));
    l.du = w.du; l.x = zlast; l.gamma = P + L.dnw; l.dres = nullptr; l.dx = w.G0; l.dgamma = grads + L.dnw; l.dbeta = grads + L.dnb;
    l.M = (int)c.Md; l.d = g.Dd; l.ld = g.Ddp; l.det_base = grads; l.det_acc = io->det_acc;
    CK(hs_ln_bwd(l, s));
    CK(emit(L.dnw, L.total, s));

    const bool fdec = fused_dec_enabled(g, sc);
    for (int i = g.ddepth - 1; i >= 0; --i) {
        BlkP bp = resolve(L.bd[i], c.W.bd[i], P, io->wpk, c.W);
        const float* xin = (i == 0) ? w.yfull : w.bd[i - 1].x2;
        if (fdec) {
            const BlkOff& o = L.bd[i];
            DecBlockGrads dg;
            dg.n1w = grads + o.n1w; dg.n1b = grads + o.n1b; dg.qw = grads + o.qw; dg.qb = grads + o.qb;
            dg.kw = grads + o.kw; dg.kb = grads + o.kb; dg.vw = grads + o.vw; dg.vb = grads + o.vb;
            dg.pw = grads + o.pw; dg.pb = grads + o.pb; dg.n2w = grads + o.n2w; dg.n2b = grads + o.n2b;
            dg.w1w = grads + o.w1w; dg.w1b = grads + o.w1b; dg.w2w = grads + o.w2w; dg.w2b = grads + o.w2b;
            dg.w3w = grads + o.w3w; dg.w3b = grads + o.w3b;
            dg.det = HsDet{grads, reinterpret_cast<long long*>(io->det_acc)};
            // MLP half then attention half, both persistent with the block's weight gradients held in registers
            // (measured equal to "row-tile kernel + wgrad operands through HBM" at d = 64, with 0.7 GB less traffic)
            // SC_DEC_SLAB clear: commit the in-register weight gradients with float atomics (rounds 1-2) instead of slab + reduce
            CK(hs_dec_block_bwd(xin, w.bd[i].x1, w.G0, w.G1, w.G0, w.bd[i].o, w.bd[i].lse, c.N, g.TL, dec_ptrs(bp, g.hdec), dg, s,
                                (sc & SC_DEC_SLAB) ? w.slab : nullptr));
        } else {
            CK(block_bwd(bp, sc, L.bd[i], grads, xin, w.bd[i], c.Md, g.Dd, g.Hd, g.hdec, g.hpd, g.TL, c.N, 0, 9, w.G0, w.sc, w.G0, 0, s, 1, nullptr, nullptr,
                         io->det_acc));
        }
        CK(emit(L.bd[i].n1w, L.bd[i].end, s));
    }
    // sequence assembly + decoder_embed
    AssembleParams as; std::memset(&as, 0, sizeof(as));
    as.N = c.N; as.K = c.K; as.TL = g.TL; as.Dd = g.Dd; as.ld = g.Ddp; as.ids_restore = io->ids_restore; as.dyfull = w.G0; as.dy = w.dyb;
    CK(hs_assemble_bwd(as, s));
    p = gp();
    p.A = w.dyb; p.lda = g.Ddp; p.M = (int)c.Me; p.N = g.Dp; p.K = g.Ddp; p.n_valid = g.D; p.W = io->wpk + c.W.deT; p.out = w.du; p.ldo = g.Dp;
    CK(hs_gemm(p, A_BF16, E_F32, s));
    {
        WgradParams wg; std::memset(&wg, 0, sizeof(wg));
        WgradTask& t = wg.t[0]; t.dO = w.dyb; t.dO_f32 = 0; t.ldo = g.Ddp; t.A = w.lat; t.lda = g.Dp; t.N = g.Dd; t.K = g.D;
        t.dW = grads + L.dew; t.ldw = g.D; t.db = grads + L.deb;
        wg.det_base = grads; wg.det_acc = io->det_acc;
        wg.ntasks = 1; wg.M = (int)c.Me; wg.msplit = wgrad_msplit(((g.Dd + 127) / 128) * ((g.D + 127) / 128), c.Me);
        CK(hs_wgrad(wg, s));
    }
    return HSIMAE_OK;
}

int hsimae_backward(const hsimae_config* cfg, const hsimae_io* io, float* grads, hsimae_bucket_cb cb, void* user,
                    void* stream) {
    Ctx c; CK(make_ctx(cfg, io, c, true));
    if (!grads || !io->ids_restore) return HSIMAE_ENULL;
    uint32_t sc_enc = 0, sc_dec = 0;                  // what the forward that filled this arena ran (never the environment's word now)
    CK(lookup_sched(io->workspace, true, true, sc_enc, sc_dec));
    hipStream_t s = S(stream);
    if (io->det_acc) CK((int)hipMemsetAsync(io->det_acc, 0, (size_t)c.L.total * 8, s));     // deterministic mode: fixed-point shadow sums
    const Geo& g = c.g; const float* P = io->params; const Ws& w = c.w; const PLayout& L = c.L;
    Emitter emit{cb, user, cb ? S(io->bucket_stream) : nullptr, 0, grads, io->det_acc};
    CK(decoder_backward(c, io, grads, s, emit, sc_dec));
    // norm (Models.py:570)
    const float* xf = g.nfus ? w.bf[g.nfus - 1].x2 : (g.has_axis ? w.b2[g.sdepth - 1].x2 : w.x0);
    LnBwdParams l; std::memset(&l, 0, sizeof(l));
    l.du = w.du; l.x = xf; l.gamma = P + L.nw; l.dres = nullptr; l.dx = w.G0; l.dgamma = grads + L.nw; l.dbeta = grads + L.nb;
    l.M = (int)c.Me; l.d = g.D; l.ld = g.Dp; l.det_base = grads; l.det_acc = io->det_acc;
    CK(hs_ln_bwd(l, s));
    CK(emit(L.nw, L.deb + g.Dd, s));
    return encoder_backward(c, io, grads, s, emit, sc_enc);
}

int hsimae_decode_backward(const hsimae_config* cfg, const hsimae_io* io, const float* dpred, float* dlatent, float* grads,
                           hsimae_bucket_cb cb, void* user, void* stream) {
    Ctx c; CK(make_ctx(cfg, io, c, true));
    if (!grads || !dpred || !dlatent || !io->ids_restore) return HSIMAE_ENULL;
    uint32_t sc_enc = 0, sc_dec = 0;
    CK(lookup_sched(io->workspace, false, true, sc_enc, sc_dec));
    hipStream_t s = S(stream);
    if (io->det_acc) CK((int)hipMemsetAsync(io->det_acc, 0, (size_t)c.L.total * 8, s));     // deterministic mode: fixed-point shadow sums
    const Geo& g = c.g; const Ws& w = c.w; const PLayout& L = c.L;
    CK(hs_rows_pad_bf16(dpred, w.dpred, c.Md, 72, 96, s));
    Emitter emit{cb, user, cb ? S(io->bucket_stream) : nullptr, 0, grads, io->det_acc};
    CK(decoder_backward(c, io, grads, s, emit, sc_dec));
    CK(emit(L.dew, L.deb + g.Dd, s));
    return (int)hipMemcpy2DAsync(dlatent, (size_t)g.D * 4, w.du, (size_t)g.Dp * 4, (size_t)g.D * 4, (size_t)c.Me, hipMemcpyDeviceToDevice, s);
}

int hsimae_encode_backward(const hsimae_config* cfg, const hsimae_io* io, const float* dlatent, float* grads,
                           hsimae_bucket_cb cb, void* user, void* stream) {
    Ctx c; CK(make_ctx(cfg, io, c, true));
    if (!grads || !dlatent) return HSIMAE_ENULL;
    uint32_t sc_enc = 0, sc_dec = 0;
    CK(lookup_sched(io->workspace, true, false, sc_enc, sc_dec));
    hipStream_t s = S(stream);
    if (io->det_acc) CK((int)hipMemsetAsync(io->det_acc, 0, (size_t)c.L.total * 8, s));     // deterministic mode: fixed-point shadow sums
    const Geo& g = c.g; const float* P = io->params; const Ws& w = c.w; const PLayout& L = c.L;
    // `norm` (Models.py:892): dlatent -> d(x of the last encoder block) in G0
    const float* xf = g.nfus ? w.bf[g.nfus - 1].x2 : (g.has_axis ? w.b2[g.sdepth - 1].x2 : w.x0);
    LnBwdParams l; std::memset(&l, 0, sizeof(l));
    const float* dlat = dlatent;
    if (g.Dp != g.D) {          // rows stored wider than the model: the caller's [Me][D] gradient into a padded buffer
        CK((int)hipMemcpy2DAsync(w.du, (size_t)g.Dp * 4, dlatent, (size_t)g.D * 4, (size_t)g.D * 4, (size_t)c.Me, hipMemcpyDeviceToDevice, s));
        dlat = w.du;
    }
    l.du = dlat; l.x = xf; l.gamma = P + L.nw; l.dres = nullptr; l.dx = w.G0; l.dgamma = grads + L.nw; l.dbeta = grads + L.nb;
    l.M = (int)c.Me; l.d = g.D; l.ld = g.Dp; l.det_base = grads; l.det_acc = io->det_acc;
    CK(hs_ln_bwd(l, s));
    Emitter emit{cb, user, cb ? S(io->bucket_stream) : nullptr, 0, grads, io->det_acc};
    CK(emit(L.nw, L.nb + g.D, s));
    return encoder_backward(c, io, grads, s, emit, sc_enc);
}

// ---------------------------------------------------------------------- per-kernel entry points
int hsimae_mask_from_noise(const hsimae_mask_params* p, void* stream) { return p ? hs_mask(*p, S(stream)) : HSIMAE_ENULL; }
int hsimae_patch_gather(const hsimae_patch_params* p, void* stream) { return p ? hs_patch_gather(*p, S(stream)) : HSIMAE_ENULL; }
int hsimae_gemm(const hsimae_gemm_params* p, int32_t a_kind, int32_t epilogue, void* stream) {
    return p ? hs_gemm(*p, a_kind, epilogue, S(stream)) : HSIMAE_ENULL;
}
int hsimae_gemm_tiled(const hsimae_gemm_params* p, int32_t a_kind, int32_t epilogue, int32_t bm, int32_t kc, void* stream) {
    return p ? hs_gemm_tiled(*p, a_kind, epilogue, bm, kc, S(stream)) : HSIMAE_ENULL;
}
int hsimae_pack_matrix(const hsimae_pack_desc* d, int32_t n, int32_t max_elems, void* stream) {
    return d ? hs_pack(d, n, max_elems, S(stream)) : HSIMAE_ENULL;
}
static EncMlpPtrs mlp_from_abi(const hsimae_mlp_weights* w) {
    EncMlpPtrs m;
    m.n2w = w->n2w; m.n2b = w->n2b; m.w1b = w->w1b; m.w3b = w->w3b; m.w2b = w->w2b;
    m.w1 = w->w1; m.w3 = w->w3; m.w2 = w->w2; m.w2T = w->w2T; m.w13T = w->w13T; m.h = w->hidden;
    return m;
}
int hsimae_enc_mlp_fwd(const float* x1, const float* res2, float* x2, int32_t M, int32_t d, const hsimae_mlp_weights* w,
                       const float* rowscale, void* stream) {
    if (M <= 0) return HSIMAE_OK;
    if (!x1 || !x2 || !w) return HSIMAE_ENULL;
    if (!hs_enc_mlp_fused_supported(d, w->hidden)) return HSIMAE_EUNSUPPORTED;
    return hs_enc_mlp_fwd(x1, res2, x2, M, d, mlp_from_abi(w), S(stream), rowscale);
}
int hsimae_enc_mlp_bwd(const float* x1, const float* dy, float* dx1, hs_bf16* u2, hs_bf16* dh13, hs_bf16* g, hs_bf16* dyb,
                       hs_bf16* dx1b, int32_t M, int32_t d, const hsimae_mlp_weights* w, float* g_n2w, float* g_n2b,
                       const float* rs_mlp, const float* rs_attn, int32_t plane_rows, void* stream) {
    if (plane_rows && plane_rows < M) return HSIMAE_EDIMS;
    if (M <= 0) return HSIMAE_OK;
    if (!x1 || !dy || !dx1 || !w || !g_n2w || !g_n2b) return HSIMAE_ENULL;
    // operand outputs are optional as groups: {u2, dyb} and {dh13, g} (NULL = not written: the data path alone), dx1b on its own
    if ((!u2) != (!dyb) || (!dh13) != (!g)) return HSIMAE_ENULL;
    if (!hs_enc_mlp_fused_supported(d, w->hidden)) return HSIMAE_EUNSUPPORTED;
    return hs_enc_mlp_bwd(x1, dy, dx1, u2, dh13, g, dyb, dx1b, M, d, mlp_from_abi(w), g_n2w, g_n2b, S(stream), rs_mlp, rs_attn,
                          HsDet{nullptr, nullptr}, plane_rows);
}
static DecBlockPtrs dec_from_abi(const hsimae_dec_block_weights* w) {
    DecBlockPtrs d; std::memset(&d, 0, sizeof(d));
    d.n1w = w->n1w; d.n1b = w->n1b; d.bqkv = w->bqkv; d.pb = w->pb; d.n2w = w->n2w; d.n2b = w->n2b;
    d.w1b = w->w1b; d.w3b = w->w3b; d.w2b = w->w2b;
    d.qkv = w->qkv; d.p = w->p; d.w1 = w->w1; d.w3 = w->w3; d.w2 = w->w2; d.w2T = w->w2T;
    d.qf = w->qf; d.kf = w->kf; d.vf = w->vf; d.pf = w->pf; d.w1f = w->w1f; d.w3f = w->w3f; d.h = w->hidden;
    return d;
}
int hsimae_dec_block_fwd(const hsimae_dec_block_weights* w, const float* x, float* x1, float* x2, hs_bf16* o, float* lse,
                         int32_t nsamples, int32_t Ts, int32_t split, void* stream) {
    if (nsamples <= 0) return HSIMAE_OK;
    if (!w || !x || !x1 || !x2 || !o || !lse) return HSIMAE_ENULL;
    if (!hs_dec_fused_supported(64, 8, w->hidden, Ts)) return HSIMAE_EUNSUPPORTED;
    const DecBlockPtrs d = dec_from_abi(w);
    if (split) {
        CK(hs_dec_attn_fwd(x, x1, o, lse, nsamples, Ts, d, S(stream)));
        EncMlpPtrs m;
        m.n2w = d.n2w; m.n2b = d.n2b; m.w1b = d.w1b; m.w3b = d.w3b; m.w2b = d.w2b;
        m.w1 = d.w1; m.w3 = d.w3; m.w2 = d.w2; m.w2T = d.w2T; m.w13T = nullptr; m.h = d.h;
        return hs_enc_mlp_fwd(x1, nullptr, x2, nsamples * Ts, 64, m, S(stream));
    }
    return hs_dec_block_fwd(x, x1, x2, o, lse, nsamples, Ts, d, S(stream));
}
int64_t hsimae_dec_block_slab_floats(void) { return kDecSlabFloats; }
int64_t hsimae_wgrad_slab_bytes(const hsimae_config* cfg, int32_t side_stream) { return wgrad_slab_bytes(cfg, side_stream); }
int hsimae_dec_block_bwd(const hsimae_dec_block_weights* w, const hsimae_dec_block_grads* g, const float* x, const float* x1,
                         const float* dy, float* dx1_tmp, float* dx, const hs_bf16* o, const float* lse, int32_t nsamples,
                         int32_t Ts, float* slab, void* stream) {
    if (nsamples <= 0) return HSIMAE_OK;
    if (!w || !g || !x || !x1 || !dy || !dx1_tmp || !dx || !o || !lse) return HSIMAE_ENULL;
    if (!hs_dec_fused_supported(64, 8, w->hidden, Ts)) return HSIMAE_EUNSUPPORTED;
    DecBlockGrads dg;
    dg.n1w = g->n1w; dg.n1b = g->n1b; dg.qw = g->qw; dg.qb = g->qb; dg.kw = g->kw; dg.kb = g->kb; dg.vw = g->vw; dg.vb = g->vb;
    dg.pw = g->pw; dg.pb = g->pb; dg.n2w = g->n2w; dg.n2b = g->n2b; dg.w1w = g->w1w; dg.w1b = g->w1b; dg.w2w = g->w2w;
    dg.w2b = g->w2b; dg.w3w = g->w3w; dg.w3b = g->w3b; dg.det = HsDet{nullptr, nullptr};
    return hs_dec_block_bwd(x, x1, dy, dx1_tmp, dx, o, lse, nsamples, Ts, dec_from_abi(w), dg, S(stream), slab);
}
int hsimae_attn_fwd(const hsimae_attn_params* p, void* stream) { return p ? hs_attn_fwd(*p, S(stream)) : HSIMAE_ENULL; }
int hsimae_attn_bwd(const hsimae_attn_params* p, void* stream) { return p ? hs_attn_bwd(*p, S(stream)) : HSIMAE_ENULL; }
int hsimae_wgrad(const hsimae_wgrad_params* p, void* stream) { return p ? hs_wgrad(*p, S(stream)) : HSIMAE_ENULL; }
int32_t hsimae_wgrad_msplit(int32_t tiles, int64_t M) { return wgrad_msplit(tiles, M); }
int hsimae_agg_pool(const float* latent, float* pooled, int32_t N, int32_t T, int32_t L, int32_t D, void* stream) {
    if (N <= 0) return HSIMAE_OK;
    if (!latent || !pooled) return HSIMAE_ENULL;
    return hs_agg_pool(latent, pooled, N, T, L, D, S(stream));
}
int hsimae_head_bwd(const float* g, const float* pooled, const float* w, float* gw, float* gb, float* dlatent, int32_t N, int32_t C,
                    int32_t T, int32_t L, int32_t D, void* stream) {
    if (N <= 0) return HSIMAE_OK;
    if (!g || !pooled || !w || !gw || !gb || !dlatent) return HSIMAE_ENULL;
    return hs_head_bwd(g, pooled, w, gw, gb, dlatent, N, C, T, L, D, S(stream));
}
int hsimae_cube_gather(const hsimae_cube_params* p, void* stream) {
    if (!p) return HSIMAE_ENULL;
    if (p->N <= 0) return HSIMAE_OK;
    if (!p->scenes || !p->scene_off || !p->scene_w || !p->cut || !p->index || !p->out) return HSIMAE_ENULL;
    return hs_cube_gather(*p, S(stream));
}
int hsimae_ln_bwd(const hsimae_lnbwd_params* p, void* stream) { return p ? hs_ln_bwd(*p, S(stream)) : HSIMAE_ENULL; }
int hsimae_ln_fwd(const float* x, const float* gamma, const float* beta, float* out, int32_t M, int32_t d, void* stream) {
    return (x && gamma && beta && out) ? hs_ln_fwd(x, gamma, beta, out, M, d, S(stream)) : HSIMAE_ENULL;
}
int hsimae_assemble_fwd(const hsimae_assemble_params* p, void* stream) { return p ? hs_assemble_fwd(*p, S(stream)) : HSIMAE_ENULL; }
int hsimae_assemble_bwd(const hsimae_assemble_params* p, void* stream) { return p ? hs_assemble_bwd(*p, S(stream)) : HSIMAE_ENULL; }
int hsimae_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* group, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !group) return HSIMAE_ENULL;
    return hs_adamw(params, grads, exp_avg, exp_avg_sq, group, n, lr, beta1, beta2, eps, weight_decay, step, S(stream));
}
int hsimae_loss_partials(int32_t N, int32_t T) { return hs_loss_partials(N, T); }
int hsimae_loss(const hsimae_loss_params* p, void* stream) { return p ? hs_loss(*p, S(stream)) : HSIMAE_ENULL; }

}  // extern "C"
