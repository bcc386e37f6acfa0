// Host-side planning of the HSIMAE kernel library: model geometry, the flat parameter layout, the packed-weight layout and
// its descriptor table, the workspace carve and the launch-shape rule of the weight-gradient kernel.  Pure C++ (no HIP):
// api.hip includes it for the real library; csrc/plan_host.cpp builds the same code with g++ -fsanitize=address,undefined
// into libhsimae_plan_asan.so, which the CPU tests drive through the same C entry points (tests/test_plan_asan_cpu.py).
#pragma once
#include "../../include/hsimae_hip.h"
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace hsplan {

typedef hsimae_pack_desc PackDesc;
constexpr int LOSS_ROWS_PER_WG = 32;       // rows per workgroup of the row-form loss kernel (elem.hip)
inline int loss_partials(int N, int T) {
    const int64_t M = (int64_t)N * T * 9;
    return (int)std::max<int64_t>((M + LOSS_ROWS_PER_WG - 1) / LOSS_ROWS_PER_WG, N);     // either kernel form of hs_loss
}

inline int rup(int x, int m) { return (x + m - 1) / m * m; }

// ------------------------------------------------------------------ parameter layout (flat fp32, registration order)
struct BlkOff { int64_t n1w, n1b, qw, qb, kw, kb, vw, vb, pw, pb, n2w, n2b, w1w, w1b, w2w, w2b, w3w, w3b, end; };

struct Geo {
    int B, T, TL, D, H, hd, h, hp, Dd, Hd, hdd, hdec, hpd, depth, sdepth, nfus, ddepth, norm_pix, prec;
    // Storage widths: every activation row, packed image and GEMM K extent uses the width rounded up to 32 (one MFMA
    // k-step); the columns past the true width are exact zeros everywhere (zero weight rows / columns, LayerNorm and
    // attention confined to the true width, a zero-filled arena), so widths like 144 / 72 (Model_Finetuning.py:66-67) or
    // 64 / 48 (Model_Pretraining.py:57-58) run on the same kernels.  Dp == D for the multiples of 32.
    int Dp, Ddp;
    bool has_axis, has_fus;
};

inline int make_geo(const hsimae_config* c, Geo& g) {
    if (!c) return HSIMAE_ENULL;
    if (c->bands <= 0 || c->bands % 8) return HSIMAE_EDIMS;
    g.B = c->bands; g.T = c->bands / 8; g.TL = g.T * 9;
    g.D = c->embed_dim; g.H = c->num_heads; g.Dd = c->dec_dim; g.Hd = c->dec_heads;
    if (g.D <= 0 || g.Dd <= 0 || g.H <= 0 || g.Hd <= 0 || g.D % g.H || g.Dd % g.Hd) return HSIMAE_EDIMS;
    g.hd = g.D / g.H; g.hdd = g.Dd / g.Hd;
    g.h = c->hidden; g.hdec = c->dec_hidden; g.hp = rup(g.h, 32); g.hpd = rup(g.hdec, 32);
    g.depth = c->depth; g.sdepth = c->s_depth; g.ddepth = c->dec_depth;
    g.has_axis = g.sdepth > 0;
    g.has_fus = g.sdepth < 12;                     // Models.py:385 (hard-coded 12)
    g.nfus = g.has_fus ? std::max(0, g.depth - g.sdepth) : 0;
    g.norm_pix = c->norm_pix_loss;
    g.prec = c->precision;
    if (g.prec != HSIMAE_PREC_BF16 && g.prec != HSIMAE_PREC_FP8) return HSIMAE_EUNSUPPORTED;
    if (g.D % 8 || g.Dd % 8 || g.D > 512 || g.Dd > 512) return HSIMAE_EUNSUPPORTED;
    g.Dp = rup(g.D, 32); g.Ddp = rup(g.Dd, 32);
    if ((g.hd != 8 && g.hd != 16) || (g.hdd != 8 && g.hdd != 16)) return HSIMAE_EUNSUPPORTED;
    if (g.h <= 0 || g.hdec <= 0 || g.h % 4 || g.hdec % 4 || g.T > 64 || g.ddepth < 1) return HSIMAE_EUNSUPPORTED;
    return HSIMAE_OK;
}

struct PLayout {
    int64_t pos, mask_token, dpos, pew, peb, nw, nb, dew, deb, dnw, dnb, dpw, dpb, total;
    std::vector<BlkOff> b1, b2, bf, bd;
    std::vector<int64_t> offs, sizes;
};

inline void make_playout(const Geo& g, PLayout& L) {
    int64_t cur = 0;
    auto add = [&](int64_t n) { int64_t o = cur; L.offs.push_back(o); L.sizes.push_back(n); cur += n; return o; };
    auto blk = [&](int d, int h) {
        BlkOff b;
        b.n1w = add(d); b.n1b = add(d);
        b.qw = add((int64_t)d * d); b.qb = add(d);
        b.kw = add((int64_t)d * d); b.kb = add(d);
        b.vw = add((int64_t)d * d); b.vb = add(d);
        b.pw = add((int64_t)d * d); b.pb = add(d);
        b.n2w = add(d); b.n2b = add(d);
        b.w1w = add((int64_t)h * d); b.w1b = add(h);
        b.w2w = add((int64_t)d * h); b.w2b = add(d);
        b.w3w = add((int64_t)h * d); b.w3b = add(h);
        b.end = cur;
        return b;
    };
    L.pos = add((int64_t)g.TL * g.D);
    L.mask_token = add(g.Dd);
    L.dpos = add((int64_t)g.TL * g.Dd);
    L.pew = add((int64_t)g.D * 72);
    L.peb = add(g.D);
    if (g.has_axis) {
        for (int i = 0; i < g.sdepth; ++i) L.b1.push_back(blk(g.D, g.h));
        for (int i = 0; i < g.sdepth; ++i) L.b2.push_back(blk(g.D, g.h));
    }
    for (int i = 0; i < g.nfus; ++i) L.bf.push_back(blk(g.D, g.h));
    L.nw = add(g.D); L.nb = add(g.D);
    L.dew = add((int64_t)g.Dd * g.D); L.deb = add(g.Dd);
    for (int i = 0; i < g.ddepth; ++i) L.bd.push_back(blk(g.Dd, g.hdec));
    L.dnw = add(g.Dd); L.dnb = add(g.Dd);
    L.dpw = add((int64_t)72 * g.Dd); L.dpb = add(72);
    L.total = cur;
}

// ------------------------------------------------------------------ packed-weight layout (bf16 images + fp32 bias packs)
struct Img8 { int64_t w, s; };            // byte offsets of an e4m3 image and of its e8m0 scale image (fp8 region)
struct BlkW {
    int64_t qkv, p, w1, w3, w2, qkvT, pT, w13T, w2T; int64_t bqkv;   // element offsets (bf16) / float offsets
    Img8 qkv8, p8, w1_8, w3_8, w2_8, qkvT8, pT8, w13T8, w2T8;         // precision = FP8: encoder blocks only
};
struct WLayout {
    int64_t pe, de, deT, dp, dpT;
    std::vector<BlkW> b1, b2, bf, bd;
    int64_t bf16_elems;      // bf16 region size (elements), multiple of 8
    int64_t f32_elems;       // fp32 region (bias packs)
    int64_t fp8_bytes;       // e4m3 images + scale images of the encoder blocks (precision = FP8), after the fp32 region
    int64_t total_elems;     // in bf16 units
};

inline void make_wlayout(const Geo& g, WLayout& W) {
    int64_t cur = 0, fcur = 0, cur8 = 0;
    auto img = [&](int N, int K) { int64_t o = cur; cur += (int64_t)N * K; return o; };
    auto img8 = [&](int N, int K) {       // [ceil(N/16)][ceil(K/128)][64 lanes][32 B] + one scale dword per (n-tile, 512-chunk, lane)
        const int64_t nt = (N + 15) / 16, ks = (K + 127) / 128, kch = (ks + 3) / 4;
        Img8 o; o.w = cur8; cur8 += nt * ks * 64 * 32; o.s = cur8; cur8 += nt * kch * 64 * 4;
        return o;
    };
    auto blk = [&](int d, int hp, bool f8) {
        BlkW b; std::memset(&b, 0, sizeof(b));
        b.qkv = img(3 * d, d); b.p = img(d, d); b.w1 = img(hp, d); b.w3 = img(hp, d); b.w2 = img(d, hp);
        b.qkvT = img(d, 3 * d); b.pT = img(d, d); b.w13T = img(d, 2 * hp); b.w2T = img(hp, d);
        b.bqkv = fcur; fcur += 3 * d;
        if (f8) {
            b.qkv8 = img8(3 * d, d); b.p8 = img8(d, d); b.w1_8 = img8(hp, d); b.w3_8 = img8(hp, d); b.w2_8 = img8(d, hp);
            b.qkvT8 = img8(d, 3 * d); b.pT8 = img8(d, d); b.w13T8 = img8(d, 2 * hp); b.w2T8 = img8(hp, d);
        }
        return b;
    };
    const bool f8 = g.prec == HSIMAE_PREC_FP8;
    W.pe = img(g.Dp, 96);
    if (g.has_axis) {
        for (int i = 0; i < g.sdepth; ++i) W.b1.push_back(blk(g.Dp, g.hp, f8));
        for (int i = 0; i < g.sdepth; ++i) W.b2.push_back(blk(g.Dp, g.hp, f8));
    }
    for (int i = 0; i < g.nfus; ++i) W.bf.push_back(blk(g.Dp, g.hp, f8));
    W.de = img(g.Ddp, g.Dp); W.deT = img(g.Dp, g.Ddp);
    for (int i = 0; i < g.ddepth; ++i) W.bd.push_back(blk(g.Ddp, g.hpd, false));
    W.dp = img(80, g.Ddp); W.dpT = img(g.Ddp, 96);
    W.bf16_elems = (cur + 7) & ~7ll;
    W.f32_elems = (fcur + 3) & ~3ll;            // keeps the fp8 region 16-B aligned
    W.fp8_bytes = cur8;
    W.total_elems = W.bf16_elems + 2 * W.f32_elems + (cur8 + 1) / 2;
}

inline void pack_descs(const Geo& g, const PLayout& L, const WLayout& W, const float* P, hs_bf16* wpk,
                std::vector<PackDesc>& out) {
    float* fbase = reinterpret_cast<float*>(wpk + W.bf16_elems);
    unsigned char* base8 = reinterpret_cast<unsigned char*>(fbase + W.f32_elems);
    auto mat = [&](int64_t src, int rows, int cols, int tr, int n_off, int k_off, int K_img, int64_t dst) {
        PackDesc d; std::memset(&d, 0, sizeof(d));
        d.src = P + src; d.rows = rows; d.cols = cols; d.transpose = tr; d.n_off = n_off; d.k_off = k_off;
        d.KS = K_img / 32; d.dst = wpk + dst; out.push_back(d);
    };
    auto mat8 = [&](int64_t src, int rows, int cols, int tr, int n_off, int k_off, int K_img, const Img8& im) {
        PackDesc d; std::memset(&d, 0, sizeof(d));
        d.src = P + src; d.rows = rows; d.cols = cols; d.transpose = tr; d.n_off = n_off; d.k_off = k_off;
        d.KS = (K_img + 127) / 128; d.dst = reinterpret_cast<hs_bf16*>(base8 + im.w); d.fp8 = 1; d.scales = base8 + im.s;
        out.push_back(d);
    };
    auto fcopy = [&](int64_t src, int n, int64_t dst_f, int off) {
        PackDesc d; std::memset(&d, 0, sizeof(d));
        d.src = P + src; d.rows = 1; d.cols = n; d.transpose = 0; d.n_off = off; d.k_off = 0; d.KS = 0;
        d.dst = reinterpret_cast<hs_bf16*>(fbase + dst_f); out.push_back(d);
    };
    // d: true width of the block, dp: its storage width (K extents and the q | k | v placement use dp; the sources keep d)
    auto blk = [&](const BlkOff& b, const BlkW& w, int d, int dp, int h, int hp, bool f8) {
        mat(b.qw, d, d, 0, 0, 0, dp, w.qkv); mat(b.kw, d, d, 0, dp, 0, dp, w.qkv); mat(b.vw, d, d, 0, 2 * dp, 0, dp, w.qkv);
        mat(b.pw, d, d, 0, 0, 0, dp, w.p);
        mat(b.w1w, h, d, 0, 0, 0, dp, w.w1); mat(b.w3w, h, d, 0, 0, 0, dp, w.w3);
        mat(b.w2w, d, h, 0, 0, 0, hp, w.w2);
        // transposed images for the data gradients
        mat(b.qw, d, d, 1, 0, 0, 3 * dp, w.qkvT); mat(b.kw, d, d, 1, 0, dp, 3 * dp, w.qkvT); mat(b.vw, d, d, 1, 0, 2 * dp, 3 * dp, w.qkvT);
        mat(b.pw, d, d, 1, 0, 0, dp, w.pT);
        mat(b.w1w, h, d, 1, 0, 0, 2 * hp, w.w13T); mat(b.w3w, h, d, 1, 0, hp, 2 * hp, w.w13T);
        mat(b.w2w, d, h, 1, 0, 0, dp, w.w2T);
        fcopy(b.qb, d, w.bqkv, 0); fcopy(b.kb, d, w.bqkv, dp); fcopy(b.vb, d, w.bqkv, 2 * dp);
        if (f8) {                                     // the same images as MX e4m3 (encoder blocks, precision = FP8)
            mat8(b.qw, d, d, 0, 0, 0, dp, w.qkv8); mat8(b.kw, d, d, 0, dp, 0, dp, w.qkv8); mat8(b.vw, d, d, 0, 2 * dp, 0, dp, w.qkv8);
            mat8(b.pw, d, d, 0, 0, 0, dp, w.p8);
            mat8(b.w1w, h, d, 0, 0, 0, dp, w.w1_8); mat8(b.w3w, h, d, 0, 0, 0, dp, w.w3_8);
            mat8(b.w2w, d, h, 0, 0, 0, hp, w.w2_8);
            mat8(b.qw, d, d, 1, 0, 0, 3 * dp, w.qkvT8); mat8(b.kw, d, d, 1, 0, dp, 3 * dp, w.qkvT8); mat8(b.vw, d, d, 1, 0, 2 * dp, 3 * dp, w.qkvT8);
            mat8(b.pw, d, d, 1, 0, 0, dp, w.pT8);
            mat8(b.w1w, h, d, 1, 0, 0, 2 * hp, w.w13T8); mat8(b.w3w, h, d, 1, 0, hp, 2 * hp, w.w13T8);
            mat8(b.w2w, d, h, 1, 0, 0, dp, w.w2T8);
        }
    };
    mat(L.pew, g.D, 72, 0, 0, 0, 96, W.pe);
    const bool f8 = g.prec == HSIMAE_PREC_FP8;
    for (size_t i = 0; i < L.b1.size(); ++i) blk(L.b1[i], W.b1[i], g.D, g.Dp, g.h, g.hp, f8);
    for (size_t i = 0; i < L.b2.size(); ++i) blk(L.b2[i], W.b2[i], g.D, g.Dp, g.h, g.hp, f8);
    for (size_t i = 0; i < L.bf.size(); ++i) blk(L.bf[i], W.bf[i], g.D, g.Dp, g.h, g.hp, f8);
    mat(L.dew, g.Dd, g.D, 0, 0, 0, g.Dp, W.de); mat(L.dew, g.Dd, g.D, 1, 0, 0, g.Ddp, W.deT);
    for (size_t i = 0; i < L.bd.size(); ++i) blk(L.bd[i], W.bd[i], g.Dd, g.Ddp, g.hdec, g.hpd, false);
    mat(L.dpw, 72, g.Dd, 0, 0, 0, g.Ddp, W.dp); mat(L.dpw, 72, g.Dd, 1, 0, 0, 96, W.dpT);
}

// ------------------------------------------------------------------ workspace
// rows between the planes of a planar weight-gradient operand beyond M (round 5): with exactly M rows the planes of C2 start
// 27 x 2^19 bytes apart and the two streams of a dW tile's operand slice meet in the same HBM channels (wgrad_dma 118.8 -> 123.5 us)
#ifndef HS_PLANE_PAD_ROWS
#define HS_PLANE_PAD_ROWS 48
#endif
constexpr int kPlanePadRows = 64;          // what the arena reserves per plane (>= HS_PLANE_PAD_ROWS)
static_assert(HS_PLANE_PAD_ROWS <= kPlanePadRows && HS_PLANE_PAD_ROWS % 16 == 0,
              "a -DHS_PLANE_PAD_ROWS variant must stay inside the rows the arena reserves per plane (ADVICE r05)");
struct BlkBuf { hs_bf16* u; hs_bf16* qkv; float* lse; hs_bf16* o; float* x1; hs_bf16* u2; hs_bf16* h13; hs_bf16* g; float* x2; };

struct Scr { float* G1; float* du; hs_bf16* dh13; hs_bf16* dob; hs_bf16* dqkv; hs_bf16* g0b; hs_bf16* g1b; float* slab; };   // per-stream backward scratch (g0b/g1b: bf16 dY / dx1; slab: weight-gradient partials)

struct Ws {
    Scr sc, sc2;                      // sc2: encoder-sized second set for the side stream (spectral stack)
    hs_bf16* a_pe; float* x0;
    std::vector<BlkBuf> b1, b2, bf, bd;
    hs_bf16* lat; float* y; float* yfull; hs_bf16* zn; float* pred; hs_bf16* dpred; float* partial;
    float *G0, *G1, *G2, *du; hs_bf16 *dh13, *dob, *dqkv, *dyb;
    float* slab;                      // weight-gradient partials of the persistent kernels: [workgroup][slot][thread] (slab_bytes())
    int64_t bytes;
};
// Weight-gradient slabs, sized from the geometry (every arena used to carry 2 x 64 MiB whatever ran in it):
//   * the 256 x 256-tile weight-gradient launch (wgrad.hip, all matrices >= 256 wide: Large / Huge): 256 workgroups x 256 KB,
//     one slab per stream — the caller's and, for the forked spectral stack, the side stream's;
//   * the fused decoder's backward: HSIMAE_DEC_BLOCK_SLAB_FLOATS (checked against fused_dec.hip's own constants there), caller's stream.
//   The caller's stream also runs the DECODER's weight-gradient launches when the decoder goes layer at a time (block_bwd with
//   w.sc): a decoder of width >= 256 takes the 256 x 256-tile path whatever the encoder's width is (embed_dim 128 +
//   decoder_embed_dim 256 wrote 247 x 256 KB into a 56.7 MB slab), so the caller's slab is sized from BOTH widths; the side
//   stream only ever runs encoder blocks.
constexpr int64_t kSlabBytes = 256ll * 256 * 256 * 4;
inline int64_t slab_bytes(const Geo& g, bool side_stream) {
    if (side_stream) return (g.has_axis && g.Dp >= 256) ? kSlabBytes : 0;
    const int64_t wg = std::max(g.Dp, g.Ddp) >= 256 ? kSlabBytes : 0;
    return std::max<int64_t>(wg, HSIMAE_DEC_BLOCK_SLAB_FLOATS * 4);
}

inline void carve(const Geo& g, int N, int K, char* base, Ws& w) {
    int64_t cur = 0;
    auto take = [&](int64_t bytes) { char* p = base ? base + cur : nullptr; cur += (bytes + 255) & ~255ll; return p; };
    const int64_t Me = (int64_t)N * K, Md = (int64_t)N * g.TL;
    // (g and dh1|dh3 are sized for the planar operand layout of round 5 — 64-column planes of M + kPlanePadRows rows,
    //  hsimae_wgrad_task — whose padded width is hp rounded up to 64)
    auto blk = [&](int64_t M, int d, int heads, int hp) {
        BlkBuf b;
        b.u = (hs_bf16*)take(M * d * 2); b.qkv = (hs_bf16*)take(M * 3 * d * 2); b.lse = (float*)take(M * heads * 4);
        b.o = (hs_bf16*)take(M * d * 2); b.x1 = (float*)take(M * d * 4); b.u2 = (hs_bf16*)take(M * d * 2);
        b.h13 = (hs_bf16*)take(M * 2 * hp * 2); b.g = (hs_bf16*)take((M + kPlanePadRows) * rup(hp, 64) * 2); b.x2 = (float*)take(M * d * 4);
        return b;
    };
    w.a_pe = (hs_bf16*)take(Me * 96 * 2);
    w.x0 = (float*)take(Me * g.Dp * 4);
    w.b1.clear(); w.b2.clear(); w.bf.clear(); w.bd.clear();
    if (g.has_axis) {
        for (int i = 0; i < g.sdepth; ++i) w.b1.push_back(blk(Me, g.Dp, g.H, g.hp));
        for (int i = 0; i < g.sdepth; ++i) w.b2.push_back(blk(Me, g.Dp, g.H, g.hp));
    }
    for (int i = 0; i < g.nfus; ++i) w.bf.push_back(blk(Me, g.Dp, g.H, g.hp));
    w.lat = (hs_bf16*)take(Me * g.Dp * 2);
    w.y = (float*)take(Me * g.Ddp * 4);
    w.yfull = (float*)take(Md * g.Ddp * 4);
    for (int i = 0; i < g.ddepth; ++i) w.bd.push_back(blk(Md, g.Ddp, g.Hd, g.hpd));
    w.zn = (hs_bf16*)take(Md * g.Ddp * 2);
    w.pred = (float*)take(Md * 72 * 4);
    w.dpred = (hs_bf16*)take(Md * 96 * 2);
    w.partial = (float*)take((int64_t)loss_partials(N, g.T) * 4);
    const int64_t gmax = std::max(Me * g.Dp, Md * g.Ddp);
    w.G0 = (float*)take(gmax * 4); w.G1 = (float*)take(gmax * 4); w.G2 = (float*)take(gmax * 4); w.du = (float*)take(gmax * 4);
    w.dh13 = (hs_bf16*)take(std::max((Me + kPlanePadRows) * 2 * rup(g.hp, 64), (Md + kPlanePadRows) * 2 * rup(g.hpd, 64)) * 2);
    w.dob = (hs_bf16*)take(gmax * 2);
    w.dqkv = (hs_bf16*)take(gmax * 3 * 2);
    w.dyb = (hs_bf16*)take(Me * g.Ddp * 2);
    w.sc.G1 = w.G1; w.sc.du = w.du; w.sc.dh13 = w.dh13; w.sc.dob = w.dob; w.sc.dqkv = w.dqkv;
    w.sc2.G1 = (float*)take(Me * g.Dp * 4); w.sc2.du = (float*)take(Me * g.Dp * 4);
    w.sc2.dh13 = (hs_bf16*)take((Me + kPlanePadRows) * 2 * rup(g.hp, 64) * 2); w.sc2.dob = (hs_bf16*)take(Me * g.Dp * 2);
    w.sc2.dqkv = (hs_bf16*)take(Me * g.Dp * 3 * 2);
    w.sc.g0b = (hs_bf16*)take(gmax * 2); w.sc.g1b = (hs_bf16*)take(gmax * 2);     // (also decoder rows on the layer-at-a-time path)
    w.sc2.g0b = (hs_bf16*)take(Me * g.Dp * 2); w.sc2.g1b = (hs_bf16*)take(Me * g.Dp * 2);
    w.slab = (float*)take(slab_bytes(g, false));
    w.sc.slab = w.slab;
    w.sc2.slab = slab_bytes(g, true) ? (float*)take(slab_bytes(g, true)) : nullptr;      // (NULL: that launch commits with atomics)
    w.bytes = cur;
}

inline int wgrad_msplit(int tiles, int64_t M, int concurrent = 1) {
    const int chunks = (int)((M + 63) / 64);
    // the kernel holds 2 workgroups per CU (196 registers: 64 accumulators + the prefetched next chunk): keep the
    // launch to one resident wave of workgroups
    const int budget = 512;
    // `concurrent` launches resident at once (the forked axis stacks) share the budget as long as each still gets whole
    // groups of 8 row slices: slice ms runs on XCD ms % 8, so fewer than 8 slices leave XCDs idle (D = 256: 52 tiles,
    // sharing made the step 13 % slower; D = 128: 13 tiles, 2 % faster).
    int ms = std::max(1, budget / std::max(1, concurrent) / std::max(1, tiles));
    if (ms < 8) ms = std::max(1, budget / std::max(1, tiles));
    if (ms >= 8) ms &= ~7;              // whole XCD groups (wgrad.hip places row slice ms on XCD ms % 8)
    return std::max(1, std::min(ms, chunks));
}

// ------------------------------------------------------------------ bodies of the host-only C entry points
inline int param_layout(const hsimae_config* cfg, int64_t* offsets, int64_t* sizes, int max_entries) {
    Geo g; int e = make_geo(cfg, g); if (e) return e;
    PLayout L; make_playout(g, L);
    const int n = (int)L.offs.size();
    for (int i = 0; i < n && i < max_entries; ++i) {
        if (offsets) offsets[i] = L.offs[i];
        if (sizes) sizes[i] = L.sizes[i];
    }
    return n;
}
inline int64_t wpk_elems(const hsimae_config* cfg) {
    Geo g; if (make_geo(cfg, g)) return -1;
    WLayout W; make_wlayout(g, W);
    return W.total_elems;
}
inline int64_t pack_table_bytes(const hsimae_config* cfg) {
    Geo g; if (make_geo(cfg, g)) return -1;
    PLayout L; make_playout(g, L); WLayout W; make_wlayout(g, W);
    std::vector<PackDesc> d; pack_descs(g, L, W, nullptr, nullptr, d);
    return (int64_t)d.size() * sizeof(PackDesc);
}
inline int build_pack_table(const hsimae_config* cfg, const float* params_dev, hs_bf16* wpk_dev, void* table_host) {
    Geo g; int e = make_geo(cfg, g); if (e) return e;
    if (!params_dev || !wpk_dev || !table_host) return HSIMAE_ENULL;
    PLayout L; make_playout(g, L); WLayout W; make_wlayout(g, W);
    std::vector<PackDesc> d; pack_descs(g, L, W, params_dev, wpk_dev, d);
    std::memcpy(table_host, d.data(), d.size() * sizeof(PackDesc));
    return HSIMAE_OK;
}
inline int64_t wgrad_slab_bytes(const hsimae_config* cfg, int32_t side_stream) {
    Geo g; if (make_geo(cfg, g)) return -1;
    return slab_bytes(g, side_stream != 0);
}
inline int64_t workspace_bytes(const hsimae_config* cfg, int32_t N, int32_t len_t, int32_t len_l) {
    Geo g; if (make_geo(cfg, g)) return -1;
    if (N <= 0 || len_t < 1 || len_l < 1) return -1;
    Ws w; carve(g, N, len_t * len_l, nullptr, w);
    return w.bytes;
}

}  // namespace hsplan
