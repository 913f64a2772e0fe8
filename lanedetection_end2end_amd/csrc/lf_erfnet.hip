// ERFNet backbone engine: a host-side plan (layer table -> kernel launches + workspace layout)
// and the two C-ABI calls that run the whole forward / backward pass on one stream.
//
// Replaces ERFNet.Net.forward (BEV/Networks/ERFNet.py:151-157; Encoder :63-95, Decoder :109-142)
// and the autograd backward PyTorch derives for it.
//
// Data layout in HBM: activations NHWC fp32, every tensor a block's backward needs (t1..t4, out,
// pre-BN conv outputs) stays resident in one caller-provided workspace; gradients ping-pong
// through three buffers.  Weights are re-packed into the MFMA operand order once per forward
// (2 M floats).  Train-mode BatchNorm = per-wave partial sums written by the producing conv's
// epilogue + a tiny finalise kernel; the apply is fused into the next conv's operand load where
// the dataflow allows (conv3x1_2) and into the residual/ReLU pass otherwise.
#include <stdlib.h>

#include <vector>

#include "lf_conv.h"
#include "lf_eltwise.h"
#include "lf_plan.h"
#include "lf_types.h"

namespace {

constexpr float BN_EPS = 1e-3f, BN_MOM = 0.1f;
enum { K_DOWN = 0, K_NB = 1, K_UP = 2 };

struct GemmOp {
    LfTapGeom geom;
    int pack;   // index into plan->packs
};
struct BNRef {
    int p_g, p_b, idx, C;
    long sc, sh, asc, ash, c1, c2;   // float offsets into the workspace
};
struct ConvRef {
    int p_w, p_b;
    GemmOp fwd;                  // also the wgrad geometry
    GemmOp dg[4];                // data-gradient launches (1, or 4 phases)
    int ndg;
    GemmOp fph[4];               // forward phases for transposed convs (nfph > 0 replaces fwd)
    int nfph;
};
struct Layer {
    int kind, Cin, Cout, Hin, Win, Hout, Wout, dil, drop_idx;
    long reg_lo;                 // first float of this layer's own region of the workspace (its activations + BN vectors)
    long x;                      // input activation (float offset); -1 = the NCHW image
    long b[5];                   // down: cat,y | nb: t1,t2,t3,t4,out | up: c,y
    ConvRef cv[4];
    BNRef bn[2];
};

}  // namespace

struct lf_erfnet_plan {
    int N, H, W, Cin, Cout;
    std::vector<Layer> layers;
    std::vector<LfPackEntry> packs;
    int n_params, n_bn, n_drop;
    int p_head_w[2], p_head_b[2], n_heads;
    long off_entries, off_packed, packed_floats;
    long off_packed16, packed16_elems;          // bf16 operand copies of the packed weights (precision mode 2)
    long off_packed48;                          // 3-piece bf16 split of the packed weights (modes 3, 4): 3 * packed16_elems
    mutable int precision = 0;                  // lf_erfnet_set_precision
    long off_stat0, off_stat1, stat_floats;     // two scratch regions for BN partial rows
    long off_wpart, wpart_floats, off_bpart, bpart_floats;
    // one region per weight gradient (batched reduce): the LAST regions of the workspace, sized by storage type ([0] fp32
    // tensors, [1] bf16 tensors: the read-once weight gradient writes up to 512 rows) -- lf_erfnet_workspace_bytes follows the
    // precision mode, every other offset is independent of it
    long off_wpart_all, wpart_all_floats[2], bpart_all_floats[2];
    long off_globals;                           // first float behind the layers' regions
    long off_gA, off_gB, off_gC, gbuf_floats;
    long total_floats;
    long head_in;                               // activation feeding the head
    std::vector<long> drop_off;                 // per dropout block: float offset into the mask buffer
    long drop_floats;
    // optional per-kernel-family timing (bench.py roofline): HIP events around every MFMA launch
    mutable int prof_on = 0;
    struct ProfRec { hipEvent_t a, b; int family; double flops; int layer; int Cs, Cd, ntaps; long npix; int epi; };
    mutable int prof_layer = -1;
    mutable std::vector<ProfRec> prof;
};

namespace {

int add_pack(lf_erfnet_plan* P, int param, int Kc, int Nc, long sk, long sn, const LfTapGeom& g, const int* tapidx) {
    LfPackEntry e;
    memset(&e, 0, sizeof(e));
    e.param = param; e.Kc = Kc; e.Nc = Nc; e.ntaps = g.ntaps; e.sk = sk; e.sn = sn;
    for (int t = 0; t < g.ntaps; ++t) e.tapidx[t] = tapidx[t];
    e.dst_off = P->packed_floats;
    P->packed_floats += (long)g.ntaps * Kc * Nc;
    e.dst16_off = P->packed16_elems;
    P->packed16_elems += lf_pack_bf16_elems(Kc, Nc, g.ntaps);
    P->packs.push_back(e);
    return (int)P->packs.size() - 1;
}

// 1-D factorised conv (3 taps along H (axis 0) or W (axis 1), dilation d): forward + data gradient
void build_conv1d(lf_erfnet_plan* P, ConvRef& c, int N, int H, int W, int C, int axis, int d) {
    LfTapGeom g = lf_base_geom(N, H, W, H, W, C, H, W, C, C, C);
    g.ntaps = 3;
    int idx_f[3], idx_d[3];
    for (int t = 0; t < 3; ++t) {
        g.tdh[t] = axis == 0 ? (t - 1) * d : 0;
        g.tdw[t] = axis == 1 ? (t - 1) * d : 0;
        idx_f[t] = t;        // Conv2d weight (Co,Ci,3,1)/(Co,Ci,1,3): tap t = kernel element t
        idx_d[t] = 2 - t;    // data gradient = correlation with the flipped kernel
    }
    c.fwd.geom = g;
    c.fwd.pack = add_pack(P, c.p_w, C, C, /*sk (ci)*/ 3, /*sn (co)*/ (long)C * 3, g, idx_f);
    c.nfph = 0;
    c.ndg = 1;
    c.dg[0].geom = g;
    c.dg[0].pack = add_pack(P, c.p_w, C, C, /*sk (co)*/ (long)C * 3, /*sn (ci)*/ 3, g, idx_d);
}

// sub-pixel phase tap sets shared by "3x3 s2 conv data-gradient" and "3x3 s2 transposed conv forward"
int phase_taps(int a, int* k, int* off) {
    if (a == 0) { k[0] = 1; off[0] = 0; return 1; }
    k[0] = 0; off[0] = 1; k[1] = 2; off[1] = 0;
    return 2;
}

// DownsamplerBlock conv: Conv2d(Cin, Cc, 3, stride 2, pad 1) writing channels [0,Cc) of the concat buffer
void build_down_conv(lf_erfnet_plan* P, ConvRef& c, int N, int H, int W, int Cin, int Cc, int Ccat) {
    const int Ho = H / 2, Wo = W / 2;
    LfTapGeom g = lf_base_geom(N, Ho, Wo, H, W, Cin, Ho, Wo, Ccat, Cin, Cc);
    g.ssh = 2; g.ssw = 2; g.ntaps = 9;
    int idx[9];
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) { g.tdh[kh * 3 + kw] = kh - 1; g.tdw[kh * 3 + kw] = kw - 1; idx[kh * 3 + kw] = kh * 3 + kw; }
    c.fwd.geom = g;
    c.fwd.pack = add_pack(P, c.p_w, Cin, Cc, /*sk (ci)*/ 9, /*sn (co)*/ (long)Cin * 9, g, idx);
    c.nfph = 0;
    c.ndg = 4;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            LfTapGeom d = lf_base_geom(N, Ho, Wo, Ho, Wo, Ccat, H, W, Cin, Cc, Cin);
            d.dsh = 2; d.dsw = 2; d.dah = a; d.daw = b;
            int kh[2], oh[2], kw[2], ow[2], ti[4];
            const int na = phase_taps(a, kh, oh), nb = phase_taps(b, kw, ow);
            for (int i = 0; i < na; ++i)
                for (int j = 0; j < nb; ++j) { d.tdh[d.ntaps] = oh[i]; d.tdw[d.ntaps] = ow[j]; ti[d.ntaps] = kh[i] * 3 + kw[j]; ++d.ntaps; }
            c.dg[a * 2 + b].geom = d;
            c.dg[a * 2 + b].pack = add_pack(P, c.p_w, Cc, Cin, /*sk (co)*/ (long)Cin * 9, /*sn (ci)*/ 9, d, ti);
        }
}

// UpsamplerBlock conv: ConvTranspose2d(Cin, Co, 3, stride 2, pad 1, output_padding 1), weight (Cin,Co,3,3)
void build_up_conv(lf_erfnet_plan* P, ConvRef& c, int N, int Hi, int Wi, int Cin, int Co) {
    c.nfph = 4;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            LfTapGeom g = lf_base_geom(N, Hi, Wi, Hi, Wi, Cin, 2 * Hi, 2 * Wi, Co, Cin, Co);
            g.dsh = 2; g.dsw = 2; g.dah = a; g.daw = b;
            int kh[2], oh[2], kw[2], ow[2], ti[4];
            const int na = phase_taps(a, kh, oh), nb = phase_taps(b, kw, ow);
            for (int i = 0; i < na; ++i)
                for (int j = 0; j < nb; ++j) { g.tdh[g.ntaps] = oh[i]; g.tdw[g.ntaps] = ow[j]; ti[g.ntaps] = kh[i] * 3 + kw[j]; ++g.ntaps; }
            c.fph[a * 2 + b].geom = g;
            c.fph[a * 2 + b].pack = add_pack(P, c.p_w, Cin, Co, /*sk (ci)*/ (long)Co * 9, /*sn (co)*/ 9, g, ti);
        }
    LfTapGeom d = lf_base_geom(N, Hi, Wi, 2 * Hi, 2 * Wi, Co, Hi, Wi, Cin, Co, Cin);
    d.ssh = 2; d.ssw = 2; d.ntaps = 9;
    int idx[9];
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) { d.tdh[kh * 3 + kw] = kh - 1; d.tdw[kh * 3 + kw] = kw - 1; idx[kh * 3 + kw] = kh * 3 + kw; }
    c.ndg = 1;
    c.dg[0].geom = d;
    c.dg[0].pack = add_pack(P, c.p_w, Co, Cin, /*sk (co)*/ 9, /*sn (ci)*/ (long)Co * 9, d, idx);
    c.fwd = c.fph[0];
}

void bn_alloc(LfBump& ws, BNRef& b) {
    b.sc = ws.take(b.C); b.sh = ws.take(b.C); b.asc = ws.take(b.C); b.ash = ws.take(b.C);
    b.c1 = ws.take(b.C); b.c2 = ws.take(b.C);
}

void account_fwd_stats(lf_erfnet_plan* P, const LfTapGeom& g) {
    P->stat_floats = lf_maxl(P->stat_floats, (long)lf_tapgemm_stat_rows(g) * 2 * g.Cd);
}
long wneed_of(const LfTapGeom& g, int s16) { return (long)lf_tapwgrad_splits_bound(g, s16) * g.ntaps * g.Cs * g.Cd; }
long bneed_of(const LfTapGeom& g, int s16) { return (long)lf_tapwgrad_splits_bound(g, s16) * g.Cd; }
void account_wgrad(lf_erfnet_plan* P, const LfTapGeom& g) {
    for (int s16 = 0; s16 < 2; ++s16) {
        P->wpart_floats = lf_maxl(P->wpart_floats, wneed_of(g, s16));
        P->bpart_floats = lf_maxl(P->bpart_floats, bneed_of(g, s16));
        P->wpart_all_floats[s16] += (wneed_of(g, s16) + 63) / 64 * 64;
        P->bpart_all_floats[s16] += (bneed_of(g, s16) + 63) / 64 * 64;
    }
}
long wpart_all_end(const lf_erfnet_plan* P, int s16) { return P->off_wpart_all + P->wpart_all_floats[s16] + P->bpart_all_floats[s16]; }

}  // namespace

extern "C" {

// Build the plan for a fixed input shape.  out_channels = channels of the selected head
// (nclasses or nclasses+1); n_heads = 2 when the model was built with pretrained=True
// (Decoder.output_conv2, ERFNet.py:125-126).  Returns NULL on error.
lf_erfnet_plan* lf_erfnet_plan_create(int N, int H, int W, int in_channels, int out_channels, int n_heads) {
    if (N < 1 || H % 16 != 0 || W % 32 != 0 || in_channels < 1 || in_channels > 4 || out_channels < 1 || out_channels + n_heads - 1 > 5 ||
        n_heads < 1 || n_heads > 2) {
        lf_fail("lf_erfnet_plan_create: unsupported shape N=%d H=%d W=%d Cin=%d Cout=%d (need H%%16==0, W%%32==0)", N, H, W,
                in_channels, out_channels);
        return nullptr;
    }
    lf_erfnet_plan* P = new lf_erfnet_plan();
    P->N = N; P->H = H; P->W = W; P->Cin = in_channels; P->Cout = out_channels; P->n_heads = n_heads;
    P->packed_floats = 0; P->packed16_elems = 0; P->stat_floats = 0; P->wpart_floats = 0; P->bpart_floats = 0; P->drop_floats = 0;
    P->wpart_all_floats[0] = P->wpart_all_floats[1] = 0; P->bpart_all_floats[0] = P->bpart_all_floats[1] = 0;
    LfBump ws;
    int param = 0, bn = 0, drop = 0;
    long cur = -1;
    int h = H, w = W;
    auto add_down = [&](int cin, int cout) {
        Layer L;
        memset(&L, 0, sizeof(L));
        L.kind = K_DOWN; L.Cin = cin; L.Cout = cout; L.Hin = h; L.Win = w; L.Hout = h / 2; L.Wout = w / 2; L.drop_idx = -1;
        L.x = cur; L.reg_lo = ws.cur;
        L.cv[0].p_w = param++; L.cv[0].p_b = param++;
        L.bn[0].p_g = param++; L.bn[0].p_b = param++; L.bn[0].idx = bn++; L.bn[0].C = cout;
        const long sz = (long)N * (h / 2) * (w / 2) * cout;
        L.b[0] = ws.take(sz); L.b[1] = ws.take(sz);
        bn_alloc(ws, L.bn[0]);
        if (cur >= 0) {
            build_down_conv(P, L.cv[0], N, h, w, cin, cout - cin, cout);
            account_fwd_stats(P, L.cv[0].fwd.geom);
            account_wgrad(P, L.cv[0].fwd.geom);
        }
        P->layers.push_back(L);
        cur = L.b[1]; h /= 2; w /= 2;
    };
    auto add_nb = [&](int c, float pdrop, int d) {
        Layer L;
        memset(&L, 0, sizeof(L));
        L.kind = K_NB; L.Cin = c; L.Cout = c; L.Hin = L.Hout = h; L.Win = L.Wout = w; L.dil = d;
        L.drop_idx = pdrop > 0.f ? drop++ : -1;
        L.x = cur; L.reg_lo = ws.cur;
        const long sz = (long)N * h * w * c;
        for (int i = 0; i < 5; ++i) L.b[i] = ws.take(sz);
        const int axis[4] = {0, 1, 0, 1}, dil[4] = {1, 1, d, d};
        for (int i = 0; i < 4; ++i) {
            if (i == 2) { L.bn[0].p_g = param++; L.bn[0].p_b = param++; L.bn[0].idx = bn++; L.bn[0].C = c; }
            L.cv[i].p_w = param++; L.cv[i].p_b = param++;
            build_conv1d(P, L.cv[i], N, h, w, c, axis[i], dil[i]);
        }
        L.bn[1].p_g = param++; L.bn[1].p_b = param++; L.bn[1].idx = bn++; L.bn[1].C = c;
        bn_alloc(ws, L.bn[0]); bn_alloc(ws, L.bn[1]);
        account_fwd_stats(P, L.cv[0].fwd.geom);
        for (int i = 0; i < 4; ++i) account_wgrad(P, L.cv[i].fwd.geom);     // one partial region per weight gradient
        if (L.drop_idx >= 0) { P->drop_off.push_back(P->drop_floats); P->drop_floats += (long)N * c; }
        P->layers.push_back(L);
        cur = L.b[4];
    };
    auto add_up = [&](int cin, int cout) {
        Layer L;
        memset(&L, 0, sizeof(L));
        L.kind = K_UP; L.Cin = cin; L.Cout = cout; L.Hin = h; L.Win = w; L.Hout = 2 * h; L.Wout = 2 * w; L.drop_idx = -1;
        L.x = cur; L.reg_lo = ws.cur;
        L.cv[0].p_w = param++; L.cv[0].p_b = param++;
        L.bn[0].p_g = param++; L.bn[0].p_b = param++; L.bn[0].idx = bn++; L.bn[0].C = cout;
        const long sz = (long)N * 2 * h * 2 * w * cout;
        L.b[0] = ws.take(sz); L.b[1] = ws.take(sz);
        bn_alloc(ws, L.bn[0]);
        build_up_conv(P, L.cv[0], N, h, w, cin, cout);
        long rows4 = 0;
        for (int ph = 0; ph < 4; ++ph) { rows4 += lf_tapgemm_stat_rows(L.cv[0].fph[ph].geom); account_wgrad(P, L.cv[0].fph[ph].geom); }
        P->stat_floats = lf_maxl(P->stat_floats, rows4 * 2 * cout);   // the 4 phases' rows are laid end to end
        P->layers.push_back(L);
        cur = L.b[1]; h *= 2; w *= 2;
    };
    // Encoder (ERFNet.py:63-84)
    add_down(in_channels, 16);
    add_down(16, 64);
    for (int i = 0; i < 5; ++i) add_nb(64, 0.03f, 1);
    add_down(64, 128);
    const int dils[8] = {2, 4, 8, 16, 2, 4, 8, 16};
    for (int i = 0; i < 8; ++i) add_nb(128, 0.3f, dils[i]);
    param += 2;   // encoder.output_conv (1x1, only used by only_encode=True; never trained, ERFNet.py:84,92-93)
    // Decoder (ERFNet.py:109-126)
    add_up(128, 64);
    add_nb(64, 0.f, 1); add_nb(64, 0.f, 1);
    add_up(64, 16);
    add_nb(16, 0.f, 1); add_nb(16, 0.f, 1);
    for (int i = 0; i < n_heads; ++i) { P->p_head_w[i] = param++; P->p_head_b[i] = param++; }
    P->head_in = cur;
    P->n_params = param; P->n_bn = bn; P->n_drop = drop;

    // stem / pool / bn-backward partial rows also use the stat scratch regions
    const long npix1 = (long)N * (H / 2) * (W / 2);
    P->stat_floats = lf_maxl(P->stat_floats, (long)lf_stem_rows(N, H, W) * 2 * 16);
    P->stat_floats = lf_maxl(P->stat_floats, (long)lf_pool_rows(npix1) * 2 * 128);
    P->stat_floats = lf_maxl(P->stat_floats, (long)lf_bn_bwd_reduce_rows(npix1) * 2 * 128);
    P->wpart_floats = lf_maxl(P->wpart_floats, (long)lf_stem_wgrad_rows(N, H, W) * 16 * 36);
    P->wpart_floats = lf_maxl(P->wpart_floats, (long)lf_head_wgrad_rows(N, H / 2, W / 2) * 16 * 5 * 4);
    P->bpart_floats = lf_maxl(P->bpart_floats, (long)lf_stem_wgrad_rows(N, H, W) * 16);
    P->bpart_floats = lf_maxl(P->bpart_floats, (long)lf_head_wgrad_rows(N, H / 2, W / 2) * 8);
    for (int s16 = 0; s16 < 2; ++s16) {       // ... and their own regions beside the convolutions', so that their row sums join the batched reduction
        P->wpart_all_floats[s16] += ((long)lf_stem_wgrad_rows(N, H, W) * 16 * 36 + 63) / 64 * 64 + ((long)lf_head_wgrad_rows(N, H / 2, W / 2) * 16 * 5 * 4 + 63) / 64 * 64;
        P->bpart_all_floats[s16] += ((long)lf_stem_wgrad_rows(N, H, W) * 16 + 63) / 64 * 64 + ((long)lf_head_wgrad_rows(N, H / 2, W / 2) * 8 + 63) / 64 * 64;
    }

    P->off_globals = ws.cur;
    P->off_entries = ws.take((long)(P->packs.size() * sizeof(LfPackEntry) + 3) / 4);
    P->off_packed = ws.take(P->packed_floats);
    P->off_packed16 = ws.take((P->packed16_elems + 1) / 2);
    P->off_packed48 = ws.take((3 * P->packed16_elems + 1) / 2);
    P->stat_floats += 2 * 128 * 4;                              // (channel-major rows: a use's leading dimension is its row count rounded up to 4)
    P->off_stat0 = ws.take((P->stat_floats + 3) / 4 * 4);       // (16-byte aligned bases: the finalise kernels load four rows at once)
    P->off_stat1 = ws.take((P->stat_floats + 3) / 4 * 4);
    P->off_wpart = ws.take(P->wpart_floats);
    P->off_bpart = ws.take(P->bpart_floats);
    P->gbuf_floats = (long)N * (H / 2) * (W / 2) * 16;          // largest activation (= N*(H/4)*(W/4)*64)
    P->off_gA = ws.take(P->gbuf_floats);
    P->off_gB = ws.take(P->gbuf_floats);
    P->off_gC = ws.take(P->gbuf_floats);
    P->off_wpart_all = ws.cur;                                  // last: its size follows the precision mode
    P->total_floats = wpart_all_end(P, 1);                      // (the larger of the two)
    return P;
}

void lf_erfnet_plan_destroy(lf_erfnet_plan* P) {
    if (!P) return;
    delete P;
}
// (follows the precision mode: call it after lf_erfnet_set_precision)
size_t lf_erfnet_workspace_bytes(const lf_erfnet_plan* P) { return (size_t)wpart_all_end(P, P->precision == 2) * sizeof(float); }
// ... or for a NAMED precision mode, independent of the plan's current setting (a plan is shared between calls: sizing by hidden
// mutable state depends on call order -- ADVICE round 5); 0 for an unknown mode
size_t lf_erfnet_workspace_bytes_for(const lf_erfnet_plan* P, int mode) {
    if (!P || !(mode == 0 || mode == 2 || mode == 3)) return 0;
    return (size_t)wpart_all_end(P, mode == 2) * sizeof(float);
}
// Precision mode (include/lanefit.h): 0 = fp32 matrix cores, fp32 tensors (default, the parity path); 2 = bf16 tensors and bf16
// matrix cores (parameters, parameter gradients, BatchNorm statistics and the logits stay fp32); 3 = fp32 tensors, products from
// exact 9-term bf16 splits on the bf16 matrix cores.  Set before a forward; the matching backward must run with the same setting.
int lf_erfnet_set_precision(const lf_erfnet_plan* P, int mode) {
    LF_REQUIRE(P && (mode == 0 || mode == 2 || mode == 3), "lf_erfnet_set_precision: mode must be 0 (fp32 cores), 2 (bf16 tensors) or "
               "3 (fp32 from 9-term split operands on the bf16 cores); modes 1 and 4 were removed in round 6");
    P->precision = mode;
    return 0;
}
// floats of the layers' own regions: every activation tensor a backward needs (+ a few per-channel vectors) -- what bench.py's
// HBM roofline counts, independent of scratch regions (packed weights, statistics rows, partial rows, gradient ping-pong)
long lf_erfnet_activation_floats(const lf_erfnet_plan* P) { return P->off_globals; }
int lf_erfnet_num_params(const lf_erfnet_plan* P) { return P->n_params; }
int lf_erfnet_num_bn(const lf_erfnet_plan* P) { return P->n_bn; }
// Dropout2d keep-masks: one (N, C) fp32 block per non_bottleneck_1d with p > 0, in module order;
// values 0 or 1/(1-p).  Returns the total float count; offsets via lf_erfnet_dropmask_offset.
long lf_erfnet_dropmask_floats(const lf_erfnet_plan* P) { return P->drop_floats; }
int lf_erfnet_num_dropout(const lf_erfnet_plan* P) { return P->n_drop; }
long lf_erfnet_dropmask_offset(const lf_erfnet_plan* P, int i) { return P->drop_off[i]; }
int lf_erfnet_dropmask_channels(const lf_erfnet_plan* P, int i) {
    int k = 0;
    for (const Layer& L : P->layers)
        if (L.drop_idx >= 0 && k++ == i) return L.Cout;
    return -1;
}
// byte offset / float count of a named activation inside the workspace (parity tests): layer index in
// module order (0 = stem), slot as in Layer::b.  Returns -1 when out of range.
long lf_erfnet_encoder_offset(const lf_erfnet_plan* P) {
    for (const Layer& L : P->layers)
        if (L.kind == K_UP) return L.x;      // input of the first UpsamplerBlock = encoder output (N,H/8,W/8,128)
    return -1;
}
long lf_erfnet_activation_offset(const lf_erfnet_plan* P, int layer, int slot) {
    if (layer < 0 || layer >= (int)P->layers.size() || slot < 0 || slot > 4) return -1;
    return P->layers[layer].b[slot];
}
// float offset of a BatchNorm's folded per-channel vector in the workspace (parity tests: the fp32 values the ReLU masks of the
// forward pass were decided with).  bn = 0 / 1 within the layer (non_bottleneck_1d: bn1, bn2); which = 0 scale (gamma * rstd),
// 1 shift (beta - mean * scale), 2 rstd, 3 -mean * rstd.  Valid after lf_erfnet_forward.  Returns -1 when out of range.
long lf_erfnet_bn_vector_offset(const lf_erfnet_plan* P, int layer, int bn, int which) {
    if (layer < 0 || layer >= (int)P->layers.size() || bn < 0 || bn > 1 || which < 0 || which > 3) return -1;
    const Layer& L = P->layers[layer];
    if (bn == 1 && L.kind != K_NB) return -1;
    const BNRef& b = L.bn[bn];
    return which == 0 ? b.sc : which == 1 ? b.sh : which == 2 ? b.asc : b.ash;
}

}  // extern "C"

namespace {

struct Ctx {
    const lf_erfnet_plan* P;
    float* ws;
    const float* const* params;      // host array of device pointers
    float* const* grads;             // host array of device pointers (may hold nulls)
    float* const* running;           // host array: [2*i] running_mean, [2*i+1] running_var
    const float* dropmask;           // device buffer or null
    int training;
    hipStream_t st;
    const float* g_enc = nullptr;    // backward: extra gradient w.r.t. the encoder output (NHWC) or null
    int s16 = 0;                     // precision mode 2: activation / gradient tensors hold bf16 elements
    // weight-gradient reductions deferred to the end of the backward pass (one batched launch per LF_REDUCE_BATCH jobs)
    mutable std::vector<LfReduceJob> reduce_jobs;
    mutable long wpart_used = 0, bpart_used = 0;
    mutable int last_rows = 0;           // BatchNorm partial rows the last run_gemm wrote (depends on the kernel it selected)
    mutable bool wgrad_launched = false; // the last run_wgrad put a weight-gradient kernel on the stream (not skipped: frozen weight)
    // A RANGE of layers may run on a compact workspace (lf_erfnet_range_workspace_bytes): its layers' regions [lo, hi) followed by
    // the globals [cut, ...) -- the full workspace is the case lo = 0, hi = cut.
    long lo = 0, hi = 0, cut = 0;
    bool no_batch = false;               // compact workspace: no per-weight-gradient partial regions, every reduction immediately
    float* at(long off) const { return off < cut ? ws + (off - lo) : ws + (hi - lo) + (off - cut); }
    const float* packed(int pack) const { return at(P->off_packed + P->packs[pack].dst_off); }
};

double gemm_flops(const LfTapGeom& g) { return 2.0 * (double)g.N * g.Hl * g.Wl * g.Cs * g.Cd * g.ntaps; }

struct ProfScope {   // records a HIP event pair on the launch stream around one kernel when profiling is on
    const Ctx& c; int idx = -1;
    hipStream_t pst;
    ProfScope(const Ctx& ctx, int family, const LfTapGeom& g, int epi, hipStream_t stream) : c(ctx), pst(stream) {
        if (!c.P->prof_on) return;
        lf_erfnet_plan::ProfRec r;
        r.family = family; r.flops = gemm_flops(g);
        r.layer = c.P->prof_layer; r.Cs = g.Cs; r.Cd = g.Cd; r.ntaps = g.ntaps; r.npix = (long)g.N * g.Hl * g.Wl; r.epi = epi;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        (void)hipEventRecord(r.a, pst);
        c.P->prof.push_back(r);
        idx = (int)c.P->prof.size() - 1;
    }
    ~ProfScope() { if (idx >= 0) (void)hipEventRecord(c.P->prof[idx].b, pst); }
};

// beside_wgrad: this data gradient directly follows the weight gradient of the same convolution.  The two read the same gradient
// tensor and write disjoint buffers, so the data gradient is launched without the in-order barrier: its workgroups start in the
// slots the weight gradient's last workgroups leave instead of after its tail (the NEXT launch is in order again and waits for both).
int run_gemm(const Ctx& c, const GemmOp& op, const float* src, float* dst, const float* bias, int pro, int epi,
             LfTapArgs extra, bool beside_wgrad = false) {
    extra.src = src; extra.dst = dst; extra.bias = bias; extra.wp = c.packed(op.pack);
    extra.s16 = c.s16;
    if (c.P->precision == 2)
        extra.wp16 = reinterpret_cast<const unsigned short*>(c.at(c.P->off_packed16)) + c.P->packs[op.pack].dst16_off;
    if (c.P->precision == 3) {      // fp32 from split operands on the bf16 matrix cores (all 9 partial products)
        extra.split = 9;
        extra.wp48 = reinterpret_cast<const unsigned short*>(c.at(c.P->off_packed48)) + 3 * c.P->packs[op.pack].dst16_off;
    }
    c.last_rows = lf_tapgemm_stat_rows_for(op.geom, extra);
    ProfScope ps(c, 0, op.geom, epi | (pro << 8), c.st);
    if (beside_wgrad && c.wgrad_launched && !c.P->prof_on) return lf_tapgemm_launch_unordered(op.geom, extra, pro, epi, c.st);
    return lf_tapgemm_launch(op.geom, extra, pro, epi, c.st);
}

int bn_finalize(const Ctx& c, const BNRef& b, const LfStatPart* parts, int nparts, double count) {
    return lf_bn_finalize_fwd(parts, nparts, b.C, count, c.params[b.p_g], c.params[b.p_b], c.running[2 * b.idx],
                              c.running[2 * b.idx + 1], BN_MOM, BN_EPS, c.training, c.at(b.sc), c.at(b.sh), c.at(b.asc),
                              c.at(b.ash), c.st);
}

int encoder_layers(const lf_erfnet_plan* P) {     // layers [0, n) = the encoder (everything before the first UpsamplerBlock)
    int n = 0;
    for (const Layer& L : P->layers) { if (L.kind == K_UP) break; ++n; }
    return n;
}

// layers [first, last) of the plan; with first > 0 the caller has put the range's input at P->layers[first].x
int forward_layers(const Ctx& c, const float* img, int nlayers, int first = 0) {
    const lf_erfnet_plan* P = c.P;
    const int N = P->N;
    float* stat0 = c.at(P->off_stat0);
    float* stat1 = c.at(P->off_stat1);
    int layer_idx = 0;
    for (const Layer& L : P->layers) {
        if (layer_idx >= nlayers) break;
        if (layer_idx < first) { ++layer_idx; continue; }
        P->prof_layer = layer_idx++;
        const long npo = (long)N * L.Hout * L.Wout;
        if (L.kind == K_DOWN) {
            LfStatPart parts[2];
            int np = 0;
            if (L.x < 0) {
                const int srows = lf_stem_rows(N, L.Hin, L.Win);
                LF_TRY(lf_stem_fwd(img, N, L.Cin, L.Hin, L.Win, c.params[L.cv[0].p_w], c.params[L.cv[0].p_b], c.at(L.b[0]),
                                   c.training ? stat0 : nullptr, lf_stat_ld(srows), c.s16, c.st));
                parts[np++] = {stat0, srows, 16, 0, lf_stat_ld(srows)};
            } else {
                LfTapArgs a = lf_no_args();
                a.stats = stat0; a.stats_ld = lf_stat_ld(lf_tapgemm_stat_rows(L.cv[0].fwd.geom));
                LF_TRY(run_gemm(c, L.cv[0].fwd, c.at(L.x), c.at(L.b[0]), c.params[L.cv[0].p_b], LF_PRO_NONE,
                                c.training ? LF_EPI_STATS_SQ : 0, a));
                parts[np++] = lf_stat_part_tiles(stat0, c.last_rows, a.stats_ld, L.Cout - L.Cin, 0, c.last_rows, npo);     // centred rows (LfStatPart)
                const int prows = lf_pool_rows(npo);
                LF_TRY(lf_pool_concat_fwd(c.at(L.x), N, L.Hin, L.Win, L.Cin, c.at(L.b[0]), L.Cout, L.Cout - L.Cin,
                                          c.training ? stat1 : nullptr, lf_stat_ld(prows), c.s16, c.st));
                parts[np++] = {stat1, prows, L.Cin, L.Cout - L.Cin, lf_stat_ld(prows)};
            }
            LF_TRY(bn_finalize(c, L.bn[0], parts, np, (double)npo));
            LF_TRY(lf_bn_act(c.at(L.b[0]), c.at(L.bn[0].sc), c.at(L.bn[0].sh), nullptr, nullptr, c.at(L.b[1]), npo, L.Cout,
                             (long)L.Hout * L.Wout, c.s16, c.st));
        } else if (L.kind == K_NB) {
            LfTapArgs a = lf_no_args();
            LF_TRY(run_gemm(c, L.cv[0].fwd, c.at(L.x), c.at(L.b[0]), c.params[L.cv[0].p_b], LF_PRO_NONE, LF_EPI_RELU, a));
            a.stats = stat0; a.stats_ld = lf_stat_ld(lf_tapgemm_stat_rows(L.cv[1].fwd.geom));
            LF_TRY(run_gemm(c, L.cv[1].fwd, c.at(L.b[0]), c.at(L.b[1]), c.params[L.cv[1].p_b], LF_PRO_NONE,
                            c.training ? LF_EPI_STATS_SQ : 0, a));
            LfStatPart p0 = lf_stat_part_tiles(stat0, c.last_rows, a.stats_ld, L.Cout, 0, c.last_rows, npo);
            LF_TRY(bn_finalize(c, L.bn[0], &p0, 1, (double)npo));
            a = lf_no_args();
            a.pro_sc = c.at(L.bn[0].sc); a.pro_sh = c.at(L.bn[0].sh);
            LF_TRY(run_gemm(c, L.cv[2].fwd, c.at(L.b[1]), c.at(L.b[2]), c.params[L.cv[2].p_b], LF_PRO_BNRELU, LF_EPI_RELU, a));
            a = lf_no_args();
            a.stats = stat0; a.stats_ld = lf_stat_ld(lf_tapgemm_stat_rows(L.cv[3].fwd.geom));
            LF_TRY(run_gemm(c, L.cv[3].fwd, c.at(L.b[2]), c.at(L.b[3]), c.params[L.cv[3].p_b], LF_PRO_NONE,
                            c.training ? LF_EPI_STATS_SQ : 0, a));
            p0.nrows = p0.seg_rows = c.last_rows; p0.ld = a.stats_ld;
            LF_TRY(bn_finalize(c, L.bn[1], &p0, 1, (double)npo));
            const float* dm = (c.training && L.drop_idx >= 0 && c.dropmask) ? c.dropmask + P->drop_off[L.drop_idx] : nullptr;
            LF_TRY(lf_bn_act(c.at(L.b[3]), c.at(L.bn[1].sc), c.at(L.bn[1].sh), dm, c.at(L.x), c.at(L.b[4]), npo, L.Cout,
                             (long)L.Hout * L.Wout, c.s16, c.st));
        } else {
            LfStatPart parts[1];
            // the 4 sub-pixel phases write disjoint pixels of c; their stat rows are laid end to end
            int rows = 0, rows4 = 0;
            for (int ph = 0; ph < 4; ++ph) rows4 += lf_tapgemm_stat_rows(L.cv[0].fph[ph].geom);
            for (int ph = 0; ph < 4; ++ph) {
                LfTapArgs a = lf_no_args();
                a.stats = stat0 + rows; a.stats_ld = lf_stat_ld(rows4);      // (the phases' rows side by side in every channel's run)
                LF_TRY(run_gemm(c, L.cv[0].fph[ph], c.at(L.x), c.at(L.b[0]), c.params[L.cv[0].p_b], LF_PRO_NONE,
                                c.training ? LF_EPI_STATS_SQ : 0, a));
                rows += c.last_rows;
            }
            parts[0] = lf_stat_part_tiles(stat0, rows, lf_stat_ld(rows4), L.Cout, 0, rows / 4, npo / 4);      // every phase: N * Hin * Win pixels, rows / 4 rows
            LF_TRY(bn_finalize(c, L.bn[0], parts, 1, (double)npo));
            LF_TRY(lf_bn_act(c.at(L.b[0]), c.at(L.bn[0].sc), c.at(L.bn[0].sh), nullptr, nullptr, c.at(L.b[1]), npo, L.Cout,
                             (long)L.Hout * L.Wout, c.s16, c.st));
        }
    }
    return 0;
}

// The four sub-pixel phases of a transposed convolution have ONE bias gradient: in batched mode their bias partial rows are laid
// end to end (the phases' own bias regions are adjacent) and summed by the last phase's reduction job
struct BiasChain { float* base = nullptr; int rows = 0; int phase = 0; };

// weight + bias gradient of one forward-geometry GEMM
int run_wgrad(const Ctx& c, const GemmOp& op, const ConvRef& cv, const float* x, const float* g, const float* pro_sc,
              const float* pro_sh, int bias_accumulate, bool batch_off = false, BiasChain* chain = nullptr) {
    const lf_erfnet_plan* P = c.P;
    c.wgrad_launched = false;
    if (!c.grads[cv.p_w]) return 0;
    hipStream_t ws = c.st;
    LfWgradArgs a;
    a.x = x; a.g = g; a.pro_sc = pro_sc; a.pro_sh = pro_sh; a.s16 = c.s16;
    a.split = 0;        // (the weight gradient of mode 3 runs on the fp32 matrix cores)
    // Batched mode (default): this weight gradient keeps its partial rows in its own region and its reduction joins the
    // one launch at the end of the pass.  Immediate mode (compact workspaces): summed on the spot from the shared region; the
    // transposed-conv phases' bias rows then accumulate in order.
    const long wneed = (wneed_of(op.geom, c.s16) + 63) / 64 * 64, bneed = (bneed_of(op.geom, c.s16) + 63) / 64 * 64;
    const bool batched = !bias_accumulate && !batch_off && !c.no_batch &&
                         c.wpart_used + wneed <= P->wpart_all_floats[c.s16] && c.bpart_used + bneed <= P->bpart_all_floats[c.s16];
    const long off_bpart_all = P->off_wpart_all + P->wpart_all_floats[c.s16];
    a.partial = batched ? c.at(P->off_wpart_all + c.wpart_used) : c.at(P->off_wpart);
    a.bias_partial = c.grads[cv.p_b] ? (batched ? c.at(off_bpart_all + c.bpart_used) : c.at(P->off_bpart)) : nullptr;
    if (chain) {
        if (!batched) return lf_fail("erfnet backward: a chained weight gradient does not fit its partial-row regions");
        if (chain->phase == 0) chain->base = c.at(off_bpart_all + c.bpart_used);
        if (a.bias_partial) a.bias_partial = chain->base + (long)chain->rows * op.geom.Cd;
    }
    {
        ProfScope ps(c, 1, op.geom, 0, ws);
        LF_TRY(lf_tapwgrad_launch(op.geom, a, pro_sc ? LF_PRO_BNRELU : LF_PRO_NONE, ws));
    }
    c.wgrad_launched = true;
    const LfPackEntry& e = P->packs[op.pack];
    const int nsplit = lf_tapwgrad_splits_for(op.geom, a, pro_sc ? LF_PRO_BNRELU : LF_PRO_NONE);   // rows this launch wrote
    if (batched) {
        c.wpart_used += wneed; c.bpart_used += bneed;
        LfReduceJob j;
        memset(&j, 0, sizeof(j));
        j.partial = a.partial; j.grad = c.grads[cv.p_w]; j.bias_rows = a.bias_partial; j.bias_grad = c.grads[cv.p_b];
        j.sk = e.sk; j.sn = e.sn; j.splits = nsplit; j.ntaps = op.geom.ntaps; j.Cs = op.geom.Cs; j.Cd = op.geom.Cd;
        j.n_bias_rows = nsplit;
        if (chain) {
            chain->rows += nsplit;
            const bool last = ++chain->phase == 4;
            j.bias_rows = (last && a.bias_partial) ? chain->base : nullptr;
            j.n_bias_rows = chain->rows;
            if (!last) j.bias_grad = nullptr;
        }
        for (int t = 0; t < op.geom.ntaps; ++t) j.tapidx[t] = e.tapidx[t];
        c.reduce_jobs.push_back(j);
        // (launching the reductions gathered so far early and WITHOUT a barrier bit, beside a later weight / data gradient, was
        // measured: same bits, no gain -- DESIGN.md section 9)
        return 0;
    }
    LF_TRY(lf_wgrad_reduce_launch(a.partial, nsplit, op.geom.ntaps, op.geom.Cs, op.geom.Cd,
                                  c.grads[cv.p_w], e.sk, e.sn, e.tapidx, a.bias_partial, nsplit,
                                  c.grads[cv.p_b], bias_accumulate, ws));
    return 0;
}

// Partial rows of the stem / head weight gradients: rows_w [rows][nw] -> gw, rows_b [rows][nb] -> gb (plain column sums).  Batched
// mode: the rows live in their own regions and the sums join the reduction launch at the end of the pass (they were four 12 us
// launches of a handful of workgroups each); compact / exhausted workspace: summed on the spot from the shared region.
struct RowSums { float* wrows; float* brows; bool batched; };
RowSums row_sum_regions(const Ctx& c, long wneed, long bneed) {
    const lf_erfnet_plan* P = c.P;
    wneed = (wneed + 63) / 64 * 64; bneed = (bneed + 63) / 64 * 64;
    const bool batched = !c.no_batch && c.wpart_used + wneed <= P->wpart_all_floats[c.s16] && c.bpart_used + bneed <= P->bpart_all_floats[c.s16];
    RowSums r;
    r.batched = batched;
    r.wrows = batched ? c.at(P->off_wpart_all + c.wpart_used) : c.at(P->off_wpart);
    r.brows = batched ? c.at(P->off_wpart_all + P->wpart_all_floats[c.s16] + c.bpart_used) : c.at(P->off_bpart);
    if (batched) { c.wpart_used += wneed; c.bpart_used += bneed; }
    return r;
}
int row_sums_finish(const Ctx& c, const RowSums& r, int rows, int nw, float* gw, int nb, float* gb) {
    if (!r.batched) {
        LF_TRY(lf_rows_reduce_launch(r.wrows, rows, nw, gw, 0, c.st));
        if (gb) LF_TRY(lf_rows_reduce_launch(r.brows, rows, nb, gb, 0, c.st));
        return 0;
    }
    LfReduceJob j;
    memset(&j, 0, sizeof(j));
    j.partial = r.wrows; j.grad = gw; j.sk = 0; j.sn = 1; j.splits = rows; j.ntaps = 1; j.Cs = 1; j.Cd = nw;      // grad[n] = sum_r rows[r][n]
    c.reduce_jobs.push_back(j);
    if (gb) { j.partial = r.brows; j.grad = gb; j.Cd = nb; c.reduce_jobs.push_back(j); }
    return 0;
}

int bn_bwd_finalize(const Ctx& c, const BNRef& b, const LfStatPart* parts, int nparts, double count) {
    // parameter gradients go straight to bn.weight.grad / bn.bias.grad; c1/c2 stay in the workspace
    if (!c.grads[b.p_g] || !c.grads[b.p_b]) return lf_fail("erfnet backward: BatchNorm weight/bias must both require grad");
    return lf_bn_bwd_finalize(parts, nparts, b.C, count, c.at(b.asc), c.at(b.ash), c.at(b.c1), c.at(b.c2), c.grads[b.p_g],
                              c.grads[b.p_b], c.training, c.st);
}

// What the LAST data-gradient launch of layer L can do on behalf of layer L-1 (whose output it differentiates):
// apply L-1's final ReLU mask, and accumulate the sums of L-1's last BatchNorm backward -- so L-1 starts with
// g_z = g * [y > 0] already in memory and needs no separate reduction pass.
struct Prep {
    const float* y = nullptr;      // L-1 output (post ReLU)
    const float* pre = nullptr;    // L-1 pre-BN tensor
    const BNRef* bn = nullptr;
    const float* dm = nullptr;
};

Prep prep_for(const Ctx& c, const Layer& Lp) {
    Prep p;
    if (Lp.kind == K_NB) {
        p.y = c.at(Lp.b[4]); p.pre = c.at(Lp.b[3]); p.bn = &Lp.bn[1];
        p.dm = (Lp.drop_idx >= 0 && c.dropmask) ? c.dropmask + c.P->drop_off[Lp.drop_idx] : nullptr;
    } else {
        p.y = c.at(Lp.b[1]); p.pre = c.at(Lp.b[0]); p.bn = &Lp.bn[0];
    }
    return p;
}

// layers [first, nlayers) in reverse; *gin (optional) receives the buffer holding d loss / d (input of layer `first`), NHWC
int backward_layers(const Ctx& c, const float* img, float* g0, float* g1, float* g2, int nlayers, int first = 0,
                    float** gin = nullptr) {
    // on entry g0 holds d loss / d (output of the last block), NHWC.  The three buffers rotate roles:
    // `in` = incoming gradient, X / Y = scratch; every layer leaves its result in one of them.
    const lf_erfnet_plan* P = c.P;
    const int N = P->N;
    float* stat0 = c.at(P->off_stat0);
    float* bufs[3] = {g0, g1, g2};
    float* in = g0;
    bool prepped = false;        // `in` already masked by the layer's output ReLU, BN sums in stat0
    int prep_rows = 0, prep_ld = 0;
    for (int li = nlayers - 1; li >= first; --li) {
        const Layer& L = P->layers[li];
        P->prof_layer = 100 + li;
        const long npo = (long)N * L.Hout * L.Wout;
        const long ppi = (long)L.Hout * L.Wout;
        const int iin = in == bufs[0] ? 0 : (in == bufs[1] ? 1 : 2);
        float* X = bufs[(iin + 1) % 3];
        float* Y = bufs[(iin + 2) % 3];
        // gradient preparation the final dgrad of THIS layer performs for the previous one
        Prep nx;
        const bool can_prep = li > first && (L.kind == K_NB || L.kind == K_UP);
        if (can_prep) nx = prep_for(c, P->layers[li - 1]);
        auto add_prep = [&](LfTapArgs& a, int& epi, const LfTapGeom& dg) {
            if (!can_prep) return;
            a.mask_src = nx.y; a.aux = nx.pre; a.asc = c.at(nx.bn->asc); a.ash = c.at(nx.bn->ash); a.dm = nx.dm;
            a.stats = stat0; a.stats_ld = lf_stat_ld(lf_tapgemm_stat_rows(dg));      // (the data gradient's own rows: one per 256 pixels of the layer's input)
            epi |= LF_EPI_MASK | LF_EPI_STATS_XHAT;
        };
        const BNRef& blast = L.kind == K_NB ? L.bn[1] : L.bn[0];
        const float* ylast = c.at(L.kind == K_NB ? L.b[4] : L.b[1]);
        const float* prelast = c.at(L.kind == K_NB ? L.b[3] : L.b[0]);
        const float* dm = (L.kind == K_NB && L.drop_idx >= 0 && c.dropmask) ? c.dropmask + P->drop_off[L.drop_idx] : nullptr;
        // ---- last BatchNorm (+dropout +residual) + ReLU backward: g_pre -> X ; g_z (masked incoming gradient)
        const float* gz;
        if (!prepped) {
            const int rrows = lf_bn_bwd_reduce_rows(npo);
            LF_TRY(lf_bn_bwd_reduce(in, ylast, prelast, c.at(blast.asc), c.at(blast.ash), dm, stat0, lf_stat_ld(rrows), npo, L.Cout, ppi, c.s16, c.st));
            LfStatPart rp = {stat0, rrows, L.Cout, 0, lf_stat_ld(rrows)};
            LF_TRY(bn_bwd_finalize(c, blast, &rp, 1, (double)npo));
            LF_TRY(lf_bn_bwd_apply(in, ylast, prelast, c.at(blast.asc), c.at(blast.ash), c.params[blast.p_g], c.at(blast.c1),
                                   c.at(blast.c2), dm, X, L.kind == K_NB ? Y : nullptr, npo, L.Cout, ppi, c.s16, c.st));
            gz = Y;
            // `in` is free from here on
        } else {
            LfStatPart rp = {stat0, prep_rows, L.Cout, 0, prep_ld};
            LF_TRY(bn_bwd_finalize(c, blast, &rp, 1, (double)npo));
            LF_TRY(lf_bn_bwd_apply(in, nullptr, prelast, c.at(blast.asc), c.at(blast.ash), c.params[blast.p_g], c.at(blast.c1),
                                   c.at(blast.c2), dm, X, nullptr, npo, L.Cout, ppi, c.s16, c.st));
            gz = in;
            // Y is free; `in` must survive until the residual add
        }
        float* F = prepped ? Y : in;       // the free scratch buffer (the other of {in, Y} holds g_z)
        float* out = nullptr;              // where this layer leaves d loss / d (its input)
        prepped = false;
        if (L.kind == K_NB) {
            const float* x = c.at(L.x);
            const float *t1 = c.at(L.b[0]), *t2 = c.at(L.b[1]), *t3 = c.at(L.b[2]);
            const BNRef& b1 = L.bn[0];
            // conv1x3_2: wgrad(t3, g_t4 = X); dgrad -> g_t3 = (.) * [t3 > 0] -> F
            LF_TRY(run_wgrad(c, L.cv[3].fwd, L.cv[3], t3, X, nullptr, nullptr, 0));
            LfTapArgs a = lf_no_args();
            a.mask_src = t3;
            LF_TRY(run_gemm(c, L.cv[3].dg[0], X, F, nullptr, LF_PRO_NONE, LF_EPI_MASK, a, true));
            // conv3x1_2: input relu(bn1(t2)) recomputed on the fly; g_y1 -> X with the bn1-backward sums
            LF_TRY(run_wgrad(c, L.cv[2].fwd, L.cv[2], t2, F, c.at(b1.sc), c.at(b1.sh), 0));
            a = lf_no_args();
            a.aux = t2; a.msc = c.at(b1.sc); a.msh = c.at(b1.sh); a.asc = c.at(b1.asc); a.ash = c.at(b1.ash);
            a.stats = stat0; a.stats_ld = lf_stat_ld(lf_tapgemm_stat_rows(L.cv[2].dg[0].geom));
            LF_TRY(run_gemm(c, L.cv[2].dg[0], F, X, nullptr, LF_PRO_NONE, LF_EPI_MASKBN | LF_EPI_STATS_XHAT, a, true));
            LfStatPart sp = {stat0, c.last_rows, L.Cout, 0, a.stats_ld};
            LF_TRY(bn_bwd_finalize(c, b1, &sp, 1, (double)npo));
            LF_TRY(lf_bn_bwd_apply(X, nullptr, t2, c.at(b1.asc), c.at(b1.ash), c.params[b1.p_g], c.at(b1.c1), c.at(b1.c2),
                                   nullptr, F /*g_t2*/, nullptr, npo, L.Cout, ppi, c.s16, c.st));
            // conv1x3_1: g_t1 -> X
            LF_TRY(run_wgrad(c, L.cv[1].fwd, L.cv[1], t1, F, nullptr, nullptr, 0));
            a = lf_no_args();
            a.mask_src = t1;
            LF_TRY(run_gemm(c, L.cv[1].dg[0], F, X, nullptr, LF_PRO_NONE, LF_EPI_MASK, a, true));
            // conv3x1_1 (+ residual branch gradient g_z) -> F, optionally prepared for the previous layer
            LF_TRY(run_wgrad(c, L.cv[0].fwd, L.cv[0], x, X, nullptr, nullptr, 0));
            a = lf_no_args();
            a.add_src = gz;
            int epi = LF_EPI_ADD;
            add_prep(a, epi, L.cv[0].dg[0].geom);
            LF_TRY(run_gemm(c, L.cv[0].dg[0], X, F, nullptr, LF_PRO_NONE, epi, a, true));
            out = F;
            if (can_prep) { prepped = true; prep_rows = c.last_rows; prep_ld = a.stats_ld; }
        } else if (L.kind == K_UP) {
            // (the four phases' reductions join the batched launch when their regions fit: eight 6 us launches per step, each
            // between two dependent weight-gradient launches, otherwise)
            long wsum = 0, bsum = 0;
            for (int ph = 0; ph < 4; ++ph) {
                wsum += (wneed_of(L.cv[0].fph[ph].geom, c.s16) + 63) / 64 * 64;
                bsum += (bneed_of(L.cv[0].fph[ph].geom, c.s16) + 63) / 64 * 64;
            }
            const bool chained = !c.no_batch && c.wpart_used + wsum <= P->wpart_all_floats[c.s16] && c.bpart_used + bsum <= P->bpart_all_floats[c.s16];
            BiasChain bc;
            for (int ph = 0; ph < 4; ++ph) {
                if (chained) LF_TRY(run_wgrad(c, L.cv[0].fph[ph], L.cv[0], c.at(L.x), X, nullptr, nullptr, 0, false, &bc));
                else LF_TRY(run_wgrad(c, L.cv[0].fph[ph], L.cv[0], c.at(L.x), X, nullptr, nullptr, ph > 0, true));
            }
            LfTapArgs a = lf_no_args();
            int epi = 0;
            if (c.g_enc && L.x == lf_erfnet_encoder_offset(P)) {   // gradient of the --clas heads joins here
                a.add_src = c.g_enc;
                epi |= LF_EPI_ADD;
            }
            add_prep(a, epi, L.cv[0].dg[0].geom);
            LF_TRY(run_gemm(c, L.cv[0].dg[0], X, F, nullptr, LF_PRO_NONE, epi, a));
            out = F;
            if (can_prep) { prepped = true; prep_rows = c.last_rows; prep_ld = a.stats_ld; }
        } else if (L.x < 0) {
            // stem: weight gradient only (the image needs no gradient)
            const int Cc = 16 - L.Cin, rows = lf_stem_wgrad_rows(N, L.Hin, L.Win);
            if (c.grads[L.cv[0].p_w]) {
                const RowSums rs = row_sum_regions(c, (long)rows * Cc * L.Cin * 9, (long)rows * Cc);
                LF_TRY(lf_stem_wgrad(img, X, N, L.Cin, L.Hin, L.Win, rs.wrows, rs.brows, c.s16, c.st));
                LF_TRY(row_sums_finish(c, rs, rows, Cc * L.Cin * 9, c.grads[L.cv[0].p_w], Cc, c.grads[L.cv[0].p_b]));
            }
            out = F;
        } else {
            LF_TRY(run_wgrad(c, L.cv[0].fwd, L.cv[0], c.at(L.x), X, nullptr, nullptr, 0));
            LF_TRY(lf_pool_bwd(c.at(L.x), X, N, L.Hin, L.Win, L.Cin, L.Cout, L.Cout - L.Cin, F, c.s16, c.st));
            for (int ph = 0; ph < 4; ++ph) {
                LfTapArgs a = lf_no_args();
                a.add_src = F;
                LF_TRY(run_gemm(c, L.cv[0].dg[ph], X, F, nullptr, LF_PRO_NONE, LF_EPI_ADD, a));
            }
            out = F;
        }
        in = out;
    }
    if (gin) *gin = in;
    return 0;
}

int upload_and_pack(const Ctx& c, const float* const* params_dev) {
    const lf_erfnet_plan* P = c.P;
    LfPackEntry* ent = reinterpret_cast<LfPackEntry*>(c.at(P->off_entries));
    // (8 KB from a pageable vector; a page-locked copy of the table was measured: no difference on the step)
    if (hipMemcpyAsync(ent, P->packs.data(), P->packs.size() * sizeof(LfPackEntry), hipMemcpyHostToDevice, c.st) != hipSuccess)
        return lf_fail("erfnet: upload of the pack table failed");
    LF_TRY(lf_pack_weights_launch(ent, (int)P->packs.size(), params_dev, c.at(P->off_packed), c.st));
    if (P->precision == 2)
        LF_TRY(lf_pack_weights_bf16_launch(ent, (int)P->packs.size(), params_dev, c.at(P->off_packed16), c.st));
    if (P->precision == 3)
        LF_TRY(lf_pack_weights_split_launch(ent, (int)P->packs.size(), params_dev, c.at(P->off_packed48), c.st));
    return 0;
}

}  // namespace

extern "C" {


// Forward.  img (N,Cin,H,W) fp32 NCHW; params_host / params_dev: the n_params parameter tensors in
// state_dict order (weights, biases, BN weight/bias; buffers excluded), as a HOST array and a DEVICE
// array of device pointers; running_host: 2*n_bn device pointers (mean, var per BN, module order);
// dropmask: device buffer of lf_erfnet_dropmask_floats() floats or NULL; head = 0 (output_conv) or 1
// (output_conv2); logits out (N,Cout,H,W) NCHW; enc_out: optional NHWC->NCHW copy target is NOT
// produced here (the encoder output stays in the workspace: lf_erfnet_export_encoder).
// head = -1: ENCODER ONLY (Net.forward(only_encode=True), ERFNet.py:151-153): the decoder layers do not run -- their BatchNorm
// running statistics stay untouched, as in the reference, where the decoder is never called -- and logits may be NULL.
int lf_erfnet_forward(const lf_erfnet_plan* P, const float* img, const float* const* params_host,
                      const float* const* params_dev, float* const* running_host, const float* dropmask, int training,
                      int head, float* logits, void* workspace, size_t workspace_bytes, void* stream) {
    LF_REQUIRE(P && img && params_host && params_dev && running_host && (logits || head < 0) && workspace, "lf_erfnet_forward: null pointer");
    LF_REQUIRE(workspace_bytes >= lf_erfnet_workspace_bytes(P), "lf_erfnet_forward: workspace too small");
    LF_REQUIRE(head >= -1 && head < P->n_heads, "lf_erfnet_forward: head %d out of range", head);
    Ctx c{P, (float*)workspace, params_host, nullptr, running_host, dropmask, training, (hipStream_t)stream};
    c.s16 = P->precision == 2;
    LF_TRY(upload_and_pack(c, params_dev));
    if (head < 0) return forward_layers(c, img, encoder_layers(P));
    LF_TRY(forward_layers(c, img, (int)P->layers.size()));
    return lf_head_fwd(c.at(P->head_in), params_host[P->p_head_w[head]], params_host[P->p_head_b[head]], logits, P->N, P->H / 2,
                       P->W / 2, P->Cout + head, c.s16, c.st);   // output_conv2 has one more channel (ERFNet.py:125-126)
}

// Backward of the forward that last used `workspace`.  grad_logits (N,Cout,H,W) NCHW; grad_encoder: optional
// (N,H/8,W/8,128) NHWC gradient w.r.t. the encoder output (the `shared_encoder` the --clas heads consume);
// training: the mode the forward ran in (0 = running statistics: BatchNorm backward is then the affine map's);
// grads_host: n_params device pointers receiving d loss / d param (NULL entries are skipped:
// encoder.output_conv, the unused head).  Gradients are WRITTEN, not accumulated.
// head = -1: backward of the encoder-only forward: grad_encoder is the incoming gradient (required), grad_logits is ignored,
// decoder parameters receive no gradient (their grads_host entries must be NULL).
int lf_erfnet_backward(const lf_erfnet_plan* P, const float* img, const float* grad_logits,
                       const float* grad_encoder, const float* const* params_host, float* const* grads_host,
                       const float* dropmask, int training, int head, void* workspace, size_t workspace_bytes, void* stream) {
    LF_REQUIRE(P && img && (grad_logits || head < 0) && params_host && grads_host && workspace, "lf_erfnet_backward: null pointer");
    LF_REQUIRE(workspace_bytes >= lf_erfnet_workspace_bytes(P), "lf_erfnet_backward: workspace too small");
    LF_REQUIRE(head >= -1 && head < P->n_heads, "lf_erfnet_backward: head %d out of range", head);
    Ctx c{P, (float*)workspace, params_host, grads_host, nullptr, dropmask, training, (hipStream_t)stream};
    c.g_enc = grad_encoder;
    c.s16 = P->precision == 2;
    LF_REQUIRE(!(c.s16 && grad_encoder), "lf_erfnet_backward: grad_encoder is not supported with bf16 tensors (mode 2)");
    float *gA = c.at(P->off_gA), *gB = c.at(P->off_gB), *gC = c.at(P->off_gC);
    if (head < 0) {       // encoder only: the incoming gradient IS d loss / d (encoder output); the pass overwrites its buffers
        LF_REQUIRE(grad_encoder, "lf_erfnet_backward: the encoder-only backward needs grad_encoder");
        const int ne = encoder_layers(P);
        const Layer& Le = P->layers[ne - 1];
        const size_t bytes = (size_t)P->N * Le.Hout * Le.Wout * Le.Cout * sizeof(float);
        if (hipMemcpyAsync(gA, grad_encoder, bytes, hipMemcpyDeviceToDevice, c.st) != hipSuccess)
            return lf_fail("lf_erfnet_backward: copy of grad_encoder failed");
        c.g_enc = nullptr;
        LF_TRY(backward_layers(c, img, gA, gB, gC, ne));
        if (!c.reduce_jobs.empty()) LF_TRY(lf_wgrad_reduce_batch_launch(c.reduce_jobs.data(), (int)c.reduce_jobs.size(), c.st));
        return 0;
    }
    const int h = P->H / 2, w = P->W / 2, K = P->Cout + head;
    const int pw = P->p_head_w[head], pb = P->p_head_b[head];
    if (grads_host[pw]) {
        const int rows = lf_head_wgrad_rows(P->N, h, w);
        const RowSums rs = row_sum_regions(c, (long)rows * 16 * K * 4, (long)rows * K);
        LF_TRY(lf_head_wgrad(c.at(P->head_in), grad_logits, rs.wrows, rs.brows, P->N, h, w, K, c.s16, c.st));
        LF_TRY(row_sums_finish(c, rs, rows, 16 * K * 4, grads_host[pw], K, grads_host[pb]));
    }
    LF_TRY(lf_head_bwd_data(grad_logits, params_host[pw], gA, P->N, h, w, K, c.s16, c.st));
    LF_TRY(backward_layers(c, img, gA, gB, gC, (int)P->layers.size()));
    if (!c.reduce_jobs.empty()) LF_TRY(lf_wgrad_reduce_batch_launch(c.reduce_jobs.data(), (int)c.reduce_jobs.size(), c.st));
    return 0;
}

// Per-kernel-family timing for the roofline report.  enable=1 starts recording a HIP event pair around
// every tap-GEMM (family 0: forward + data-gradient) and weight-gradient (family 1) launch of subsequent
// forward/backward calls; lf_erfnet_profile_read waits for them, adds up elapsed ms, algorithmic FLOPs
// (2 * pixels * Cs * Cd * taps) and launch counts per family (3 doubles each), and clears the records.
int lf_erfnet_profile(const lf_erfnet_plan* P, int enable) { P->prof_on = enable; return 0; }
int lf_erfnet_profile_read(const lf_erfnet_plan* P, double* out6, const char* csv_path) {
    for (int i = 0; i < 6; ++i) out6[i] = 0.0;
    FILE* f = nullptr;
    if (csv_path) {
        f = fopen(csv_path, "w");
        if (f) fprintf(f, "family,layer,Cs,Cd,ntaps,npix,epi,us,tflops\n");
    }
    for (auto& r : P->prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            out6[r.family * 3 + 0] += ms; out6[r.family * 3 + 1] += r.flops; out6[r.family * 3 + 2] += 1.0;
            if (f) fprintf(f, "%d,%d,%d,%d,%d,%ld,%d,%.2f,%.2f\n", r.family, r.layer, r.Cs, r.Cd, r.ntaps, r.npix, r.epi,
                           ms * 1e3, r.flops / (ms * 1e-3) / 1e12);
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    if (f) fclose(f);
    P->prof.clear();
    return 0;
}

// Copy an NHWC activation of the workspace to an NCHW tensor (encoder output for the drop-in tuple,
// per-layer parity tests).
int lf_nhwc_to_nchw(const float* src, float* dst, int N, int H, int W, int C, void* stream);

}  // extern "C"

namespace {
// T = the storage type of the NHWC workspace tensor (float, or lf_bf16 in precision mode 2); the NCHW side is always fp32
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ s, float* __restrict__ d, int H, int W,
                                                          int C, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int w = (int)(i % W);
        long r = i / W;
        const int h = (int)(r % H);
        r /= H;
        const int ch = (int)(r % C);
        const long n = r / C;
        d[i] = lf_ld1(s + ((n * H + h) * W + w) * C + ch);
    }
}
int nhwc_to_nchw(const float* src, float* dst, int N, int H, int W, int C, int s16, hipStream_t st) {
    const long total = (long)N * H * W * C;
    int grid = lf_cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    if (s16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<lf_bf16>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const lf_bf16*>(src), dst, H, W, C, total);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid), dim3(256), 0, st, src, dst, H, W, C, total);
    LF_CHECK_LAUNCH("nhwc_to_nchw");
    return 0;
}
}  // namespace

extern "C" int lf_nhwc_to_nchw(const float* src, float* dst, int N, int H, int W, int C, void* stream) {
    return nhwc_to_nchw(src, dst, N, H, W, C, 0, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------
// Block-level surface (round 4): a contiguous RANGE of the plan's layers as one call, so that the reference's sub-modules are
// callable on their own -- DownsamplerBlock / non_bottleneck_1d / UpsamplerBlock.forward(input), Encoder.forward(input,
// predict), Decoder.forward(input, flag) (BEV/Networks/ERFNet.py:19-22,44-60,86-95,104-107,129-142).  The range runs inside
// the plan of the whole network at the matching input size: same kernels, same workspace slots, same BatchNorm /
// Dropout2d handling as the full pass; only the range's parameters are read and only their gradients written.
// Tensors cross this boundary in the reference's NCHW fp32.  In precision mode 2 (bf16 tensors, round 6) the range's input and
// the incoming gradient are rounded to bf16 on their way into the workspace (round to nearest even, as every store of that mode)
// and the output / input gradient are widened on their way out.
// ---------------------------------------------------------------------------------------
namespace {
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ s, T* __restrict__ d, int H, int W, int C,
                                                          long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % C);
        long r = i / C;
        const int w = (int)(r % W);
        r /= W;
        const int h = (int)(r % H);
        const long n = r / H;
        lf_st1(d + i, s[((n * C + ch) * H + h) * W + w]);
    }
}
int nchw_to_nhwc(const float* src, float* dst, int N, int H, int W, int C, int s16, hipStream_t st) {
    const long total = (long)N * H * W * C;
    int grid = lf_cdiv(total, 256);
    if (grid > 8192) grid = 8192;
    if (s16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<lf_bf16>, dim3(grid), dim3(256), 0, st, src, reinterpret_cast<lf_bf16*>(dst), H, W, C, total);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid), dim3(256), 0, st, src, dst, H, W, C, total);
    LF_CHECK_LAUNCH("nchw_to_nhwc");
    return 0;
}
long layer_out_offset(const Layer& L) { return L.kind == K_NB ? L.b[4] : L.b[1]; }
// The part of the workspace layers [first, last) touch: the input slot of `first` (the previous layer's output) through the end of
// `last - 1`'s region, and the globals behind the layers (packed weights, statistics rows, ONE partial-row region, the three
// gradient buffers) -- not the other layers' activations and not the per-weight-gradient partial regions of the full pass.
void range_span(const lf_erfnet_plan* P, int first, int last, long* lo, long* hi) {
    const Layer& Lf = P->layers[first];
    *lo = first > 0 ? (Lf.x < Lf.reg_lo ? Lf.x : Lf.reg_lo) : Lf.reg_lo;
    *hi = last < (int)P->layers.size() ? P->layers[last].reg_lo : P->off_globals;
}
void compact(Ctx& c, int first, int last) {
    range_span(c.P, first, last, &c.lo, &c.hi);
    c.cut = c.P->off_globals;
    c.no_batch = true;
}
int check_range(const lf_erfnet_plan* P, int first, int last, int head, const char* who) {
    LF_REQUIRE(P && first >= 0 && last > first && last <= (int)P->layers.size(), "%s: bad layer range [%d, %d)", who, first, last);
    LF_REQUIRE(head < 0 || (last == (int)P->layers.size() && head < P->n_heads), "%s: the head follows the last layer only", who);
    return 0;
}
}  // namespace

extern "C" {

int lf_erfnet_num_layers(const lf_erfnet_plan* P) { return (int)P->layers.size(); }
// Bytes of the COMPACT workspace lf_erfnet_forward_range / _backward_range run layers [first, last) on: the range's own activations
// + the globals.  (A loop over the blocks of a network -- `for l in net.encoder.layers: x = l(x)` -- keeps one such workspace per
// block until its backward: together the activations of ONE network, not one whole-network workspace per block.)
size_t lf_erfnet_range_workspace_bytes(const lf_erfnet_plan* P, int first, int last) {
    if (!P || first < 0 || last <= first || last > (int)P->layers.size()) return 0;
    long lo, hi;
    range_span(P, first, last, &lo, &hi);
    return (size_t)((hi - lo) + (P->off_wpart_all - P->off_globals)) * sizeof(float);
}
// out6 = {Cin, Hin, Win, Cout, Hout, Wout} of a layer (module order: 0 = encoder.initial_block, 1.. = encoder.layers, then decoder.layers)
int lf_erfnet_layer_io(const lf_erfnet_plan* P, int layer, int* out6) {
    LF_REQUIRE(P && out6 && layer >= 0 && layer < (int)P->layers.size(), "lf_erfnet_layer_io: layer %d out of range", layer);
    const Layer& L = P->layers[layer];
    out6[0] = L.Cin; out6[1] = L.Hin; out6[2] = L.Win; out6[3] = L.Cout; out6[4] = L.Hout; out6[5] = L.Wout;
    return 0;
}

// x: (N, Cin, Hin, Win) NCHW input of layer `first` (the image when first = 0); y: (N, Cout, Hout, Wout) NCHW output of layer
// last - 1, or, with head >= 0 (last = the layer count), the logits (N, out_channels + head, H, W).
int lf_erfnet_forward_range(const lf_erfnet_plan* P, int first, int last, int head, const float* x,
                            const float* const* params_host, const float* const* params_dev, float* const* running_host,
                            const float* dropmask, int training, float* y, void* workspace, size_t workspace_bytes, void* stream) {
    LF_TRY(check_range(P, first, last, head, "lf_erfnet_forward_range"));
    LF_REQUIRE(x && y && params_host && params_dev && running_host && workspace, "lf_erfnet_forward_range: null pointer");
    LF_REQUIRE(workspace_bytes >= lf_erfnet_range_workspace_bytes(P, first, last), "lf_erfnet_forward_range: workspace too small");
    Ctx c{P, (float*)workspace, params_host, nullptr, running_host, dropmask, training, (hipStream_t)stream};
    compact(c, first, last);
    c.s16 = P->precision == 2;
    LF_TRY(upload_and_pack(c, params_dev));
    const Layer& Lf = P->layers[first];
    if (first > 0) LF_TRY(nchw_to_nhwc(x, c.at(Lf.x), P->N, Lf.Hin, Lf.Win, Lf.Cin, c.s16, c.st));
    LF_TRY(forward_layers(c, x, last, first));
    const Layer& Ll = P->layers[last - 1];
    if (head >= 0)
        return lf_head_fwd(c.at(P->head_in), params_host[P->p_head_w[head]], params_host[P->p_head_b[head]], y, P->N, P->H / 2,
                           P->W / 2, P->Cout + head, c.s16, c.st);
    return nhwc_to_nchw(c.at(layer_out_offset(Ll)), y, P->N, Ll.Hout, Ll.Wout, Ll.Cout, c.s16, c.st);
}

// Backward of lf_erfnet_forward_range on the same workspace: gy = d loss / d y (NCHW, the forward's output shape); gx (NCHW,
// the forward's input shape) or NULL (always NULL-able; ignored when first = 0: the image takes no gradient).
// grads_host: n_params device pointers; only the range's (and the head's) entries are written, NULL entries skipped.
int lf_erfnet_backward_range(const lf_erfnet_plan* P, int first, int last, int head, const float* x, const float* gy,
                             const float* const* params_host, float* const* grads_host, const float* dropmask, int training,
                             float* gx, void* workspace, size_t workspace_bytes, void* stream) {
    LF_TRY(check_range(P, first, last, head, "lf_erfnet_backward_range"));
    LF_REQUIRE(x && gy && params_host && grads_host && workspace, "lf_erfnet_backward_range: null pointer");
    LF_REQUIRE(workspace_bytes >= lf_erfnet_range_workspace_bytes(P, first, last), "lf_erfnet_backward_range: workspace too small");
    Ctx c{P, (float*)workspace, params_host, grads_host, nullptr, dropmask, training, (hipStream_t)stream};
    compact(c, first, last);
    c.s16 = P->precision == 2;
    float *gA = c.at(P->off_gA), *gB = c.at(P->off_gB), *gC = c.at(P->off_gC);
    const Layer& Ll = P->layers[last - 1];
    if (head >= 0) {
        const int h = P->H / 2, w = P->W / 2, K = P->Cout + head;
        const int pw = P->p_head_w[head], pb = P->p_head_b[head];
        if (grads_host[pw]) {
            const int rows = lf_head_wgrad_rows(P->N, h, w);
            const RowSums rs = row_sum_regions(c, (long)rows * 16 * K * 4, (long)rows * K);
            LF_TRY(lf_head_wgrad(c.at(P->head_in), gy, rs.wrows, rs.brows, P->N, h, w, K, c.s16, c.st));
            LF_TRY(row_sums_finish(c, rs, rows, 16 * K * 4, grads_host[pw], K, grads_host[pb]));
        }
        LF_TRY(lf_head_bwd_data(gy, params_host[pw], gA, P->N, h, w, K, c.s16, c.st));
    } else {
        LF_TRY(nchw_to_nhwc(gy, gA, P->N, Ll.Hout, Ll.Wout, Ll.Cout, c.s16, c.st));
    }
    float* gin = nullptr;
    LF_TRY(backward_layers(c, x, gA, gB, gC, last, first, &gin));
    if (!c.reduce_jobs.empty()) LF_TRY(lf_wgrad_reduce_batch_launch(c.reduce_jobs.data(), (int)c.reduce_jobs.size(), c.st));
    const Layer& Lf = P->layers[first];
    if (gx && first > 0) LF_TRY(nhwc_to_nchw(gin, gx, P->N, Lf.Hin, Lf.Win, Lf.Cin, c.s16, c.st));
    return 0;
}

}  // extern "C"
