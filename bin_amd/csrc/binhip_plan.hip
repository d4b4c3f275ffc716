// binhip_plan.hip — one whole RDN sub-network (reference RDN.py:210-222 / 268-280 / 322-334) as a
// fixed sequence of 1 + 66 kernel launches issued from C on the caller's stream: no Python, no
// allocation, no host sync between layers (so the sequence is hipGraph-capturable).
//
// Workspace (chunk planes, fp16; each tensor = hi planes followed by lo planes when nterms == 3):
//   X0   [kc0]      pixel-unshuffled input frames, half resolution (h = H/2, w = W/2)
//   F1   [6]        SFENet1 output (f__1)
//   BLK  [13][14]   dense-block buffers: BLK[d][0:6] = input of RDB d (= output of RDB d-1),
//                   BLK[d][6+2c : 8+2c] = output of conv c of RDB d  -> "cat" is free
//   G0,G1 [6]       GFF.0 / GFF.1(+f__1)
//   U    [4]        UPNet.0 output after PixelShuffle, full resolution
// (numbers for bin_stage4's shape G0 = 96, D = 12, C = 4, G = 32; in general c0 = G0/16 planes per feature map, cg = G/16
//  per conv output, BLK [D + 1][c0 + C cg], layer index 2 + d (C + 1) + c — include/binhip.h, BinRdnShape)
#include "binhip_conv_common.h"
// 1 (default since round 6) = the forward's 1x1 layers (GFF.0, an unfused LFF) run on the pixel grid reshaped to 32-pixel rows (see
// mk()): bit-identical, 720p window -0.3 % (GFF.0 reads 1152 channel planes: 8 KiB contiguous per plane and tile instead of 8 runs of
// 1 KiB at a 21 KiB pitch).  The same reshape in the BACKWARD plan (1x1 backward-data and weight gradients at 128-pixel rows) measured
// +0.2 / +0.3 / -0.05 % on the training step and is not there (profiles/r06_experiments.md).  0 = side builds, the old geometry.
#ifndef BINHIP_PLAN_LINEAR_1X1
#define BINHIP_PLAN_LINEAR_1X1 1
#endif

namespace {

// resolved network shape (BinRdnShape with the defaults filled in)
struct Shp {
    int G0, D, C, G;
    int c0, cg, cb;       // planes per feature map / per conv output / per dense-block buffer
    int L;                // layers
    bool stage4;          // bin_stage4's shape: the fused tail / three-phase kernels exist for it
};
bool resolve_shape(const BinRdnShape* s, Shp* o) {
    Shp h;
    const bool dflt = !s || (s->G0 == 0 && s->D == 0 && s->C == 0 && s->G == 0);
    h.G0 = dflt ? 96 : s->G0; h.D = dflt ? 12 : s->D; h.C = dflt ? 4 : s->C; h.G = dflt ? 32 : s->G;
    if (h.G0 < 32 || h.G0 > 256 || h.G0 % 32 || h.G < 32 || h.G > 128 || h.G % 32) return false;
    if (h.C < 1 || h.C > BINHIP_RDN_MAX_CONVS || h.D < 1 || h.D > 20) return false;
    h.c0 = h.G0 / 16; h.cg = h.G / 16; h.cb = h.c0 + h.C * h.cg;
    h.L = 2 + h.D * (h.C + 1) + 4;
    if (h.L > BINHIP_RDN_MAX_LAYERS) return false;
    h.stage4 = (h.G0 == 96 && h.C == 4 && h.G == 32);
    *o = h;
    return true;
}
inline int layer_conv(const Shp& sh, int d, int c) { return 2 + d * (sh.C + 1) + c; }      // c == C: the block's LFF

struct Ws {
    int64_t P, PF;        // plane elems at half / full res
    int kc0;
    int64_t x0, f1, blk, g0, g1, u, total;   // element offsets of the hi part
    int64_t sync;                            // element offset of the dense-block sync words (32 reserved + 2 T flags, uint32)
    int tiles;                               // 16x32 tiles of one dense-block conv launch
    int64_t s_x0, s_f1, s_blk, s_g, s_u;     // sizes (elements) of each tensor's hi part
    int nt;
};

Ws make_ws(int N, int H, int W, int nin, int nt, const Shp& sh) {
    Ws w;
    const int h = H / 2, ww = W / 2;
    w.P = (int64_t)N * h * ww * 16;
    w.PF = (int64_t)N * H * W * 16;
    w.kc0 = (12 * nin + 15) / 16;
    w.nt = nt;
    const int mul = (nt == 3) ? 2 : 1;
    w.s_x0 = w.kc0 * w.P; w.s_f1 = sh.c0 * w.P; w.s_blk = (int64_t)(sh.D + 1) * sh.cb * w.P; w.s_g = sh.c0 * w.P;
    w.s_u = 4 * w.PF;
    int64_t o = 0;
    w.x0 = o; o += mul * w.s_x0;
    w.f1 = o; o += mul * w.s_f1;
    w.blk = o; o += mul * w.s_blk;
    w.g0 = o; o += mul * w.s_g;
    w.g1 = o; o += mul * w.s_g;
    w.u = o; o += mul * w.s_u;
    // sync words of the one-launch dense blocks (binhip_conv_x3.hip): 32 reserved words (the round-2 work-queue heads) + two
    // flag words per tile; 4-byte words kept in the fp16-element address space (2 elements each), 256-B aligned
    w.tiles = N * ((h + 15) / 16) * ((ww + 31) / 32);
    o = (o + 127) & ~(int64_t)127;
    w.sync = o; o += 2 * (32 + 2 * (int64_t)w.tiles);
    w.total = o;
    return w;
}

}  // namespace

extern "C" {

size_t binhip_rdn_workspace_bytes(int N, int H, int W, int n_inputs, int nterms, const BinRdnShape* shape) {
    if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return 0;
    if (n_inputs != 2 && n_inputs != 3 && n_inputs != 5) return 0;
    Shp sh;
    if (!resolve_shape(shape, &sh)) return 0;
    return (size_t)make_ws(N, H, W, n_inputs, nterms, sh).total * 2 + 256;
}

int binhip_rdn_workspace_layout(int N, int H, int W, int n_inputs, int nterms, const BinRdnShape* shape, int64_t* out,
                                int n_out) {
    if (!out || n_out < BINHIP_RDN_LAYOUT_WORDS) return BINHIP_E_ARG;
    Shp sh;
    if (!resolve_shape(shape, &sh) || binhip_rdn_workspace_bytes(N, H, W, n_inputs, nterms, shape) == 0) return BINHIP_E_SHAPE;
    const Ws w = make_ws(N, H, W, n_inputs, nterms, sh);
    const int64_t v[BINHIP_RDN_LAYOUT_WORDS] = {w.P, w.PF, w.kc0, w.x0, w.s_x0, w.f1, w.s_f1, w.blk, w.s_blk,
                                                 w.g0, w.s_g, w.g1, w.s_g, w.u, w.s_u, nterms == 3 ? 1 : 0};
    for (int i = 0; i < BINHIP_RDN_LAYOUT_WORDS; ++i) out[i] = v[i];
    return 0;
}

int binhip_rdn_forward(const BinRdnPlan* p, const float* const* inputs, float* out, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (!p || !inputs || !out || !workspace) return BINHIP_E_ARG;
    const int N = p->N, H = p->H, W = p->W, nin = p->n_inputs, nt = p->nterms;
    if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return BINHIP_E_SHAPE;
    if (nin != 2 && nin != 3 && nin != 5) return BINHIP_E_SHAPE;
    if (nt != 1 && nt != 3) return BINHIP_E_ARG;
    Shp sh;
    if (!resolve_shape(&p->shape, &sh)) return BINHIP_E_SHAPE;
    const size_t need = binhip_rdn_workspace_bytes(N, H, W, nin, nt, &p->shape);
    if (workspace_bytes < need) return BINHIP_E_WORKSPACE;
    for (int i = 0; i < nin; ++i) if (!inputs[i]) return BINHIP_E_ARG;
    for (int i = 0; i < sh.L; ++i)
        if (!p->w_hi[i] || !p->bias[i] || (nt == 3 && !p->w_lo[i])) return BINHIP_E_ARG;

    hipStream_t s = (hipStream_t)stream;
    const Ws w = make_ws(N, H, W, nin, nt, sh);
    const int h = H / 2, ww = W / 2;
    // align the workspace base to 256 B
    _Float16* base = (_Float16*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    auto HI = [&](int64_t off) { return (void*)(base + off); };
    auto LO = [&](int64_t off, int64_t size) { return nt == 3 ? (void*)(base + off + size) : (void*)nullptr; };

    int rc = binhip_pack_inputs(inputs, nin, N, H, W, HI(w.x0), LO(w.x0, w.s_x0), p->status, stream);
    if (rc) return rc;

    auto mk = [&](int layer, int ks, int cin_chunks, int cout, int cout_pad, int epi, int relu, int Hc, int Wc,
                  int64_t x_off, int64_t x_size, int cpg, int64_t gstride,
                  int64_t y_off, int64_t y_size, int64_t r_off, int64_t r_size) -> BhConvCall {
        BhConvCall c;
        c.d.N = N; c.d.H = Hc; c.d.W = Wc; c.d.ksize = ks; c.d.cin_chunks = cin_chunks; c.d.cout = cout;
        c.d.cout_pad = cout_pad; c.d.nterms = nt; c.d.epilogue = epi; c.d.relu = relu;
        c.d.x_cpg = cpg; c.d.x_group_stride = gstride; c.d.n_images = 0; c.d.reserved = 0; c.d.status = p->status;
#if BINHIP_PLAN_LINEAR_1X1
        // a 1x1 convolution is pointwise: any reshape of the pixel grid computes the same values.  [H][W] -> [H * W / 32][32] makes a
        // workgroup's TH x 32 tile TH KiB of CONTIGUOUS bytes per plane instead of TH runs of 1 KiB at a row pitch of W * 32 B
        if (ks == 1 && epi == BINHIP_EPI_PLANES && Wc > 32 && Wc % 32 == 0) { c.d.H = Hc * (Wc / 32); c.d.W = 32; }
#endif
        // SFENet1 on 2 / 3 input frames: 24 / 36 channels = the last chunk's upper half is zero padding (packer and relayout)
        if (layer == 0 && ks == 5 && (12 * nin) % 16 >= 1 && (12 * nin) % 16 <= 8) c.d.reserved = BINHIP_CONV_HALF_LAST_CHUNK;
        c.x_hi = HI(x_off); c.x_lo = LO(x_off, x_size);
        c.w_hi = p->w_hi[layer]; c.w_lo = p->w_lo[layer]; c.bias = p->bias[layer];
        c.r_hi = (r_off >= 0) ? HI(r_off) : nullptr;
        c.r_lo = (r_off >= 0) ? LO(r_off, r_size) : nullptr;
        c.y_hi = (y_off >= 0) ? HI(y_off) : nullptr;
        c.y_lo = (y_off >= 0) ? LO(y_off, y_size) : nullptr;
        c.y_f32 = nullptr;
        c.status = p->status;
        c.prof = p->profiler;
        for (int i = 0; i < 5; ++i) c.images[i] = nullptr;
        if (epi == BINHIP_EPI_FINAL || epi == BINHIP_EPI_FINAL_SUBPIX) {
            c.y_f32 = out;
            c.d.n_images = nin;
            for (int i = 0; i < nin; ++i) c.images[i] = inputs[i];
        }
        return c;
    };
    auto conv = [&](int layer, int ks, int cin_chunks, int cout, int cout_pad, int epi, int relu, int Hc, int Wc,
                    int64_t x_off, int64_t x_size, int cpg, int64_t gstride,
                    int64_t y_off, int64_t y_size, int64_t r_off, int64_t r_size) -> int {
        return bh_launch_conv(mk(layer, ks, cin_chunks, cout, cout_pad, epi, relu, Hc, Wc, x_off, x_size, cpg, gstride,
                                 y_off, y_size, r_off, r_size), s);
    };
    // one-launch dense blocks (BINHIP_PLAN_RDB3, opt-in, fp32-class path): one memset per call zeroes the per-tile flags
    // (block d publishes the value d + 1, so the blocks of one call never confuse
    // each other's flags; the memset only removes what an earlier call or uninitialised memory left behind)
    const bool rdb3 = (p->reserved & BINHIP_PLAN_RDB3) && nt == 3 && sh.stage4;
    unsigned* sync_words = (unsigned*)(base + w.sync);
    const int cus = binhip_device_cus();
    if (rdb3) {
        hipError_t me = hipMemsetAsync(sync_words, 0, (size_t)(32 + 2 * (size_t)w.tiles) * 4, s);
        if (me != hipSuccess) return (int)me;
    }
    const int P_ = BINHIP_EPI_PLANES;
    const int64_t P = w.P;
    const int G0 = sh.G0, G = sh.G, C = sh.C, D = sh.D, c0 = sh.c0, cg = sh.cg, cb = sh.cb;
    // SFENet1 5x5 (RDN.py:187/245/299) and SFENet2 3x3 (:188)
    if ((rc = conv(0, 5, w.kc0, G0, G0, P_, 0, h, ww, w.x0, w.s_x0, 0, 0, w.f1, w.s_f1, -1, 0))) return rc;
    if ((rc = conv(1, 3, c0, G0, G0, P_, 0, h, ww, w.f1, w.s_f1, 0, 0, w.blk, w.s_blk, -1, 0))) return rc;
    // D residual dense blocks (RDN.py:149-165)
    for (int d = 0; d < D; ++d) {
        const int64_t b = w.blk + (int64_t)d * cb * P;
        const bool fuse = sh.stage4 && !(p->reserved & BINHIP_PLAN_NO_FUSE);
        bool done3 = false;
        if (rdb3 && fuse) {
            // convs 0-2 as three phases of one launch (static tile ownership + neighbour flags instead of two kernel boundaries)
            ConvKArgs ka[3];
            bool ok3 = true;
            for (int c = 0; c < 3 && ok3; ++c) {
                BhConvCall cc = mk(layer_conv(sh, d, c), 3, c0 + cg * c, G, G, P_, 1, h, ww, b, w.s_blk, 0, 0,
                                   b + (int64_t)(c0 + cg * c) * P, w.s_blk, -1, 0);
                if ((rc = bh_prepare_conv(cc, &ka[c]))) return rc;
                ok3 = ok3 && ka[c].wt;
            }
            if (ok3) {
                rc = bh_launch_rdb3_x3(ka, sync_words + 32, (unsigned)(d + 1), cus, s);
                if (rc == 0) done3 = true;
                else if (rc != BINHIP_E_SHAPE) return rc;          // E_SHAPE: not co-resident on this device -> per-conv launches
            }
        }
        for (int c = 0; c < (fuse ? C - 1 : C) && !done3; ++c) {
            if ((rc = conv(layer_conv(sh, d, c), 3, c0 + cg * c, G, G, P_, 1, h, ww, b, w.s_blk, 0, 0,
                           b + (int64_t)(c0 + cg * c) * P, w.s_blk, -1, 0))) return rc;
        }
        if (fuse) {
            // conv #3 + LFF + residual in one kernel (binhip_fused.hip); o3 is only written out for training
            const int L3 = layer_conv(sh, d, 3), LF = layer_conv(sh, d, 4);
            if ((rc = binhip_rdb_tail_fwd(N, h, ww, nt, HI(b), LO(b, w.s_blk), p->w_hi[L3], p->w_lo[L3], p->bias[L3],
                                          p->w_hi[LF], p->w_lo[LF], p->bias[LF], HI(b + (int64_t)cb * P),
                                          LO(b + (int64_t)cb * P, w.s_blk), (p->reserved & BINHIP_PLAN_KEEP_ACTS) ? 1 : 0, p->status, stream)))
                return rc;
        } else if ((rc = conv(layer_conv(sh, d, C), 1, cb, G0, G0, P_, 0, h, ww, b, w.s_blk, 0, 0, b + (int64_t)cb * P, w.s_blk, b, w.s_blk)))
            return rc;
    }
    const int LG = sh.L - 4;
    // GFF.0 1x1 over cat(RDBs_out) (RDN.py:199, 218): D groups of c0 chunks, one per dense block
    if ((rc = conv(LG, 1, D * c0, G0, G0, P_, 0, h, ww, w.blk + (int64_t)cb * P, w.s_blk, c0, (int64_t)cb * P, w.g0, w.s_g, -1, 0))) return rc;
    // GFF.1 3x3, x += f__1 (RDN.py:200, 219)
    if ((rc = conv(LG + 1, 3, c0, G0, G0, P_, 0, h, ww, w.g0, w.s_g, 0, 0, w.g1, w.s_g, w.f1, w.s_f1))) return rc;
    // UPNet as ONE 5x5 convolution G0 -> 12 sub-pixel channels (BINHIP_PLAN_FUSED_UPNET, include/binhip.h): inference only — training keeps
    // the two layers, whose activations and separate weight gradients its backward needs
    const bool fused_up = (p->reserved & BINHIP_PLAN_FUSED_UPNET) &&
                          (!(p->reserved & BINHIP_PLAN_KEEP_ACTS) || (p->reserved & BINHIP_PLAN_FUSED_UPNET_TRAIN)) &&
                          sh.L + 1 < BINHIP_RDN_MAX_LAYERS && p->w_hi[sh.L] && (nt == 1 || p->w_lo[sh.L]) && p->bias[sh.L] &&
                          p->w_hi[sh.L + 1] && p->bias[sh.L + 1];
    if (fused_up) {
        if ((rc = bh_launch_conv(mk(sh.L, 5, c0, 12, 32, BINHIP_EPI_FINAL_SUBPIX, 0, h, ww, w.g1, w.s_g, 0, 0, -1, 0, -1, 0), s))) return rc;
        // ... and the one-pixel full-resolution border ring from its own operators (UPNet.2 pads the intermediate, not the input)
        return bh_launch_upnet_ring(HI(w.g1), LO(w.g1, w.s_g), (const float*)p->w_hi[sh.L + 1], p->bias[sh.L + 1], out, inputs, nin, N, h, ww,
                                    G0, s);
    }
    // UPNet.0 3x3 G0->256 + PixelShuffle(2) (RDN.py:205-206)
    if ((rc = conv(LG + 2, 3, c0, 256, 256, BINHIP_EPI_SHUFFLE, 0, h, ww, w.g1, w.s_g, 0, 0, w.u, w.s_u, -1, 0))) return rc;
    // UPNet.2 3x3 64->3 + mean(inputs) (RDN.py:207, 221/279/333)
    if ((rc = conv(LG + 3, 3, 4, 3, 32, BINHIP_EPI_FINAL, 0, H, W, w.u, w.s_u, 0, 0, -1, 0, -1, 0))) return rc;
    return 0;
}

}  // extern "C"

// ====================================================================================================
// Backward of one RDN sub-network (autograd of RDN.py:210-222 / 268-280 / 322-334), again as one fixed launch
// sequence from C.  Gradient activations are chunk planes like the forward ones, stored multiplied by a per-call
// power-of-two scale (fp16 range); weight gradients come out as fp32 OIHW.
//   gOut [1 @HxW] <- gout*scale        gU [4 @HxW]     gUu [16]  (un-shuffled)      gG1, gG0, gF1 [6]
//   GY [13][6]: GY[0] = grad of SFENet2's output, GY[d+1] = grad of RDB d's output (GFF.0 dgrad + chained RDBs)
//   gcat [14]: gradient of the current dense block's concat buffer        gX0 [<=4]
namespace {

struct Bws {
    int64_t P, PF;
    int kc0, gx0_chunks;
    int64_t gout, gu, guu, gg1, gg0, gf1, gy, gcat, gcat2, gx0, total_halfs;
    int64_t s_gout, s_gu, s_guu, s_g, s_gy, s_gcat, s_gx0;
    size_t wg_bytes, sc_off_bytes, wg_off_bytes, total_bytes;
};

Bws make_bws(int N, int H, int W, int nin, int nt, const Shp& sh) {
    Bws b;
    const int h = H / 2, w = W / 2;
    b.P = (int64_t)N * h * w * 16;
    b.PF = (int64_t)N * H * W * 16;
    b.kc0 = (12 * nin + 15) / 16;
    b.gx0_chunks = ((12 * nin + 31) / 32) * 2;
    const int mul = (nt == 3) ? 2 : 1;
    b.s_gout = b.PF; b.s_gu = 4 * b.PF; b.s_guu = 16 * b.P; b.s_g = sh.c0 * b.P; b.s_gy = (int64_t)(sh.D + 1) * sh.c0 * b.P;
    b.s_gcat = (int64_t)sh.cb * b.P; b.s_gx0 = (int64_t)b.gx0_chunks * b.P;
    int64_t o = 0;
    b.gout = o; o += mul * b.s_gout;
    b.gu = o; o += mul * b.s_gu;
    b.guu = o; o += mul * b.s_guu;
    b.gg1 = o; o += mul * b.s_g;
    b.gg0 = o; o += mul * b.s_g;
    b.gf1 = o; o += mul * b.s_g;
    b.gy = o; o += mul * b.s_gy;
    b.gcat = o; o += mul * b.s_gcat;
    b.gcat2 = o; o += mul * b.s_gcat;      // dense blocks alternate between the two: block d's weight gradients (side
                                           // stream) still read one while block d-1's backward-data fills the other
    b.gx0 = o; o += mul * b.s_gx0;
    b.total_halfs = o;
    // weight-gradient partial workspace: max over the layer shapes
    size_t wg = 0;
    auto mx = [&](size_t v) { if (v > wg) wg = v; };
    mx(binhip_wgrad_workspace_bytes(3, N, H, W, 4, 3));                       // UPNet.2
    mx(binhip_wgrad_workspace_bytes(3, N, h, w, sh.c0, 256));                 // UPNet.0
    mx(binhip_wgrad_workspace_bytes(3, N, h, w, sh.c0, sh.G0));               // GFF.1 / SFENet2
    mx(binhip_wgrad_workspace_bytes(1, N, h, w, sh.D * sh.c0, sh.G0));        // GFF.0
    mx(binhip_wgrad_workspace_bytes(1, N, h, w, sh.cb, sh.G0));               // LFF
    for (int c = 0; c < sh.C; ++c) mx(binhip_wgrad_workspace_bytes(3, N, h, w, sh.c0 + sh.cg * c, sh.G));
    mx(binhip_wgrad_workspace_bytes(5, N, h, w, b.kc0, sh.G0));               // SFENet1
    b.wg_bytes = wg;
    size_t bytes = ((size_t)b.total_halfs * 2 + 255) & ~(size_t)255;
    b.sc_off_bytes = bytes; bytes += 8192;                       // scale[2] + 1024 amax partials (+pad)
    b.wg_off_bytes = bytes; bytes += (size_t)(sh.C + 1) * wg;    // one partial region per layer of a dense block (batched reduce)
    b.total_bytes = bytes + 256;
    return b;
}

}  // namespace

extern "C" {

size_t binhip_rdn_backward_workspace_bytes(int N, int H, int W, int n_inputs, int nterms, const BinRdnShape* shape) {
    if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return 0;
    if (n_inputs != 2 && n_inputs != 3 && n_inputs != 5) return 0;
    Shp sh;
    if (!resolve_shape(shape, &sh)) return 0;
    return make_bws(N, H, W, n_inputs, nterms, sh).total_bytes;
}

int binhip_rdn_backward_workspace_layout(int N, int H, int W, int n_inputs, int nterms, const BinRdnShape* shape, int64_t* out,
                                         int n_out) {
    if (!out || n_out < BINHIP_RDN_BWD_LAYOUT_WORDS) return BINHIP_E_ARG;
    Shp sh;
    if (!resolve_shape(shape, &sh) || binhip_rdn_backward_workspace_bytes(N, H, W, n_inputs, nterms, shape) == 0)
        return BINHIP_E_SHAPE;
    const Bws b = make_bws(N, H, W, n_inputs, nterms, sh);
    const int64_t v[BINHIP_RDN_BWD_LAYOUT_WORDS] = {b.P, b.PF, b.gx0_chunks, b.gout, b.s_gout, b.gu, b.s_gu, b.guu, b.s_guu,
                                                     b.gg1, b.s_g, b.gg0, b.s_g, b.gf1, b.s_g, b.gy, b.s_gy, b.gcat, b.s_gcat,
                                                     b.gcat2, b.s_gcat, b.gx0, b.s_gx0, (int64_t)b.sc_off_bytes};
    for (int i = 0; i < BINHIP_RDN_BWD_LAYOUT_WORDS; ++i) out[i] = v[i];
    return 0;
}

int binhip_rdn_backward(const BinRdnBwdPlan* p, const void* saved, size_t saved_bytes, const float* gout,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!p || !saved || !gout || !workspace || !p->zero_bias) return BINHIP_E_ARG;
    const int N = p->N, H = p->H, W = p->W, nin = p->n_inputs, nt = p->nterms;
    if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return BINHIP_E_SHAPE;
    if (nin != 2 && nin != 3 && nin != 5) return BINHIP_E_SHAPE;
    if (nt != 1 && nt != 3) return BINHIP_E_ARG;
    // BINHIP_BWD_SAVED_X3: the forward ran in the hi/lo (nterms = 3) layout but this backward computes single-product
    // (nterms = 1): it reads the HI planes of the saved activations (an fp16 rounding of them; the ReLU masks are the
    // sign of hi, which is the sign of hi + lo) at the offsets of the 3-term layout
    const int nt_saved = (p->reserved & BINHIP_BWD_SAVED_X3) ? 3 : nt;
    if (nt_saved == 3 && nt == 3 && (p->reserved & BINHIP_BWD_SAVED_X3)) return BINHIP_E_ARG;
    Shp sh;
    if (!resolve_shape(&p->shape, &sh)) return BINHIP_E_SHAPE;
    if (saved_bytes < binhip_rdn_workspace_bytes(N, H, W, nin, nt_saved, &p->shape)) return BINHIP_E_WORKSPACE;
    const Bws b = make_bws(N, H, W, nin, nt, sh);
    if (workspace_bytes < b.total_bytes) return BINHIP_E_WORKSPACE;
    for (int i = 0; i < sh.L; ++i)
        if (!p->wt_hi[i] || (nt == 3 && !p->wt_lo[i]) || !p->dw[i] || !p->db[i]) return BINHIP_E_ARG;
    const int G0 = sh.G0, G = sh.G, C = sh.C, D = sh.D, c0 = sh.c0, cg = sh.cg, cb = sh.cb, LG = sh.L - 4;

    hipStream_t s = (hipStream_t)stream;
    // Optional second stream (plan->aux_stream): the weight-gradient kernels of a layer depend only on that layer's
    // output gradient and the saved activations, not on the backward-data chain that continues behind it, so they run
    // on the side stream and overlap the chain (both kernel families are latency-bound at training sizes and fit a CU
    // together).  Ordering is by events created, recorded, waited on and destroyed inside this call (no state is kept):
    //   fork  — side stream waits for everything the main stream has queued so far (the gradient a wgrad reads);
    //   b_done[d] — main stream waits, before it overwrites a gradient-concat buffer, for the side-stream weight
    //   gradients of the block that read it two blocks earlier; the main stream joins the side stream before returning.
    hipStream_t sb = p->aux_stream ? (hipStream_t)p->aux_stream : s;
    const bool two = (sb != s);
    auto order = [&](hipStream_t from, hipStream_t to) -> int {      // `to` waits for `from`'s queue up to here
        if (!two) return 0;
        hipEvent_t e;
        hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (r != hipSuccess) return (int)r;
        r = hipEventRecord(e, from);
        if (r == hipSuccess) r = hipStreamWaitEvent(to, e, 0);
        (void)hipEventDestroy(e);                                    // released once the recorded work completes
        return r == hipSuccess ? 0 : (int)r;
    };
    const Ws w = make_ws(N, H, W, nin, nt_saved, sh);
    const int h = H / 2, ww = W / 2;
    const int64_t P = w.P;
    _Float16* sbase = (_Float16*)(((uintptr_t)saved + 255) & ~(uintptr_t)255);
    char* wbytes = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    _Float16* gbase = (_Float16*)wbytes;
    float* sc = (float*)(wbytes + b.sc_off_bytes);
    float* amax_part = sc + 16;
    void* wgws = wbytes + b.wg_off_bytes;
    const float* inv = sc + 1;

    auto SH = [&](int64_t off) { return (const void*)(sbase + off); };
    auto SL = [&](int64_t off, int64_t size) { return nt == 3 ? (const void*)(sbase + off + size) : (const void*)nullptr; };
    auto GH = [&](int64_t off) { return (void*)(gbase + off); };
    auto GL = [&](int64_t off, int64_t size) { return nt == 3 ? (void*)(gbase + off + size) : (void*)nullptr; };

    const int accumulate = (p->reserved & BINHIP_BWD_ACCUMULATE) ? 1 : 0;   // dw/db += instead of =
    int rc;
    if ((rc = binhip_grad_scale(gout, (int64_t)N * 3 * H * W, 16.f, amax_part, sc, stream))) return rc;
    // BINHIP_BWD_FUSED_UPNET: the forward ran UPNet as one 5x5 convolution on 12 sub-pixel channels (+ the border ring)
    const bool fused_up = (p->reserved & BINHIP_BWD_FUSED_UPNET) != 0;
    if (fused_up) {
        if (sh.L + 1 >= BINHIP_RDN_MAX_LAYERS || !p->wt_hi[sh.L] || (nt == 3 && !p->wt_lo[sh.L]) || !p->wt_hi[sh.L + 1] || !p->dw[sh.L] ||
            !p->db[sh.L] || !p->dw[sh.L + 1] || !p->db[sh.L + 1]) return BINHIP_E_ARG;
        // gsub: the pixel-unshuffled, scaled gradient with the ring zeroed — ONE half-resolution chunk, hi at b.gout, lo one plane further
        if ((rc = bh_upnet_gsub(gout, N, H / 2, W / 2, sc, GH(b.gout), GL(b.gout, (int64_t)N * (H / 2) * (W / 2) * 16), p->status, s))) return rc;
    } else
    if ((rc = binhip_nchw_to_planes_scaled(gout, N, 3, H, W, sc, GH(b.gout), GL(b.gout, b.s_gout), p->status, stream))) return rc;

    // weight gradient of forward layer `layer`: X = saved activations, gY = gradient planes.  slot < 0: reduce at once;
    // slot 0..4: keep the partials in region `slot` and queue the reduction for flush_reduces() (one launch per dense block)
    BhWgradReduce pending[BH_WGRAD_BATCH];
    int npending = 0;
    auto wgrad = [&](int layer, int ks, int Hc, int Wc, int cin_chunks, int cin, int cout, int64_t x_off, int64_t x_size,
                     int cpg, int64_t gstride, int64_t g_off, int64_t g_size, int shuffle, int slot = -1) -> int {
        BinConvDesc d;
        d.N = N; d.H = Hc; d.W = Wc; d.ksize = ks; d.cin_chunks = cin_chunks; d.cout = cout; d.cout_pad = 0;
        d.nterms = nt; d.epilogue = 0; d.relu = 0; d.x_cpg = cpg; d.x_group_stride = gstride; d.n_images = 0; d.reserved = 0;
        d.status = p->status;
        if (int rf = order(s, sb)) return rf;                        // its gY (and the scale) are queued on the main stream
        BhWgradReduce r;
        char* region = (char*)wgws + (size_t)(slot < 0 ? 0 : slot) * b.wg_bytes;
        const bool timed = bh_prof_begin(p->profiler, ks, cout, BINHIP_PROF_WGRAD, sb);
        const int rw = bh_wgrad_partials(&d, SH(x_off), SL(x_off, x_size), GH(g_off), GL(g_off, g_size), region, b.wg_bytes,
                                         p->dw[layer], p->db[layer], cin, shuffle, &r, (void*)sb);
        if (timed) bh_prof_end(p->profiler, sb);
        if (rw) return rw;
        if (slot < 0) return bh_wgrad_reduce_batch(&r, 1, inv, layer >= sh.L ? 0 : accumulate, (void*)sb);   // (fused UPNet: written)
        pending[npending++] = r;
        return 0;
    };
    auto flush_reduces = [&]() -> int {
        const int rr = bh_wgrad_reduce_batch(pending, npending, inv, accumulate, (void*)sb);
        npending = 0;
        return rr;
    };
    // data gradient through forward layer `layer`: conv with the transposed/flipped weights
    auto dgrad = [&](int layer, int ks, int Hc, int Wc, int gin_chunks, int gout_ch, int64_t g_off, int64_t g_size,
                     int64_t y_off, int64_t y_size, int64_t r_off, int64_t r_size, int res_chunks, bool acc_inplace,
                     int64_t m_off, int mask_from, int y_cpg, int64_t y_gstride, int y_unshuf = 0) -> int {
        BhConvCall c;
        c.d.N = N; c.d.H = Hc; c.d.W = Wc; c.d.ksize = ks; c.d.cin_chunks = gin_chunks; c.d.cout = gout_ch;
        c.d.cout_pad = binhip_dgrad_rows_pad(ks, gout_ch); c.d.nterms = nt; c.d.epilogue = BINHIP_EPI_PLANES; c.d.relu = 0;
        c.d.x_cpg = 0; c.d.x_group_stride = 0; c.d.n_images = 0; c.d.reserved = 0; c.d.status = p->status;
        c.x_hi = GH(g_off); c.x_lo = GL(g_off, g_size);
        c.w_hi = p->wt_hi[layer]; c.w_lo = p->wt_lo[layer]; c.bias = p->zero_bias;
        c.r_hi = (r_off >= 0) ? GH(r_off) : nullptr; c.r_lo = (r_off >= 0) ? GL(r_off, r_size) : nullptr;
        c.res_chunks = res_chunks;
        c.y_hi = GH(y_off); c.y_lo = GL(y_off, y_size);
        c.r2_hi = acc_inplace ? c.y_hi : nullptr; c.r2_lo = acc_inplace ? c.y_lo : nullptr;
        c.m_hi = (m_off >= 0) ? SH(m_off) : nullptr; c.mask_from = mask_from;
        c.y_cpg = y_cpg; c.y_group_stride = y_gstride;
        c.y_unshuf = y_unshuf;
        c.y_f32 = nullptr;
        c.status = p->status;
        for (int i = 0; i < 5; ++i) c.images[i] = nullptr;
        return bh_launch_conv(c, s);
    };

    // Everything below may have work in flight on the side stream: every exit — error or not — goes through ONE epilogue
    // that destroys the pending per-block events and joins the side stream into the main stream (the caller reuses
    // `workspace` / frees `saved` in main-stream order, also after a failed call).
    hipEvent_t b_done[20] = {};
    auto chain = [&]() -> int {
        if (fused_up) {
            // ---- the fused UPNet (G0 -> 12 sub-pixel channels, 5x5): X = G1, gY = gsub; then the ring's share of both gradients
            const int64_t s_sub = (int64_t)N * h * ww * 16;
            if ((rc = wgrad(sh.L, 5, h, ww, c0, G0, 12, w.g1, w.s_g, 0, 0, b.gout, s_sub, 0))) return rc;
            if ((rc = order(s, sb))) return rc;
            if ((rc = bh_upnet_ring_wgrad(gout, SH(w.g1), SL(w.g1, w.s_g), p->dw[sh.L + 1], p->db[sh.L + 1], N, h, ww, G0, 0, sb))) return rc;
            if ((rc = dgrad(sh.L, 5, h, ww, 1, G0, b.gout, s_sub, b.gg1, b.s_g, -1, 0, 0, false, -1, 0, 0, 0))) return rc;
            if ((rc = bh_upnet_ring_dgrad(gout, (const float*)p->wt_hi[sh.L + 1], sc, GH(b.gg1), GL(b.gg1, b.s_g), p->status, N, h, ww, G0, s)))
                return rc;
        } else {
        // ---- UPNet.2 (64 -> 3 at full res): X = U
        if ((rc = wgrad(LG + 3, 3, H, W, 4, 64, 3, w.u, w.s_u, 0, 0, b.gout, b.s_gout, 0))) return rc;
        // its backward-data writes straight through the inverse PixelShuffle (round 4: y_unshuf; before, a 64-channel
        // full-resolution gradient went to b.gu and a separate layout pass — 856 MB per launch at N = 40 — turned it
        // into the 256 half-resolution channels UPNet.0's backward reads)
        if ((rc = dgrad(LG + 3, 3, H, W, 1, 64, b.gout, b.s_gout, b.guu, b.s_guu, -1, 0, 0, false, -1, 0, 0, 0, 4))) return rc;
        // ---- UPNet.0 (G0 -> 256): X = G1
        if ((rc = wgrad(LG + 2, 3, h, ww, c0, G0, 256, w.g1, w.s_g, 0, 0, b.guu, b.s_guu, 1))) return rc;
        if ((rc = dgrad(LG + 2, 3, h, ww, 16, G0, b.guu, b.s_guu, b.gg1, b.s_g, -1, 0, 0, false, -1, 0, 0, 0))) return rc;
        }
        // ---- GFF.1 (+ f__1 skip): X = G0
        if ((rc = wgrad(LG + 1, 3, h, ww, c0, G0, G0, w.g0, w.s_g, 0, 0, b.gg1, b.s_g, 0))) return rc;
        if ((rc = dgrad(LG + 1, 3, h, ww, c0, G0, b.gg1, b.s_g, b.gg0, b.s_g, -1, 0, 0, false, -1, 0, 0, 0))) return rc;
        // ---- GFF.0 over cat(RDB outputs): X = BLK[1..D][0:c0]; gradient scattered to GY[1..D]
        if ((rc = wgrad(LG, 1, h, ww, D * c0, D * G0, G0, w.blk + (int64_t)cb * P, w.s_blk, c0, (int64_t)cb * P, b.gg0, b.s_g, 0))) return rc;
        if ((rc = dgrad(LG, 1, h, ww, c0, D * G0, b.gg0, b.s_g, b.gy + (int64_t)c0 * P, b.s_gy, -1, 0, 0, false, -1, 0, c0, (int64_t)c0 * P))) return rc;
        // ---- the D residual dense blocks, last to first
        for (int d = D - 1; d >= 0; --d) {
            const int64_t blk = w.blk + (int64_t)d * cb * P;          // saved forward buffer of RDB d
            const int64_t gy = b.gy + (int64_t)(d + 1) * c0 * P;      // grad of RDB d's output
            const int64_t gcat = (d & 1) ? b.gcat2 : b.gcat;          // this block's gradient-concat buffer
            const int L = layer_conv(sh, d, 0);
            // LFF 1x1 (G0 + C G) -> G0 (+x): gcat = W'^T gy (+ gy on the first c0 chunks); ReLU mask of the last conv's output
            if ((rc = wgrad(L + C, 1, h, ww, cb, G0 + C * G, G0, blk, w.s_blk, 0, 0, gy, b.s_gy, 0, C))) return rc;
            // block d+2 used this gcat buffer: its weight gradients (side stream) must have read it before it is refilled.
            // Everything queued on the side stream up to here is older than block d+1's wgrads, so a plain join suffices
            // only every other block would over-serialise; the side stream is in order, so "block d+2 done" = an event
            // recorded there right after block d+2's last wgrad.
            if (two && d + 2 <= D - 1 && b_done[d + 2]) {
                hipError_t r = hipStreamWaitEvent(s, b_done[d + 2], 0);
                (void)hipEventDestroy(b_done[d + 2]);
                b_done[d + 2] = nullptr;
                if (r != hipSuccess) return (int)r;
            }
            if ((rc = dgrad(L + C, 1, h, ww, c0, G0 + C * G, gy, b.s_gy, gcat, b.s_gcat, gy, b.s_gy, c0, false, blk, c0 + cg * (C - 1), 0, 0))) return rc;
            // The C 3x3 convs in gather form (binhip_weights_relayout_rdb_gather): every group of gcat is produced
            // ONCE as L_g + conv(stacked G_c of the later convs) instead of being read-modified-written by each of them.
            for (int c = C - 1; c >= 0; --c) {
                const int64_t gyc = gcat + (int64_t)(c0 + cg * c) * P;     // G_c .. G_{C-1}, contiguous chunks
                if ((rc = wgrad(L + c, 3, h, ww, c0 + cg * c, G0 + G * c, G, blk, w.s_blk, 0, 0, gyc, b.s_gcat, 0, c))) return rc;
                if (c > 0) {
                    // group c = conv c-1's output slot: G_{c-1} = relu'( L_c + sum_{c' >= c} dgrad_c' )
                    const int64_t slot = gcat + (int64_t)(c0 + cg * (c - 1)) * P;
                    if ((rc = dgrad(L + c, 3, h, ww, cg * (C - c), G, gyc, b.s_gcat, slot, b.s_gcat, slot, b.s_gcat, 0, false,
                                    blk + (int64_t)(c0 + cg * (c - 1)) * P, 0, 0, 0))) return rc;
                } else {
                    // group 0: L_0 + all C convs -> grad of the block input = GY[d] (already holds GFF.0's share when d >= 1)
                    if ((rc = dgrad(L, 3, h, ww, cg * C, G0, gyc, b.s_gcat, b.gy + (int64_t)d * c0 * P, b.s_gy, gcat, b.s_gcat, 0,
                                    d >= 1, -1, 0, 0, 0))) return rc;
                }
            }
            if ((rc = flush_reduces())) return rc;                       // the block's C + 1 layers in one reduce launch
            if (two) {
                hipError_t r = hipEventCreateWithFlags(&b_done[d], hipEventDisableTiming);
                if (r == hipSuccess) r = hipEventRecord(b_done[d], sb);
                if (r != hipSuccess) return (int)r;
            }
        }
        // ---- SFENet2: X = F1; gF1 = dgrad + gG1 (the `x += f__1` skip)
        if ((rc = wgrad(1, 3, h, ww, c0, G0, G0, w.f1, w.s_f1, 0, 0, b.gy, b.s_gy, 0))) return rc;
        if ((rc = dgrad(1, 3, h, ww, c0, G0, b.gy, b.s_gy, b.gf1, b.s_g, b.gg1, b.s_g, 0, false, -1, 0, 0, 0))) return rc;
        // ---- SFENet1 5x5: X = X0
        if ((rc = wgrad(0, 5, h, ww, w.kc0, 12 * nin, G0, w.x0, w.s_x0, 0, 0, b.gf1, b.s_g, 0))) return rc;
        bool need_in = false;
        for (int i = 0; i < nin; ++i) need_in = need_in || (p->gin[i] != nullptr);
        if (need_in) {
            if ((rc = dgrad(0, 5, h, ww, c0, 12 * nin, b.gf1, b.s_g, b.gx0, b.s_gx0, -1, 0, 0, false, -1, 0, 0, 0))) return rc;
            if ((rc = binhip_unpack_input_grads(GH(b.gx0), GL(b.gx0, b.s_gx0), gout, sc, nin, N, H, W, p->gin, stream))) return rc;
        }
        return 0;
    };
    rc = chain();
    for (int d = 0; d < 20; ++d)
        if (b_done[d]) { (void)hipEventDestroy(b_done[d]); b_done[d] = nullptr; }
    const int rj = order(sb, s);
    return rc ? rc : rj;
}

}  // extern "C"
