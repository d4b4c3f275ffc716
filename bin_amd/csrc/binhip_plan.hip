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
#include "binhip_internal.h"

namespace {

struct Ws {
    int64_t P, PF;        // plane elems at half / full res
    int kc0;
    int64_t x0, f1, blk, g0, g1, u, total;   // element offsets of the hi part
    int64_t s_x0, s_f1, s_blk, s_g, s_u;     // sizes (elements) of each tensor's hi part
    int nt;
};

Ws make_ws(int N, int H, int W, int nin, int nt) {
    Ws w;
    const int h = H / 2, ww = W / 2;
    w.P = (int64_t)N * h * ww * 16;
    w.PF = (int64_t)N * H * W * 16;
    w.kc0 = (12 * nin + 15) / 16;
    w.nt = nt;
    const int mul = (nt == 3) ? 2 : 1;
    w.s_x0 = w.kc0 * w.P; w.s_f1 = 6 * w.P; w.s_blk = (int64_t)13 * 14 * w.P; w.s_g = 6 * w.P; w.s_u = 4 * w.PF;
    int64_t o = 0;
    w.x0 = o; o += mul * w.s_x0;
    w.f1 = o; o += mul * w.s_f1;
    w.blk = o; o += mul * w.s_blk;
    w.g0 = o; o += mul * w.s_g;
    w.g1 = o; o += mul * w.s_g;
    w.u = o; o += mul * w.s_u;
    w.total = o;
    return w;
}

}  // namespace

extern "C" {

size_t binhip_rdn_workspace_bytes(int N, int H, int W, int n_inputs, int nterms) {
    if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return 0;
    if (n_inputs != 2 && n_inputs != 3 && n_inputs != 5) return 0;
    return (size_t)make_ws(N, H, W, n_inputs, nterms).total * 2 + 256;
}

int binhip_rdn_forward(const BinRdnPlan* p, const float* const* inputs, float* out, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (!p || !inputs || !out || !workspace) return BINHIP_E_ARG;
    const int N = p->N, H = p->H, W = p->W, nin = p->n_inputs, nt = p->nterms;
    if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return BINHIP_E_SHAPE;
    if (nin != 2 && nin != 3 && nin != 5) return BINHIP_E_SHAPE;
    if (nt != 1 && nt != 3) return BINHIP_E_ARG;
    const size_t need = binhip_rdn_workspace_bytes(N, H, W, nin, nt);
    if (workspace_bytes < need) return BINHIP_E_WORKSPACE;
    for (int i = 0; i < nin; ++i) if (!inputs[i]) return BINHIP_E_ARG;
    for (int i = 0; i < BINHIP_RDN_LAYERS; ++i)
        if (!p->w_hi[i] || !p->bias[i] || (nt == 3 && !p->w_lo[i])) return BINHIP_E_ARG;

    hipStream_t s = (hipStream_t)stream;
    const Ws w = make_ws(N, H, W, nin, nt);
    const int h = H / 2, ww = W / 2;
    // align the workspace base to 256 B
    _Float16* base = (_Float16*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    auto HI = [&](int64_t off) { return (void*)(base + off); };
    auto LO = [&](int64_t off, int64_t size) { return nt == 3 ? (void*)(base + off + size) : (void*)nullptr; };

    int rc = binhip_pack_inputs(inputs, nin, N, H, W, HI(w.x0), LO(w.x0, w.s_x0), stream);
    if (rc) return rc;

    auto conv = [&](int layer, int ks, int cin_chunks, int cout, int cout_pad, int epi, int relu, int Hc, int Wc,
                    int64_t x_off, int64_t x_size, int cpg, int64_t gstride,
                    int64_t y_off, int64_t y_size, int64_t r_off, int64_t r_size) -> int {
        BhConvCall c;
        c.d.N = N; c.d.H = Hc; c.d.W = Wc; c.d.ksize = ks; c.d.cin_chunks = cin_chunks; c.d.cout = cout;
        c.d.cout_pad = cout_pad; c.d.nterms = nt; c.d.epilogue = epi; c.d.relu = relu;
        c.d.x_cpg = cpg; c.d.x_group_stride = gstride; c.d.n_images = 0; c.d.reserved = 0;
        c.x_hi = HI(x_off); c.x_lo = LO(x_off, x_size);
        c.w_hi = p->w_hi[layer]; c.w_lo = p->w_lo[layer]; c.bias = p->bias[layer];
        c.r_hi = (r_off >= 0) ? HI(r_off) : nullptr;
        c.r_lo = (r_off >= 0) ? LO(r_off, r_size) : nullptr;
        c.y_hi = (y_off >= 0) ? HI(y_off) : nullptr;
        c.y_lo = (y_off >= 0) ? LO(y_off, y_size) : nullptr;
        c.y_f32 = nullptr;
        for (int i = 0; i < 5; ++i) c.images[i] = nullptr;
        if (epi == BINHIP_EPI_FINAL) {
            c.y_f32 = out;
            c.d.n_images = nin;
            for (int i = 0; i < nin; ++i) c.images[i] = inputs[i];
        }
        return bh_launch_conv(c, s);
    };
    const int P_ = BINHIP_EPI_PLANES;
    const int64_t P = w.P;
    // SFENet1 5x5 (RDN.py:187/245/299) and SFENet2 3x3 (:188)
    if ((rc = conv(0, 5, w.kc0, 96, 96, P_, 0, h, ww, w.x0, w.s_x0, 0, 0, w.f1, w.s_f1, -1, 0))) return rc;
    if ((rc = conv(1, 3, 6, 96, 96, P_, 0, h, ww, w.f1, w.s_f1, 0, 0, w.blk, w.s_blk, -1, 0))) return rc;
    // 12 residual dense blocks (RDN.py:149-165)
    for (int d = 0; d < 12; ++d) {
        const int64_t b = w.blk + (int64_t)d * 14 * P;
        for (int c = 0; c < 4; ++c) {
            if ((rc = conv(2 + 5 * d + c, 3, 6 + 2 * c, 32, 32, P_, 1, h, ww, b, w.s_blk, 0, 0,
                           b + (int64_t)(6 + 2 * c) * P, w.s_blk, -1, 0))) return rc;
        }
        if ((rc = conv(2 + 5 * d + 4, 1, 14, 96, 96, P_, 0, h, ww, b, w.s_blk, 0, 0, b + 14 * P, w.s_blk, b, w.s_blk)))
            return rc;
    }
    // GFF.0 1x1 over cat(RDBs_out) (RDN.py:199, 218): 12 groups of 6 chunks, one per dense block
    if ((rc = conv(62, 1, 72, 96, 96, P_, 0, h, ww, w.blk + 14 * P, w.s_blk, 6, 14 * P, w.g0, w.s_g, -1, 0))) return rc;
    // GFF.1 3x3, x += f__1 (RDN.py:200, 219)
    if ((rc = conv(63, 3, 6, 96, 96, P_, 0, h, ww, w.g0, w.s_g, 0, 0, w.g1, w.s_g, w.f1, w.s_f1))) return rc;
    // UPNet.0 3x3 96->256 + PixelShuffle(2) (RDN.py:205-206)
    if ((rc = conv(64, 3, 6, 256, 256, BINHIP_EPI_SHUFFLE, 0, h, ww, w.g1, w.s_g, 0, 0, w.u, w.s_u, -1, 0))) return rc;
    // UPNet.2 3x3 64->3 + mean(inputs) (RDN.py:207, 221/279/333)
    if ((rc = conv(65, 3, 4, 3, 32, BINHIP_EPI_FINAL, 0, H, W, w.u, w.s_u, 0, 0, -1, 0, -1, 0))) return rc;
    return 0;
}

}  // extern "C"
