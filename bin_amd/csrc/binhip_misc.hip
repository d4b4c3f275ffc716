// binhip_misc.hip — the HBM-bound glue kernels of the bin_stage4 path:
//   * layout: fp32 NCHW <-> fp16 chunk planes; pixel_reshuffle(cat(frames),2) (RDN.py:107-132)
//   * ConvLSTMCell.forward (RDN.py:50-95) as ONE fused kernel (conv 6->12 + gates), fp32
//   * CharbonnierLoss (loss.py:137-141) forward (deterministic two-pass) and backward
// All are one-pass streaming kernels: every input byte is read once, every output byte written once.
#include "binhip_internal.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// ---- fp32 NCHW -> chunk planes -----------------------------------------------------------------
// one thread = one 16-byte slot (8 channels of one pixel)
__global__ void nchw_to_planes_kernel(const float* __restrict__ x, int N, int C, int H, int W,
                                      _Float16* __restrict__ y_hi, _Float16* __restrict__ y_lo,
                                      const float* __restrict__ scale, unsigned* __restrict__ flags) {
    const float sc = scale ? scale[0] : 1.f;
    const long long HW = (long long)H * W;
    const long long total = (long long)bh_chunks_dev(C) * N * HW * 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int s = (int)(t & 1);
    long long u = t >> 1;
    const long long pix = u % HW; u /= HW;
    const int n = (int)(u % N);
    const int ch = (int)(u / N);
    half8 hv, lv;
    unsigned sat = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = ch * 16 + s * 8 + e;
        const float v = (c < C) ? x[((long long)n * C + c) * HW + pix] * sc : 0.f;
        hv[e] = split_hi(v, sat);
        lv[e] = split_lo(v, hv[e]);
    }
    *reinterpret_cast<half8*>(y_hi + t * 8) = hv;
    if (y_lo) *reinterpret_cast<half8*>(y_lo + t * 8) = lv;
    if (sat != 0 && flags) atomicOr(flags, BINHIP_FLAG_SATURATED);
}

// one thread = one (n, c, pixel) output element; reads are 2-byte gathers (test/boundary glue only)
__global__ void planes_to_nchw_kernel(const _Float16* __restrict__ x_hi, const _Float16* __restrict__ x_lo,
                                      int N, int C, int H, int W, float* __restrict__ y) {
    const long long HW = (long long)H * W;
    const long long total = (long long)N * C * HW;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const long long pix = t % HW;
    const int c = (int)((t / HW) % C);
    const int n = (int)(t / (HW * C));
    const long long o = (((long long)(c >> 4) * N + n) * HW + pix) * 16 + (c & 15);
    float v = (float)x_hi[o];
    if (x_lo) v += (float)x_lo[o];
    y[t] = v;
}

// ---- exact fp32 space-to-depth (the standalone pixel_reshuffle of the reference's API, RDN.py:107-132) --------------
__global__ void pixel_unshuffle_f32_kernel(const float* __restrict__ x, int N, int C, int H, int W, int r,
                                           float* __restrict__ y) {
    const int h = H / r, w = W / r;
    const long long total = (long long)N * C * H * W;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int ox = (int)(t % w);
    const int oy = (int)((t / w) % h);
    const int oc = (int)((t / ((long long)w * h)) % (C * r * r));
    const int n = (int)(t / ((long long)w * h * C * r * r));
    const int c = oc / (r * r), i = (oc / r) % r, j = oc % r;
    y[t] = x[(((long long)n * C + c) * H + (oy * r + i)) * W + (ox * r + j)];
}

// ---- K1: pixel_reshuffle(cat(images), 2) -> chunk planes at half resolution ---------------------
struct PackArgs {
    const float* img[5];
    int nimg, N, H, W;   // full-res H, W
};
__global__ void pack_inputs_kernel(PackArgs a, _Float16* __restrict__ y_hi, _Float16* __restrict__ y_lo,
                                   unsigned* __restrict__ flags) {
    const int h = a.H / 2, w = a.W / 2;
    const long long hw = (long long)h * w;
    const int C = 12 * a.nimg;
    const int nch = (C + 15) / 16;
    const long long total = (long long)nch * a.N * hw * 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int s = (int)(t & 1);
    long long u = t >> 1;
    const long long pix = u % hw; u /= hw;
    const int n = (int)(u % a.N);
    const int ch = (int)(u / a.N);
    const int y = (int)(pix / w), x = (int)(pix % w);
    half8 hv, lv;
    unsigned sat = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = ch * 16 + s * 8 + e;     // = 4*cc + 2*i + j   (RDN.py:128-132)
        float v = 0.f;
        if (c < C) {
            const int cc = c >> 2, i = (c >> 1) & 1, j = c & 1;
            const int im = cc / 3, rgb = cc - im * 3;
            v = a.img[im][(((long long)n * 3 + rgb) * a.H + (2 * y + i)) * a.W + (2 * x + j)];
        }
        hv[e] = split_hi(v, sat);
        lv[e] = split_lo(v, hv[e]);
    }
    *reinterpret_cast<half8*>(y_hi + t * 8) = hv;
    if (y_lo) *reinterpret_cast<half8*>(y_lo + t * 8) = lv;
    if (sat != 0 && flags) atomicOr(flags, BINHIP_FLAG_SATURATED);
}

// ---- harness glue: the per-frame host work of test.py moved onto the device -----------------
// u8 HWC BGR image -> fp32 CHW RGB in [0,1] (read_image, test.py:44-56) + ReplicationPad2d (test.py:348-371)
__global__ void u8_to_frame_kernel(const unsigned char* __restrict__ img, int H, int W, int pl, int pt, int Hp, int Wp,
                                   float* __restrict__ out) {
    const long long total = (long long)3 * Hp * Wp;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int x = (int)(t % Wp), y = (int)((t / Wp) % Hp), c = (int)(t / ((long long)Wp * Hp));
    int sy = y - pt, sx = x - pl;
    sy = sy < 0 ? 0 : (sy >= H ? H - 1 : sy);
    sx = sx < 0 ? 0 : (sx >= W ? W - 1 : sx);
    out[t] = (float)img[((long long)sy * W + sx) * 3 + (2 - c)] / 255.f;
}
// fp32 CHW RGB -> cropped u8 HWC BGR: clamp [0,1], x255, round-half-even (utils/util.py:113-137), crop (test.py:394-402)
__global__ void frame_to_u8_kernel(const float* __restrict__ x, int Hp, int Wp, int top, int left, int H, int W,
                                   unsigned char* __restrict__ out) {
    const long long total = (long long)H * W * 3;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % 3), xx = (int)((t / 3) % W), yy = (int)(t / (3LL * W));
    float v = x[((long long)(2 - c) * Hp + (yy + top)) * Wp + (xx + left)];
    v = fminf(fmaxf(v, 0.f), 1.f);
    out[t] = (unsigned char)rintf(v * 255.0f);
}

// ---- ConvLSTM cell ------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }
// The per-pixel gate arithmetic of the cell (RDN.py:74-92) and of its backward, ONE definition for the one-pixel and the
// four-pixel kernels, with the multiply-adds spelled out (fmaf / __fmul_rn) so that both compile to the same roundings.
__device__ __forceinline__ void lstm_point_fwd(float gi, float gj, float gf, float go, float cprev, float fb, float& c1, float& h1) {
    c1 = fmaf(cprev, sigmoidf_(gf + fb), __fmul_rn(sigmoidf_(gi), tanhf(gj)));
    h1 = __fmul_rn(tanhf(c1), sigmoidf_(go));
}
__device__ __forceinline__ void lstm_point_bwd(float gi, float gj, float gf, float go, float cprev, float fb, float ghv, float gcv,
                                               float& di, float& dj, float& df, float& dob, float& dcp) {
    const float si = sigmoidf_(gi), tj = tanhf(gj), sf = sigmoidf_(gf + fb), so = sigmoidf_(go);
    const float c1 = fmaf(cprev, sf, __fmul_rn(si, tj));
    const float tc = tanhf(c1);
    const float gct = fmaf(__fmul_rn(ghv, so), fmaf(-tc, tc, 1.f), gcv);
    di = __fmul_rn(__fmul_rn(gct, tj), __fmul_rn(si, 1.f - si));
    dj = __fmul_rn(__fmul_rn(gct, si), fmaf(-tj, tj, 1.f));
    df = __fmul_rn(__fmul_rn(gct, cprev), __fmul_rn(sf, 1.f - sf));
    dob = __fmul_rn(__fmul_rn(ghv, tc), __fmul_rn(so, 1.f - so));
    dcp = __fmul_rn(gct, sf);
}

// one thread = one pixel; 3x3x6 neighbourhood from global (L1/L2 resident), weights via scalar loads
__global__ void __launch_bounds__(256)
convlstm_kernel(const float* __restrict__ x, const float* __restrict__ cp, const float* __restrict__ hp,
                const float* __restrict__ w, const float* __restrict__ b, float fb, int N, int H, int W,
                float* __restrict__ cn, float* __restrict__ hn) {
    __shared__ float ws[12 * 6 * 9 + 12];
    for (int i = threadIdx.x; i < 12 * 54 + 12; i += blockDim.x) ws[i] = (i < 648) ? w[i] : b[i - 648];
    __syncthreads();
    const long long HW = (long long)H * W;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * HW) return;
    const int n = (int)(t / HW);
    const long long pix = t - (long long)n * HW;
    const int y = (int)(pix / W), xx = (int)(pix - (long long)y * W);
    float g[12];
#pragma unroll
    for (int o = 0; o < 12; ++o) g[o] = ws[648 + o];
    const int nin = hp ? 6 : 3;
    for (int ci = 0; ci < nin; ++ci) {
        const float* src = (ci < 3) ? (x + ((long long)n * 3 + ci) * HW) : (hp + ((long long)n * 3 + (ci - 3)) * HW);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xq = xx + dx - 1;
                if (xq < 0 || xq >= W) continue;
                const float v = src[(long long)yy * W + xq];
#pragma unroll
                for (int o = 0; o < 12; ++o) g[o] = fmaf(ws[(o * 6 + ci) * 9 + dy * 3 + dx], v, g[o]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {           // i = g[0:3], j = g[3:6], f = g[6:9], o = g[9:12]  (RDN.py:79)
        const long long o = ((long long)n * 3 + k) * HW + pix;
        const float cprev = cp ? cp[o] : 0.f;
        float c1, h1;
        lstm_point_fwd(g[k], g[3 + k], g[6 + k], g[9 + k], cprev, fb, c1, h1);
        if (cn) cn[o] = c1;
        hn[o] = h1;
    }
}

// ---- the same cell, FOUR pixels per thread (round 5; W % 4 == 0 and 16-byte aligned planes, else the kernels above/below) ----
// The one-pixel kernel issues 27 (54 with state) 4-byte loads and 324 (648) broadcast LDS reads per pixel and, with state, fetches
// c_prev between its stores: loads and stores share vmcnt on gfx9, so each of those loads waits for the stores before it
// (36.9 us per 768x1344 cell = 1.0 TB/s of a 37 MB pass).  Here a thread owns 4 consecutive pixels of a row: one 16-byte load + two
// edge loads per (channel, row) serve 3 taps x 4 pixels, the 12 gate weights of a tap are three ds_read_b128 (layout
// [channel][tap][gate]) reused by 4 pixels, and EVERY load of the epilogue (c_prev; in the backward also g_h, g_c) is issued
// and pinned before the first store.  Each pixel's fmaf chain runs in the order of the one-pixel kernel (bias, then channel,
// dy, dx; an out-of-image tap contributes fma(w, 0, g) = g), so the two kernels agree bit for bit.
__device__ __forceinline__ void convlstm_load_weights4(float* ws, const float* __restrict__ w, const float* __restrict__ b) {
    for (int i = threadIdx.x; i < 12 * 54 + 12; i += blockDim.x) {
        if (i < 648) {
            const int o = i / 54, r = i % 54;              // w[(o * 6 + ci) * 9 + tap]  ->  ws[(ci * 9 + tap) * 12 + o]
            ws[r * 12 + o] = w[i];
        } else {
            ws[i] = b[i - 648];
        }
    }
}
__device__ __forceinline__ void convlstm_gates4(const float* __restrict__ x, const float* __restrict__ hp, const float* ws,
                                                int n, int y, int x0, int H, int W, long long HW, float (&g)[12][4]) {
#pragma unroll
    for (int o = 0; o < 12; ++o)
#pragma unroll
        for (int p = 0; p < 4; ++p) g[o][p] = ws[648 + o];
    const int nin = hp ? 6 : 3;
    for (int ci = 0; ci < nin; ++ci) {
        const float* src = (ci < 3) ? (x + ((long long)n * 3 + ci) * HW) : (hp + ((long long)n * 3 + (ci - 3)) * HW);
        float v[3][6];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {                   // the three rows' loads first, then their 9 x 12 x 4 fmas
            const int yy = y + dy - 1;
            const bool ok = yy >= 0 && yy < H;
            const float* r = src + (long long)(ok ? yy : y) * W + x0;
            const float4 c = *reinterpret_cast<const float4*>(r);
            const float l = (x0 > 0) ? r[-1] : 0.f, rt = (x0 + 4 < W) ? r[4] : 0.f;
            v[dy][0] = ok ? l : 0.f; v[dy][1] = ok ? c.x : 0.f; v[dy][2] = ok ? c.y : 0.f;
            v[dy][3] = ok ? c.z : 0.f; v[dy][4] = ok ? c.w : 0.f; v[dy][5] = ok ? rt : 0.f;
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float4* wq = reinterpret_cast<const float4*>(ws + ((ci * 9) + dy * 3 + dx) * 12);
                const float4 w0 = wq[0], w1 = wq[1], w2 = wq[2];
                const float wv[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
#pragma unroll
                for (int o = 0; o < 12; ++o)
#pragma unroll
                    for (int p = 0; p < 4; ++p) g[o][p] = fmaf(wv[o], v[dy][p + dx], g[o][p]);
            }
    }
}
__device__ __forceinline__ float4 ld4_or_zero(const float* p, long long o) {
    return p ? *reinterpret_cast<const float4*>(p + o) : float4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void pin4(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

__global__ void __launch_bounds__(256)
convlstm4_kernel(const float* __restrict__ x, const float* __restrict__ cp, const float* __restrict__ hp,
                 const float* __restrict__ w, const float* __restrict__ b, float fb, int N, int H, int W,
                 float* __restrict__ cn, float* __restrict__ hn) {
    __shared__ __attribute__((aligned(16))) float ws[12 * 6 * 9 + 12];
    convlstm_load_weights4(ws, w, b);
    __syncthreads();
    const long long HW = (long long)H * W;
    const int W4 = W >> 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * H * W4) return;
    const int xq = (int)(t % W4), y = (int)((t / W4) % H), n = (int)(t / ((long long)W4 * H));
    const int x0 = xq * 4;
    float g[12][4];
    convlstm_gates4(x, hp, ws, n, y, x0, H, W, HW, g);
    const long long o0 = ((long long)n * 3) * HW + (long long)y * W + x0;
    float4 cprev[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) cprev[k] = ld4_or_zero(cp, o0 + k * HW);
#pragma unroll
    for (int k = 0; k < 3; ++k) pin4(cprev[k]);
    __builtin_amdgcn_sched_barrier(0);
    float4 c1v[3], h1v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {           // i = g[0:3], j = g[3:6], f = g[6:9], o = g[9:12]  (RDN.py:79)
        float c1[4], h1[4];
        const float cpv[4] = {cprev[k].x, cprev[k].y, cprev[k].z, cprev[k].w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            lstm_point_fwd(g[k][p], g[3 + k][p], g[6 + k][p], g[9 + k][p], cpv[p], fb, c1[p], h1[p]);
        }
        c1v[k] = float4{c1[0], c1[1], c1[2], c1[3]};
        h1v[k] = float4{h1[0], h1[1], h1[2], h1[3]};
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (cn) *reinterpret_cast<float4*>(cn + o0 + k * HW) = c1v[k];
        *reinterpret_cast<float4*>(hn + o0 + k * HW) = h1v[k];
    }
}

__global__ void __launch_bounds__(256)
convlstm4_bwd_gates_kernel(const float* __restrict__ x, const float* __restrict__ cp, const float* __restrict__ hp,
                           const float* __restrict__ w, const float* __restrict__ b, float fb, int N, int H, int W,
                           const float* __restrict__ gh, const float* __restrict__ gc, float* __restrict__ dgates,
                           float* __restrict__ gcp) {
    __shared__ __attribute__((aligned(16))) float ws[12 * 6 * 9 + 12];
    convlstm_load_weights4(ws, w, b);
    __syncthreads();
    const long long HW = (long long)H * W;
    const int W4 = W >> 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * H * W4) return;
    const int xq = (int)(t % W4), y = (int)((t / W4) % H), n = (int)(t / ((long long)W4 * H));
    const int x0 = xq * 4;
    float g[12][4];
    convlstm_gates4(x, hp, ws, n, y, x0, H, W, HW, g);
    const long long pix = (long long)y * W + x0;
    const long long o0 = ((long long)n * 3) * HW + pix;
    float4 cprev[3], ghv[3], gcv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        cprev[k] = ld4_or_zero(cp, o0 + k * HW);
        ghv[k] = ld4_or_zero(gh, o0 + k * HW);
        gcv[k] = ld4_or_zero(gc, o0 + k * HW);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { pin4(cprev[k]); pin4(ghv[k]); pin4(gcv[k]); }
    __builtin_amdgcn_sched_barrier(0);
    const long long d0 = ((long long)n * 12) * HW + pix;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float cpv[4] = {cprev[k].x, cprev[k].y, cprev[k].z, cprev[k].w};
        const float gh4[4] = {ghv[k].x, ghv[k].y, ghv[k].z, ghv[k].w};
        const float gc4[4] = {gcv[k].x, gcv[k].y, gcv[k].z, gcv[k].w};
        float di[4], dj[4], df[4], dob[4], dcp[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            lstm_point_bwd(g[k][p], g[3 + k][p], g[6 + k][p], g[9 + k][p], cpv[p], fb, gh4[p], gc4[p], di[p], dj[p], df[p], dob[p], dcp[p]);
        }
        *reinterpret_cast<float4*>(dgates + d0 + (long long)(k) * HW) = float4{di[0], di[1], di[2], di[3]};
        *reinterpret_cast<float4*>(dgates + d0 + (long long)(3 + k) * HW) = float4{dj[0], dj[1], dj[2], dj[3]};
        *reinterpret_cast<float4*>(dgates + d0 + (long long)(6 + k) * HW) = float4{df[0], df[1], df[2], df[3]};
        *reinterpret_cast<float4*>(dgates + d0 + (long long)(9 + k) * HW) = float4{dob[0], dob[1], dob[2], dob[3]};
        if (gcp) *reinterpret_cast<float4*>(gcp + o0 + k * HW) = float4{dcp[0], dcp[1], dcp[2], dcp[3]};
    }
}

// ---- ConvLSTM gate arithmetic for cells OTHER than the (3, 3) cell of the live path (reference RDN.py:74-82, any
// input_size / hidden_size: RDN.py:14-24).  The gates conv of such a cell runs on the general convolution kernels; these two
// elementwise kernels are the rest: gates [N, 4h, H, W] (i, j, f, o) -> c', h' and its backward.
__global__ void __launch_bounds__(256)
lstm_gates_fwd_kernel(const float* __restrict__ gates, const float* __restrict__ cp, float fb, int hid, long long HW,
                      long long total, float* __restrict__ cn, float* __restrict__ hn) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // over [N, hid, H, W]
    if (t >= total) return;
    const long long chw = (long long)hid * HW;
    const long long n = t / chw, r = t - n * chw;
    const float* g = gates + n * 4 * chw + r;
    const float cprev = cp ? cp[t] : 0.f;
    const float c1 = cprev * sigmoidf_(g[2 * chw] + fb) + sigmoidf_(g[0]) * tanhf(g[chw]);
    cn[t] = c1;
    hn[t] = tanhf(c1) * sigmoidf_(g[3 * chw]);
}
__global__ void __launch_bounds__(256)
lstm_gates_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ cp, const float* __restrict__ gh,
                      const float* __restrict__ gc, float fb, int hid, long long HW, long long total,
                      float* __restrict__ dg, float* __restrict__ gcp) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const long long chw = (long long)hid * HW;
    const long long n = t / chw, r = t - n * chw;
    const float* g = gates + n * 4 * chw + r;
    float* d = dg + n * 4 * chw + r;
    const float cprev = cp ? cp[t] : 0.f;
    const float si = sigmoidf_(g[0]), tj = tanhf(g[chw]), sf = sigmoidf_(g[2 * chw] + fb), so = sigmoidf_(g[3 * chw]);
    const float c1 = cprev * sf + si * tj;
    const float tc = tanhf(c1);
    const float ghv = gh ? gh[t] : 0.f;
    const float dc = (gc ? gc[t] : 0.f) + ghv * so * (1.f - tc * tc);
    d[0] = dc * tj * si * (1.f - si);
    d[chw] = dc * si * (1.f - tj * tj);
    d[2 * chw] = dc * cprev * sf * (1.f - sf);
    d[3 * chw] = ghv * tc * so * (1.f - so);
    if (gcp) gcp[t] = dc * sf;
}

// ---- pixel criteria (bin_model.py:52-60): Charbonnier mean (loss.py:137-141), L1 sum, L2 sum --------------------
#define CHARB_BLOCKS 1024
template <int KIND>
__device__ __forceinline__ float crit_term(float d, float eps) {
    if constexpr (KIND == BINHIP_LOSS_CHARBONNIER) return sqrtf(d * d + eps);
    else if constexpr (KIND == BINHIP_LOSS_L1_SUM) return fabsf(d);
    else return d * d;
}
template <int KIND>
__device__ __forceinline__ float crit_grad(float d, float eps) {
    if constexpr (KIND == BINHIP_LOSS_CHARBONNIER) return d / sqrtf(d * d + eps);
    else if constexpr (KIND == BINHIP_LOSS_L1_SUM) return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);      // torch: sign(0) = 0
    else return 2.f * d;
}
template <int KIND>
__global__ void __launch_bounds__(256)
charb_partial_kernel(const float* __restrict__ x, const float* __restrict__ y, long long n, float eps,
                     float* __restrict__ partials) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        acc += crit_term<KIND>(x[i] - y[i], eps);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
// `denom`: numel for the mean criterion, 1 for the sum criteria
__global__ void __launch_bounds__(256)
charb_final_kernel(const float* __restrict__ partials, int nb, double denom, float* __restrict__ loss) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += (double)partials[i];
    __shared__ double sm[256];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(sm[0] / denom);
}
template <int KIND>
__global__ void __launch_bounds__(256)
charb_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, long long n, float eps, float inv_denom,
                 const float* __restrict__ gl, float* __restrict__ gx, float* __restrict__ gy) {
    const float s = gl[0] * inv_denom;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float g = s * crit_grad<KIND>(x[i] - y[i], eps);
        if (gx) gx[i] = g;
        if (gy) gy[i] = -g;
    }
}

// ---- all terms of bin_model.get_loss at once (binhip_multi_loss_fwd / _bwd, include/binhip.h) -------------------------
// blockIdx.y = term; per term the SAME partial sums as charb_partial_kernel with the same grid -> the same bits
template <int KIND>
__global__ void __launch_bounds__(256)
multi_loss_partial_kernel(const BinLossTerms t, long long n, float eps, float* __restrict__ partials) {
    const float* __restrict__ x = t.x[blockIdx.y];
    const float* __restrict__ y = t.y[blockIdx.y];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        acc += crit_term<KIND>(x[i] - y[i], eps);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[(long long)blockIdx.y * gridDim.x + blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
// one block: every term's final reduction (as charb_final_kernel), then the left-to-right fp32 sum of the terms / T
__global__ void __launch_bounds__(256)
multi_loss_final_kernel(const float* __restrict__ partials, int nb, int nterms, double denom, float* __restrict__ terms,
                        float* __restrict__ loss) {
    __shared__ double sm[256];
    for (int t = 0; t < nterms; ++t) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < nb; i += 256) acc += (double)partials[(long long)t * nb + i];
        sm[threadIdx.x] = acc;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) terms[t] = (float)(sm[0] / denom);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float s = terms[0];
        for (int t = 1; t < nterms; ++t) s = s + terms[t];
        // (ATen divides a device tensor by a host scalar as a product with the fp32 reciprocal; the same here, so that the fused
        //  loss and the per-term path — torch ops over binhip_pixel_loss_fwd results — agree bit for bit)
        loss[0] = s * (1.0f / (float)nterms);
    }
}
// blockIdx.y = output tensor k: out[k] = s * (sign_a * crit'(x_a - y_a) [+ sign_b * crit'(x_b - y_b)])
template <int KIND>
__global__ void __launch_bounds__(256)
multi_loss_bwd_kernel(const BinLossTerms t, const BinLossGrads g, long long n, float eps, float scale, float inv_terms,
                      const float* __restrict__ gl) {
    const int k = blockIdx.y;
    const int ta = g.term_a[k], tb = g.term_b[k];
    const float s = (gl[0] * inv_terms) * scale;       // d loss / d term = gloss * (1 / T) (as autograd's division node), then / numel
    const float sa = g.sign_a[k], sb = g.sign_b[k];
    const float* __restrict__ xa = t.x[ta];
    const float* __restrict__ ya = t.y[ta];
    const float* __restrict__ xb = tb >= 0 ? t.x[tb] : nullptr;
    const float* __restrict__ yb = tb >= 0 ? t.y[tb] : nullptr;
    float* __restrict__ out = g.out[k];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        // each term's gradient is rounded on its own (s * crit') and the two are then added: what autograd's accumulation of
        // the per-term gradients computes
        float v = sa * (s * crit_grad<KIND>(xa[i] - ya[i], eps));
        if (xb) v += sb * (s * crit_grad<KIND>(xb[i] - yb[i], eps));
        out[i] = v;
    }
}

// ---- gradient scaling: scale = 2^floor(log2(target / amax)) so fp16 gradient planes neither overflow nor
// underflow; sc[0] = scale, sc[1] = 1/scale.  Two-pass amax (deterministic).
__global__ void __launch_bounds__(256)
amax_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ partials) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}
__global__ void __launch_bounds__(256)
grad_scale_final_kernel(const float* __restrict__ partials, int nb, float target, float* __restrict__ sc) {
    __shared__ float sm[256];
    float m = 0.f;                                        // max is order-independent: a parallel sweep is exact
    for (int i = threadIdx.x; i < nb; i += 256) m = fmaxf(m, partials[i]);
    sm[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    m = sm[0];
    float s = 1.f;
    if (m > 0.f && isfinite(m)) {
        int e = (int)floorf(log2f(target / m));
        e = e > 40 ? 40 : (e < -40 ? -40 : e);
        s = exp2f((float)e);
    }
    sc[0] = s;
    sc[1] = 1.f / s;
}

// ---- inverse PixelShuffle on chunk planes: [C/16] planes at 2H x 2W -> [4*C/16] planes at H x W, output chunk
// sub*(C/16) + c (the channel order UPNet.0's permuted rows use).  Pure 16-byte slot copy.
__global__ void unshuffle_planes_kernel(const _Float16* __restrict__ x, int N, int H, int W, int nch,
                                        _Float16* __restrict__ y) {
    const long long hw = (long long)H * W;
    const long long total = (long long)4 * nch * N * hw * 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int s = (int)(t & 1);
    long long u = t >> 1;
    const long long pix = u % hw; u /= hw;
    const int n = (int)(u % N); u /= N;
    const int oc = (int)u;                       // output chunk = sub*nch + c
    const int sub = oc / nch, c = oc - sub * nch;
    const int yy = (int)(pix / W), xx = (int)(pix - (long long)yy * W);
    const long long src = ((((long long)c * N + n) * (2 * H) + (2 * yy + (sub >> 1))) * (2 * W) + (2 * xx + (sub & 1))) * 16 + s * 8;
    *reinterpret_cast<half8*>(y + t * 8) = *reinterpret_cast<const half8*>(x + src);
}

// ---- gradients w.r.t. the RDN's input frames: inverse of pack_inputs (pixel-shuffle of the SFENet1 input
// gradient) un-scaled, plus the mean skip path gout / k (RDN.py:221/279/333).
struct UnpackArgs {
    float* out[5];
    int nimg, N, H, W;
};
__global__ void unpack_input_grads_kernel(UnpackArgs a, const _Float16* __restrict__ g_hi, const _Float16* __restrict__ g_lo,
                                          const float* __restrict__ gout, const float* __restrict__ sc) {
    const long long HW = (long long)a.H * a.W;
    const long long total = (long long)a.N * 3 * HW;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const long long pix = t % HW;
    const int rgb = (int)((t / HW) % 3);
    const int n = (int)(t / (3 * HW));
    const int Y = (int)(pix / a.W), X = (int)(pix - (long long)Y * a.W);
    const int h = a.H / 2, w = a.W / 2;
    const float inv = sc ? sc[1] : 1.f;
    const float skip = gout[t] / (float)a.nimg;
    for (int im = 0; im < a.nimg; ++im) {
        if (!a.out[im]) continue;
        float v = skip;
        if (g_hi) {
            const int c = 4 * (im * 3 + rgb) + 2 * (Y & 1) + (X & 1);
            const long long o = ((((long long)(c >> 4) * a.N + n) * h + (Y >> 1)) * w + (X >> 1)) * 16 + (c & 15);
            float g = (float)g_hi[o];
            if (g_lo) g += (float)g_lo[o];
            v += g * inv;
        }
        a.out[im][t] = v;
    }
}

// ---- ConvLSTM backward -----------------------------------------------------------------------------
// pass 1: recompute the gates per pixel, emit dgates [N,12,H,W] (i,j,f,o order) and gc_prev
__global__ void __launch_bounds__(256)
convlstm_bwd_gates_kernel(const float* __restrict__ x, const float* __restrict__ cp, const float* __restrict__ hp,
                          const float* __restrict__ w, const float* __restrict__ b, float fb, int N, int H, int W,
                          const float* __restrict__ gh, const float* __restrict__ gc, float* __restrict__ dgates,
                          float* __restrict__ gcp) {
    __shared__ float ws[12 * 6 * 9 + 12];
    for (int i = threadIdx.x; i < 12 * 54 + 12; i += blockDim.x) ws[i] = (i < 648) ? w[i] : b[i - 648];
    __syncthreads();
    const long long HW = (long long)H * W;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * HW) return;
    const int n = (int)(t / HW);
    const long long pix = t - (long long)n * HW;
    const int y = (int)(pix / W), xx = (int)(pix - (long long)y * W);
    float g[12];
#pragma unroll
    for (int o = 0; o < 12; ++o) g[o] = ws[648 + o];
    const int nin = hp ? 6 : 3;
    for (int ci = 0; ci < nin; ++ci) {
        const float* src = (ci < 3) ? (x + ((long long)n * 3 + ci) * HW) : (hp + ((long long)n * 3 + (ci - 3)) * HW);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xq = xx + dx - 1;
                if (xq < 0 || xq >= W) continue;
                const float v = src[(long long)yy * W + xq];
#pragma unroll
                for (int o = 0; o < 12; ++o) g[o] = fmaf(ws[(o * 6 + ci) * 9 + dy * 3 + dx], v, g[o]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const long long o = ((long long)n * 3 + k) * HW + pix;
        const float cprev = cp ? cp[o] : 0.f;
        const float ghv = gh ? gh[o] : 0.f;
        const float gcv = gc ? gc[o] : 0.f;
        float di, dj, df, dob, dcp;
        lstm_point_bwd(g[k], g[3 + k], g[6 + k], g[9 + k], cprev, fb, ghv, gcv, di, dj, df, dob, dcp);
        const long long d0 = ((long long)n * 12) * HW + pix;
        dgates[d0 + (long long)(k) * HW] = di;
        dgates[d0 + (long long)(3 + k) * HW] = dj;
        dgates[d0 + (long long)(6 + k) * HW] = df;
        dgates[d0 + (long long)(9 + k) * HW] = dob;
        if (gcp) gcp[o] = dcp;
    }
}
// pass 2: dx / dh_prev = conv_transpose(dgates, w)
__global__ void __launch_bounds__(256)
convlstm_bwd_input_kernel(const float* __restrict__ dgates, const float* __restrict__ w, int N, int H, int W,
                          float* __restrict__ gx, float* __restrict__ ghp) {
    __shared__ float ws[648];
    for (int i = threadIdx.x; i < 648; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    const long long HW = (long long)H * W;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * HW) return;
    const int n = (int)(t / HW);
    const long long pix = t - (long long)n * HW;
    const int y = (int)(pix / W), xx = (int)(pix - (long long)y * W);
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y - (dy - 1);
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xq = xx - (dx - 1);
            if (xq < 0 || xq >= W) continue;
            for (int o = 0; o < 12; ++o) {
                const float d = dgates[((long long)n * 12 + o) * HW + (long long)yy * W + xq];
#pragma unroll
                for (int ci = 0; ci < 6; ++ci) acc[ci] = fmaf(ws[(o * 6 + ci) * 9 + dy * 3 + dx], d, acc[ci]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const long long o = ((long long)n * 3 + k) * HW + pix;
        if (gx) gx[o] = acc[k];
        if (ghp) ghp[o] = acc[3 + k];
    }
}
// pass 3: dW / db partials per pixel strip (LDS tiles), then a fixed-order final sum
#define CL_TW 64
#define CL_TH 8
__global__ void __launch_bounds__(256)
convlstm_bwd_weight_kernel(const float* __restrict__ dgates, const float* __restrict__ x, const float* __restrict__ hp,
                           int N, int H, int W, int tiles_x, int tiles_y, float* __restrict__ partials) {
    __shared__ float sd[12][CL_TH][CL_TW];
    __shared__ float sx[6][CL_TH + 2][CL_TW + 2];
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int x0 = tx * CL_TW, y0 = ty * CL_TH;
    const long long HW = (long long)H * W;
    for (int i = threadIdx.x; i < 12 * CL_TH * CL_TW; i += 256) {
        const int o = i / (CL_TH * CL_TW), r = (i / CL_TW) % CL_TH, c = i % CL_TW;
        const int yy = y0 + r, xx = x0 + c;
        sd[o][r][c] = (yy < H && xx < W) ? dgates[((long long)n * 12 + o) * HW + (long long)yy * W + xx] : 0.f;
    }
    // (no previous state — every cell of bin_stage4's two-window schedule, RDN.py:57-68 — means the recurrent half of the gates
    //  conv saw zeros: its 324 weight gradients are exactly zero and neither their inputs nor their sums are formed; round 5)
    const int nci = hp ? 6 : 3;
    for (int i = threadIdx.x; i < nci * (CL_TH + 2) * (CL_TW + 2); i += 256) {
        const int ci = i / ((CL_TH + 2) * (CL_TW + 2)), r = (i / (CL_TW + 2)) % (CL_TH + 2), c = i % (CL_TW + 2);
        const int yy = y0 + r - 1, xx = x0 + c - 1;
        float v = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            if (ci < 3) v = x[((long long)n * 3 + ci) * HW + (long long)yy * W + xx];
            else if (hp) v = hp[((long long)n * 3 + (ci - 3)) * HW + (long long)yy * W + xx];
        }
        sx[ci][r][c] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 660; k += 256) {
        float acc = 0.f;
        if (k < 648) {
            const int o = k / 54, ci = (k / 9) % 6, dy = (k % 9) / 3, dx = k % 3;
            if (ci < nci)
                for (int r = 0; r < CL_TH; ++r)
                    for (int c = 0; c < CL_TW; ++c) acc = fmaf(sd[o][r][c], sx[ci][r + dy][c + dx], acc);
        } else {
            const int o = k - 648;
            for (int r = 0; r < CL_TH; ++r)
                for (int c = 0; c < CL_TW; ++c) acc += sd[o][r][c];
        }
        partials[(long long)blockIdx.x * 660 + k] = acc;
    }
}
// one 256-thread block per output k (648 weights + 12 biases): thread t adds the partials t, t + 256, ... in double, the
// 256 sums are combined by a fixed tree -> deterministic (a single thread per output walking all `nb` partials took 250 us)
__global__ void __launch_bounds__(256)
convlstm_bwd_weight_final_kernel(const float* __restrict__ partials, int nb, float* __restrict__ dw, float* __restrict__ db) {
    __shared__ double sm[256];
    const int k = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += (double)partials[(long long)i * 660 + k];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (k < 648) dw[k] = (float)sm[0]; else db[k - 648] = (float)sm[0];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of the FUSED UPNet (BINHIP_BWD_FUSED_UPNET; forward: BINHIP_PLAN_FUSED_UPNET, binhip_conv_x3.hip).  The forward is
//   O = Main(x; W[4]) on every full-resolution pixel, then the outermost pixel ring overwritten by Ring(x; W[v]), v = border variant,
// so with g = dL/dO split into g_int (ring zeroed) and g_ring:
//   dL/dx = Main^T(g_int; W[4]) + Ring^T(g_ring; W[v]),   dW[4] = wgrad5x5(x, g_int),   dW[v] = sum over the ring pixels of variant v.
// Main^T and wgrad5x5 are the ordinary 5x5 kernels on `gsub`, the pixel-unshuffled (12 sub-pixel channels, one chunk), scaled,
// ring-zeroed gradient that upnet_gsub_kernel packs; the two ring kernels below add the rest (fp32, ~30 MFLOP each).  dW[*] -> dW0, dW2
// is the host's job (torch autograd through rdn_plan.fused_upnet_weights).

// one thread = one 16-byte slot (8 of the 16 plane channels) of one half-resolution pixel: channels c * 4 + i * 2 + j = g[c][2y+i][2x+j]
__global__ void upnet_gsub_kernel(const float* __restrict__ g, int N, int H, int W, const float* __restrict__ scale,
                                  _Float16* __restrict__ y_hi, _Float16* __restrict__ y_lo, unsigned* __restrict__ flags) {
    const float sc = scale ? scale[0] : 1.f;
    const long long HW = (long long)H * W, total = (long long)N * HW * 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int s = (int)(t & 1);
    long long u = t >> 1;
    const int x = (int)(u % W); u /= W;
    const int y = (int)(u % H);
    const int n = (int)(u / H);
    const int H2 = 2 * H, W2 = 2 * W;
    half8 hv, lv;
    unsigned sat = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = s * 8 + e, c = ch >> 2, Y = 2 * y + ((ch >> 1) & 1), X = 2 * x + (ch & 1);
        const bool ring = (Y == 0) || (Y == H2 - 1) || (X == 0) || (X == W2 - 1);
        const float v = (c < 3 && !ring) ? g[(((long long)n * 3 + c) * H2 + Y) * W2 + X] * sc : 0.f;
        hv[e] = split_hi(v, sat);
        lv[e] = split_lo(v, hv[e]);
    }
    *reinterpret_cast<half8*>(y_hi + t * 8) = hv;
    if (y_lo) *reinterpret_cast<half8*>(y_lo + t * 8) = lv;
    if (sat != 0 && flags) atomicOr(flags, BINHIP_FLAG_SATURATED);
}

struct RingBwdArgs {
    const float* g;            // dL/dO, fp32 [N, 3, 2H, 2W]
    const float* wvar;         // forward ring operators, fp32 [9][12][25][cin]
    const float* scale;        // scale[0]: the gradient planes carry dL/dx * scale
    const _Float16* x_hi;      // saved input of UPNet (G1), planes [cin / 16][N][H][W][16]
    const _Float16* x_lo;
    _Float16* gx_hi;           // dL/dx planes (same layout): read-modify-write by ring_dgrad
    _Float16* gx_lo;
    float* dwvar;              // [N][9][12][25][cin]: per-image partial sums
    float* dbvar;              // [N][9][12]
    unsigned* flags;
    int N, H, W, cin, accumulate;
};

// Ring^T: one thread = (band pixel, 8 input channels).  A half-resolution pixel (y, x) receives from ring output pixels (Y, X) with
// |Y / 2 - y| <= 2 and |X / 2 - x| <= 2, so only pixels within two of the border are touched: the band is enumerated as the rows
// {0, 1, 2, H-3, H-2, H-1} in full and the columns {0, 1, 2, W-3, W-2, W-1} of the remaining rows (all rows / columns when there are
// six or fewer).  Single writer per slot: plain read-modify-write of the hi / lo planes, no atomics.
__global__ void __launch_bounds__(256) upnet_ring_dgrad_kernel(const RingBwdArgs a) {
    const int H = a.H, W = a.W, H2 = 2 * H, W2 = 2 * W, cin = a.cin, ng = cin >> 3;
    const int nrow = H < 6 ? H : 6, ncolx = W < 6 ? W : 6;              // band rows / band columns
    const int mid = H > 6 ? H - 6 : 0;                                  // rows that only contribute their border columns
    const long long per_img = (long long)nrow * W + (long long)mid * ncolx;
    const long long total = per_img * a.N * ng;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int grp = (int)(t % ng);
    long long u = t / ng;
    const int n = (int)(u / per_img);
    long long q = u - (long long)n * per_img;
    int y, x;
    if (q < (long long)nrow * W) {
        const int r = (int)(q / W);
        x = (int)(q - (long long)r * W);
        y = (H <= 6) ? r : (r < 3 ? r : H - 6 + r);
    } else {
        q -= (long long)nrow * W;
        const int r = (int)(q / ncolx), k = (int)(q - (long long)r * ncolx);
        y = 3 + r;
        x = (W <= 6) ? k : (k < 3 ? k : W - 6 + k);
    }
    const float sc = a.scale ? a.scale[0] : 1.f;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int ci = grp << 3;
    // ring output pixels whose 5x5 window covers (y, x): half-resolution position (py, px) = (y - ty + 2, x - tx + 2), tap (ty, tx)
    for (int ty = 0; ty < 5; ++ty) {
        const int py = y - ty + 2;
        if (py < 0 || py >= H) continue;
        for (int tx = 0; tx < 5; ++tx) {
            const int px = x - tx + 2;
            if (px < 0 || px >= W) continue;
            const int tap = ty * 5 + tx;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const int Y = 2 * py + (sub >> 1), X = 2 * px + (sub & 1);
                const bool ring = (Y == 0) || (Y == H2 - 1) || (X == 0) || (X == W2 - 1);
                if (!ring) continue;
                const int vy = (Y == 0) ? 0 : (Y == H2 - 1 ? 2 : 1), vx = (X == 0) ? 0 : (X == W2 - 1 ? 2 : 1);
                const float* w = a.wvar + ((((long long)(3 * vy + vx) * 12 + sub) * 25 + tap) * cin) + ci;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float gv = a.g[(((long long)n * 3 + c) * H2 + Y) * W2 + X];
                    const float* wc = w + (long long)4 * c * 25 * cin;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = fmaf(wc[e], gv, acc[e]);
                }
            }
        }
    }
    const long long o = (((((long long)(ci >> 4) * a.N + n) * H + y) * W + x) << 4) + (ci & 15);
    half8 hv = *reinterpret_cast<const half8*>(a.gx_hi + o), lv;
    if (a.gx_lo) lv = *reinterpret_cast<const half8*>(a.gx_lo + o);
    unsigned sat = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = (float)hv[e] + acc[e] * sc;
        if (a.gx_lo) v += (float)lv[e];
        hv[e] = split_hi(v, sat);
        lv[e] = split_lo(v, hv[e]);
    }
    *reinterpret_cast<half8*>(a.gx_hi + o) = hv;
    if (a.gx_lo) *reinterpret_cast<half8*>(a.gx_lo + o) = lv;
    if (sat != 0 && a.flags) atomicOr(a.flags, BINHIP_FLAG_SATURATED);
}

// dW[n][v][o][tap][ci], dB[n][v][o] of the eight border variants, per image: one workgroup = (image, variant, sub-pixel, tap), thread =
// input channel; the ring pixels of the variant with that sub-pixel (an edge's every second pixel, or one corner) are walked in a fixed
// order and the caller sums the images (deterministic).  Variant 4 (interior) is written as zeros: its gradient is the 5x5
// weight-gradient kernel's.
__global__ void __launch_bounds__(256) upnet_ring_wgrad_kernel(const RingBwdArgs a) {
    const int H = a.H, W = a.W, H2 = 2 * H, W2 = 2 * W, cin = a.cin;
    int id = blockIdx.x;
    const int ncg = (cin + 31) >> 5;                   // 32 input channels per workgroup
    const int cg = id % ncg; id /= ncg;
    const int n = id / (12 * 25);                      // one image per workgroup: per-image partials, summed by the caller (deterministic)
    id -= n * (12 * 25);
    const int tap = id % 25; id /= 25;
    // the twelve (variant, sub-pixel) pairs that have ring pixels: an edge row / column carries one parity of its fixed coordinate and both
    // of the running one, a corner one pair.  The caller zero-fills the buffers: the other 24 pairs (and variant 4) stay zero.
    const int var = (int)((0x862055337711ull >> (4 * id)) & 15), sub = (int)((0x321031203210ull >> (4 * id)) & 3);   // (1,0) (1,1) (7,2) (7,3) (3,0) (3,2) (5,1) (5,3) + corners
    const int vy = var / 3, vx = var - 3 * vy, i = sub >> 1, j = sub & 1;
    const int ty = tap / 5 - 2, tx = tap % 5 - 2;
    const int ci = cg * 32 + threadIdx.x, sl = threadIdx.y, S = blockDim.y;      // 8 slices of the pixel walk per channel (an edge of 127 pixels = 16 per
    // thread; with cin x 2 threads and 64 pixels per thread the launch took 359 us; 1024-thread workgroups measured slower in the step:
    // the kernel shares the chip with the backward-data chain)
    // the pixels (Y, X) of this variant with parity (i, j): a border row / column has ONE parity (row 0: i = 0, row 2H - 1: i = 1), the
    // free coordinate of an edge runs over [1, L - 2] in steps of two
    int Y0, Y1, X0, X1;      // inclusive ranges, step 2; an empty range has Y0 > Y1 (X0 > X1)
    if (vy == 0) { Y0 = 0; Y1 = (i == 0) ? 0 : -1; }
    else if (vy == 2) { Y0 = H2 - 1; Y1 = (i == 1) ? H2 - 1 : -1; }
    else { Y0 = i ? 1 : 2; Y1 = H2 - 2; }
    if (vx == 0) { X0 = 0; X1 = (j == 0) ? 0 : -1; }
    else if (vx == 2) { X0 = W2 - 1; X1 = (j == 1) ? W2 - 1 : -1; }
    else { X0 = j ? 1 : 2; X1 = W2 - 2; }
    const bool none = (var == 4);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
    if (!none && ci < cin) {
            const int ny = Y1 >= Y0 ? (Y1 - Y0) / 2 + 1 : 0, nx = X1 >= X0 ? (X1 - X0) / 2 + 1 : 0;
            for (int pix = sl; pix < ny * nx; pix += S) {
                    const int Y = Y0 + 2 * (pix / nx), X = X0 + 2 * (pix % nx);
                    const float g0 = a.g[(((long long)n * 3 + 0) * H2 + Y) * W2 + X];
                    const float g1 = a.g[(((long long)n * 3 + 1) * H2 + Y) * W2 + X];
                    const float g2 = a.g[(((long long)n * 3 + 2) * H2 + Y) * W2 + X];
                    b0 += g0; b1 += g1; b2 += g2;
                    const int yy = (Y >> 1) + ty, xx = (X >> 1) + tx;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    const long long o = (((((long long)(ci >> 4) * a.N + n) * H + yy) * W + xx) << 4) + (ci & 15);
                    float xv = (float)a.x_hi[o];
                    if (a.x_lo) xv += (float)a.x_lo[o];
                    s0 = fmaf(g0, xv, s0); s1 = fmaf(g1, xv, s1); s2 = fmaf(g2, xv, s2);
                }
    }
    __shared__ float red[6][256];
    {
        const int t = sl * blockDim.x + threadIdx.x;
        red[0][t] = s0; red[1][t] = s1; red[2][t] = s2; red[3][t] = b0; red[4][t] = b1; red[5][t] = b2;
    }
    __syncthreads();
    if (sl != 0) return;
    for (int q = 1; q < S; ++q) {                       // fixed order: deterministic
        const int t = q * blockDim.x + threadIdx.x;
        s0 += red[0][t]; s1 += red[1][t]; s2 += red[2][t]; b0 += red[3][t]; b1 += red[4][t]; b2 += red[5][t];
    }
    if (ci < cin) {
        const float sv[3] = {s0, s1, s2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a.dwvar[(((((long long)n * 9 + var) * 12 + 4 * c + sub) * 25 + tap) * cin) + ci] = sv[c];
        }
    }
    if (tap == 0 && ci == 0) {                            // (channel group 0, thread 0)
        const float bv[3] = {b0, b1, b2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a.dbvar[((long long)n * 9 + var) * 12 + 4 * c + sub] = bv[c];
        }
    }
}

int bh_upnet_gsub(const float* g, int N, int H, int W, const float* scale, void* y_hi, void* y_lo, void* status, hipStream_t s) {
    if (!g || !y_hi || N <= 0 || H <= 0 || W <= 0) return BINHIP_E_ARG;
    const long long total = (long long)N * H * W * 2;
    upnet_gsub_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(g, N, H, W, scale, (_Float16*)y_hi, (_Float16*)y_lo,
                                                                             (unsigned*)status);
    BH_CHECK_LAUNCH();
    return 0;
}
static int ring_bwd_args(RingBwdArgs& a, const float* g, const float* wvar, const float* scale, const void* x_hi, const void* x_lo,
                         void* gx_hi, void* gx_lo, float* dwvar, float* dbvar, void* status, int N, int H, int W, int cin, int accumulate) {
    if (!g || N <= 0 || H <= 0 || W <= 0 || cin <= 0 || (cin & 15) || cin > 256) return BINHIP_E_ARG;
    a.g = g; a.wvar = wvar; a.scale = scale; a.x_hi = (const _Float16*)x_hi; a.x_lo = (const _Float16*)x_lo;
    a.gx_hi = (_Float16*)gx_hi; a.gx_lo = (_Float16*)gx_lo; a.dwvar = dwvar; a.dbvar = dbvar; a.flags = (unsigned*)status;
    a.N = N; a.H = H; a.W = W; a.cin = cin; a.accumulate = accumulate;
    return 0;
}
int bh_upnet_ring_dgrad(const float* g, const float* wvar, const float* scale, void* gx_hi, void* gx_lo, void* status, int N, int H, int W,
                        int cin, hipStream_t s) {
    RingBwdArgs a;
    if (!wvar || !gx_hi) return BINHIP_E_ARG;
    if (int rc = ring_bwd_args(a, g, wvar, scale, nullptr, nullptr, gx_hi, gx_lo, nullptr, nullptr, status, N, H, W, cin, 0)) return rc;
    const int nrow = H < 6 ? H : 6, ncolx = W < 6 ? W : 6, mid = H > 6 ? H - 6 : 0;
    const long long total = ((long long)nrow * W + (long long)mid * ncolx) * N * (cin >> 3);
    upnet_ring_dgrad_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(a);
    BH_CHECK_LAUNCH();
    return 0;
}
int bh_upnet_ring_wgrad(const float* g, const void* x_hi, const void* x_lo, float* dwvar, float* dbvar, int N, int H, int W, int cin,
                        int accumulate, hipStream_t s) {
    RingBwdArgs a;
    if (!x_hi || !dwvar || !dbvar) return BINHIP_E_ARG;
    if (int rc = ring_bwd_args(a, g, nullptr, nullptr, x_hi, x_lo, nullptr, nullptr, dwvar, dbvar, nullptr, N, H, W, cin, accumulate)) return rc;
    const int ncg = (cin + 31) / 32;
    upnet_ring_wgrad_kernel<<<dim3((unsigned)(12 * 25 * N * ncg)), dim3(32, 8), 0, s>>>(a);
    BH_CHECK_LAUNCH();
    return 0;
}


extern "C" {


int binhip_version(void) { return BINHIP_VERSION; }

int binhip_device_cus(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}

int binhip_nchw_to_planes(const float* x, int N, int C, int H, int W, void* y_hi, void* y_lo, void* status, void* stream) {
    if (!x || !y_hi) return BINHIP_E_ARG;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return BINHIP_E_SHAPE;
    const long long total = (long long)bh_chunks(C) * N * H * W * 2;
    hipLaunchKernelGGL(nchw_to_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, N, C, H, W, (_Float16*)y_hi, (_Float16*)y_lo, (const float*)nullptr, (unsigned*)status);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_nchw_to_planes_scaled(const float* x, int N, int C, int H, int W, const float* scale, void* y_hi,
                                 void* y_lo, void* status, void* stream) {
    if (!x || !y_hi || !scale) return BINHIP_E_ARG;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return BINHIP_E_SHAPE;
    const long long total = (long long)bh_chunks(C) * N * H * W * 2;
    hipLaunchKernelGGL(nchw_to_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, N, C, H, W, (_Float16*)y_hi, (_Float16*)y_lo, scale, (unsigned*)status);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_planes_to_nchw(const void* x_hi, const void* x_lo, int N, int C, int H, int W, float* y, void* stream) {
    if (!x_hi || !y) return BINHIP_E_ARG;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return BINHIP_E_SHAPE;
    const long long total = (long long)N * C * H * W;
    hipLaunchKernelGGL(planes_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)x_hi, (const _Float16*)x_lo, N, C, H, W, y);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_pixel_unshuffle_f32(const float* x, int N, int C, int H, int W, int r, float* y, void* stream) {
    if (!x || !y) return BINHIP_E_ARG;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || r < 1 || H % r || W % r) return BINHIP_E_SHAPE;
    const long long total = (long long)N * C * H * W;
    hipLaunchKernelGGL(pixel_unshuffle_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       N, C, H, W, r, y);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_pack_inputs(const float* const* images, int n_images, int N, int H, int W, void* y_hi, void* y_lo,
                       void* status, void* stream) {
    if (!images || !y_hi) return BINHIP_E_ARG;
    if (n_images < 1 || n_images > 5 || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return BINHIP_E_SHAPE;
    PackArgs a;
    for (int i = 0; i < 5; ++i) a.img[i] = (i < n_images) ? images[i] : nullptr;
    for (int i = 0; i < n_images; ++i) if (!images[i]) return BINHIP_E_ARG;
    a.nimg = n_images; a.N = N; a.H = H; a.W = W;
    const long long total = (long long)bh_chunks(12 * n_images) * N * (H / 2) * (W / 2) * 2;
    hipLaunchKernelGGL(pack_inputs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       a, (_Float16*)y_hi, (_Float16*)y_lo, (unsigned*)status);
    BH_CHECK_LAUNCH();
    return 0;
}

// the four-pixel ConvLSTM kernels: rows of whole float4s, every plane pointer 16-byte aligned (null = absent = fine)
static inline bool convlstm_vec4_ok(int W, const void* a, const void* b, const void* c, const void* d, const void* e, const void* f,
                                    const void* g) {
    uintptr_t m = 0;
    for (const void* p : {a, b, c, d, e, f, g}) m |= (uintptr_t)p;
    return (W & 3) == 0 && (m & 15) == 0;
}

int binhip_convlstm_fwd(const float* x, const float* c_prev, const float* h_prev, const float* w, const float* b,
                        float forget_bias, int N, int H, int W, float* c_new, float* h_new, void* stream) {
    if (!x || !w || !b || !h_new) return BINHIP_E_ARG;
    if ((c_prev == nullptr) != (h_prev == nullptr)) return BINHIP_E_ARG;
    if (N <= 0 || H <= 0 || W <= 0) return BINHIP_E_SHAPE;
    const long long total = (long long)N * H * W;
    if (convlstm_vec4_ok(W, x, c_prev, h_prev, c_new, h_new, nullptr, nullptr))
        hipLaunchKernelGGL(convlstm4_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           x, c_prev, h_prev, w, b, forget_bias, N, H, W, c_new, h_new);
    else
        hipLaunchKernelGGL(convlstm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           x, c_prev, h_prev, w, b, forget_bias, N, H, W, c_new, h_new);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_lstm_gates_fwd(const float* gates, const float* c_prev, float forget_bias, int N, int hidden, int H, int W,
                          float* c_new, float* h_new, void* stream) {
    if (!gates || !c_new || !h_new) return BINHIP_E_ARG;
    if (N <= 0 || hidden <= 0 || H <= 0 || W <= 0) return BINHIP_E_SHAPE;
    const long long HW = (long long)H * W, total = (long long)N * hidden * HW;
    hipLaunchKernelGGL(lstm_gates_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gates,
                       c_prev, forget_bias, hidden, HW, total, c_new, h_new);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_lstm_gates_bwd(const float* gates, const float* c_prev, const float* g_h, const float* g_c, float forget_bias, int N,
                          int hidden, int H, int W, float* g_gates, float* g_cprev, void* stream) {
    if (!gates || !g_gates || (!g_h && !g_c)) return BINHIP_E_ARG;
    if (N <= 0 || hidden <= 0 || H <= 0 || W <= 0) return BINHIP_E_SHAPE;
    const long long HW = (long long)H * W, total = (long long)N * hidden * HW;
    hipLaunchKernelGGL(lstm_gates_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gates,
                       c_prev, g_h, g_c, forget_bias, hidden, HW, total, g_gates, g_cprev);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_charbonnier_partials(int64_t numel) { (void)numel; return CHARB_BLOCKS; }

int binhip_pixel_loss_fwd(int kind, const float* x, const float* y, int64_t numel, float eps, float* partials, float* loss,
                          void* stream) {
    if (!x || !y || !partials || !loss) return BINHIP_E_ARG;
    if (kind < BINHIP_LOSS_CHARBONNIER || kind > BINHIP_LOSS_L2_SUM) return BINHIP_E_ARG;
    if (numel <= 0) return BINHIP_E_SHAPE;
    long long nb = (numel + 255) / 256;
    if (nb > CHARB_BLOCKS) nb = CHARB_BLOCKS;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)nb), block(256);
    if (kind == BINHIP_LOSS_CHARBONNIER)
        hipLaunchKernelGGL(charb_partial_kernel<BINHIP_LOSS_CHARBONNIER>, grid, block, 0, s, x, y, (long long)numel, eps, partials);
    else if (kind == BINHIP_LOSS_L1_SUM)
        hipLaunchKernelGGL(charb_partial_kernel<BINHIP_LOSS_L1_SUM>, grid, block, 0, s, x, y, (long long)numel, eps, partials);
    else
        hipLaunchKernelGGL(charb_partial_kernel<BINHIP_LOSS_L2_SUM>, grid, block, 0, s, x, y, (long long)numel, eps, partials);
    hipLaunchKernelGGL(charb_final_kernel, dim3(1), dim3(256), 0, s, partials, (int)nb,
                       kind == BINHIP_LOSS_CHARBONNIER ? (double)numel : 1.0, loss);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_pixel_loss_bwd(int kind, const float* x, const float* y, int64_t numel, float eps, const float* gloss, float* gx,
                          float* gy, void* stream) {
    if (!x || !y || !gloss || (!gx && !gy)) return BINHIP_E_ARG;
    if (kind < BINHIP_LOSS_CHARBONNIER || kind > BINHIP_LOSS_L2_SUM) return BINHIP_E_ARG;
    if (numel <= 0) return BINHIP_E_SHAPE;
    long long nb = (numel + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)nb), block(256);
    const float inv = kind == BINHIP_LOSS_CHARBONNIER ? 1.f / (float)numel : 1.f;
    if (kind == BINHIP_LOSS_CHARBONNIER)
        hipLaunchKernelGGL(charb_bwd_kernel<BINHIP_LOSS_CHARBONNIER>, grid, block, 0, s, x, y, (long long)numel, eps, inv, gloss, gx, gy);
    else if (kind == BINHIP_LOSS_L1_SUM)
        hipLaunchKernelGGL(charb_bwd_kernel<BINHIP_LOSS_L1_SUM>, grid, block, 0, s, x, y, (long long)numel, eps, inv, gloss, gx, gy);
    else
        hipLaunchKernelGGL(charb_bwd_kernel<BINHIP_LOSS_L2_SUM>, grid, block, 0, s, x, y, (long long)numel, eps, inv, gloss, gx, gy);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_multi_loss_fwd(int kind, const BinLossTerms* t, int64_t numel, float eps, float* partials, float* terms, float* loss,
                          void* stream) {
    if (!t || !partials || !terms || !loss) return BINHIP_E_ARG;
    if (kind < BINHIP_LOSS_CHARBONNIER || kind > BINHIP_LOSS_L2_SUM) return BINHIP_E_ARG;
    if (t->n_terms <= 0 || t->n_terms > BINHIP_LOSS_MAX_TERMS) return BINHIP_E_ARG;
    for (int i = 0; i < t->n_terms; ++i)
        if (!t->x[i] || !t->y[i]) return BINHIP_E_ARG;
    if (numel <= 0) return BINHIP_E_SHAPE;
    long long nb = (numel + 255) / 256;
    if (nb > CHARB_BLOCKS) nb = CHARB_BLOCKS;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)nb, (unsigned)t->n_terms), block(256);
    if (kind == BINHIP_LOSS_CHARBONNIER)
        hipLaunchKernelGGL(multi_loss_partial_kernel<BINHIP_LOSS_CHARBONNIER>, grid, block, 0, s, *t, (long long)numel, eps, partials);
    else if (kind == BINHIP_LOSS_L1_SUM)
        hipLaunchKernelGGL(multi_loss_partial_kernel<BINHIP_LOSS_L1_SUM>, grid, block, 0, s, *t, (long long)numel, eps, partials);
    else
        hipLaunchKernelGGL(multi_loss_partial_kernel<BINHIP_LOSS_L2_SUM>, grid, block, 0, s, *t, (long long)numel, eps, partials);
    hipLaunchKernelGGL(multi_loss_final_kernel, dim3(1), dim3(256), 0, s, partials, (int)nb, (int)t->n_terms,
                       kind == BINHIP_LOSS_CHARBONNIER ? (double)numel : 1.0, terms, loss);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_multi_loss_bwd(int kind, const BinLossTerms* t, int64_t numel, float eps, const float* gloss, const BinLossGrads* g,
                          void* stream) {
    if (!t || !g || !gloss) return BINHIP_E_ARG;
    if (kind < BINHIP_LOSS_CHARBONNIER || kind > BINHIP_LOSS_L2_SUM) return BINHIP_E_ARG;
    if (t->n_terms <= 0 || t->n_terms > BINHIP_LOSS_MAX_TERMS || g->n_out <= 0 || g->n_out > BINHIP_LOSS_MAX_TERMS) return BINHIP_E_ARG;
    for (int k = 0; k < g->n_out; ++k)
        if (!g->out[k] || g->term_a[k] < 0 || g->term_a[k] >= t->n_terms || g->term_b[k] >= t->n_terms) return BINHIP_E_ARG;
    if (numel <= 0) return BINHIP_E_SHAPE;
    long long nb = (numel + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)nb, (unsigned)g->n_out), block(256);
    // gloss / T, and for the mean criterion / numel: the per-term rounding order of the per-term path (g / T first, then / numel)
    const float scale = (kind == BINHIP_LOSS_CHARBONNIER ? 1.f / (float)numel : 1.f);
    const float inv_t = 1.0f / (float)t->n_terms;
    if (kind == BINHIP_LOSS_CHARBONNIER)
        hipLaunchKernelGGL(multi_loss_bwd_kernel<BINHIP_LOSS_CHARBONNIER>, grid, block, 0, s, *t, *g, (long long)numel, eps, scale, inv_t, gloss);
    else if (kind == BINHIP_LOSS_L1_SUM)
        hipLaunchKernelGGL(multi_loss_bwd_kernel<BINHIP_LOSS_L1_SUM>, grid, block, 0, s, *t, *g, (long long)numel, eps, scale, inv_t, gloss);
    else
        hipLaunchKernelGGL(multi_loss_bwd_kernel<BINHIP_LOSS_L2_SUM>, grid, block, 0, s, *t, *g, (long long)numel, eps, scale, inv_t, gloss);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_charbonnier_fwd(const float* x, const float* y, int64_t numel, float eps, float* partials, float* loss,
                           void* stream) {
    return binhip_pixel_loss_fwd(BINHIP_LOSS_CHARBONNIER, x, y, numel, eps, partials, loss, stream);
}

int binhip_charbonnier_bwd(const float* x, const float* y, int64_t numel, float eps, const float* gloss, float* gx,
                           float* gy, void* stream) {
    return binhip_pixel_loss_bwd(BINHIP_LOSS_CHARBONNIER, x, y, numel, eps, gloss, gx, gy, stream);
}

int binhip_u8_to_frame(const unsigned char* bgr_hwc, int H, int W, int pad_left, int pad_right, int pad_top,
                       int pad_bottom, float* out_chw, void* stream) {
    if (!bgr_hwc || !out_chw) return BINHIP_E_ARG;
    if (H <= 0 || W <= 0 || pad_left < 0 || pad_right < 0 || pad_top < 0 || pad_bottom < 0) return BINHIP_E_SHAPE;
    const int Hp = H + pad_top + pad_bottom, Wp = W + pad_left + pad_right;
    const long long total = (long long)3 * Hp * Wp;
    hipLaunchKernelGGL(u8_to_frame_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bgr_hwc, H,
                       W, pad_left, pad_top, Hp, Wp, out_chw);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_frame_to_u8(const float* chw, int Hp, int Wp, int top, int left, int H, int W, unsigned char* bgr_hwc,
                       void* stream) {
    if (!chw || !bgr_hwc) return BINHIP_E_ARG;
    if (H <= 0 || W <= 0 || top < 0 || left < 0 || top + H > Hp || left + W > Wp) return BINHIP_E_SHAPE;
    const long long total = (long long)H * W * 3;
    hipLaunchKernelGGL(frame_to_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, chw, Hp, Wp,
                       top, left, H, W, bgr_hwc);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_grad_scale(const float* g, int64_t numel, float target, float* partials, float* scale_out, void* stream) {
    if (!g || !partials || !scale_out) return BINHIP_E_ARG;
    if (numel <= 0 || !(target > 0.f)) return BINHIP_E_SHAPE;
    long long nb = (numel + 255) / 256;
    if (nb > CHARB_BLOCKS) nb = CHARB_BLOCKS;
    hipLaunchKernelGGL(amax_partial_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, g, (long long)numel, partials);
    hipLaunchKernelGGL(grad_scale_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, (int)nb, target, scale_out);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_unshuffle_planes(const void* x_hi, const void* x_lo, int N, int H, int W, int nchunks, void* y_hi, void* y_lo,
                            void* stream) {
    if (!x_hi || !y_hi || ((x_lo == nullptr) != (y_lo == nullptr))) return BINHIP_E_ARG;
    if (N <= 0 || H <= 0 || W <= 0 || nchunks <= 0) return BINHIP_E_SHAPE;
    const long long total = (long long)4 * nchunks * N * H * W * 2;
    const unsigned nb = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(unshuffle_planes_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x_hi, N, H, W,
                       nchunks, (_Float16*)y_hi);
    if (x_lo)
        hipLaunchKernelGGL(unshuffle_planes_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x_lo, N, H,
                           W, nchunks, (_Float16*)y_lo);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_unpack_input_grads(const void* gx0_hi, const void* gx0_lo, const float* gout, const float* scale,
                              int n_images, int N, int H, int W, float* const* outs, void* stream) {
    if (!gout || !outs) return BINHIP_E_ARG;
    if (n_images < 1 || n_images > 5 || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return BINHIP_E_SHAPE;
    UnpackArgs a;
    for (int i = 0; i < 5; ++i) a.out[i] = (i < n_images) ? outs[i] : nullptr;
    a.nimg = n_images; a.N = N; a.H = H; a.W = W;
    const long long total = (long long)N * 3 * H * W;
    hipLaunchKernelGGL(unpack_input_grads_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       a, (const _Float16*)gx0_hi, (const _Float16*)gx0_lo, gout, scale);
    BH_CHECK_LAUNCH();
    return 0;
}

size_t binhip_convlstm_bwd_workspace_bytes(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const size_t tiles = (size_t)((W + CL_TW - 1) / CL_TW) * ((H + CL_TH - 1) / CL_TH) * N;
    return ((size_t)N * 12 * H * W + tiles * 660) * sizeof(float) + 256;
}

int binhip_convlstm_bwd(const float* x, const float* c_prev, const float* h_prev, const float* w, const float* b,
                        float forget_bias, int N, int H, int W, const float* g_h, const float* g_c, void* workspace,
                        size_t workspace_bytes, float* gx, float* g_hprev, float* g_cprev, float* dw, float* db,
                        void* stream) {
    if (!x || !w || !b || !workspace || (!g_h && !g_c)) return BINHIP_E_ARG;
    if ((c_prev == nullptr) != (h_prev == nullptr)) return BINHIP_E_ARG;
    if (N <= 0 || H <= 0 || W <= 0) return BINHIP_E_SHAPE;
    if (workspace_bytes < binhip_convlstm_bwd_workspace_bytes(N, H, W)) return BINHIP_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* dg = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float* part = dg + (size_t)N * 12 * H * W;
    const long long total = (long long)N * H * W;
    const unsigned nb = (unsigned)((total + 255) / 256);
    if (convlstm_vec4_ok(W, x, c_prev, h_prev, g_h, g_c, dg, g_cprev))
        hipLaunchKernelGGL(convlstm4_bwd_gates_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, x, c_prev, h_prev,
                           w, b, forget_bias, N, H, W, g_h, g_c, dg, g_cprev);
    else
        hipLaunchKernelGGL(convlstm_bwd_gates_kernel, dim3(nb), dim3(256), 0, s, x, c_prev, h_prev, w, b, forget_bias, N, H, W,
                           g_h, g_c, dg, g_cprev);
    if (gx || g_hprev)
        hipLaunchKernelGGL(convlstm_bwd_input_kernel, dim3(nb), dim3(256), 0, s, dg, w, N, H, W, gx, g_hprev);
    if (dw && db) {
        const int tiles_x = (W + CL_TW - 1) / CL_TW, tiles_y = (H + CL_TH - 1) / CL_TH;
        const int nblk = tiles_x * tiles_y * N;
        hipLaunchKernelGGL(convlstm_bwd_weight_kernel, dim3((unsigned)nblk), dim3(256), 0, s, dg, x, h_prev, N, H, W, tiles_x,
                           tiles_y, part);
        hipLaunchKernelGGL(convlstm_bwd_weight_final_kernel, dim3(660), dim3(256), 0, s, part, nblk, dw, db);
    }
    BH_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
