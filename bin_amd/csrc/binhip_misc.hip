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
                                      _Float16* __restrict__ y_hi, _Float16* __restrict__ y_lo) {
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
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = ch * 16 + s * 8 + e;
        const float v = (c < C) ? x[((long long)n * C + c) * HW + pix] : 0.f;
        hv[e] = (_Float16)v;
        lv[e] = (_Float16)(v - (float)hv[e]);
    }
    *reinterpret_cast<half8*>(y_hi + t * 8) = hv;
    if (y_lo) *reinterpret_cast<half8*>(y_lo + t * 8) = lv;
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

// ---- K1: pixel_reshuffle(cat(images), 2) -> chunk planes at half resolution ---------------------
struct PackArgs {
    const float* img[5];
    int nimg, N, H, W;   // full-res H, W
};
__global__ void pack_inputs_kernel(PackArgs a, _Float16* __restrict__ y_hi, _Float16* __restrict__ y_lo) {
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
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = ch * 16 + s * 8 + e;     // = 4*cc + 2*i + j   (RDN.py:128-132)
        float v = 0.f;
        if (c < C) {
            const int cc = c >> 2, i = (c >> 1) & 1, j = c & 1;
            const int im = cc / 3, rgb = cc - im * 3;
            v = a.img[im][(((long long)n * 3 + rgb) * a.H + (2 * y + i)) * a.W + (2 * x + j)];
        }
        hv[e] = (_Float16)v;
        lv[e] = (_Float16)(v - (float)hv[e]);
    }
    *reinterpret_cast<half8*>(y_hi + t * 8) = hv;
    if (y_lo) *reinterpret_cast<half8*>(y_lo + t * 8) = lv;
}

// ---- ConvLSTM cell ------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

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
        const float c1 = cprev * sigmoidf_(g[6 + k] + fb) + sigmoidf_(g[k]) * tanhf(g[3 + k]);
        const float h1 = tanhf(c1) * sigmoidf_(g[9 + k]);
        if (cn) cn[o] = c1;
        hn[o] = h1;
    }
}

// ---- Charbonnier ---------------------------------------------------------------------------------
#define CHARB_BLOCKS 1024
__global__ void __launch_bounds__(256)
charb_partial_kernel(const float* __restrict__ x, const float* __restrict__ y, long long n, float eps,
                     float* __restrict__ partials) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = x[i] - y[i];
        acc += sqrtf(d * d + eps);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__global__ void __launch_bounds__(256)
charb_final_kernel(const float* __restrict__ partials, int nb, long long n, float* __restrict__ loss) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += (double)partials[i];
    __shared__ double sm[256];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(sm[0] / (double)n);
}
__global__ void __launch_bounds__(256)
charb_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, long long n, float eps,
                 const float* __restrict__ gl, float* __restrict__ gx, float* __restrict__ gy) {
    const float s = gl[0] / (float)n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = x[i] - y[i];
        const float g = s * d / sqrtf(d * d + eps);
        if (gx) gx[i] = g;
        if (gy) gy[i] = -g;
    }
}

extern "C" {

int binhip_version(void) { return BINHIP_VERSION; }

int binhip_device_cus(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
    return p.multiProcessorCount;
}

int binhip_nchw_to_planes(const float* x, int N, int C, int H, int W, void* y_hi, void* y_lo, void* stream) {
    if (!x || !y_hi) return BINHIP_E_ARG;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return BINHIP_E_SHAPE;
    const long long total = (long long)bh_chunks(C) * N * H * W * 2;
    hipLaunchKernelGGL(nchw_to_planes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, N, C, H, W, (_Float16*)y_hi, (_Float16*)y_lo);
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

int binhip_pack_inputs(const float* const* images, int n_images, int N, int H, int W, void* y_hi, void* y_lo,
                       void* stream) {
    if (!images || !y_hi) return BINHIP_E_ARG;
    if (n_images < 1 || n_images > 5 || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return BINHIP_E_SHAPE;
    PackArgs a;
    for (int i = 0; i < 5; ++i) a.img[i] = (i < n_images) ? images[i] : nullptr;
    for (int i = 0; i < n_images; ++i) if (!images[i]) return BINHIP_E_ARG;
    a.nimg = n_images; a.N = N; a.H = H; a.W = W;
    const long long total = (long long)bh_chunks(12 * n_images) * N * (H / 2) * (W / 2) * 2;
    hipLaunchKernelGGL(pack_inputs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       a, (_Float16*)y_hi, (_Float16*)y_lo);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_convlstm_fwd(const float* x, const float* c_prev, const float* h_prev, const float* w, const float* b,
                        float forget_bias, int N, int H, int W, float* c_new, float* h_new, void* stream) {
    if (!x || !w || !b || !h_new) return BINHIP_E_ARG;
    if ((c_prev == nullptr) != (h_prev == nullptr)) return BINHIP_E_ARG;
    if (N <= 0 || H <= 0 || W <= 0) return BINHIP_E_SHAPE;
    const long long total = (long long)N * H * W;
    hipLaunchKernelGGL(convlstm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, c_prev, h_prev, w, b, forget_bias, N, H, W, c_new, h_new);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_charbonnier_partials(int64_t numel) { (void)numel; return CHARB_BLOCKS; }

int binhip_charbonnier_fwd(const float* x, const float* y, int64_t numel, float eps, float* partials, float* loss,
                           void* stream) {
    if (!x || !y || !partials || !loss) return BINHIP_E_ARG;
    if (numel <= 0) return BINHIP_E_SHAPE;
    long long nb = (numel + 255) / 256;
    if (nb > CHARB_BLOCKS) nb = CHARB_BLOCKS;
    hipLaunchKernelGGL(charb_partial_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, y,
                       (long long)numel, eps, partials);
    hipLaunchKernelGGL(charb_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, (int)nb,
                       (long long)numel, loss);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_charbonnier_bwd(const float* x, const float* y, int64_t numel, float eps, const float* gloss, float* gx,
                           float* gy, void* stream) {
    if (!x || !y || !gloss || (!gx && !gy)) return BINHIP_E_ARG;
    if (numel <= 0) return BINHIP_E_SHAPE;
    long long nb = (numel + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(charb_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, y,
                       (long long)numel, eps, gloss, gx, gy);
    BH_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
