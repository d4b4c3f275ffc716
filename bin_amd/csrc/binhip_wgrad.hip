// binhip_wgrad.hip — weight/bias gradients of the bin_stage4 convolutions on the matrix cores.
//
// Stands in for autograd's conv2d weight/bias backward of every F.conv2d on the path (reference
// models/archs/RDN.py:141,162,187-207, triggered by bin_model.py:140 `l_pix.backward()`):
//     dW[co][ci][dy][dx] = sum_{n,y,x} gY[co][n,y,x] * X[ci][n,y+dy-p,x+dx-p],   db[co] = sum gY[co]
// as a GEMM whose contraction dimension is the PIXEL index: per tap, D[ci 32][co 32] += X^T[ci][px16] * gY[px16][co]
// with v_mfma_f32_32x32x16_f16.  Both operands live in chunk planes ([pixel][16 ch], pixel-major), i.e. with the
// contraction index on the slow axis, so fragments are fetched with the LDS transpose read ds_read_b64_tr_b16
// (lane t of a 16-lane group addresses pixel t/4, 8-byte piece t%4 and receives channel t of pixels 0..3 — mapping
// verified on hardware by tools/probe_tr16.hip).
//
// Work split: block (pb, cp, z) owns input-channel pair cp (32 ci), output tile z (32 co, and for 5x5 one tap
// row), and walks pixel tiles pb, pb+PB, ... (8x32 pixels, halo patch + gY tile DMA'd to LDS, double buffered),
// keeping all its taps' 32x32 accumulators in registers (K-split over the 4 waves by pixel row).  At the end the
// 4 waves are reduced through LDS and ONE partial per block is written; a second kernel sums the PB partials in a
// fixed order (deterministic), un-scales and scatters to OIHW fp32.
#include "binhip_internal.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef short short4_ __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(3))) short4_ lds_short4_t;

struct WgradKArgs {
    const _Float16* x_hi;
    const _Float16* x_lo;
    const _Float16* g_hi;
    const _Float16* g_lo;
    float* partial;
    float* partial_b;
    long long x_group_stride;
    int x_cpg;
    int N, H, W;
    int cin_chunks, cout_chunks;
    int tiles_x, tiles_y, ntiles;
    int PB, ncp, ncot;
    int dbg;      // ablation (timing experiments): 1 skip DMA, 2 skip MFMA/LDS reads, 4 skip the final reduction+store
};

template <int KS, int TR, int NT>
struct WgCfg {
    static constexpr int PAD = KS / 2;
    static constexpr int TH = 8;
    static constexpr int PH = TH + TR - 1;
    static constexpr int PW = 32 + KS - 1;
    static constexpr int NTAP = TR * KS;
    static constexpr int XP = (PH * PW * 2 + 63) / 64;   // 1-KiB pieces per X chunk patch
    static constexpr int GP = TH * 32 * 2 / 64;          // 1-KiB pieces per gY chunk tile (= 8)
    static constexpr int XBYTES = XP * 1024, GBYTES = GP * 1024;
    static constexpr int PLANE_BYTES = 2 * XBYTES + 2 * GBYTES;
    static constexpr int NPL = (NT == 3) ? 2 : 1;
    static constexpr int BUF_BYTES = NPL * PLANE_BYTES;
    static constexpr int LDS_BYTES = (2 * BUF_BYTES > 16384) ? 2 * BUF_BYTES : 16384;
    static constexpr int NXJ = (XP + 3) / 4, NGJ = GP / 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// fragment via two transpose reads: lane -> (channel col = lane&15 of chunk (lane>>4)&1, pixels p0 + 8*(lane>>5) + 0..7)
__device__ __forceinline__ half8 tr_frag(const char* img, int chunk_bytes, int p0, int lane) {
    const int t = lane & 15;
    const int ch = (lane >> 4) & 1;
    const int kg = lane >> 5;
    const char* base = img + ch * chunk_bytes;
    half8 r;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int p = p0 + kg * 8 + rd * 4 + (t >> 2);
        const int off = p * 32 + (((((t & 3) >> 1)) ^ ((p >> 3) & 1)) << 4) + ((t & 1) << 3);
        short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_t*)(base + off));
        union { short4_ s; _Float16 h[4]; } u;
        u.s = v;
        r[rd * 4 + 0] = u.h[0]; r[rd * 4 + 1] = u.h[1]; r[rd * 4 + 2] = u.h[2]; r[rd * 4 + 3] = u.h[3];
    }
    return r;
}

template <int KS, int TR, int NT>
__device__ __forceinline__ void wg_issue(const WgradKArgs& a, char* smem, int buf, int tile, int cp, int cot, int dy0,
                                         int wave, int lane, long long plane_elems, unsigned plane_bytes) {
    using C = WgCfg<KS, TR, NT>;
    int b = tile;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const int img = b / a.tiles_y;
    const int tx0 = tx * 32, ty0 = ty * C::TH;
    const int H = a.H, W = a.W;
#pragma unroll
    for (int pl = 0; pl < C::NPL; ++pl) {
        char* pbase = smem + buf * C::BUF_BYTES + pl * C::PLANE_BYTES;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // ---- X patch of input chunk 2*cp + h (zeros when the chunk does not exist)
            const int c = 2 * cp + h;
            const _Float16* xb = pl ? a.x_lo : a.x_hi;
            const long long coff = (a.x_cpg > 0)
                ? (long long)(c / a.x_cpg) * a.x_group_stride + (long long)(c % a.x_cpg) * plane_elems
                : (long long)c * plane_elems;
            const bool have = c < a.cin_chunks;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(have ? xb + coff : xb), 0, have ? plane_bytes : 0u, 0x00020000);
            char* lds = pbase + h * C::XBYTES;
#pragma unroll
            for (int j = 0; j < C::NXJ; ++j) {
                const int i = wave + 4 * j;
                if (i < C::XP) {
                    const int q = i * 64 + lane;
                    const int p = q >> 1, s = q & 1;
                    const int py = p / C::PW, px = p - py * C::PW;
                    const int gy = ty0 + py + dy0 - C::PAD, gx = tx0 + px - C::PAD;
                    const int cg = s ^ ((p >> 3) & 1);
                    const bool ok = (p < C::PH * C::PW) && gy >= 0 && gy < H && gx >= 0 && gx < W;
                    const unsigned vo = ok ? (unsigned)((((long long)img * H + gy) * W + gx) * 32 + cg * 16) : 0x80000000u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(lds + i * 1024), 16, vo, 0, 0, 0);
                }
            }
            // ---- gY tile of output chunk 2*cot + h
            const int gc = 2 * cot + h;
            const bool haveg = gc < a.cout_chunks;
            const _Float16* gb = pl ? a.g_lo : a.g_hi;
            __amdgpu_buffer_rsrc_t gs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(haveg ? gb + (long long)gc * plane_elems : gb), 0, haveg ? plane_bytes : 0u, 0x00020000);
            char* gl = pbase + 2 * C::XBYTES + h * C::GBYTES;
#pragma unroll
            for (int j = 0; j < C::NGJ; ++j) {
                const int i = wave + 4 * j;
                const int q = i * 64 + lane;
                const int p = q >> 1, s = q & 1;
                const int py = p >> 5, px = p & 31;
                const int gy = ty0 + py, gx = tx0 + px;
                const int cg = s ^ ((p >> 3) & 1);
                const bool ok = gy < H && gx < W;
                const unsigned vo = ok ? (unsigned)((((long long)img * H + gy) * W + gx) * 32 + cg * 16) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(gs, (lds_void_t*)(gl + i * 1024), 16, vo, 0, 0, 0);
            }
        }
    }
}

template <int KS, int TR, int NT>
__global__ void __launch_bounds__(256)
wgrad_mfma_kernel(const WgradKArgs a) {
    using C = WgCfg<KS, TR, NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pb = blockIdx.x, cp = blockIdx.y;
    const int cot = blockIdx.z % a.ncot;
    const int dyg = blockIdx.z / a.ncot;
    const int dy0 = dyg * TR;
    const long long plane_elems = (long long)a.N * a.H * a.W * 16;
    const unsigned plane_bytes = (unsigned)(plane_elems * 2);
    const bool do_bias = (cp == 0) && (dyg == 0);

    floatx16 acc[C::NTAP];
    floatx16 accb;
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[e] = 0.f;
#pragma unroll
    for (int t = 0; t < C::NTAP; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    half8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (_Float16)1.0f;

    int tile = pb;
    if (tile < a.ntiles && !(a.dbg & 1)) wg_issue<KS, TR, NT>(a, smem, 0, tile, cp, cot, dy0, wave, lane, plane_elems, plane_bytes);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (; tile < a.ntiles; tile += a.PB) {
        const int nxt = tile + a.PB;
        if (nxt < a.ntiles && !(a.dbg & 1)) wg_issue<KS, TR, NT>(a, smem, cur ^ 1, nxt, cp, cot, dy0, wave, lane, plane_elems, plane_bytes);
        if (a.dbg & 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); cur ^= 1; continue; }
        const char* xb = smem + cur * C::BUF_BYTES;
        const char* gb = xb + 2 * C::XBYTES;
        // 4 K-steps per wave and tile (2 rows x 2 half-rows of 16 pixels), software-pipelined: the transpose reads of
        // K-step s+1 are issued before the MFMAs of K-step s (a workgroup is alone on its CU, one wave per SIMD, so
        // nothing else hides the LDS latency); sched_barriers keep the scheduler from sinking the reads again.
        half8 Bh[2], Bl[2], Ah[2][C::NTAP], Al[2][C::NTAP];
        {
            const int row = wave * 2, x0 = 0;
            Bh[0] = tr_frag(gb, C::GBYTES, row * 32 + x0, lane);
            if constexpr (NT == 3) Bl[0] = tr_frag(gb + C::PLANE_BYTES, C::GBYTES, row * 32 + x0, lane);
#pragma unroll
            for (int t = 0; t < C::NTAP; ++t) {
                const int p0 = (row + t / KS) * C::PW + x0 + t % KS;
                Ah[0][t] = tr_frag(xb, C::XBYTES, p0, lane);
                if constexpr (NT == 3) Al[0][t] = tr_frag(xb + C::PLANE_BYTES, C::XBYTES, p0, lane);
            }
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            if (s4 + 1 < 4) {
                const int row = wave * 2 + ((s4 + 1) >> 1), x0 = ((s4 + 1) & 1) * 16;
                Bh[(s4 + 1) & 1] = tr_frag(gb, C::GBYTES, row * 32 + x0, lane);
                if constexpr (NT == 3) Bl[(s4 + 1) & 1] = tr_frag(gb + C::PLANE_BYTES, C::GBYTES, row * 32 + x0, lane);
#pragma unroll
                for (int t = 0; t < C::NTAP; ++t) {
                    const int p0 = (row + t / KS) * C::PW + x0 + t % KS;
                    Ah[(s4 + 1) & 1][t] = tr_frag(xb, C::XBYTES, p0, lane);
                    if constexpr (NT == 3) Al[(s4 + 1) & 1][t] = tr_frag(xb + C::PLANE_BYTES, C::XBYTES, p0, lane);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (do_bias) {
                accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, Bh[s4 & 1], accb, 0, 0, 0);
                if constexpr (NT == 3) accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, Bl[s4 & 1], accb, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < C::NTAP; ++t) {
                if constexpr (NT == 3) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[s4 & 1][t], Bh[s4 & 1], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s4 & 1][t], Bl[s4 & 1], acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s4 & 1][t], Bh[s4 & 1], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    if (a.dbg & 4) { if (acc[0][0] == 12345.f) a.partial[0] = accb[0]; return; }
    // ---- reduce the 4 waves through LDS, one partial per block ----------------------------------
    float* red = reinterpret_cast<float*>(smem);          // [4 waves][32 m][32 n]
    const int n = lane & 31, hi = lane >> 5;
    const long long blk = ((long long)blockIdx.z * a.ncp + cp) * a.PB + pb;
#pragma unroll
    for (int t = 0; t < C::NTAP; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = (e & 3) + 8 * (e >> 2) + 4 * hi;
            red[wave * 1024 + m * 32 + n] = acc[t][e];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            a.partial[(blk * C::NTAP + t) * 1024 + idx] =
                (red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx]);
        }
    }
    if (do_bias) {
        __syncthreads();
        // bias: every row m of accb holds sum_k gY[k][n]; keep row 0 (e = 0 of the lanes with hi == 0)
        if (hi == 0) red[wave * 32 + n] = accb[0];
        __syncthreads();
        if (tid < 32)
            a.partial_b[((long long)cot * a.PB + pb) * 32 + tid] =
                (red[tid] + red[32 + tid]) + (red[64 + tid] + red[96 + tid]);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Lean variant of the kernel above: ONE LDS buffer (no intra-workgroup double buffering), fragments fetched per tap,
// bias as per-lane VALU sums, <= 256 registers -> two workgroups share a CU and overlap each other's DMA waits.
template <int KS, int TR, int NT>
__global__ void __launch_bounds__(256, 2)
wgrad_mfma_sb_kernel(const WgradKArgs a) {
    using C = WgCfg<KS, TR, NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pb = blockIdx.x, cp = blockIdx.y;
    const int cot = blockIdx.z % a.ncot;
    const int dyg = blockIdx.z / a.ncot;
    const int dy0 = dyg * TR;
    const long long plane_elems = (long long)a.N * a.H * a.W * 16;
    const unsigned plane_bytes = (unsigned)(plane_elems * 2);
    const bool do_bias = (cp == 0) && (dyg == 0);

    floatx16 acc[C::NTAP];
    float bsum = 0.f;
#pragma unroll
    for (int t = 0; t < C::NTAP; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    for (int tile = pb; tile < a.ntiles; tile += a.PB) {
        if (!(a.dbg & 1)) wg_issue<KS, TR, NT>(a, smem, 0, tile, cp, cot, dy0, wave, lane, plane_elems, plane_bytes);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!(a.dbg & 2)) {
            const char* xb = smem;
            const char* gb = xb + 2 * C::XBYTES;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int row = wave * 2 + (s4 >> 1), x0 = (s4 & 1) * 16;
                const half8 Bh = tr_frag(gb, C::GBYTES, row * 32 + x0, lane);
                half8 Bl;
                if constexpr (NT == 3) Bl = tr_frag(gb + C::PLANE_BYTES, C::GBYTES, row * 32 + x0, lane);
                if (do_bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        bsum += (float)Bh[e];
                        if constexpr (NT == 3) bsum += (float)Bl[e];
                    }
                }
#pragma unroll
                for (int t = 0; t < C::NTAP; ++t) {
                    const int p0 = (row + t / KS) * C::PW + x0 + t % KS;
                    const half8 Ah = tr_frag(xb, C::XBYTES, p0, lane);
                    if constexpr (NT == 3) {
                        const half8 Al = tr_frag(xb + C::PLANE_BYTES, C::XBYTES, p0, lane);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, acc[t], 0, 0, 0);
                    }
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (a.dbg & 4) { if (acc[0][0] == 12345.f) a.partial[0] = bsum; return; }
    float* red = reinterpret_cast<float*>(smem);
    const int n = lane & 31, hi = lane >> 5;
    const long long blk = ((long long)blockIdx.z * a.ncp + cp) * a.PB + pb;
#pragma unroll
    for (int t = 0; t < C::NTAP; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = (e & 3) + 8 * (e >> 2) + 4 * hi;
            red[wave * 1024 + m * 32 + n] = acc[t][e];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            a.partial[(blk * C::NTAP + t) * 1024 + idx] = (red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx]);
        }
    }
    if (do_bias) {
        __syncthreads();
        red[tid] = bsum;                              // [wave][kg][co]
        __syncthreads();
        if (tid < 32) {
            float tsum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) tsum += red[k * 32 + tid];
            a.partial_b[((long long)cot * a.PB + pb) * 32 + tid] = tsum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 convolutions (LFF 224->96, GFF.0 1152->96): with a single tap the whole [NCPB x 32 ci] x [96 co] gradient tile
// fits in registers (NCPB*3 accumulators), so one workgroup keeps ALL three output tiles and NCPB = 4 input-channel
// pairs: X and gY are each streamed through LDS exactly once (the generic kernel re-reads X per output tile and gY
// per channel pair: 705 MB instead of 168 MB for LFF at 8x128x128).  Pixel tile 4 x 32, wave w owns row w.
constexpr int W1_NCPB = 4, W1_NCOT = 3, W1_TH = 4;   // 12 accumulator tiles = 192 registers

template <int NT>
struct W1Cfg {
    static constexpr int NPL = (NT == 3) ? 2 : 1;
    static constexpr int CH_BYTES = W1_TH * 32 * 32;                 // one chunk tile: 4 KiB
    static constexpr int G_BYTES = NPL * 2 * W1_NCOT * CH_BYTES;     // gY tile, 6 chunks per plane
    static constexpr int X_BYTES = NPL * 2 * CH_BYTES;               // one channel pair of X
    static constexpr int LDS_BYTES = 2 * G_BYTES + 2 * X_BYTES;      // both double buffered
    static_assert(LDS_BYTES >= 16384 && LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int NT>
__device__ __forceinline__ void w1_issue_chunk(const _Float16* base, long long off, bool have, unsigned plane_bytes, char* lds,
                                               int img, int ty0, int tx0, int H, int W, int wave, int lane) {
    // one 4 x 32 pixel chunk tile = 4 KiB = 4 DMA pieces, one per wave
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(have ? base + off : base), 0,
                                                                  have ? plane_bytes : 0u, 0x00020000);
    const int q = wave * 64 + lane;
    const int p = q >> 1, s = q & 1;
    const int gy = ty0 + (p >> 5), gx = tx0 + (p & 31);
    const int cg = s ^ ((p >> 3) & 1);
    const bool ok = gy < H && gx < W;
    const unsigned vo = ok ? (unsigned)((((long long)img * H + gy) * W + gx) * 32 + cg * 16) : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(lds + wave * 1024), 16, vo, 0, 0, 0);
}

template <int NT>
__global__ void __launch_bounds__(256)
wgrad1x1_kernel(const WgradKArgs a) {
    using C = W1Cfg<NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pb = blockIdx.x;
    const int cpg = blockIdx.y;                   // group of W1_NCPB channel pairs
    const int cp0 = cpg * W1_NCPB;
    const long long plane_elems = (long long)a.N * a.H * a.W * 16;
    const unsigned plane_bytes = (unsigned)(plane_elems * 2);
    const bool do_bias = (cpg == 0);
    char* gbuf = smem;                            // [2][NPL][6 chunks][4 KiB]
    char* xbuf = smem + 2 * C::G_BYTES;           // [2][NPL][2 chunks][4 KiB]

    floatx16 acc[W1_NCPB][W1_NCOT];
    float bsum[W1_NCOT] = {0.f, 0.f, 0.f};   // bias: per-lane sum of this lane's gY fragment elements (co = lane & 31)
#pragma unroll
    for (int i = 0; i < W1_NCPB; ++i)
#pragma unroll
        for (int j = 0; j < W1_NCOT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // issue helpers are inlined by hand (no lambdas in kernels): gY tile of `tile` into g-buffer gb, X pair cp into x-buffer xb
#define W1_TILE_COORDS(tile_)                                   \
    int b_ = (tile_);                                           \
    const int tx_ = b_ % a.tiles_x; b_ /= a.tiles_x;            \
    const int ty_ = b_ % a.tiles_y;                             \
    const int img_ = b_ / a.tiles_y;                            \
    const int tx0_ = tx_ * 32, ty0_ = ty_ * W1_TH;
#define W1_ISSUE_G(tile_, gb_)                                                                                         \
    {                                                                                                                  \
        W1_TILE_COORDS(tile_)                                                                                          \
        _Pragma("unroll") for (int pl = 0; pl < C::NPL; ++pl)                                                          \
        _Pragma("unroll") for (int c6 = 0; c6 < 2 * W1_NCOT; ++c6)                                                     \
            w1_issue_chunk<NT>(pl ? a.g_lo : a.g_hi, (long long)c6 * plane_elems, c6 < a.cout_chunks, plane_bytes,     \
                               gbuf + (gb_) * C::G_BYTES + (pl * 2 * W1_NCOT + c6) * C::CH_BYTES, img_, ty0_, tx0_,    \
                               a.H, a.W, wave, lane);                                                                  \
    }
#define W1_ISSUE_X(tile_, cp_, xb_)                                                                                    \
    {                                                                                                                  \
        W1_TILE_COORDS(tile_)                                                                                          \
        _Pragma("unroll") for (int pl = 0; pl < C::NPL; ++pl)                                                          \
        _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                             \
            const int c = 2 * (cp_) + hh;                                                                              \
            const long long coff = (a.x_cpg > 0)                                                                       \
                ? (long long)(c / a.x_cpg) * a.x_group_stride + (long long)(c % a.x_cpg) * plane_elems                 \
                : (long long)c * plane_elems;                                                                          \
            w1_issue_chunk<NT>(pl ? a.x_lo : a.x_hi, coff, c < a.cin_chunks, plane_bytes,                              \
                               xbuf + (xb_) * C::X_BYTES + (pl * 2 + hh) * C::CH_BYTES, img_, ty0_, tx0_, a.H, a.W,    \
                               wave, lane);                                                                            \
        }                                                                                                              \
    }

    int tile = pb;
    int gb = 0, xb = 0;
    if (tile < a.ntiles && !(a.dbg & 1)) {
        W1_ISSUE_G(tile, 0)
        W1_ISSUE_X(tile, cp0, 0)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (; tile < a.ntiles; tile += a.PB) {
        const int nxt_tile = tile + a.PB;
        half8 Bh[2][W1_NCOT], Bl[2][W1_NCOT];
        if (!(a.dbg & 2)) {
            const char* gbase = gbuf + gb * C::G_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < W1_NCOT; ++j) {
                    Bh[ks][j] = tr_frag(gbase + j * 2 * C::CH_BYTES, C::CH_BYTES, wave * 32 + ks * 16, lane);
                    if constexpr (NT == 3)
                        Bl[ks][j] = tr_frag(gbase + (2 * W1_NCOT + j * 2) * C::CH_BYTES, C::CH_BYTES, wave * 32 + ks * 16, lane);
                }
            if (do_bias) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < W1_NCOT; ++j) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            bsum[j] += (float)Bh[ks][j][e];
                            if constexpr (NT == 3) bsum[j] += (float)Bl[ks][j][e];
                        }
                    }
            }
        }
#pragma unroll
        for (int cpl = 0; cpl < W1_NCPB; ++cpl) {
            // prefetch: next channel pair of this tile, or the next tile's gY + first pair
            if (!(a.dbg & 1)) {
                if (cpl + 1 < W1_NCPB) {
                    if (cp0 + cpl + 1 < a.ncp) W1_ISSUE_X(tile, cp0 + cpl + 1, xb ^ 1)
                } else if (nxt_tile < a.ntiles) {
                    W1_ISSUE_G(nxt_tile, gb ^ 1)
                    W1_ISSUE_X(nxt_tile, cp0, xb ^ 1)
                }
            }
            if (cp0 + cpl < a.ncp && !(a.dbg & 2)) {
                const char* xbase = xbuf + xb * C::X_BYTES;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const half8 Ah = tr_frag(xbase, C::CH_BYTES, wave * 32 + ks * 16, lane);
                    half8 Al;
                    if constexpr (NT == 3) Al = tr_frag(xbase + 2 * C::CH_BYTES, C::CH_BYTES, wave * 32 + ks * 16, lane);
#pragma unroll
                    for (int j = 0; j < W1_NCOT; ++j) {
                        if constexpr (NT == 3) {
                            acc[cpl][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh[ks][j], acc[cpl][j], 0, 0, 0);
                            acc[cpl][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl[ks][j], acc[cpl][j], 0, 0, 0);
                        }
                        acc[cpl][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh[ks][j], acc[cpl][j], 0, 0, 0);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            xb ^= 1;
        }
        gb ^= 1;
    }
#undef W1_ISSUE_G
#undef W1_ISSUE_X
#undef W1_TILE_COORDS

    if (a.dbg & 4) { if (acc[0][0][0] == 12345.f) a.partial[0] = bsum[0]; return; }
    // ---- reduce the 4 waves through LDS; partial layout identical to the generic kernel (ntap = 1, z = co tile)
    float* red = reinterpret_cast<float*>(smem);
    const int n = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int cpl = 0; cpl < W1_NCPB; ++cpl) {
        const bool cp_valid = cp0 + cpl < a.ncp;          // block-uniform
#pragma unroll
        for (int j = 0; j < W1_NCOT; ++j) {
            if (!cp_valid) continue;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (e & 3) + 8 * (e >> 2) + 4 * hi;
                red[wave * 1024 + m * 32 + n] = acc[cpl][j][e];
            }
            __syncthreads();
            if (j < a.ncot) {
                const long long blk = ((long long)j * a.ncp + (cp0 + cpl)) * a.PB + pb;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = tid + 256 * i;
                    a.partial[blk * 1024 + idx] = (red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx]);
                }
            }
        }
    }
    if (do_bias) {
#pragma unroll
        for (int j = 0; j < W1_NCOT; ++j) {
            __syncthreads();
            red[tid] = bsum[j];                       // [wave][kg][co]
            __syncthreads();
            if (tid < 32 && j < a.ncot) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) t += red[k * 32 + tid];
                a.partial_b[((long long)j * a.PB + pb) * 32 + tid] = t;
            }
        }
    }
}

// final deterministic reduction over the PB partials + un-scale + scatter to OIHW, for a BATCH of layers in one launch
// (the backward plan reduces the five layers of a dense block together: 1 122 -> 306 reduce launches per training step).
// One 256-thread block per (z, cp, tap, m) row of 32 outputs of a layer: thread (nn, ps) sums every 8th partial
// (coalesced 128-B reads), the 8 slices are combined through LDS in a fixed order.  Rows >= nrows of a layer handle
// its bias (one per co tile).  Summation order per output is independent of the batching.
struct ReduceBatch {
    BhWgradReduce it[BH_WGRAD_BATCH];
    long long row_start[BH_WGRAD_BATCH + 1];
    int n;
};

__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const ReduceBatch rb, const float* __restrict__ inv_scale, int accumulate) {
    __shared__ float sm[8][32];
    long long row = blockIdx.x;
    int li = 0;
#pragma unroll
    for (int i = 1; i < BH_WGRAD_BATCH; ++i)
        if (i < rb.n && row >= rb.row_start[i]) li = i;
    const BhWgradReduce& L = rb.it[li];
    row -= rb.row_start[li];
    const float* __restrict__ partial = L.partial;
    const float* __restrict__ partial_b = L.partial_b;
    const int PB = L.PB, ncp = L.ncp, ncot = L.ncot, ks = L.ks, tr = L.tr, cout = L.cout, cin = L.cin;
    const long long nrows = L.nrows;
    const int ntap_blk = tr * ks;
    const int nn = threadIdx.x & 31, ps = threadIdx.x >> 5;
    const float is = inv_scale ? inv_scale[0] : 1.f;
    float s = 0.f;
    int co, ci = 0, dy = 0, dx = 0;
    bool is_bias = row >= nrows;
    if (!is_bias) {
        const int m = (int)(row & 31);
        long long u = row >> 5;
        const int tap = (int)(u % ntap_blk); u /= ntap_blk;
        const int cp = (int)(u % ncp); u /= ncp;
        const int z = (int)u;                       // = dyg * ncot + cot
        const int cot = z % ncot, dyg = z / ncot;
        co = cot * 32 + nn; ci = cp * 32 + m;
        dy = dyg * tr + tap / ks; dx = tap % ks;
        const long long base = (((long long)z * ncp + cp) * PB) * ntap_blk + tap;
        for (int p = ps; p < PB; p += 8) s += partial[(base + (long long)p * ntap_blk) * 1024 + m * 32 + nn];
    } else {
        const int cot = (int)(row - nrows);
        co = cot * 32 + nn;
        for (int p = ps; p < PB; p += 8) s += partial_b[((long long)cot * PB + p) * 32 + nn];
    }
    sm[ps][nn] = s;
    __syncthreads();
    if (ps != 0) return;
    const float tot = ((sm[0][nn] + sm[1][nn]) + (sm[2][nn] + sm[3][nn])) + ((sm[4][nn] + sm[5][nn]) + (sm[6][nn] + sm[7][nn]));
    if (co >= cout) return;
    int cr = co;
    if (L.shuffle) { const int cq = cout / 4; cr = (co % cq) * 4 + co / cq; }
    if (is_bias) {
        if (L.db) L.db[cr] = accumulate ? L.db[cr] + tot * is : tot * is;
        return;
    }
    if (ci >= cin) return;
    float* o = L.dw + (((long long)cr * cin + ci) * ks + dy) * ks + dx;
    *o = accumulate ? *o + tot * is : tot * is;
}

namespace {

struct WgGeom { int ncp, ncot, ndyg, tr, ntap, tiles_x, tiles_y, ntiles, PB; size_t partial_floats, bias_floats; };

bool use_w1(int ksize, int cout) { return ksize == 1 && cout <= 32 * W1_NCOT; }

WgGeom wg_geom(int ksize, int N, int H, int W, int cin_chunks, int cout, int cus) {
    WgGeom g;
    if (use_w1(ksize, cout)) {
        g.tr = 1; g.ndyg = 1; g.ntap = 1;
        g.ncp = (cin_chunks + 1) / 2;
        g.ncot = (cout + 31) / 32;
        g.tiles_x = (W + 31) / 32;
        g.tiles_y = (H + W1_TH - 1) / W1_TH;
        g.ntiles = g.tiles_x * g.tiles_y * N;
        const int cgroups = (g.ncp + W1_NCPB - 1) / W1_NCPB;
        int pb = (cus > 0 ? cus : 256) / cgroups;
        if (pb < 1) pb = 1;
        if (pb > g.ntiles) pb = g.ntiles;
        g.PB = pb;
        g.partial_floats = (size_t)g.ncp * g.ncot * pb * 1024;
        g.bias_floats = (size_t)g.ncot * pb * 32;
        return g;
    }
    g.tr = (ksize == 5) ? 1 : ksize;
    g.ndyg = ksize / g.tr;
    g.ntap = g.tr * ksize;
    g.ncp = (cin_chunks + 1) / 2;
    g.ncot = (cout + 31) / 32;
    g.tiles_x = (W + 31) / 32;
    g.tiles_y = (H + 7) / 8;
    g.ntiles = g.tiles_x * g.tiles_y * N;
    const int groups = g.ncp * g.ncot * g.ndyg;
    int pb = (2 * (cus > 0 ? cus : 256)) / groups;     // floor: never one straggler workgroup in an extra round
    if (pb < 1) pb = 1;
    if (pb > g.ntiles) pb = g.ntiles;
    g.PB = pb;
    g.partial_floats = (size_t)groups * pb * g.ntap * 1024;
    g.bias_floats = (size_t)g.ncot * pb * 32;
    return g;
}

template <int KS, int TR, int NT>
int launch_wg(const WgradKArgs& a, const WgGeom& g, hipStream_t s) {
    using C = WgCfg<KS, TR, NT>;
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = bh_set_max_lds(&wgrad_mfma_kernel<KS, TR, NT>, C::LDS_BYTES, lds_set)) return rc;
    dim3 grid((unsigned)g.PB, (unsigned)g.ncp, (unsigned)(g.ncot * g.ndyg));
    wgrad_mfma_kernel<KS, TR, NT><<<grid, dim3(256), C::LDS_BYTES, s>>>(a);
    BH_CHECK_LAUNCH();
    return 0;
}

template <int KS, int TR, int NT>
int launch_wg_sb(const WgradKArgs& a, const WgGeom& g, hipStream_t s) {
    using C = WgCfg<KS, TR, NT>;
    constexpr int LDS = C::BUF_BYTES > 16384 ? C::BUF_BYTES : 16384;
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = bh_set_max_lds(&wgrad_mfma_sb_kernel<KS, TR, NT>, LDS, lds_set)) return rc;
    dim3 grid((unsigned)g.PB, (unsigned)g.ncp, (unsigned)(g.ncot * g.ndyg));
    wgrad_mfma_sb_kernel<KS, TR, NT><<<grid, dim3(256), LDS, s>>>(a);
    BH_CHECK_LAUNCH();
    return 0;
}

#if BINHIP_TUNING
int g_wg_dbg = 0;      // side builds only: ablation switches (1 skip DMA, 2 skip MFMA, 4 skip reduce/store, 16 double-buffered 3x3)
#define WG_DBG g_wg_dbg
#else
#define WG_DBG 0
#endif
// CU count of the current device (sizes the pixel-block split); looked up per call — no cached global
int cus() {
    const int n = binhip_device_cus();
    return n > 0 ? n : 256;
}

}  // namespace

extern "C" {

#if BINHIP_TUNING
int binhip_wgrad_set_debug(int flags) { g_wg_dbg = flags; return 0; }
#endif

size_t binhip_wgrad_workspace_bytes(int ksize, int N, int H, int W, int cin_chunks, int cout) {
    if (N <= 0 || H <= 0 || W <= 0 || cin_chunks <= 0 || cout <= 0) return 0;
    const WgGeom g = wg_geom(ksize, N, H, W, cin_chunks, cout, cus());
    return (g.partial_floats + g.bias_floats) * sizeof(float) + 256;
}

}  // extern "C"

// main kernel of one layer's weight gradient: writes the per-block partials into `workspace` and fills `out` for the
// (batched) reduction
int bh_wgrad_partials(const BinConvDesc* d, const void* x_hi, const void* x_lo, const void* gy_hi, const void* gy_lo,
                      void* workspace, size_t workspace_bytes, float* dw_oihw, float* dbias, int cin, int shuffle_perm,
                      BhWgradReduce* out, void* stream) {
    if (!d || !x_hi || !gy_hi || !workspace || !dw_oihw) return BINHIP_E_ARG;
    if (d->nterms != 1 && d->nterms != 3) return BINHIP_E_ARG;
    if (d->nterms == 3 && (!x_lo || !gy_lo)) return BINHIP_E_ARG;
    if (d->ksize != 1 && d->ksize != 3 && d->ksize != 5) return BINHIP_E_SHAPE;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->cin_chunks <= 0 || d->cout <= 0) return BINHIP_E_SHAPE;
    if ((long long)d->N * d->H * d->W >= (1ll << 26)) return BINHIP_E_SHAPE;
    if (cin <= 0 || cin > d->cin_chunks * 16) return BINHIP_E_SHAPE;
    if (shuffle_perm && d->cout % 4) return BINHIP_E_SHAPE;
    const WgGeom g = wg_geom(d->ksize, d->N, d->H, d->W, d->cin_chunks, d->cout, cus());
    const size_t need = (g.partial_floats + g.bias_floats) * sizeof(float) + 256;
    if (workspace_bytes < need) return BINHIP_E_WORKSPACE;
    float* part = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    WgradKArgs a;
    a.x_hi = (const _Float16*)x_hi; a.x_lo = (const _Float16*)x_lo;
    a.g_hi = (const _Float16*)gy_hi; a.g_lo = (const _Float16*)gy_lo;
    a.partial = part; a.partial_b = part + g.partial_floats;
    a.x_group_stride = d->x_group_stride; a.x_cpg = d->x_cpg;
    a.N = d->N; a.H = d->H; a.W = d->W;
    a.cin_chunks = d->cin_chunks; a.cout_chunks = (d->cout + 15) / 16;
    a.tiles_x = g.tiles_x; a.tiles_y = g.tiles_y; a.ntiles = g.ntiles;
    a.PB = g.PB; a.ncp = g.ncp; a.ncot = g.ncot;
    a.dbg = WG_DBG & 15;
    hipStream_t s = (hipStream_t)stream;
    int rc = BINHIP_E_SHAPE;
    if (use_w1(d->ksize, d->cout)) {
        const int cgroups = (g.ncp + W1_NCPB - 1) / W1_NCPB;
        dim3 grid((unsigned)g.PB, (unsigned)cgroups);
        if (d->nterms == 1) {
            static std::atomic<unsigned long long> set1{0};
            if (int rc1 = bh_set_max_lds(&wgrad1x1_kernel<1>, W1Cfg<1>::LDS_BYTES, set1)) return rc1;
            wgrad1x1_kernel<1><<<grid, dim3(256), W1Cfg<1>::LDS_BYTES, s>>>(a);
        } else {
            static std::atomic<unsigned long long> set3{0};
            if (int rc3 = bh_set_max_lds(&wgrad1x1_kernel<3>, W1Cfg<3>::LDS_BYTES, set3)) return rc3;
            wgrad1x1_kernel<3><<<grid, dim3(256), W1Cfg<3>::LDS_BYTES, s>>>(a);
        }
        BH_CHECK_LAUNCH();
        rc = 0;
    } else if (d->ksize == 3 && !(WG_DBG & 16)) {   // default: lean 2-workgroup/CU kernel; flag 16 = the double-buffered one
        rc = (d->nterms == 1) ? launch_wg_sb<3, 3, 1>(a, g, s) : launch_wg_sb<3, 3, 3>(a, g, s);
    } else if (d->nterms == 1) {
        if (d->ksize == 3) rc = launch_wg<3, 3, 1>(a, g, s);
        else if (d->ksize == 1) rc = launch_wg<1, 1, 1>(a, g, s);
        else rc = launch_wg<5, 1, 1>(a, g, s);
    } else {
        if (d->ksize == 3) rc = launch_wg<3, 3, 3>(a, g, s);
        else if (d->ksize == 1) rc = launch_wg<1, 1, 3>(a, g, s);
        else rc = launch_wg<5, 1, 3>(a, g, s);
    }
    if (rc) return rc;
    out->partial = a.partial; out->partial_b = a.partial_b;
    out->PB = g.PB; out->ncp = g.ncp; out->ncot = g.ncot; out->ks = d->ksize; out->tr = g.tr;
    out->cout = d->cout; out->cin = cin; out->shuffle = shuffle_perm;
    out->dw = dw_oihw; out->db = dbias;
    out->nrows = (long long)g.ndyg * g.ncot * g.ncp * g.ntap * 32;
    return 0;
}

int bh_wgrad_reduce_batch(const BhWgradReduce* items, int n, const float* inv_scale, int accumulate, void* stream) {
    if (n <= 0) return 0;
    if (n > BH_WGRAD_BATCH) return BINHIP_E_ARG;
    ReduceBatch rb;
    long long rows = 0;
    for (int i = 0; i < n; ++i) {
        rb.it[i] = items[i];
        rb.row_start[i] = rows;
        rows += items[i].nrows + items[i].ncot;
    }
    for (int i = n; i <= BH_WGRAD_BATCH; ++i) rb.row_start[i] = rows;
    for (int i = n; i < BH_WGRAD_BATCH; ++i) rb.it[i] = items[0];
    rb.n = n;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, rb, inv_scale, accumulate);
    BH_CHECK_LAUNCH();
    return 0;
}

extern "C" {

int binhip_conv2d_bwd_weight(const BinConvDesc* d, const void* x_hi, const void* x_lo, const void* gy_hi,
                             const void* gy_lo, const float* inv_scale, void* workspace, size_t workspace_bytes,
                             float* dw_oihw, float* dbias, int cin, int shuffle_perm, int accumulate, void* stream) {
    BhWgradReduce r;
    if (int rc = bh_wgrad_partials(d, x_hi, x_lo, gy_hi, gy_lo, workspace, workspace_bytes, dw_oihw, dbias, cin,
                                   shuffle_perm, &r, stream)) return rc;
    return bh_wgrad_reduce_batch(&r, 1, inv_scale, accumulate, stream);
}

}  // extern "C"
