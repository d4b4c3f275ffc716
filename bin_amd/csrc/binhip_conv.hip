// binhip_conv.hip — im2col-free direct convolution on MI355X (gfx950) matrix cores.
//
// Stands in for every F.conv2d on the bin_stage4 path (reference models/archs/RDN.py:141 RDB_Conv,
// :162 LFF, :187-188 SFENet1/2, :199-200 GFF, :205-207 UPNet) with the surrounding elementwise work
// fused into the epilogue: bias, ReLU + "cat" (RDN.py:145-147: the 32 new channels are simply the
// next two chunk planes of the dense block), residual add (RDN.py:165, :219), PixelShuffle(2)
// (RDN.py:206) and the input-mean skip (RDN.py:221/279/333).
//
// Formulation: D[cout][pixel] += W[cout][tap][cin16] * X[cin16][pixel+tap] with
// v_mfma_f32_32x32x16_f16 (A = weights, B = activations, fp32 accumulate).  A workgroup (4, 8 or 16
// wave64) owns a TH x 32 pixel tile and COUTB output channels; per K-stage it DMA-copies
// (buffer_load ... lds, 16 B/lane, zero-fill outside the image = the conv's zero padding) the
// (TH+k-1) x (32+k-1) x 16-channel input patch and the k*k x COUTB x 16 weight slab into LDS,
// double-buffered, then every wave runs k*k*MT*R MFMAs straight out of LDS with ds_read_b128: the
// patch image is [channel half][pixel][16 B] (a 16-lane read group = 256 contiguous bytes, one per-lane
// base + immediates), the weight slab keeps 16-byte slots XOR-swizzled by (row>>3)&1 from the relayout.
// Precision: NT=1 -> one fp16 product; NT=3 -> hi/lo split, Ah*Bh + Al*Bh + Ah*Bl (fp32 class).
#include "binhip_conv_common.h"
#include <vector>
#include <cstdio>
#include <cstdlib>

template <int KS, int MT, int WM, int R, int WN, int KC, int NT, int NBUF, int EPI>
struct ConvCfg {
    static constexpr int PAD = KS / 2;
    static constexpr int COUTB = 32 * MT * WM;
    static constexpr int TH = R * WN;
    static constexpr int PH = TH + KS - 1;
    static constexpr int PW = 32 + KS - 1;
    static constexpr int PP = (PH * PW * 2 + 63) / 64;   // 1-KiB DMA pieces of the patch, per chunk
    static constexpr int WP = KS * KS * MT * WM;         // 1-KiB pieces of the weight slab, per chunk
    static constexpr int CHUNK_BYTES = (PP + WP) * 1024;
    static constexpr int NPL = (NT == 3) ? 2 : 1;
    static constexpr int PLANE_BYTES = KC * CHUNK_BYTES;
    static constexpr int BUF_BYTES = NPL * PLANE_BYTES;
    static constexpr int NW = WM * WN;                   // waves per workgroup (4 or 8)
    static constexpr int LDS_BYTES = NBUF * BUF_BYTES + 1024;   // + 1 KiB dummy DMA target
    static constexpr int NPJ = (PP + NW - 1) / NW;
    static constexpr int NWJ = (WP + NW - 1) / NW;
    static constexpr int PS = NPL * KC * (NPJ + NWJ);    // DMA instructions per wave per full stage
    static_assert(NW == 4 || NW == 8 || NW == 16, "4, 8 or 16 waves per workgroup");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(NBUF <= 2 || (NBUF - 2) * PS <= 63, "vmcnt immediate range");
};

// BINHIP_ABLATE (side builds only; 0 in the product): timing-only variants of the K-loop — 1 no MFMA (fragment loads
// kept alive), 2 no fragment loads (MFMA on undefined registers), 3 no per-stage barrier, 4 no DMA instructions.
// Results are garbage by construction (round-1 findings: profiles/r01_layer_variants.md).
#ifndef BINHIP_ABLATE
#define BINHIP_ABLATE 0
#endif
// runtime ablation switches exist in BINHIP_TUNING side builds only; in the product the tests fold to `false`
#if BINHIP_TUNING
#define BH_DBG(a, bit) ((a).dbg & (bit))
#else
#define BH_DBG(a, bit) false
#endif
__device__ __forceinline__ half8 lds_ld8(const char* p) {
#if BINHIP_ABLATE == 2
    half8 v;
    asm volatile("" : "=v"(v));
    return v;
#else
    return *reinterpret_cast<const half8*>(p);
#endif
}
__device__ __forceinline__ floatx16 mfma16(half8 a, half8 b, floatx16 c) {
#if BINHIP_ABLATE == 1
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

// Issue the LDS-DMA of K-stage `st` (KC chunks x NPL planes: input patch + weight slab) into buffer `buf`.
// Every wave issues exactly NPJ + NWJ instructions per (plane, chunk) so that s_waitcnt vmcnt(N) can count
// whole stages; surplus lanes read out of range (zero fill, no memory traffic) into a dummy 1-KiB LDS area.
template <class C, int KS, int KC>
__device__ __forceinline__ void issue_stage(const ConvKArgs& a, char* smem, int st, int buf, int wave, int lane, int z,
                                            const unsigned* voff, long long plane_elems, unsigned plane_bytes) {
    char* bbase = smem + buf * C::BUF_BYTES;
    char* dummy = smem + (C::LDS_BYTES - 1024);
    const int nchunks = a.nchunks;
#pragma unroll
    for (int pl = 0; pl < C::NPL; ++pl) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int c = st * KC + kc;
            if (c < nchunks) {
                const _Float16* xb = pl ? a.x_lo : a.x_hi;
                const long long coff = (a.cpg > 0)
                    ? (long long)(c / a.cpg) * a.group_stride + (long long)(c % a.cpg) * plane_elems
                    : (long long)c * plane_elems;
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(xb + coff), 0, plane_bytes, 0x00020000);
                char* lds = bbase + pl * C::PLANE_BYTES + kc * C::CHUNK_BYTES;
#pragma unroll
                for (int j = 0; j < C::NPJ; ++j) {
                    const int i = wave + C::NW * j;
                    const bool real = ((C::PP % C::NW == 0) || (i < C::PP)) && !BH_DBG(a, 2);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(real ? lds + i * 1024 : dummy), 16,
                                                             real ? voff[j] : 0x80000000u, 0, 0, 0);
                }
                const _Float16* wb = (pl ? a.w_lo : a.w_hi) +
                                     ((long long)z * nchunks + c) * (KS * KS * C::COUTB * 16);
                __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)wb, 0, KS * KS * C::COUTB * 32, 0x00020000);
#pragma unroll
                for (int j = 0; j < C::NWJ; ++j) {
                    const int i = wave + C::NW * j;
                    const bool real = ((C::WP % C::NW == 0) || (i < C::WP)) && !BH_DBG(a, 1);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ws, (lds_void_t*)(real ? lds + (C::PP + i) * 1024 : dummy), 16,
                                                             real ? (unsigned)(lane * 16) : 0x80000000u,
                                                             real ? i * 1024 : 0, 0, 0);
                }
            }
        }
    }
}

// Operand fragments of one (chunk, dx) step: R+KS-1 patch rows (B) and the KS x MT weight tiles of this tap column (A).
template <class C, int KS, int MT, int R, int NT>
__device__ __forceinline__ void load_frags(const char* bbase, int kc, int dx, int wm, int a_lane_off, int b_lane_p, int kg,
                                           half8 (&Bh)[R + KS - 1], half8 (&Bl)[R + KS - 1],
                                           half8 (&Ah)[KS][MT], half8 (&Al)[KS][MT]) {
    const char* pb = bbase + kc * C::CHUNK_BYTES;
    const char* wb = pb + C::PP * 1024 + wm * (MT * 32 * 32);
#pragma unroll
    for (int rr = 0; rr < R + KS - 1; ++rr) {
        const int off = (kg * (C::PH * C::PW) + b_lane_p) * 16 + (rr * C::PW + dx) * 16;
        Bh[rr] = lds_ld8(pb + off);
        if constexpr (NT == 3) Bl[rr] = lds_ld8(pb + C::PLANE_BYTES + off);
    }
#pragma unroll
    for (int dy = 0; dy < KS; ++dy)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int off = ((dy * KS + dx) * C::COUTB + mt * 32) * 32 + a_lane_off;
            Ah[dy][mt] = lds_ld8(wb + off);
            if constexpr (NT == 3) Al[dy][mt] = lds_ld8(wb + C::PLANE_BYTES + off);
        }
}

// All MFMAs of K-stage `st` out of LDS buffer `buf`, software-pipelined over its KC*KS (chunk, dx) steps: the
// fragments of step s+1 are fetched into the other register set before the MFMAs of step s issue.  The
// sched_barriers pin that order — left alone, the scheduler sinks every ds_read next to its MFMA to save registers
// (it assumes 8 waves/SIMD because the LDS size is dynamic) and the loop degenerates into read -> wait -> MFMA.
template <class C, int KS, int MT, int R, int KC, int NT>
__device__ __forceinline__ void compute_stage(const char* smem, int st, int buf, int nchunks, int wm,
                                              int a_lane_off, int b_lane_p, int kg, floatx16 (&acc)[MT][R]) {
    const char* bbase = smem + buf * C::BUF_BYTES;
    constexpr int NSTEP = KC * KS;
    // register estimate: accumulators + two fragment sets; fall back to one set (no prefetch) when it would spill
    constexpr int FRAG_REGS = ((R + KS - 1) + KS * MT) * 4 * ((NT == 3) ? 2 : 1);
    constexpr bool PIPE = (MT * R * 16 + 2 * FRAG_REGS + 40) <= 232;
    constexpr int NSET = PIPE ? 2 : 1;
    half8 Bh[NSET][R + KS - 1], Bl[NSET][R + KS - 1], Ah[NSET][KS][MT], Al[NSET][KS][MT];
    int nvalid = NSTEP;
    if constexpr (KC > 1) {
        const int left = nchunks - st * KC;
        nvalid = (left < KC ? left : KC) * KS;
    }
    if constexpr (PIPE)
        load_frags<C, KS, MT, R, NT>(bbase, 0, 0, wm, a_lane_off, b_lane_p, kg, Bh[0], Bl[0], Ah[0], Al[0]);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        if (KC == 1 || s < nvalid) {
            if constexpr (PIPE) {
                if (s + 1 < NSTEP && (KC == 1 || s + 1 < nvalid))
                    load_frags<C, KS, MT, R, NT>(bbase, (s + 1) / KS, (s + 1) % KS, wm, a_lane_off, b_lane_p, kg,
                                                 Bh[(s + 1) & 1], Bl[(s + 1) & 1], Ah[(s + 1) & 1], Al[(s + 1) & 1]);
            } else {
                load_frags<C, KS, MT, R, NT>(bbase, s / KS, s % KS, wm, a_lane_off, b_lane_p, kg, Bh[0], Bl[0], Ah[0], Al[0]);
            }
            if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);   // (single-set kernels: leave the interleaving to hipcc)
#pragma unroll
            for (int dy = 0; dy < KS; ++dy)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if constexpr (NT == 3) {
                            acc[mt][r] = mfma16(Al[s & (NSET - 1)][dy][mt], Bh[s & (NSET - 1)][r + dy], acc[mt][r]);
                            acc[mt][r] = mfma16(Ah[s & (NSET - 1)][dy][mt], Bl[s & (NSET - 1)][r + dy], acc[mt][r]);
                        }
                        acc[mt][r] = mfma16(Ah[s & (NSET - 1)][dy][mt], Bh[s & (NSET - 1)][r + dy], acc[mt][r]);
                    }
            if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// XTRA: the epilogue also reads residual / accumulator / ReLU-mask planes (backward-data layers, layers with a skip
// connection); `bias`: the layer's bias again as the kernel's own restrict parameter (scalar loads, binhip_conv_common.h)
// (at least two waves per SIMD = at most 256 registers: with no waits left between the epilogue's stores the scheduler would
//  otherwise overlap more of its iterations and take the 4-wave 3-tile kernels from 240 to 272 registers — one workgroup per CU)
template <int KS, int MT, int WM, int R, int WN, int KC, int NT, int NBUF, int EPI, bool XTRA>
__global__ void __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2)))
conv_mfma_kernel(const ConvKArgs a, const float* __restrict__ bias) {
    using C = ConvCfg<KS, MT, WM, R, WN, KC, NT, NBUF, EPI>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int n = lane & 31;     // pixel column (B/N index) and cout row (A/M index) of this lane
    const int kg = lane >> 5;    // which 8-channel half of the 16-channel chunk

    if (BH_DBG(a, 16)) return;   // timing experiments: empty kernel (launch + boundary only)
    // 1-D grid of tiles x output-channel blocks, block index fastest: the workgroups that share an input patch are
    // neighbours on ONE XCD after the banding and fetch it into that L2 once (as a (tiles, blocks) grid every block was a
    // separate sweep: GFF.0's backward-data read its 214 MB gradient six times from HBM, profiles/r02_train_kernel_stats.md)
    int bid = blockIdx.x;
    if (a.xcd_remap) bid = xcd_band(bid, gridDim.x);
    const int z = bid % a.ncol;
    bid /= a.ncol;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int tx0 = tx * 32, ty0 = ty * C::TH;
    const int H = a.H, W = a.W;
    const long long plane_elems = (long long)a.N * H * W * 16;
    const unsigned plane_bytes = (unsigned)(plane_elems * 2);

    // ---- per-lane source offsets of the patch DMA pieces (stage independent) -------------------
    unsigned voff[C::NPJ];
#pragma unroll
    for (int j = 0; j < C::NPJ; ++j) {
        const int i = wave + C::NW * j;
        const int q = i * 64 + lane;          // 16-byte slot index in the LDS patch image
        // image = [channel half cg][patch pixel p][16 B]: a 16-lane ds_read_b128 group reads 16 consecutive pixels of
        // one half = 256 contiguous bytes (every bank once), and a fragment address is ONE per-lane base + an
        // immediate (row, dx) offset
        const int cg = q >= C::PH * C::PW ? 1 : 0;
        const int p = q - cg * (C::PH * C::PW);
        const int py = p / C::PW;
        const int px = p - py * C::PW;
        const int gy = ty0 + py - C::PAD;
        const int gx = tx0 + px - C::PAD;
        const bool ok = (p < C::PH * C::PW) && (gy >= 0) && (gy < H) && (gx >= 0) && (gx < W);
        voff[j] = ok ? (unsigned)((((long long)img * H + gy) * W + gx) * 32 + cg * 16) : 0x80000000u;
    }

    floatx16 acc[MT][R];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][r][e] = 0.f;

    const int nchunks = a.nchunks;
    const int nst = (nchunks + KC - 1) / KC;
    const int a_lane_off = n * 32 + ((kg ^ ((n >> 3) & 1)) << 4);
    const int b_lane_p = wn * R * C::PW + n;

    if constexpr (NBUF >= 2) {
        // ring of NBUF stage buffers; up to NBUF-1 stages of DMA in flight, counted waits, one barrier per stage
        const int nfull = nchunks / KC;       // stages that issue the full C::PS instructions per wave
#pragma unroll
        for (int s0 = 0; s0 < NBUF - 1; ++s0)
            if (s0 < nst && BINHIP_ABLATE != 4) issue_stage<C, KS, KC>(a, smem, s0, s0, wave, lane, z, voff, plane_elems, plane_bytes);
        int cur = 0, nxt = NBUF - 1;
        for (int st = 0; st < nst; ++st) {
            // younger full-size stages still allowed in flight while stage st must have landed
            int yf = nfull - 1 - st;
            yf = yf < 0 ? 0 : (yf > NBUF - 2 ? NBUF - 2 : yf);
            if (NBUF >= 4 && yf >= 2) wait_vmcnt<(NBUF >= 4 ? 2 : 0) * C::PS>();
            else if (NBUF >= 3 && yf >= 1) wait_vmcnt<(NBUF >= 3 ? 1 : 0) * C::PS>();
            else wait_vmcnt<0>();
#if BINHIP_ABLATE != 3
            __builtin_amdgcn_s_barrier();
#endif
            __builtin_amdgcn_sched_barrier(0);
#if BINHIP_ABLATE != 4
            if (st + NBUF - 1 < nst)
                issue_stage<C, KS, KC>(a, smem, st + NBUF - 1, nxt, wave, lane, z, voff, plane_elems, plane_bytes);
#endif
            compute_stage<C, KS, MT, R, KC, NT>(smem, st, cur, nchunks, wm, a_lane_off, b_lane_p, kg, acc);
            cur = (cur + 1 == NBUF) ? 0 : cur + 1;
            nxt = (nxt + 1 == NBUF) ? 0 : nxt + 1;
        }
    } else {
        for (int st = 0; st < nst; ++st) {
            issue_stage<C, KS, KC>(a, smem, st, 0, wave, lane, z, voff, plane_elems, plane_bytes);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute_stage<C, KS, MT, R, KC, NT>(smem, st, 0, nchunks, wm, a_lane_off, b_lane_p, kg, acc);
            __syncthreads();
        }
    }

    // ---- epilogue (binhip_conv_common.h) ----------------------------------------------------------
    if (BH_DBG(a, 8)) { if (acc[0][0][0] == 12345.f) a.y_hi[0] = (_Float16)1.f; return; }   // timing: no epilogue
    conv_epilogue<MT, R, NT, EPI, XTRA>(a, bias, acc, img, ty0 + wn * R, tx0, z * C::COUTB + wm * MT * 32, z == 0 && wm == 0, n, kg,
                                  plane_elems);
}

// ---------------------------------------------------------------------------------------------------
template <int KS, int MT, int WM, int R, int WN, int KC, int NT, int NBUF, int EPI, bool XTRA>
static int launch_cfg_x(const ConvKArgs& ka, int cout_pad, hipStream_t s) {
    using C = ConvCfg<KS, MT, WM, R, WN, KC, NT, NBUF, EPI>;
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = bh_set_max_lds(&conv_mfma_kernel<KS, MT, WM, R, WN, KC, NT, NBUF, EPI, XTRA>, C::LDS_BYTES, lds_set)) return rc;
    ConvKArgs a = ka;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + C::TH - 1) / C::TH;
    a.ncol = cout_pad / C::COUTB;
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.N * a.ncol));
    conv_mfma_kernel<KS, MT, WM, R, WN, KC, NT, NBUF, EPI, XTRA><<<grid, dim3(64 * C::NW), C::LDS_BYTES, s>>>(a, a.bias);
    BH_CHECK_LAUNCH();
    return 0;
}
// the epilogue with residual / accumulator / mask reads only where a call has them (plane epilogue)
template <int KS, int MT, int WM, int R, int WN, int KC, int NT, int NBUF, int EPI>
static int launch_cfg(const ConvKArgs& ka, int cout_pad, hipStream_t s) {
    if constexpr (EPI == BINHIP_EPI_PLANES) {
        if (ka.has_res || ka.r2_hi || ka.m_hi || ka.y_unshuf)
            return launch_cfg_x<KS, MT, WM, R, WN, KC, NT, NBUF, EPI, true>(ka, cout_pad, s);
    }
    return launch_cfg_x<KS, MT, WM, R, WN, KC, NT, NBUF, EPI, false>(ka, cout_pad, s);
}

// ---- optional live timing of ONE kernel class with HIP events on the launch stream ---------------------
// (bench.py's roofline leg: average duration of the dominant kernel inside the timed region).  An explicit handle
// the caller creates and passes with each plan — the library keeps no process-global state.
struct BinhipProfiler {
    int ks = 0, cout_pad = 0, epi = 0;
    std::vector<hipEvent_t> ev;   // pairs: start, stop
    size_t used = 0;
};
namespace { constexpr size_t PROF_MAX_PAIRS = 16384; }

#if BINHIP_TUNING
static int g_x3_wide = 1;     // side builds: 0 = round-1 routing of the wide nterms = 3 3x3 layers (generic kernel, 64/96-row blocks)
#define BH_X3_WIDE g_x3_wide
#else
#define BH_X3_WIDE 1
#endif

int bh_conv_cout_block(int ksize, int cout_pad, int nterms) {
    if (cout_pad <= 0 || cout_pad % 32) return -1;
    if (ksize == 3 && nterms == 3 && BH_X3_WIDE) return 32;  // every fp32-class 3x3 conv: plane-split kernel, 32-row columns
    if (cout_pad == 256) return nterms == 3 ? 64 : 128;     // UPNet.0 (PixelShuffle epilogue)
    if (ksize == 5) return 32;
    if (ksize == 1 && cout_pad == 224) return 224;            // LFF backward-data: all 224 rows in one workgroup column
    if (ksize == 1 && cout_pad == 1152) return 192;           // GFF.0 backward-data: 6 columns instead of 12
    if (cout_pad % 96 == 0) return 96;
    if (ksize == 3 && cout_pad % 64 == 0) return 64;
    return 32;
}

static int bh_dispatch_conv(const ConvKArgs& a, int k, int cp, int nt, int e, hipStream_t s);
int bh_launch_conv_x3(const ConvKArgs& a, int cout_pad, int epilogue, hipStream_t s);   // binhip_conv_x3.hip
int bh_launch_final_dot2(const ConvKArgs& a, int nterms, hipStream_t s);                 // binhip_conv_x3.hip
int bh_launch_conv_x3_k5(const ConvKArgs& a, int cout_pad, hipStream_t s);               // binhip_conv_x3.hip
int bh_launch_conv_x3_k5_subpix(const ConvKArgs& a, hipStream_t s);                      // binhip_conv_x3.hip
int bh_launch_final_m16(const ConvKArgs& a, hipStream_t s);                              // binhip_conv_x3.hip
#ifndef BINHIP_LFFD_EPI
#define BINHIP_LFFD_EPI 1     // 0 (side builds): the LFF backward-data tile on the generic extras grouping (rounds 2-3)
#endif
#ifndef BINHIP_FINAL_M16
#define BINHIP_FINAL_M16 1    // 0 (side builds): UPNet.2 of the fp32-class mode on the 32-row tile of conv_x3_kernel (rounds 2-3)
#endif
#ifndef BINHIP_K5_PAIR
#define BINHIP_K5_PAIR 1      // 0 (side builds): ignore BINHIP_CONV_HALF_LAST_CHUNK (SFENet1's last chunk on the plain 25-tap loop)
#endif
#ifndef BINHIP_K5_X3
#define BINHIP_K5_X3 1        // 0 (side builds): the 5x5 layers of the fp32-class mode on the generic single-buffered kernel (rounds 1-3)
#endif

// validate a call and fill the kernel argument block (everything but the tile counts, which the launcher of the chosen
// tile shape sets)
int bh_prepare_conv(const BhConvCall& c, ConvKArgs* out) {
    const BinConvDesc& d = c.d;
    if (!c.x_hi || !c.w_hi || !c.bias) return BINHIP_E_ARG;
    if (d.nterms != 1 && d.nterms != 3) return BINHIP_E_ARG;
    if (d.nterms == 3 && (!c.x_lo || !c.w_lo)) return BINHIP_E_ARG;
    if (d.N <= 0 || d.H <= 0 || d.W <= 0 || d.cin_chunks <= 0) return BINHIP_E_SHAPE;
    if ((long long)d.N * d.H * d.W >= (1ll << 26)) return BINHIP_E_SHAPE;   // plane < 2 GiB (DMA OOB sentinel)
    ConvKArgs a;
    a.x_hi = (const _Float16*)c.x_hi; a.x_lo = (const _Float16*)c.x_lo;
    a.w_hi = (const _Float16*)c.w_hi; a.w_lo = (const _Float16*)c.w_lo;
    a.bias = c.bias;
    a.y_hi = (_Float16*)c.y_hi; a.y_lo = (_Float16*)c.y_lo;
    a.r_hi = (const _Float16*)c.r_hi; a.r_lo = (const _Float16*)c.r_lo;
    a.r2_hi = (const _Float16*)c.r2_hi; a.r2_lo = (const _Float16*)c.r2_lo;
    a.m_hi = (const _Float16*)c.m_hi;
    a.flags = (unsigned*)c.status;
    a.res_chunks = c.res_chunks > 0 ? c.res_chunks : (1 << 30);
    a.mask_from = c.mask_from;
    a.y_cpg = c.y_cpg; a.y_group_stride = c.y_group_stride;
    a.y_cpg_inv = c.y_cpg > 0 ? (unsigned)(((1u << 20) + c.y_cpg - 1) / c.y_cpg) : 0u;
    if (c.y_cpg > 128 || (c.y_cpg > 0 && d.cout_pad / 16 >= 4096)) return BINHIP_E_SHAPE;
    a.half_last = ((d.reserved & BINHIP_CONV_HALF_LAST_CHUNK) && BINHIP_K5_PAIR) ? 1 : 0;
    a.y_unshuf = c.y_unshuf;
    if (c.y_unshuf && (d.epilogue != BINHIP_EPI_PLANES || c.y_cpg > 0 || (d.H & 1) || (d.W & 1) || c.y_unshuf * 16 < d.cout))
        return BINHIP_E_SHAPE;
    if (c.r2_hi && d.nterms == 3 && !c.r2_lo) return BINHIP_E_ARG;
    a.out_f32 = c.y_f32;
    for (int i = 0; i < 5; ++i) a.img[i] = c.images[i];
    a.group_stride = d.x_group_stride;
    a.N = d.N; a.H = d.H; a.W = d.W;
    a.nchunks = d.cin_chunks;
    a.cpg = d.x_cpg;
    a.relu = d.relu; a.has_res = (c.r_hi != nullptr); a.nimg = d.n_images; a.cout = d.cout;
    a.och_limit = (d.epilogue == BINHIP_EPI_PLANES) ? (d.cout + 15) / 16 : (1 << 30);
    a.tiles_x = a.tiles_y = 0;
    a.ncol = 1;
    a.xcd_remap = 1;
    a.dbg = 0;
    a.wt = 0;
    a.prog_prio = 0;
#if BINHIP_TIMELINE
    a.tl = nullptr; a.tl_launch = 0; a.tl_base = 0;
#endif
    const int P = BINHIP_EPI_PLANES, S = BINHIP_EPI_SHUFFLE, F = BINHIP_EPI_FINAL;
    if (d.epilogue == P) {
        if (!c.y_hi || (d.nterms == 3 && !c.y_lo)) return BINHIP_E_ARG;
        if (a.has_res && d.nterms == 3 && !c.r_lo) return BINHIP_E_ARG;
    } else if (d.epilogue == S) {
        if (!c.y_hi || (d.nterms == 3 && !c.y_lo)) return BINHIP_E_ARG;
        if (d.cout % 4) return BINHIP_E_SHAPE;
    } else if (d.epilogue == F) {
        if (!c.y_f32 || d.cout > 4 || d.n_images < 0 || d.n_images > 5) return BINHIP_E_ARG;
    } else if (d.epilogue == BINHIP_EPI_FINAL_SUBPIX) {      // the fused UPNet: 4 sub-pixel channels per colour, 5x5, one 32-row block
        if (!c.y_f32 || d.cout <= 0 || (d.cout & 3) || d.cout > 12 || d.n_images < 0 || d.n_images > 5) return BINHIP_E_ARG;
        if (d.ksize != 5 || d.cout_pad != 32) return BINHIP_E_SHAPE;
    } else {
        return BINHIP_E_ARG;
    }
    {   // 32-bit buffer offsets: write-through stores only while the whole output tensor stays below 4 GiB
        const int e = d.epilogue, cp = d.cout_pad;
        const long long out_chunks = (e == BINHIP_EPI_SHUFFLE) ? (a.cout / 4 + 15) / 16 : (cp + 15) / 16;
        const long long px = (long long)a.N * a.H * a.W * ((e == BINHIP_EPI_SHUFFLE) ? 4 : 1);
        const long long span = (a.y_cpg > 0 ? ((out_chunks + a.y_cpg - 1) / a.y_cpg) * a.y_group_stride * 2 : out_chunks * px * 32);
        // PLANES only: the PixelShuffle store (16 B per lane at a 64-B stride) relies on L2 to merge partial lines
        a.wt = (e == BINHIP_EPI_PLANES && span < (1ll << 32) - 64) ? 1 : 0;
    }
    *out = a;
    return 0;
}

// shared by the conv launcher and the RDN backward plan (weight-gradient launches: epi == BINHIP_PROF_WGRAD)
bool bh_prof_begin(BinhipProfiler* pr, int ks, int cout_pad, int epi, hipStream_t s) {
    if (!pr || ks != pr->ks || cout_pad != pr->cout_pad || epi != pr->epi || pr->used + 2 > pr->ev.size()) return false;
    (void)hipEventRecord(pr->ev[pr->used], s);
    return true;
}
void bh_prof_end(BinhipProfiler* pr, hipStream_t s) {
    (void)hipEventRecord(pr->ev[pr->used + 1], s);
    pr->used += 2;
}

int bh_launch_conv(const BhConvCall& c, hipStream_t s) {
    ConvKArgs a;
    if (int rc = bh_prepare_conv(c, &a)) return rc;
    const BinConvDesc& d = c.d;
    const int k = d.ksize, cp = d.cout_pad, nt = d.nterms, e = d.epilogue;
    if (bh_prof_begin(c.prof, k, cp, e, s)) {
        const int rc = bh_dispatch_conv(a, k, cp, nt, e, s);
        bh_prof_end(c.prof, s);
        return rc;
    }
    return bh_dispatch_conv(a, k, cp, nt, e, s);
}

// Kernel configuration per layer class: the measured-best tile shapes on MI355X at the 720p working size
// (profiles/r01_layer_variants.md, profiles/r02_*).  BINHIP_TUNING side builds (tools/) additionally compile the
// alternatives below and the process-global switches that pick them; the product library has neither.
enum { CLS_K3C32 = 0, CLS_K1C96 = 1, CLS_K3C96 = 2, CLS_SHUFFLE = 3, CLS_FINAL = 4, CLS_K5 = 5 };
#if BINHIP_TUNING
static int g_variant[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
static int g_xcd_remap = 1, g_dbg = 0, g_wt = 1;
#define BH_VARIANT(cls) g_variant[cls]
#else
#define BH_VARIANT(cls) (-1)
#endif

// waves (= tile rows) of the wide backward-data 1x1 tiles (LFF: 224 rows, GFF.0: 192): 8 = one workgroup per CU, 4 = two
// GFF.0 in the fp32-class mode (1x1, 1152 -> 96): rows per wave and waves of its tile (side builds; the product's 2 x 4 = 8 x 32 pixels)
#ifndef BINHIP_GFF0_R
#define BINHIP_GFF0_R 2
#endif
#ifndef BINHIP_GFF0_WN
#define BINHIP_GFF0_WN 4
#endif
#ifndef BINHIP_K1_DGRAD_WN
#define BINHIP_K1_DGRAD_WN 8
#endif
static int bh_dispatch_conv(const ConvKArgs& a0, int k, int cp, int nt, int e, hipStream_t s) {
    const int P = BINHIP_EPI_PLANES, S = BINHIP_EPI_SHUFFLE, F = BINHIP_EPI_FINAL;
    const int cb = bh_conv_cout_block(k, cp, nt);
    if (cb <= 0 || cp % cb) return BINHIP_E_SHAPE;
    ConvKArgs a = a0;
#if BINHIP_TUNING
    a.xcd_remap = g_xcd_remap;
    a.dbg = g_dbg;
    if (!g_wt) a.wt = 0;
#endif
    if (e == BINHIP_EPI_FINAL_SUBPIX)                                                // the fused UPNet (shape checked in bh_prepare_conv)
        return nt == 3 ? bh_launch_conv_x3_k5_subpix(a, s) : launch_cfg<5, 1, 1, 2, 8, 1, 1, 2, BINHIP_EPI_FINAL_SUBPIX>(a, cp, s);
    // UPNet.2 (64 -> 3 + mean of the frames) in the single-product mode: three output channels as VALU dot products instead of
    // a 32-row MFMA tile (f16 720p window 33.02 -> 32.71 ms).  In the fp32-class mode the same kernel needs 648 dot2 per lane
    // and chunk behind 112 scalar weight loads and measured 224 us against the MFMA kernel's 110 (profiles/r03_experiments.md):
    // -DBINHIP_DOT2_X3=1 side builds only.
#ifndef BINHIP_NO_DOT2        // (side builds: the round-2 MFMA path, for the A/B)
#ifndef BINHIP_DOT2_X3
#define BINHIP_DOT2_X3 0
#endif
    if (e == F && k == 3 && cp == 32 && a.cout <= 3 && (nt == 1 || BINHIP_DOT2_X3) && BH_VARIANT(CLS_FINAL) < 0)
        return bh_launch_final_dot2(a, nt, s);
#endif
    // UPNet.2 in the fp32-class mode: 16-row matrix tile over tap pairs (binhip_conv_x3.hip, round 4)
    // (its LDS-resident weight slab holds 40 taps = 4 input chunks, UPNet.2's 64 channels; a wider FINAL conv — only reachable
    //  through the per-op ABI — takes the 32-row tile below.  Round 5: the guard was `nchunks <= 5`, and 5 chunks silently
    //  multiplied taps 40-44 with zeros; found by tests/test_gpu_round5.py::test_final_m16_kernel_vs_float64[80-3])
    if (e == F && k == 3 && cp == 32 && a.cout <= 3 && nt == 3 && a.nchunks <= 4 && BINHIP_FINAL_M16 && BH_VARIANT(CLS_FINAL) < 0)
        return bh_launch_final_m16(a, s);
    //                                   KS MT WM R WN KC NT NBUF EPI
    if (nt == 1) {
        if (e == F && k == 3 && cp == 32) return launch_cfg<3, 1, 1, 2, 8, 1, 1, 2, F>(a, cp, s);     // 8 waves, 16x32 tile
        if (e == S && k == 3 && cp == 256) return launch_cfg<3, 2, 2, 4, 4, 1, 1, 2, S>(a, cp, s);    // 8 waves, 16x32 tile, 128 rows
        if (e == P && k == 3 && cb == 32) {
#if BINHIP_TUNING
            if (BH_VARIANT(CLS_K3C32) == 0) return launch_cfg<3, 1, 1, 4, 4, 1, 1, 2, P>(a, cp, s);    // 4 waves x 4 rows
            if (BH_VARIANT(CLS_K3C32) == 12) return launch_cfg<3, 1, 1, 2, 16, 1, 1, 2, P>(a, cp, s);  // 16 waves, 32x32 tile
#endif
            return launch_cfg<3, 1, 1, 2, 8, 1, 1, 2, P>(a, cp, s);                                     // 8 waves x 2 rows, 16x32 tile
        }
        if (e == P && k == 3 && cb == 64)  return launch_cfg<3, 2, 1, 2, 4, 1, 1, 2, P>(a, cp, s);
        if (e == P && k == 3 && cb == 96)  return launch_cfg<3, 3, 1, 2, 4, 1, 1, 2, P>(a, cp, s);
        if (e == P && k == 1 && cb == 224) return launch_cfg<1, 7, 1, 1, 8, 2, 1, 2, P>(a, cp, s);   // LFF dgrad, 8 waves x 1 row
        if (e == P && k == 1 && cb == 192) return launch_cfg<1, 6, 1, 1, 8, 2, 1, 2, P>(a, cp, s);   // GFF.0 dgrad
        if (e == P && k == 1 && cb == 32)  return launch_cfg<1, 1, 1, 4, 4, 4, 1, 2, P>(a, cp, s);
        if (e == P && k == 1 && cb == 96) {
            // short K (LFF, LFF dgrad): 8 waves x 1 row, 38.6 vs 49.6 us; long K (GFF.0, 72 chunks): the 4-wave tile
            if (a.nchunks >= 32) return launch_cfg<1, 3, 1, 2, 4, 2, 1, 2, P>(a, cp, s);
            return launch_cfg<1, 3, 1, 1, 8, 2, 1, 2, P>(a, cp, s);
        }
        if (e == P && k == 5 && cb == 32)  return launch_cfg<5, 1, 1, 2, 8, 1, 1, 2, P>(a, cp, s);   // 8 waves, 16x32 tile
    } else {
        if (k == 3 && cb == 32) {
#if BINHIP_TUNING
            // the generic both-planes-per-stage kernel (117 KB LDS, 157 VGPRs, 1 workgroup/CU): round-1 default
            if (BH_VARIANT(e == F ? CLS_FINAL : CLS_K3C32) == 1 && e != S)
                return e == F ? launch_cfg<3, 1, 1, 2, 8, 1, 3, 2, F>(a, cp, s) : launch_cfg<3, 1, 1, 2, 8, 1, 3, 2, P>(a, cp, s);
#endif
            return bh_launch_conv_x3(a, cp, e, s);      // plane-split stages, 2 workgroups/CU (binhip_conv_x3.hip)
        }
        if (e == S && k == 3 && cp == 256) return launch_cfg<3, 1, 2, 4, 2, 1, 3, 2, S>(a, cp, s);
        if (e == P && k == 3 && cb == 64)  return launch_cfg<3, 2, 1, 2, 4, 1, 3, 2, P>(a, cp, s);
        if (e == P && k == 3 && cb == 96)  return launch_cfg<3, 3, 1, 1, 8, 1, 3, 2, P>(a, cp, s);   // 8 waves x 1 row: 168 vs 184 us
        if (e == P && k == 1 && cb == 224) {   // LFF backward-data
            // bin_stage4's own pattern (residual gy on output chunks 0-5, ReLU mask from chunk 12, one output group): the
            // instantiation whose epilogue fetches its extras in two groups instead of four (binhip_conv_common.h)
            if (a.has_res && a.m_hi && !a.r2_hi && a.res_chunks == 6 && a.mask_from == 12 && a.y_cpg <= 0 && !a.y_unshuf &&
                a.och_limit >= 14 && cp == 224 && BINHIP_LFFD_EPI)
                return launch_cfg_x<1, 7, 1, 1, BINHIP_K1_DGRAD_WN, 1, 3, 2, BINHIP_EPI_PLANES_LFFD, true>(a, cp, s);
            return launch_cfg<1, 7, 1, 1, BINHIP_K1_DGRAD_WN, 1, 3, 2, P>(a, cp, s);
        }
        if (e == P && k == 1 && cb == 192) return launch_cfg<1, 6, 1, 1, BINHIP_K1_DGRAD_WN, 1, 3, 2, P>(a, cp, s);   // GFF.0 dgrad
        if (e == P && k == 1 && cb == 32)  return launch_cfg<1, 1, 1, 4, 4, 2, 3, 2, P>(a, cp, s);
        if (e == P && k == 1 && cb == 96) {
            if (a.nchunks >= 32) return launch_cfg<1, 3, 1, BINHIP_GFF0_R, BINHIP_GFF0_WN, 1, 3, 2, P>(a, cp, s);   // GFF.0
            return launch_cfg<1, 3, 1, 1, 8, 1, 3, 2, P>(a, cp, s);    // LFF 90 vs 108 us
        }
        if (e == P && k == 5 && cb == 32 && BINHIP_K5_X3) return bh_launch_conv_x3_k5(a, cp, s);   // plane-split stages, double-buffered
        if (e == P && k == 5 && cb == 32)  return launch_cfg<5, 1, 1, 2, 8, 1, 3, 1, P>(a, cp, s);
    }
    return BINHIP_E_SHAPE;
}

// ---------------------------------------------------------------------------------------------------
// Weight relayout: OIHW fp32 -> [cout_pad/cb][cin_chunks][k*k][cb][16] fp16 hi/lo, slot-swizzled.
__device__ __forceinline__ void relayout_body(long long t, const float* __restrict__ w, const float* __restrict__ bias,
                                              int cout, int cin, int ks, int cout_pad, int nchunks, int cb, int shuffle,
                                              _Float16* __restrict__ w_hi, _Float16* __restrict__ w_lo,
                                              float* __restrict__ bias_out) {
    const long long total = (long long)cout_pad * nchunks * ks * ks * 2;   // 16-byte slots
    if (t < cout_pad) {
        int src = (int)t;
        if (shuffle) { const int cq = cout / 4; const int sub = (int)t / cq, cc = (int)t % cq; src = cc * 4 + sub; }
        bias_out[t] = (src < cout && bias) ? bias[src] : 0.f;
    }
    if (t >= total) return;
    const int s = (int)(t & 1);
    long long u = t >> 1;
    const int row = (int)(u % cb); u /= cb;
    const int tap = (int)(u % (ks * ks)); u /= (ks * ks);
    const int c = (int)(u % nchunks); u /= nchunks;
    const int zb = (int)u;
    const int rg = zb * cb + row;
    int src = rg;
    if (shuffle) { const int cq = cout / 4; const int sub = rg / cq, cc = rg % cq; src = cc * 4 + sub; }
    const int cg = s ^ ((row >> 3) & 1);
    const int dy = tap / ks, dx = tap % ks;
    half8 hv, lv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = c * 16 + cg * 8 + e;
        float v = 0.f;
        if (src < cout && ci < cin) v = w[(((long long)src * cin + ci) * ks + dy) * ks + dx];
        hv[e] = (_Float16)v;
        lv[e] = (_Float16)(v - (float)hv[e]);
    }
    *reinterpret_cast<half8*>(w_hi + t * 8) = hv;
    if (w_lo) *reinterpret_cast<half8*>(w_lo + t * 8) = lv;
}
__global__ void relayout_kernel(const float* __restrict__ w, const float* __restrict__ bias, int cout, int cin,
                                int ks, int cout_pad, int nchunks, int cb, int shuffle,
                                _Float16* __restrict__ w_hi, _Float16* __restrict__ w_lo,
                                float* __restrict__ bias_out) {
    relayout_body((long long)blockIdx.x * blockDim.x + threadIdx.x, w, bias, cout, cin, ks, cout_pad, nchunks, cb, shuffle,
                  w_hi, w_lo, bias_out);
}

// Backward-data weights: the dgrad of a stride-1 "same" conv is the same conv with in/out channels swapped and the
// taps flipped: W'[r = ci][j = co][dy][dx] = W[co][ci][k-1-dy][k-1-dx].  shuffle != 0: the dgrad input channels j
// are in the PixelShuffle-permuted order of UPNet.0's rows (j = sub*(cout/4) + c  <->  co = c*4 + sub).
__device__ __forceinline__ void relayout_dgrad_body(long long t, const float* __restrict__ w, int cout, int cin, int ks,
                                                    int rows_pad, int nchunks, int cb, int shuffle,
                                                    _Float16* __restrict__ w_hi, _Float16* __restrict__ w_lo,
                                                    float* __restrict__ bias_out) {
    const long long total = (long long)rows_pad * nchunks * ks * ks * 2;
    if (t < rows_pad) bias_out[t] = 0.f;
    if (t >= total) return;
    const int s = (int)(t & 1);
    long long u = t >> 1;
    const int row = (int)(u % cb); u /= cb;
    const int tap = (int)(u % (ks * ks)); u /= (ks * ks);
    const int c = (int)(u % nchunks); u /= nchunks;
    const int zb = (int)u;
    const int r = zb * cb + row;                      // = original input channel
    const int cg = s ^ ((row >> 3) & 1);
    const int dy = ks - 1 - tap / ks, dx = ks - 1 - tap % ks;
    half8 hv, lv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int j = c * 16 + cg * 8 + e;            // dgrad input channel = original output channel (maybe permuted)
        int co = j;
        if (shuffle) { const int cq = cout / 4; const int sub = j / cq, cc = j % cq; co = cc * 4 + sub; }
        float v = 0.f;
        if (r < cin && j < cout) v = w[(((long long)co * cin + r) * ks + dy) * ks + dx];
        hv[e] = (_Float16)v;
        lv[e] = (_Float16)(v - (float)hv[e]);
    }
    *reinterpret_cast<half8*>(w_hi + t * 8) = hv;
    if (w_lo) *reinterpret_cast<half8*>(w_lo + t * 8) = lv;
}
__global__ void relayout_dgrad_kernel(const float* __restrict__ w, int cout, int cin, int ks, int rows_pad,
                                      int nchunks, int cb, int shuffle, _Float16* __restrict__ w_hi,
                                      _Float16* __restrict__ w_lo, float* __restrict__ bias_out) {
    relayout_dgrad_body((long long)blockIdx.x * blockDim.x + threadIdx.x, w, cout, cin, ks, rows_pad, nchunks, cb, shuffle,
                        w_hi, w_lo, bias_out);
}

// Backward-data of a residual dense block in GATHER form.  Group g of the block's concat buffer (g = 0: the 96 input
// channels, g = 1..3: the 32 outputs of conv g-1) receives dgrad contributions from every later conv c >= cmin
// (cmin = g for g >= 1, 0 for g = 0).  Stacking those convs' masked output gradients G_c (32 channels each, contiguous
// in the gradient buffer) along K turns the sum into ONE forward-shaped conv per group:
//   Wg[r][32 (c - cmin) + co][dy][dx] = W_c[co][base_g + r][2-dy][2-dx],   base_0 = 0, base_g = 96 + 32 (g - 1).
//   General block (G0, C, G): rows = G0 (group 0) or G, K = (C - group) G stacked output gradients, base_g = G0 + G (g - 1).
struct GatherSrc { const float* w[BINHIP_RDN_MAX_CONVS + 1]; int G0, G; };
__device__ __forceinline__ void relayout_rdb_gather_body(long long t, const GatherSrc& src, int group, int rows, int nchunks,
                                                         int cb, _Float16* __restrict__ w_hi, _Float16* __restrict__ w_lo,
                                                         float* __restrict__ bias_out) {
    const long long total = (long long)rows * nchunks * 9 * 2;
    if (t < rows) bias_out[t] = 0.f;
    if (t >= total) return;
    const int s = (int)(t & 1);
    long long u = t >> 1;
    const int row = (int)(u % cb); u /= cb;
    const int tap = (int)(u % 9); u /= 9;
    const int c = (int)(u % nchunks); u /= nchunks;
    const int r = (int)u * cb + row;
    const int cg = s ^ ((row >> 3) & 1);
    const int dy = 2 - tap / 3, dx = 2 - tap % 3;
    const int G0 = src.G0, G = src.G;
    const int cmin = group;                            // group 0 and group 1.. both start at conv index == group
    const int ci = (group == 0) ? r : G0 + G * (group - 1) + r;
    const int j0 = c * 16 + cg * 8;                    // 8 consecutive K entries never straddle a conv (G is a multiple of 32)
    const int conv = cmin + j0 / G;
    const int cin_c = G0 + G * conv;
    const float* w = src.w[conv];
    half8 hv, lv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int co = (j0 % G) + e;
        const float v = w[(((long long)co * cin_c + ci) * 3 + dy) * 3 + dx];
        hv[e] = (_Float16)v;
        lv[e] = (_Float16)(v - (float)hv[e]);
    }
    *reinterpret_cast<half8*>(w_hi + t * 8) = hv;
    if (w_lo) *reinterpret_cast<half8*>(w_lo + t * 8) = lv;
}
__global__ void relayout_rdb_gather_kernel(GatherSrc src, int group, int rows, int nchunks, int cb,
                                           _Float16* __restrict__ w_hi, _Float16* __restrict__ w_lo,
                                           float* __restrict__ bias_out) {
    relayout_rdb_gather_body((long long)blockIdx.x * blockDim.x + threadIdx.x, src, group, rows, nchunks, cb, w_hi, w_lo, bias_out);
}

// ---- batched relayout: a training step re-lays-out every weight of a set after each optimizer update (66 forward + 66
// backward layouts per set); one launch per <= RELAYOUT_BATCH layers instead of one per layer (round 3: 528 -> 24 launches
// per step of ~4.5 us each).  Block b of the grid belongs to the item whose block range contains it.
#define RELAYOUT_BATCH 22
struct RelayoutBatch {
    BinRelayoutItem it[RELAYOUT_BATCH];
    unsigned start[RELAYOUT_BATCH + 1];
    int n;
};
__global__ void __launch_bounds__(256)
relayout_batch_kernel(const RelayoutBatch rb) {
    const unsigned b = blockIdx.x;
    int li = 0;
#pragma unroll
    for (int i = 1; i < RELAYOUT_BATCH; ++i)
        if (i < rb.n && b >= rb.start[i]) li = i;
    const BinRelayoutItem& L = rb.it[li];
    const long long t = (long long)(b - rb.start[li]) * 256 + threadIdx.x;
    if (L.kind == BINHIP_RELAYOUT_FWD) {
        relayout_body(t, L.w[0], L.bias, L.cout, L.cin, L.ksize, L.rows_pad, L.cin_chunks, L.cout_block, L.shuffle_or_group,
                      (_Float16*)L.w_hi, (_Float16*)L.w_lo, L.bias_out);
    } else if (L.kind == BINHIP_RELAYOUT_DGRAD) {
        relayout_dgrad_body(t, L.w[0], L.cout, L.cin, L.ksize, L.rows_pad, L.cin_chunks, L.cout_block, L.shuffle_or_group,
                            (_Float16*)L.w_hi, (_Float16*)L.w_lo, L.bias_out);
    } else {
        GatherSrc src;
#pragma unroll
        for (int i = 0; i <= BINHIP_RDN_MAX_CONVS; ++i) src.w[i] = L.w[i];
        src.G0 = L.shape.G0 > 0 ? L.shape.G0 : 96;
        src.G = L.shape.G > 0 ? L.shape.G : 32;
        relayout_rdb_gather_body(t, src, L.shuffle_or_group, L.rows_pad, L.cin_chunks, L.cout_block, (_Float16*)L.w_hi,
                                 (_Float16*)L.w_lo, L.bias_out);
    }
}

extern "C" {

int binhip_weights_relayout_batch(const BinRelayoutItem* items, int n, void* stream) {
    if (!items || n <= 0) return BINHIP_E_ARG;
    for (int i = 0; i < n; ++i) {                      // validate everything before the first launch
        const BinRelayoutItem& L = items[i];
        if (!L.w[0] || !L.w_hi || !L.bias_out) return BINHIP_E_ARG;
        if (L.kind == BINHIP_RELAYOUT_FWD) {
            if (L.rows_pad % 32 || L.cout_block <= 0 || L.rows_pad % L.cout_block || L.cout_block % 32) return BINHIP_E_SHAPE;
            if (L.cin > L.cin_chunks * 16 || L.cout > L.rows_pad) return BINHIP_E_SHAPE;
            if (L.shuffle_or_group && (L.cout % 4 || L.cout != L.rows_pad)) return BINHIP_E_SHAPE;
        } else if (L.kind == BINHIP_RELAYOUT_DGRAD) {
            if (L.rows_pad % 32 || L.cout_block <= 0 || L.rows_pad % L.cout_block || L.cout_block % 32) return BINHIP_E_SHAPE;
            if (L.cin > L.rows_pad || L.cout > L.cin_chunks * 16) return BINHIP_E_SHAPE;
            if (L.shuffle_or_group && L.cout % 4) return BINHIP_E_SHAPE;
        } else if (L.kind == BINHIP_RELAYOUT_RDB_GATHER) {
            const int g = L.shuffle_or_group;
            const int G0 = L.shape.G0 > 0 ? L.shape.G0 : 96, G = L.shape.G > 0 ? L.shape.G : 32, C = L.shape.C > 0 ? L.shape.C : 4;
            if (C > BINHIP_RDN_MAX_CONVS || G0 % 32 || G % 32) return BINHIP_E_SHAPE;
            if (g < 0 || g >= C) return BINHIP_E_ARG;
            if (L.rows_pad != (g == 0 ? G0 : G) || L.cin_chunks != (C - g) * G / 16 || L.ksize != 3) return BINHIP_E_SHAPE;
            if (L.cout_block <= 0 || L.rows_pad % L.cout_block || L.cout_block % 32) return BINHIP_E_SHAPE;
            for (int k = g; k < C; ++k) if (!L.w[k]) return BINHIP_E_ARG;
        } else {
            return BINHIP_E_ARG;
        }
    }
    for (int i0 = 0; i0 < n; i0 += RELAYOUT_BATCH) {
        RelayoutBatch rb;
        const int m = (n - i0 < RELAYOUT_BATCH) ? n - i0 : RELAYOUT_BATCH;
        unsigned blocks = 0;
        for (int i = 0; i < m; ++i) {
            rb.it[i] = items[i0 + i];
            rb.start[i] = blocks;
            const BinRelayoutItem& L = rb.it[i];
            const long long total = (long long)L.rows_pad * L.cin_chunks * L.ksize * L.ksize * 2;
            blocks += (unsigned)((total + 255) / 256);
        }
        for (int i = m; i < RELAYOUT_BATCH; ++i) rb.it[i] = rb.it[0];
        for (int i = m; i <= RELAYOUT_BATCH; ++i) rb.start[i] = blocks;
        rb.n = m;
        hipLaunchKernelGGL(relayout_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rb);
        BH_CHECK_LAUNCH();
    }
    return 0;
}

int binhip_dgrad_rows_pad(int ksize, int cin) {
    // rows of the backward-data conv = original input channels, padded to the kernel's cout granularity (the LFF
    // dgrad's 224 rows form ONE 224-row workgroup column, so its input — the 96-channel output gradient — is read once)
    return ((cin + 31) / 32) * 32;
}

int binhip_weights_relayout_dgrad(const float* w_oihw, int cout, int cin, int ksize, int rows_pad, int cin_chunks,
                                  int cout_block, int shuffle_perm, void* w_hi, void* w_lo, float* bias_out,
                                  void* stream) {
    if (!w_oihw || !w_hi || !bias_out) return BINHIP_E_ARG;
    if (rows_pad % 32 || cout_block <= 0 || rows_pad % cout_block || cout_block % 32) return BINHIP_E_SHAPE;
    if (cin > rows_pad || cout > cin_chunks * 16) return BINHIP_E_SHAPE;
    if (shuffle_perm && cout % 4) return BINHIP_E_SHAPE;
    const long long total = (long long)rows_pad * cin_chunks * ksize * ksize * 2;
    const long long nb = (total + 255) / 256;
    hipLaunchKernelGGL(relayout_dgrad_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, w_oihw, cout, cin,
                       ksize, rows_pad, cin_chunks, cout_block, shuffle_perm, (_Float16*)w_hi, (_Float16*)w_lo, bias_out);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_weights_relayout_rdb_gather(const float* const* w_oihw4, int group, int cout_block, void* w_hi, void* w_lo,
                                       float* bias_out, void* stream) {
    if (!w_oihw4 || !w_hi || !bias_out || group < 0 || group > 3) return BINHIP_E_ARG;
    const int rows = group == 0 ? 96 : 32;
    const int nchunks = 2 * (4 - group);
    if (cout_block <= 0 || rows % cout_block || cout_block % 32) return BINHIP_E_SHAPE;
    GatherSrc src;
    for (int i = 0; i <= BINHIP_RDN_MAX_CONVS; ++i) src.w[i] = nullptr;
    src.G0 = 96; src.G = 32;
    for (int i = 0; i < 4; ++i) {
        src.w[i] = w_oihw4[i];
        if (i >= group && !src.w[i]) return BINHIP_E_ARG;
    }
    const long long total = (long long)rows * nchunks * 9 * 2;
    hipLaunchKernelGGL(relayout_rdb_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, src, group, rows, nchunks, cout_block, (_Float16*)w_hi, (_Float16*)w_lo,
                       bias_out);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_conv2d_bwd_data(const BinConvDesc* d, const void* gy_hi, const void* gy_lo, const void* wt_hi,
                           const void* wt_lo, const float* zero_bias, const void* res_hi, const void* res_lo,
                           int res_chunks, const void* acc_hi, const void* acc_lo, const void* mask_hi, int mask_from,
                           int y_cpg, int64_t y_group_stride, void* gx_hi, void* gx_lo, void* stream) {
    if (!d) return BINHIP_E_ARG;
    if (d->epilogue != BINHIP_EPI_PLANES) return BINHIP_E_ARG;
    if (d->reserved < 0) return BINHIP_E_ARG;               // y_unshuf is a plane count
    BhConvCall c;
    c.d = *d;
    c.x_hi = gy_hi; c.x_lo = gy_lo; c.w_hi = wt_hi; c.w_lo = wt_lo; c.bias = zero_bias;
    c.r_hi = res_hi; c.r_lo = res_lo; c.res_chunks = res_chunks;
    c.r2_hi = acc_hi; c.r2_lo = acc_lo;
    c.m_hi = mask_hi; c.mask_from = mask_from;
    c.y_cpg = y_cpg; c.y_group_stride = y_group_stride;
    c.y_unshuf = d->reserved > 0 ? d->reserved : 0;      // (BinConvDesc.reserved: the fused inverse PixelShuffle of the store)
    c.d.reserved = 0;
    c.y_hi = gx_hi; c.y_lo = gx_lo; c.y_f32 = nullptr;
    c.status = d->status;
    for (int i = 0; i < 5; ++i) c.images[i] = nullptr;
    return bh_launch_conv(c, (hipStream_t)stream);
}

#if BINHIP_TUNING
BINHIP_API int binhip_set_variant(int layer_class, int variant) {
    if (layer_class == -1) { g_xcd_remap = variant; return 0; }
    if (layer_class == -2) { g_dbg = variant; return 0; }
    if (layer_class == -3) { g_wt = variant; return 0; }
    if (layer_class == -4) { g_x3_wide = variant; return 0; }     // takes effect for weights relayouted afterwards
    if (layer_class < 0 || layer_class >= 8) return BINHIP_E_ARG;
    g_variant[layer_class] = variant;
    return 0;
}
#endif

#if BINHIP_TIMELINE
// side builds only: `buf` = zeroed device memory holding a BhTlBuf with room for `cap` records, null = stamps off.
// Returns the number of records reserved since the previous call.
BINHIP_API int binhip_set_timeline(void* buf, unsigned cap) {
    const unsigned used = g_bh_tl_next.exchange(0);
    g_bh_tl_buf = buf;
    g_bh_tl_cap = cap;
    g_bh_tl_serial.store(0);
    return (int)(used < g_bh_tl_cap || !buf ? used : used);
}
#endif

int binhip_profiler_create(int ksize, int cout_pad, int epilogue, int max_launches, BinhipProfiler** out) {
    if (!out || max_launches <= 0 || (size_t)max_launches > PROF_MAX_PAIRS) return BINHIP_E_ARG;
    BinhipProfiler* p = new BinhipProfiler();
    p->ks = ksize; p->cout_pad = cout_pad; p->epi = epilogue;
    for (int i = 0; i < 2 * max_launches; ++i) {
        hipEvent_t ev;
        hipError_t e = hipEventCreate(&ev);
        if (e != hipSuccess) { binhip_profiler_destroy(p); return (int)e; }
        p->ev.push_back(ev);
    }
    *out = p;
    return 0;
}

int binhip_profiler_read(BinhipProfiler* p, double* total_ms, int* launches) {
    if (!p) return BINHIP_E_ARG;
    double tot = 0.0;
    int n = 0;
    for (size_t i = 0; i + 1 < p->used; i += 2) {
        hipError_t e = hipEventSynchronize(p->ev[i + 1]);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]);
        if (e != hipSuccess) return (int)e;
        tot += ms;
        ++n;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    p->used = 0;
    return 0;
}

void binhip_profiler_destroy(BinhipProfiler* p) {
    if (!p) return;
    for (hipEvent_t ev : p->ev) (void)hipEventDestroy(ev);
    delete p;
}

int binhip_conv_cout_block(int ksize, int cout_pad, int nterms) { return bh_conv_cout_block(ksize, cout_pad, nterms); }

size_t binhip_weights_bytes(int cout_pad, int cin_chunks, int ksize) {
    return (size_t)cout_pad * cin_chunks * ksize * ksize * 32;
}

int binhip_weights_relayout(const float* w_oihw, const float* bias, int cout, int cin, int ksize,
                            int cout_pad, int cin_chunks, int cout_block, int shuffle_perm,
                            void* w_hi, void* w_lo, float* bias_out, void* stream) {
    if (!w_oihw || !w_hi || !bias_out) return BINHIP_E_ARG;
    if (cout_pad % 32 || cout_block <= 0 || cout_pad % cout_block || cout_block % 32) return BINHIP_E_SHAPE;
    if (cin > cin_chunks * 16 || cout > cout_pad) return BINHIP_E_SHAPE;
    if (shuffle_perm && (cout % 4 || cout != cout_pad)) return BINHIP_E_SHAPE;
    const long long total = (long long)cout_pad * cin_chunks * ksize * ksize * 2;
    const int th = 256;
    const long long nb = (total + th - 1) / th;
    hipLaunchKernelGGL(relayout_kernel, dim3((unsigned)nb), dim3(th), 0, (hipStream_t)stream, w_oihw, bias, cout, cin,
                       ksize, cout_pad, cin_chunks, cout_block, shuffle_perm, (_Float16*)w_hi, (_Float16*)w_lo,
                       bias_out);
    BH_CHECK_LAUNCH();
    return 0;
}

int binhip_conv2d_fwd(const BinConvDesc* d, const void* x_hi, const void* x_lo, const void* w_hi,
                      const void* w_lo, const float* bias, const void* res_hi, const void* res_lo,
                      void* y_hi, void* y_lo, float* y_f32, const float* const* images, void* stream) {
    if (!d) return BINHIP_E_ARG;
    // `reserved`: the one defined bit, and only where it is legal (the promise concerns the 5x5 layer's last chunk; a wrong promise
    // would give wrong sums silently because the real cin is not in the descriptor) — advisor r05
    if (d->reserved & ~BINHIP_CONV_HALF_LAST_CHUNK) return BINHIP_E_ARG;
    if ((d->reserved & BINHIP_CONV_HALF_LAST_CHUNK) && d->ksize != 5) return BINHIP_E_ARG;
    BhConvCall c;
    c.d = *d;
    c.x_hi = x_hi; c.x_lo = x_lo; c.w_hi = w_hi; c.w_lo = w_lo; c.bias = bias;
    c.r_hi = res_hi; c.r_lo = res_lo; c.y_hi = y_hi; c.y_lo = y_lo; c.y_f32 = y_f32;
    c.status = d->status;
    for (int i = 0; i < 5; ++i) c.images[i] = (images && i < d->n_images) ? images[i] : nullptr;
    if (d->epilogue == BINHIP_EPI_FINAL && d->n_images > 0 && !images) return BINHIP_E_ARG;
    return bh_launch_conv(c, (hipStream_t)stream);
}

}  // extern "C"
