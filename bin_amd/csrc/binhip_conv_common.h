// Shared device-side pieces of the convolution kernels (binhip_conv.hip, binhip_conv_x3.hip): kernel argument block,
// LDS-DMA / store helpers and the epilogue (bias, residuals, ReLU, ReLU-mask, fp16 hi/lo split, 16-byte plane stores,
// PixelShuffle scatter, fp32 NCHW final output).  Internal — not part of the C ABI.
#pragma once
#include "binhip_internal.h"
#ifndef BINHIP_TIMELINE
#define BINHIP_TIMELINE 0     // side builds only: per-workgroup time stamps (BhTl below)
#endif

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

int bh_launch_upnet_ring(const void* x_hi, const void* x_lo, const float* wvar, const float* bvar, float* out, const float* const* images,
                         int nimg, int N, int H, int W, int cin, hipStream_t s);      // binhip_conv_x3.hip (BINHIP_PLAN_FUSED_UPNET)
// backward of the fused UPNet (binhip_misc.hip)
int bh_upnet_gsub(const float* g, int N, int H, int W, const float* scale, void* y_hi, void* y_lo, void* status, hipStream_t s);
int bh_upnet_ring_dgrad(const float* g, const float* wvar, const float* scale, void* gx_hi, void* gx_lo, void* status, int N, int H, int W,
                        int cin, hipStream_t s);
int bh_upnet_ring_wgrad(const float* g, const void* x_hi, const void* x_lo, float* dwvar, float* dbvar, int N, int H, int W, int cin,
                        int accumulate, hipStream_t s);
struct ConvKArgs {
    const _Float16* x_hi;
    const _Float16* x_lo;
    const _Float16* w_hi;
    const _Float16* w_lo;
    const float* bias;
    _Float16* y_hi;
    _Float16* y_lo;
    const _Float16* r_hi;
    const _Float16* r_lo;
    const _Float16* r2_hi;    // second residual (backward-data accumulation), same indexing as y
    const _Float16* r2_lo;
    const _Float16* m_hi;     // ReLU mask source (saved forward activation, hi plane), same indexing as y
    float* out_f32;
    const float* img[5];
    unsigned* flags;          // device status word (bit 0: an output left the fp16 range and was clamped); may be null
    long long group_stride;   // elements
    int N, H, W;
    int nchunks;
    int cpg;
    int tiles_x, tiles_y;
    int ncol;                 // plane-split kernel: output columns of 32 rows sharing a tile (1-D grid, column fastest)
    int relu, has_res, nimg, cout;
    int xcd_remap;
    int wt;                   // write-through (sc1) output stores
    int och_limit;            // output chunks that exist at the destination (rows beyond are padding: not stored)
    int dbg;                  // BINHIP_TUNING side builds only: ablation switches (timing experiments, results invalid)
    int res_chunks;           // residual r applies to output chunks < res_chunks
    int mask_from;            // mask applies to output chunks >= mask_from (when m_hi != null)
    int y_cpg;                // output chunk grouping (<=0: one group)
    unsigned y_cpg_inv;       // ceil(2^20 / y_cpg): och / y_cpg == (och * y_cpg_inv) >> 20 for och < 4096, y_cpg <= 128 (no SALU division per slot)
    long long y_group_stride;
    int prog_prio;            // plane-split kernel: progress-ordered wave priority (launches that fit the chip in one round)
    int half_last;            // the last input chunk carries 8 real channels at most (BINHIP_CONV_HALF_LAST_CHUNK): 5x5 plane-split kernel
#if BINHIP_TIMELINE
    void* tl;                 // BINHIP_TIMELINE side builds: per-workgroup time-stamp records (BhTlBuf) or null
    unsigned tl_launch;       // (kernel kind << 24) | launch serial
    unsigned tl_base;         // first record of this launch (records are pre-assigned: base + tile index, no atomics)
#endif
    int y_unshuf;             // XTRA kernels only: > 0 = the output goes out through an inverse PixelShuffle(2) — full-resolution
                              // pixel (Y, X), chunk c -> plane (2 (Y & 1) + (X & 1)) * y_unshuf + c at (Y / 2, X / 2) of the half-
                              // resolution tensor (the channel order UPNet.0's permuted rows use); y_unshuf = chunks per sub-position
};

// ---- BINHIP_TIMELINE side builds (tools/wg_timeline.py; never in the product): wave 0 of every workgroup of the dominant
// fp32-class kernels stamps the constant-rate 100 MHz counter (s_memrealtime: one time base for all XCDs, whose shader clocks
// differ under the package limit) at entry / first MFMA / last MFMA / last store issued / stores drained, plus the shader-clock
// counter (s_memtime) at entry and exit and the hardware ids of where it ran.
#if BINHIP_TIMELINE
#include <atomic>
struct BhTlRec {
    unsigned kind_launch, bid, hwid, xcc;
    unsigned long long rt[5];
    unsigned long long clk[2];
    unsigned long long pad;
};
static_assert(sizeof(BhTlRec) == 80, "record size (tools/wg_timeline.py)");
struct BhTlBuf { unsigned cursor, cap, pad[2]; BhTlRec rec[1]; };
inline void* g_bh_tl_buf = nullptr;                       // host side: set by binhip_set_timeline()
inline unsigned g_bh_tl_cap = 0;
inline std::atomic<unsigned> g_bh_tl_serial{0}, g_bh_tl_next{0};
// reserve `n` records for one launch; returns the buffer (or null when full / off) and the first record's index
inline void* bh_tl_reserve(unsigned n, unsigned* base) {
    if (!g_bh_tl_buf) return nullptr;
    const unsigned b = g_bh_tl_next.fetch_add(n);
    *base = b;
    return (b + n <= g_bh_tl_cap) ? g_bh_tl_buf : nullptr;
}
struct BhTl {
    unsigned long long rt[5], clk[2];
    __device__ __forceinline__ void stamp(int i) {
        __builtin_amdgcn_sched_barrier(0);
        rt[i] = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void begin() { clk[0] = __builtin_amdgcn_s_memtime(); stamp(0); }
    __device__ __forceinline__ void finish(void* tl, unsigned kind_launch, unsigned i) {
        stamp(3);
        if (threadIdx.x < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // wave 0 only: the others end as in the product
        stamp(4);
        clk[1] = __builtin_amdgcn_s_memtime();
        if (tl && threadIdx.x == 0) {
            BhTlBuf* b = (BhTlBuf*)tl;
            {
                BhTlRec& r = b->rec[i];
                r.kind_launch = kind_launch; r.bid = blockIdx.x;
                r.hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
                r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
                for (int k = 0; k < 5; ++k) r.rt[k] = rt[k];
                r.clk[0] = clk[0]; r.clk[1] = clk[1];
            }
        }
    }
};
#define BH_TL_DECL BhTl tl__
#define BH_TL_BEGIN() tl__.begin()
#define BH_TL_STAMP(i) tl__.stamp(i)
#define BH_TL_FINISH(a, i) tl__.finish((a).tl, (a).tl_launch, (a).tl_base + (unsigned)(i))
#else
#define BH_TL_DECL
#define BH_TL_BEGIN()
#define BH_TL_STAMP(i)
#define BH_TL_FINISH(a, i)
#endif

// internal epilogue code (never crosses the ABI): chunk planes with the extras pattern of the LFF backward-data tile
#define BINHIP_EPI_PLANES_LFFD 3

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// 16-byte plane store.  wt != 0: write-through (sc1) so the XCD's L2 holds no dirty output lines at the end of the
// kernel: the kernel-boundary release then has nothing to write back (MI355X: 8 XCDs with private, mutually
// non-coherent L2s => every boundary flushes dirty lines; 16.5 MB of output costs ~2.8 us there).
__device__ __forceinline__ void store16(_Float16* base, long long off_elems, uint4 v, int wt) {
    if (wt) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xFFFFFFFFu, 0x00020000);
        u32x4 d = {v.x, v.y, v.z, v.w};
        __builtin_amdgcn_raw_buffer_store_b128(d, rs, (int)(off_elems * 2), 0, 16);
    } else {
        *reinterpret_cast<uint4*>(base + off_elems) = v;
    }
}

// consecutive workgroup ids land on different XCDs (private L2s): give each XCD a contiguous band of tiles so the
// halo rows/columns neighbouring tiles share are L2 hits (bijective for any grid size)
__device__ __forceinline__ int xcd_band(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// ---- epilogue ---------------------------------------------------------------------------------------
// acc[mt][r][4g+j] = D[cout = 8g + 4*kg + j][pixel = n]  (32x32 MFMA C/D layout): a lane holds 4 consecutive
// channels of one pixel per g, and lanes n / n+32 hold the two halves of each 8-channel slot.  One
// v_permlane32_swap per dword turns a (g even, g odd) pair into full 16-byte slots — lanes 0-31 get slot 0, lanes
// 32-63 slot 1 of the same pixel — so every store instruction writes 32 pixels x 32 B = 1 KiB contiguous.
// row0 = first image row of this wave's R rows, co0 = first output channel of this wave's MT x 32 rows (wave-uniform).
//
// Loads and stores share ONE counter on gfx9 (vmcnt) and retire out of order with each other, so the compiler has to
// drain every earlier store (s_waitcnt vmcnt(0): a full trip to L2) before it may use a later load.  Round 2's epilogue
// loaded the bias — and, in the backward-data kernels, residual / accumulator / ReLU-mask values — per 8-channel slot
// between the plane stores: up to 28 such drains per pixel row of the 224-row LFF backward-data tile.  Now
//   * the bias comes through SCALAR loads (lgkmcnt): `bias` must be the kernel's own `const float* __restrict__`
//     parameter (a pointer inside the by-value argument struct is not known to be unclobbered and gets vector loads); the
//     8 values of a slot pair are wave-uniform, the lane's half is picked with kg;
//   * XTRA = false (every forward layer without a residual): no vector load at all, the stores go out back to back;
//   * XTRA = true: residual, in-place accumulation and mask planes are loaded for a whole GROUP of slots (MTG weight
//     tiles x 4 slots) before the group's first store: one round trip per group.
template <int MT, int R, int NT, int EPI, bool XTRA>
__device__ __forceinline__ void conv_epilogue(const ConvKArgs& a, const float* __restrict__ bias, floatx16 (&acc)[MT][R],
                                              int img, int row0, int tx0, int co0, bool first_col, int n, int kg,
                                              long long plane_elems) {
    const int H = a.H, W = a.W;
    const int gx = tx0 + n;
    auto bias4 = [&](int co_u, float (&bv)[4]) {       // co_u: wave-uniform first channel of an 8-channel slot
        float b8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) b8[j] = bias[co_u + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = kg ? b8[4 + j] : b8[j];
    };
    if constexpr (EPI == BINHIP_EPI_FINAL_SUBPIX) {
        // The fused UPNet (BINHIP_PLAN_FUSED_UPNET): this convolution runs at HALF resolution and its output channel c * 4 + i * 2 + j is
        // colour c at sub-pixel (i, j) of the full-resolution pixel block (2 gy + i, 2 gx + j).  In the MFMA's C layout lane (n, kg)
        // holds channels 8 g + 4 kg + 0..3 in acc[..][4 g + 0..3]: slot g = 0 is colour kg's 2 x 2 block, slot g = 1 colour 2 + kg's
        // (colour 2 for kg = 0; nothing for kg = 1) — two 2-float row pieces per block, 32 lanes = 256 contiguous bytes per row.
        // + bias + mean of the input frames (RDN.py:221/279/333) -> fp32 NCHW [N, cout / 4, 2 H, 2 W].  Loads first, then stores.
        const int ncolour = a.cout >> 2;
        const int H2 = 2 * H, W2 = 2 * W;
        float bv[2][4];
        bias4(co0, bv[0]);
        bias4(co0 + 8, bv[1]);
        const bool col_ok = first_col && (gx < W);
        float sum[R][2][4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int gy = row0 + r;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int c = 2 * g + kg;
                const bool ok = col_ok && gy < H && c < ncolour;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float sj = 0.f;
                    if (ok && a.nimg > 0) {
                        const long long idx = (((long long)img * ncolour + c) * H2 + 2 * gy + (q >> 1)) * W2 + 2 * gx + (q & 1);
                        sj = a.img[0][idx];
                        for (int t = 1; t < a.nimg; ++t) sj += a.img[t][idx];
                        sj = sj / (float)a.nimg;
                    }
                    sum[r][g][q] = sj;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int gy = row0 + r;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int c = 2 * g + kg;
                if (!(col_ok && gy < H && c < ncolour)) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long long idx = (((long long)img * ncolour + c) * H2 + 2 * gy + (q >> 1)) * W2 + 2 * gx + (q & 1);
                    a.out_f32[idx] = (acc[0][r][4 * g + q] + bv[g][q]) + sum[r][g][q];
                }
            }
        }
    } else if constexpr (EPI == BINHIP_EPI_FINAL) {
        float bv[4];
        bias4(co0, bv);
        // every frame load of the wave's R rows first, then the stores (a load behind a store would wait for the store)
        const bool mine = (kg == 0) && first_col && (gx < W);
        float sum[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int gy = row0 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sj = 0.f;
                if (mine && gy < H && j < a.cout && a.nimg > 0) {
                    const long long idx = (((long long)img * a.cout + j) * H + gy) * W + gx;
                    sj = a.img[0][idx];
                    for (int t = 1; t < a.nimg; ++t) sj += a.img[t][idx];
                    sj = sj / (float)a.nimg;
                }
                sum[r][j] = sj;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int gy = row0 + r;
            if (!(mine && gy < H)) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j >= a.cout) break;
                const long long idx = (((long long)img * a.cout + j) * H + gy) * W + gx;
                a.out_f32[idx] = (acc[0][r][j] + bv[j]) + sum[r][j];
            }
        }
    } else {
        union H4 { half4 h; unsigned u[2]; };
        unsigned sat = 0;
        const int gxc = gx < W ? gx : W - 1;     // clamped coordinates: loads need no branch, stores are predicated
        constexpr bool X = XTRA && (EPI == BINHIP_EPI_PLANES || EPI == BINHIP_EPI_PLANES_LFFD);
        constexpr int MTG = (X && MT >= 6) ? 2 : 1;      // weight tiles per load group (<= 10 registers per slot)
        const bool use_res = X && a.has_res;
        const bool use_r2 = X && (a.r2_hi != nullptr);
        const bool use_m = X && (a.m_hi != nullptr);
        // The LFF backward-data tile (224 rows = 7 weight tiles; residual gy on output chunks 0-5, ReLU mask on chunks 12-13,
        // nothing on 6-11): the generic grouping below fetches extras per PAIR of weight tiles, four load groups per pixel row, and
        // every group after the first waits for the previous group's stores to drain before it may use what it loaded (loads and
        // stores share vmcnt).  With the pattern known, ALL extras of the row — 24 residual slots + 4 mask slots = 56 registers,
        // fewer than the generic group's 80 — are fetched before the first store: one round trip per row instead of four.
        // (Its own instantiation — EPI = BINHIP_EPI_PLANES_LFFD, chosen by the launcher when the call has exactly this pattern:
        // compiled into the generic kernel beside the generic path the two together spill; alone it needs 174 registers, the
        // generic tile 233.)
        constexpr bool LFFD = EPI == BINHIP_EPI_PLANES_LFFD;
        static_assert(!LFFD || (XTRA && MT == 7 && NT == 3), "the LFF backward-data epilogue");
        if constexpr (LFFD) {
            {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int gy = row0 + r;
                    const bool ok = (gy < H) && (gx < W);
                    const int gyc = gy < H ? gy : H - 1;
                    const long long pix = (((long long)img * H + gyc) * W + gxc) << 4;
                    // every extra of the row before its first store: tiles 0-2 (residual), 6 (mask); tiles 3-5 carry nothing
                    auto emit = [&](const int mt, const half4 (&xr)[4], const half4 (&xrl)[4], const half4 (&xm)[4], const int kind)
                        __attribute__((always_inline)) {                 // kind: 0 nothing, 1 residual, 2 mask
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            H4 hv[2], lv[2];
                            long long o_slot = 0;
#pragma unroll
                            for (int ge = 0; ge < 2; ++ge) {
                                const int g = 2 * gp + ge;
                                float bv[4];
                                bias4(mt * 32 + 8 * g, bv);
                                float v[4] = {acc[mt][r][4 * g + 0] + bv[0], acc[mt][r][4 * g + 1] + bv[1],
                                              acc[mt][r][4 * g + 2] + bv[2], acc[mt][r][4 * g + 3] + bv[3]};
                                const int co = mt * 32 + 8 * g + 4 * kg;
                                const long long o = (long long)(co >> 4) * plane_elems + pix + (co & 15);
                                if (kind == 1) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) v[j] += (float)xr[g][j];
#pragma unroll
                                    for (int j = 0; j < 4; ++j) v[j] += (float)xrl[g][j];
                                }
                                if (a.relu) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                                }
                                if (kind == 2) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) v[j] = ((float)xm[g][j] > 0.f) ? v[j] : 0.f;
                                }
                                if (ge == kg) o_slot = o - 4 * kg;
                                split_pair(v[0], v[1], sat, hv[ge].u[0], lv[ge].u[0]);
                                split_pair(v[2], v[3], sat, hv[ge].u[1], lv[ge].u[1]);
                            }
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                auto sw = __builtin_amdgcn_permlane32_swap(hv[0].u[k], hv[1].u[k], false, false);
                                hv[0].u[k] = sw[0]; hv[1].u[k] = sw[1];
                                auto sl = __builtin_amdgcn_permlane32_swap(lv[0].u[k], lv[1].u[k], false, false);
                                lv[0].u[k] = sl[0]; lv[1].u[k] = sl[1];
                            }
                            if (ok) {
                                store16(a.y_hi, o_slot, make_uint4(hv[0].u[0], hv[0].u[1], hv[1].u[0], hv[1].u[1]), a.wt);
                                store16(a.y_lo, o_slot, make_uint4(lv[0].u[0], lv[0].u[1], lv[1].u[0], lv[1].u[1]), a.wt);
                            }
                        }
                    };
                    auto fetch = [&](const int mt, const _Float16* hi, const _Float16* lo, half4 (&xh)[4], half4 (&xl)[4])
                        __attribute__((always_inline)) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int co = mt * 32 + 8 * g + 4 * kg;
                            const long long o = (long long)(co >> 4) * plane_elems + pix + (co & 15);
                            xh[g] = *reinterpret_cast<const half4*>(hi + o);
                            if (lo) xl[g] = *reinterpret_cast<const half4*>(lo + o);
                        }
                    };
                    half4 ra[4], ral[4], rb[4], rbl[4], rc[4], rcl[4], rm[4], none[4] = {};
                    fetch(0, a.r_hi, a.r_lo, ra, ral);
                    fetch(1, a.r_hi, a.r_lo, rb, rbl);
                    fetch(2, a.r_hi, a.r_lo, rc, rcl);
                    fetch(6, a.m_hi, nullptr, rm, none);
                    // pin them: left alone, hipcc waits for each tile's values right before THAT tile's stores, i.e. behind the
                    // stores of the tiles before it — and vmcnt counts those stores too
                    auto pin = [](half4 (&x)[4]) __attribute__((always_inline)) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            H4 t;
                            t.h = x[g];
                            asm volatile("" : "+v"(t.u[0]), "+v"(t.u[1]));
                            x[g] = t.h;
                        }
                    };
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    pin(ra); pin(ral); pin(rb); pin(rbl); pin(rc); pin(rcl); pin(rm);
                    __builtin_amdgcn_sched_barrier(0);
                    emit(0, ra, ral, none, 1);
                    emit(1, rb, rbl, none, 1);
                    emit(2, rc, rcl, none, 1);
                    emit(6, none, none, rm, 2);
                    emit(3, none, none, none, 0);
                    emit(4, none, none, none, 0);
                    emit(5, none, none, none, 0);
                }
            }
        }
        if constexpr (!LFFD) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int gy = row0 + r;
            const bool ok = (gy < H) && (gx < W);
            const int gyc = gy < H ? gy : H - 1;
#pragma unroll
            for (int mt0 = 0; mt0 < MT; mt0 += MTG) {
                // ---- phase 1: addresses and (XTRA) every load of the group
                long long off[MTG][4];
                // (zero-initialised: the pins below touch every slot, also those no load filled — a slot beyond res_chunks, a
                //  call without mask — and must not read indeterminate registers; advisor r04)
                half4 xr[MTG][4] = {}, xrl[MTG][4] = {}, x2[MTG][4] = {}, x2l[MTG][4] = {}, xm[MTG][4] = {};
#pragma unroll
                for (int mi = 0; mi < MTG; ++mi) {
                    const int mt = mt0 + mi;
                    if (mt >= MT) continue;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int co = co0 + mt * 32 + 8 * g + 4 * kg;
                        long long o;
                        if constexpr (EPI == BINHIP_EPI_SHUFFLE) {
                            const int cq = (a.cout + 3) / 4;      // channels after the shuffle
                            const int sub = co / cq, cc = co - sub * cq;
                            const int oy = 2 * gyc + (sub >> 1), ox = 2 * gxc + (sub & 1);
                            o = (long long)(cc >> 4) * (plane_elems * 4) +
                                ((((long long)img * 2 * H + oy) * (2 * W) + ox) << 4) + (cc & 15);
                        } else {
                            const int och = co >> 4;
                            const long long pix16 = ((((long long)img * H + gyc) * W + gxc) << 4) + (co & 15);
                            if (a.y_cpg > 0) {
                                const int q = (int)(((unsigned)och * a.y_cpg_inv) >> 20);     // och / y_cpg
                                o = (long long)q * a.y_group_stride + (long long)(och - q * a.y_cpg) * plane_elems + pix16;
                            } else {
                                o = (long long)och * plane_elems + pix16;
                            }
                            if constexpr (X) {
                                if (a.y_unshuf > 0) {     // PixelShuffle backward fused into this store (UPNet.2's backward-data)
                                    const int sub = ((gyc & 1) << 1) | (gxc & 1);
                                    o = (long long)(sub * a.y_unshuf + och) * (plane_elems >> 2) +
                                        ((((long long)img * (H >> 1) + (gyc >> 1)) * (W >> 1) + (gxc >> 1)) << 4) + (co & 15);
                                }
                            }
                            if constexpr (X) {
                                const bool live = och < a.och_limit;
                                if (use_res && och < a.res_chunks && live) {
                                    xr[mi][g] = *reinterpret_cast<const half4*>(a.r_hi + o);
                                    if constexpr (NT == 3) xrl[mi][g] = *reinterpret_cast<const half4*>(a.r_lo + o);
                                }
                                if (use_r2 && live) {
                                    x2[mi][g] = *reinterpret_cast<const half4*>(a.r2_hi + o);
                                    if constexpr (NT == 3) x2l[mi][g] = *reinterpret_cast<const half4*>(a.r2_lo + o);
                                }
                                if (use_m && och >= a.mask_from && live) xm[mi][g] = *reinterpret_cast<const half4*>(a.m_hi + o);
                            }
                        }
                        off[mi][g] = o;
                    }
                }
                if constexpr (X) {
                    // Every extra of the group is in its register BEFORE the group's first store (round 4): left alone, hipcc waits
                    // for a slot's values right before THAT slot's stores — behind the stores of the slots before it, which vmcnt
                    // counts too (the UPNet.2 epilogue went 122 -> 89 us on exactly this, profiles/r04_experiments.md section 9).
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int mi = 0; mi < MTG; ++mi)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            auto pin = [](half4& x) __attribute__((always_inline)) {
                                H4 t;
                                t.h = x;
                                asm volatile("" : "+v"(t.u[0]), "+v"(t.u[1]));
                                x = t.h;
                            };
                            pin(xr[mi][g]); pin(x2[mi][g]); pin(xm[mi][g]);
                            if constexpr (NT == 3) { pin(xrl[mi][g]); pin(x2l[mi][g]); }
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- phase 2: bias, extras, ReLU / mask, hi / lo split, lane swap, stores
#pragma unroll
                for (int mi = 0; mi < MTG; ++mi) {
                    const int mt = mt0 + mi;
                    if (mt >= MT) continue;
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        H4 hv[2], lv[2];
                        long long o_slot = 0;        // element offset of this lane's 16-byte slot after the swap
#pragma unroll
                        for (int ge = 0; ge < 2; ++ge) {
                            const int g = 2 * gp + ge;
                            float bv[4];
                            bias4(co0 + mt * 32 + 8 * g, bv);
                            float v[4] = {acc[mt][r][4 * g + 0] + bv[0], acc[mt][r][4 * g + 1] + bv[1],
                                          acc[mt][r][4 * g + 2] + bv[2], acc[mt][r][4 * g + 3] + bv[3]};
                            const long long o = off[mi][g];
                            if constexpr (EPI != BINHIP_EPI_SHUFFLE) {
                                const int och = (co0 + mt * 32 + 8 * g + 4 * kg) >> 4;
                                const bool live = och < a.och_limit;
                                if constexpr (X) {
                                    if (use_res && och < a.res_chunks && live) {
#pragma unroll
                                        for (int j = 0; j < 4; ++j) v[j] += (float)xr[mi][g][j];
                                        if constexpr (NT == 3) {
#pragma unroll
                                            for (int j = 0; j < 4; ++j) v[j] += (float)xrl[mi][g][j];
                                        }
                                    }
                                    if (use_r2 && live) {
#pragma unroll
                                        for (int j = 0; j < 4; ++j) v[j] += (float)x2[mi][g][j];
                                        if constexpr (NT == 3) {
#pragma unroll
                                            for (int j = 0; j < 4; ++j) v[j] += (float)x2l[mi][g][j];
                                        }
                                    }
                                }
                                if constexpr (X) {               // (without extras the ReLU rides on split_pair's clamp)
                                    if (a.relu) {
#pragma unroll
                                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                                    }
                                }
                                if constexpr (X) {
                                    if (use_m && och >= a.mask_from && live) {
#pragma unroll
                                        for (int j = 0; j < 4; ++j) v[j] = ((float)xm[mi][g][j] > 0.f) ? v[j] : 0.f;
                                    }
                                }
                            }
                            // after the swap, lanes 0-31 own slot 0 (g even) and lanes 32-63 slot 1 (g odd) of pixel n:
                            // the slot start is this (g, kg=0) element offset
                            if (ge == kg) o_slot = o - 4 * kg;
                            const bool fold_relu = !X && a.relu && (EPI != BINHIP_EPI_SHUFFLE);
                            split_pair(v[0], v[1], sat, hv[ge].u[0], lv[ge].u[0], fold_relu);
                            split_pair(v[2], v[3], sat, hv[ge].u[1], lv[ge].u[1], fold_relu);
                        }
                        // vdst = even group, src = odd group: upper half of vdst <-> lower half of src
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            auto sw = __builtin_amdgcn_permlane32_swap(hv[0].u[k], hv[1].u[k], false, false);
                            hv[0].u[k] = sw[0]; hv[1].u[k] = sw[1];
                            if constexpr (NT == 3) {
                                auto sl = __builtin_amdgcn_permlane32_swap(lv[0].u[k], lv[1].u[k], false, false);
                                lv[0].u[k] = sl[0]; lv[1].u[k] = sl[1];
                            }
                        }
                        if (ok && ((co0 + mt * 32 + 16 * gp) >> 4) < a.och_limit) {
                            store16(a.y_hi, o_slot, make_uint4(hv[0].u[0], hv[0].u[1], hv[1].u[0], hv[1].u[1]), a.wt);
                            if constexpr (NT == 3)
                                store16(a.y_lo, o_slot, make_uint4(lv[0].u[0], lv[0].u[1], lv[1].u[0], lv[1].u[1]), a.wt);
                        }
                    }
                }
            }
        }
        }   // !LFFD
        if (a.flags && __builtin_amdgcn_ballot_w64(sat != 0) != 0 && (threadIdx.x & 63) == 0)
            atomicOr(a.flags, BINHIP_FLAG_SATURATED);
    }
}

