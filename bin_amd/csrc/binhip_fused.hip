// binhip_fused.hip — the tail of a residual dense block as ONE kernel:
//     o3 = relu(conv3x3(blk[0:192]) + b3)            (RDB_Conv #3, reference RDN.py:141-147)
//     y  = conv1x1(cat(blk[0:192], o3)) + b + blk[0:96]   (LFF + residual, RDN.py:162-165)
// Unfused, LFF re-reads all 224 channels that conv #3 has just streamed through LDS (and o3 makes a round trip
// through HBM).  Here every 16-channel K-stage of conv #3 also feeds the LFF accumulators — the 1x1's B operand
// is exactly the conv's centre-tap fragment, already in registers — and o3 goes registers -> LDS -> MFMA for the
// last two LFF K-steps.  HBM traffic of an RDB drops by ~30 % (fp16: 2240 -> 1536 B/pixel).
//
// 512 threads (8 wave64), tile 16 rows x 32 cols, wave w owns rows 2w, 2w+1: 2 conv accumulators + 6 LFF accumulators
// (128 regs).  K-stage = patch 18x34x16ch (20 KiB) + conv weights (9 KiB) + LFF weights (3 KiB) per precision plane,
// LDS-DMA ring with counted vmcnt (see binhip_conv.hip for the layout / swizzle conventions, which are shared).
#include "binhip_conv_common.h"

#include "binhip_fused.h"

template <int NT, int NBUF>
struct TailCfg {
    static constexpr int R = 2, NW = 8, TH = 16, PH = 18, PW = 34;
    static constexpr int PP = 20, CWP = 9, LWP = 3;                 // 1-KiB pieces: patch / conv weights / LFF weights
    static constexpr int NPL = (NT == 3) ? 2 : 1;
    static constexpr int PLANE_BYTES = (PP + CWP + LWP) * 1024;     // 32 KiB
    static constexpr int BUF_BYTES = NPL * PLANE_BYTES;
    static constexpr int TAILW_BYTES = NPL * 2 * LWP * 1024;        // LFF weights of chunks 12, 13
    static constexpr int LDS_BYTES = NBUF * BUF_BYTES + TAILW_BYTES + 1024;
    static constexpr int NPJ = 3, NCJ = 2, NLJ = 1;
    static constexpr int PS = NPL * (NPJ + NCJ + NLJ);
    static constexpr int NCHUNK = 12;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(NBUF <= 2 || (NBUF - 2) * PS <= 63, "vmcnt immediate range");
    static_assert(NW * R * 32 * 2 * 32 * NPL <= BUF_BYTES, "o3 staging tile must fit one stage buffer");
};

__device__ __forceinline__ half8 ld8(const char* p) { return *reinterpret_cast<const half8*>(p); }

template <class C>
__device__ __forceinline__ void tail_issue(const TailKArgs& a, char* smem, int c, int buf, int wave, int lane,
                                           const unsigned (&voff)[C::NPJ], long long plane_elems, unsigned plane_bytes) {
    char* dummy = smem + (C::LDS_BYTES - 1024);
#pragma unroll
    for (int pl = 0; pl < C::NPL; ++pl) {
        char* lds = smem + buf * C::BUF_BYTES + pl * C::PLANE_BYTES;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((pl ? a.x_lo : a.x_hi) + (long long)c * plane_elems), 0, plane_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < C::NPJ; ++j) {
            const int i = wave + C::NW * j;
            const bool real = i < C::PP;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(real ? lds + i * 1024 : dummy), 16,
                                                     real ? voff[j] : 0x80000000u, 0, 0, 0);
        }
        __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((pl ? a.wc_lo : a.wc_hi) + (long long)c * (9 * 32 * 16)), 0, 9 * 1024, 0x00020000);
#pragma unroll
        for (int j = 0; j < C::NCJ; ++j) {
            const int i = wave + C::NW * j;
            const bool real = i < C::CWP;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ws, (lds_void_t*)(real ? lds + (C::PP + i) * 1024 : dummy), 16,
                                                     real ? (unsigned)(lane * 16) : 0x80000000u, real ? i * 1024 : 0, 0, 0);
        }
        __amdgpu_buffer_rsrc_t ls = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((pl ? a.wl_lo : a.wl_hi) + (long long)c * (96 * 16)), 0, 3 * 1024, 0x00020000);
        {
            const bool real = wave < C::LWP;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ls, (lds_void_t*)(real ? lds + (C::PP + C::CWP + wave) * 1024 : dummy), 16,
                                                     real ? (unsigned)(lane * 16) : 0x80000000u, real ? wave * 1024 : 0, 0, 0);
        }
    }
}

template <int NT, int NBUF>
__global__ void __launch_bounds__(512)
rdb_tail_kernel(const TailKArgs a, const float* __restrict__ bias_c, const float* __restrict__ bias_l) {
    // (the biases again as the kernel's own restrict parameters: scalar loads in the epilogues instead of vector loads that would
    //  have to wait for every earlier plane store — binhip_conv_common.h, conv_epilogue)
    using C = TailCfg<NT, NBUF>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, kg = lane >> 5;

    int bid = blockIdx.x;
    if (a.xcd_remap) bid = xcd_band(bid, gridDim.x);
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int tx0 = tx * 32, ty0 = ty * C::TH;
    const int H = a.H, W = a.W;
    const long long plane_elems = (long long)a.N * H * W * 16;
    const unsigned plane_bytes = (unsigned)(plane_elems * 2);

    unsigned voff[C::NPJ];
#pragma unroll
    for (int j = 0; j < C::NPJ; ++j) {
        const int i = wave + C::NW * j;
        const int q = i * 64 + lane;
        const int p = q >> 1, s = q & 1;
        const int py = p / C::PW, px = p - py * C::PW;
        const int gy = ty0 + py - 1, gx = tx0 + px - 1;
        const int cg = s ^ ((p >> 3) & 1);
        const bool ok = (p < C::PH * C::PW) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        voff[j] = ok ? (unsigned)((((long long)img * H + gy) * W + gx) * 32 + cg * 16) : 0x80000000u;
    }

    floatx16 accc[C::R];
    floatx16 accl[3][C::R];
#pragma unroll
    for (int r = 0; r < C::R; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            accc[r][e] = 0.f;
            accl[0][r][e] = 0.f; accl[1][r][e] = 0.f; accl[2][r][e] = 0.f;
        }

    // ---- prologue: LFF weights of chunks 12/13 (older than every stage => landed whenever a stage has), then ring
    char* tailw = smem + NBUF * C::BUF_BYTES;
#pragma unroll
    for (int pl = 0; pl < C::NPL; ++pl) {
        __amdgpu_buffer_rsrc_t ls = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((pl ? a.wl_lo : a.wl_hi) + (long long)12 * (96 * 16)), 0, 6 * 1024, 0x00020000);
        if (wave < 6)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ls, (lds_void_t*)(tailw + pl * (2 * C::LWP * 1024) + wave * 1024), 16,
                                                     lane * 16, wave * 1024, 0, 0);
    }
#pragma unroll
    for (int s0 = 0; s0 < NBUF - 1; ++s0)
        tail_issue<C>(a, smem, s0, s0, wave, lane, voff, plane_elems, plane_bytes);

    const int a_lane_off = n * 32 + ((kg ^ ((n >> 3) & 1)) << 4);
    const int b_lane_p = wave * C::R * C::PW + n;
    // identity A fragments: row m selects input channel k of the chunk when m == 16*half + k.  The RDB residual
    // (`+ x`, RDN.py:165) is then one extra MFMA per row in the K-steps of chunks 0..5 — x is already in registers as
    // the centre-tap fragment — instead of a second pass over x from HBM in the epilogue (49.5 MB per launch).
    half8 ident[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int e = 0; e < 8; ++e) ident[hf][e] = (n == hf * 16 + kg * 8 + e) ? (_Float16)1.0f : (_Float16)0.0f;
    int cur = 0, nxt = NBUF - 1;
    for (int st = 0; st < C::NCHUNK; ++st) {
        int yf = C::NCHUNK - 1 - st;
        yf = yf > NBUF - 2 ? NBUF - 2 : yf;
        if (NBUF >= 4 && yf >= 2) wait_vmcnt<(NBUF >= 4 ? 2 : 0) * C::PS>();
        else if (NBUF >= 3 && yf >= 1) wait_vmcnt<(NBUF >= 3 ? 1 : 0) * C::PS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (st + NBUF - 1 < C::NCHUNK)
            tail_issue<C>(a, smem, st + NBUF - 1, nxt, wave, lane, voff, plane_elems, plane_bytes);
        const char* pb = smem + cur * C::BUF_BYTES;
        const char* wb = pb + C::PP * 1024;
        const char* lb = wb + C::CWP * 1024;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            half8 Bh[C::R + 2], Bl[C::R + 2];
#pragma unroll
            for (int rr = 0; rr < C::R + 2; ++rr) {
                const int p = b_lane_p + rr * C::PW + dx;
                const int off = p * 32 + ((kg ^ ((p >> 3) & 1)) << 4);
                Bh[rr] = ld8(pb + off);
                if constexpr (NT == 3) Bl[rr] = ld8(pb + C::PLANE_BYTES + off);
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int off = ((dy * 3 + dx) * 32) * 32 + a_lane_off;
                const half8 Ah = ld8(wb + off);
                half8 Al;
                if constexpr (NT == 3) Al = ld8(wb + C::PLANE_BYTES + off);
#pragma unroll
                for (int r = 0; r < C::R; ++r) {
                    if constexpr (NT == 3) {
                        accc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh[r + dy], accc[r], 0, 0, 0);
                        accc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl[r + dy], accc[r], 0, 0, 0);
                    }
                    accc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh[r + dy], accc[r], 0, 0, 0);
                }
            }
            if (dx == 1) {   // the 1x1 LFF sees the centre tap's fragment
                if (st < 6) {        // residual: output channels 16*st .. 16*st+15 += x (exact: 1.0 * x, fp32 accumulate)
#pragma unroll
                    for (int r = 0; r < C::R; ++r) {
#pragma unroll
                        for (int mt = 0; mt < 3; ++mt) {
                            if (mt == (st >> 1)) {
                                accl[mt][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ident[st & 1], Bh[r + 1], accl[mt][r], 0, 0, 0);
                                if constexpr (NT == 3)
                                    accl[mt][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ident[st & 1], Bl[r + 1], accl[mt][r], 0, 0, 0);
                            }
                        }
                    }
                }
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) {
                    const int off = (mt * 32) * 32 + a_lane_off;
                    const half8 Ah = ld8(lb + off);
                    half8 Al;
                    if constexpr (NT == 3) Al = ld8(lb + C::PLANE_BYTES + off);
#pragma unroll
                    for (int r = 0; r < C::R; ++r) {
                        if constexpr (NT == 3) {
                            accl[mt][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh[r + 1], accl[mt][r], 0, 0, 0);
                            accl[mt][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl[r + 1], accl[mt][r], 0, 0, 0);
                        }
                        accl[mt][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh[r + 1], accl[mt][r], 0, 0, 0);
                    }
                }
            }
        }
        cur = (cur + 1 == NBUF) ? 0 : cur + 1;
        nxt = (nxt + 1 == NBUF) ? 0 : nxt + 1;
    }

    // ---- conv #3 epilogue: bias + ReLU, keep o3 in a wave-private LDS tile ([chunk][row][pixel][16 ch], swizzled) ----
    // `cur` now names the buffer after the last stage's: every wave left it at least one barrier ago.
    char* o3 = smem + cur * C::BUF_BYTES + wave * (C::R * 2 * 32 * 32);
    constexpr int O3_PLANE = C::NW * C::R * 2 * 32 * 32;     // hi tiles of all waves, then lo tiles
    const int gx = tx0 + n;
    union H4 { half4 h; unsigned u[2]; };
    unsigned sat = 0;
#pragma unroll
    for (int r = 0; r < C::R; ++r) {
        const int gy = ty0 + wave * C::R + r;
        const bool ok = (gy < H) && (gx < W);
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            H4 hv[2], lv[2];
#pragma unroll
            for (int ge = 0; ge < 2; ++ge) {
                const int g = 2 * gp + ge;
                const int co = 8 * g + 4 * kg;
                float b8[8];                              // wave-uniform slot of 8 biases, the lane's half picked by kg
#pragma unroll
                for (int j = 0; j < 8; ++j) b8[j] = bias_c[8 * g + j];
                // (the ReLU rides on split_pair's clamp)
                const float v[4] = {accc[r][4 * g + 0] + (kg ? b8[4] : b8[0]), accc[r][4 * g + 1] + (kg ? b8[5] : b8[1]),
                                    accc[r][4 * g + 2] + (kg ? b8[6] : b8[2]), accc[r][4 * g + 3] + (kg ? b8[7] : b8[3])};
                split_pair(v[0], v[1], sat, hv[ge].u[0], lv[ge].u[0], true);
                split_pair(v[2], v[3], sat, hv[ge].u[1], lv[ge].u[1], true);
                const int off = ((gp * C::R + r) * 32 + n) * 32 + ((ge ^ ((n >> 3) & 1)) << 4) + kg * 8;
                *reinterpret_cast<half4*>(o3 + off) = hv[ge].h;
                if constexpr (NT == 3) *reinterpret_cast<half4*>(o3 + O3_PLANE + off) = lv[ge].h;
            }
            if (a.o3_hi) {      // training only: keep o3 for the backward pass (16-byte coalesced stores, see binhip_conv.hip)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    auto sw = __builtin_amdgcn_permlane32_swap(hv[0].u[k], hv[1].u[k], false, false);
                    hv[0].u[k] = sw[0]; hv[1].u[k] = sw[1];
                    if constexpr (NT == 3) {
                        auto sl = __builtin_amdgcn_permlane32_swap(lv[0].u[k], lv[1].u[k], false, false);
                        lv[0].u[k] = sl[0]; lv[1].u[k] = sl[1];
                    }
                }
                if (ok) {
                    const long long o = (long long)gp * plane_elems + ((((long long)img * H + gy) * W + gx) << 4) + kg * 8;
                    *reinterpret_cast<uint4*>(a.o3_hi + o) = make_uint4(hv[0].u[0], hv[0].u[1], hv[1].u[0], hv[1].u[1]);
                    if constexpr (NT == 3)
                        *reinterpret_cast<uint4*>(a.o3_lo + o) = make_uint4(lv[0].u[0], lv[0].u[1], lv[1].u[0], lv[1].u[1]);
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- LFF K-steps 12, 13: o3 straight from LDS ------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        half8 Bh[C::R], Bl[C::R];
#pragma unroll
        for (int r = 0; r < C::R; ++r) {
            const int off = ((t * C::R + r) * 32 + n) * 32 + ((kg ^ ((n >> 3) & 1)) << 4);
            Bh[r] = ld8(o3 + off);
            if constexpr (NT == 3) Bl[r] = ld8(o3 + O3_PLANE + off);
        }
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            const int off = (t * 96 + mt * 32) * 32 + a_lane_off;
            const half8 Ah = ld8(tailw + off);
            half8 Al;
            if constexpr (NT == 3) Al = ld8(tailw + 2 * C::LWP * 1024 + off);
#pragma unroll
            for (int r = 0; r < C::R; ++r) {
                if constexpr (NT == 3) {
                    accl[mt][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh[r], accl[mt][r], 0, 0, 0);
                    accl[mt][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl[r], accl[mt][r], 0, 0, 0);
                }
                accl[mt][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh[r], accl[mt][r], 0, 0, 0);
            }
        }
    }

    // ---- LFF epilogue: bias + block input (the RDB residual) -> next block's first 6 planes ----------------------
    const int gxc = gx < W ? gx : W - 1;
#pragma unroll
    for (int r = 0; r < C::R; ++r) {
        const int gy = ty0 + wave * C::R + r;
        const bool ok = (gy < H) && (gx < W);
        const int gyc = gy < H ? gy : H - 1;
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                H4 hv[2], lv[2];
                long long o_slot = 0;
#pragma unroll
                for (int ge = 0; ge < 2; ++ge) {
                    const int g = 2 * gp + ge;
                    const int co = mt * 32 + 8 * g + 4 * kg;
                    float b8[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) b8[j] = bias_l[mt * 32 + 8 * g + j];
                    float v[4] = {accl[mt][r][4 * g + 0] + (kg ? b8[4] : b8[0]), accl[mt][r][4 * g + 1] + (kg ? b8[5] : b8[1]),
                                  accl[mt][r][4 * g + 2] + (kg ? b8[6] : b8[2]), accl[mt][r][4 * g + 3] + (kg ? b8[7] : b8[3])};
                    const long long o = (long long)(co >> 4) * plane_elems + ((((long long)img * H + gyc) * W + gxc) << 4) + (co & 15);
                    if (ge == kg) o_slot = o - 4 * kg;
                    split_pair(v[0], v[1], sat, hv[ge].u[0], lv[ge].u[0]);
                    split_pair(v[2], v[3], sat, hv[ge].u[1], lv[ge].u[1]);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    auto sw = __builtin_amdgcn_permlane32_swap(hv[0].u[k], hv[1].u[k], false, false);
                    hv[0].u[k] = sw[0]; hv[1].u[k] = sw[1];
                    if constexpr (NT == 3) {
                        auto sl = __builtin_amdgcn_permlane32_swap(lv[0].u[k], lv[1].u[k], false, false);
                        lv[0].u[k] = sl[0]; lv[1].u[k] = sl[1];
                    }
                }
                if (ok) {
                    store16(a.y_hi, o_slot, make_uint4(hv[0].u[0], hv[0].u[1], hv[1].u[0], hv[1].u[1]), a.wt);
                    if constexpr (NT == 3)
                        store16(a.y_lo, o_slot, make_uint4(lv[0].u[0], lv[0].u[1], lv[1].u[0], lv[1].u[1]), a.wt);
                }
            }
        }
    }
    if (a.flags && __builtin_amdgcn_ballot_w64(sat != 0) != 0 && lane == 0) atomicOr(a.flags, BINHIP_FLAG_SATURATED);
}

template <int NT, int NBUF>
static int launch_tail(const TailKArgs& a0, hipStream_t s) {
    using C = TailCfg<NT, NBUF>;
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = bh_set_max_lds(&rdb_tail_kernel<NT, NBUF>, C::LDS_BYTES, lds_set)) return rc;
    TailKArgs a = a0;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + C::TH - 1) / C::TH;
    rdb_tail_kernel<NT, NBUF><<<dim3((unsigned)(a.tiles_x * a.tiles_y * a.N)), dim3(512), C::LDS_BYTES, s>>>(a, a.bc, a.bl);
    BH_CHECK_LAUNCH();
    return 0;
}

#if BINHIP_TUNING
static int g_tail_depth = 0;   // 0 = default
#endif

extern "C" {

int binhip_rdb_tail_fwd(int N, int H, int W, int nterms, const void* blk_hi, const void* blk_lo, const void* wc_hi,
                        const void* wc_lo, const float* bias_c, const void* wl_hi, const void* wl_lo, const float* bias_l,
                        void* y_hi, void* y_lo, int store_o3, void* status, void* stream) {
    if (!blk_hi || !wc_hi || !wl_hi || !bias_c || !bias_l || !y_hi) return BINHIP_E_ARG;
    if (nterms != 1 && nterms != 3) return BINHIP_E_ARG;
    if (nterms == 3 && (!blk_lo || !wc_lo || !wl_lo || !y_lo)) return BINHIP_E_ARG;
    if (N <= 0 || H <= 0 || W <= 0) return BINHIP_E_SHAPE;
    if ((long long)N * H * W >= (1ll << 26)) return BINHIP_E_SHAPE;
    TailKArgs a;
    a.x_hi = (const _Float16*)blk_hi; a.x_lo = (const _Float16*)blk_lo;
    a.wc_hi = (const _Float16*)wc_hi; a.wc_lo = (const _Float16*)wc_lo;
    a.wl_hi = (const _Float16*)wl_hi; a.wl_lo = (const _Float16*)wl_lo;
    a.bc = bias_c; a.bl = bias_l;
    a.y_hi = (_Float16*)y_hi; a.y_lo = (_Float16*)y_lo;
    const long long plane = (long long)N * H * W * 16;
    a.o3_hi = store_o3 ? (_Float16*)blk_hi + 12 * plane : nullptr;
    a.o3_lo = (store_o3 && nterms == 3) ? (_Float16*)blk_lo + 12 * plane : nullptr;
    a.flags = (unsigned*)status;
    a.N = N; a.H = H; a.W = W; a.tiles_x = a.tiles_y = 0; a.xcd_remap = 1;
    a.wt = (6 * plane * 2 < (1ll << 32) - 64) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (nterms == 1) {
#if BINHIP_TUNING
        if (g_tail_depth == 3) return launch_tail<1, 3>(a, s);
        if (g_tail_depth == 4) return launch_tail<1, 4>(a, s);
#endif
        return launch_tail<1, 2>(a, s);     // ring depth 2: measured best on MI355X (56.7 vs 63 us at 384x672)
    }
#if BINHIP_TUNING
    if (g_tail_depth == 1) return launch_tail<3, 2>(a, s);      // round-1 kernel: both planes per stage, 1 workgroup/CU
#endif
    return bh_launch_tail_x3(a, s);                              // plane-split stages, half-CU footprint (binhip_fused_x3.hip)
}

#if BINHIP_TUNING
BINHIP_API int binhip_set_tail_depth(int depth) { g_tail_depth = depth; return 0; }
#endif

}  // extern "C"
