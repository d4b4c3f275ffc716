// binhip_fused_x3.hip — the fused dense-block tail (conv #3 + ReLU + LFF 1x1 + residual; reference RDN.py:141-147,
// 162-165) in the fp32-class precision (hi/lo split, three MFMA products) with a HALF-CU footprint.
//
// Same arithmetic as rdb_tail_kernel<3, 2> (binhip_fused.hip): o3 = relu(conv3x3(blk[0:192]) + b3);
// y = LFF(cat(blk[0:192], o3)) + b + blk[0:96], with every 16-channel K-stage of conv #3 also feeding the LFF
// accumulators (the 1x1's B operand is the conv's centre-tap fragment); the residual is a VALU add from that same
// fragment (round 4; rounds 1-3: an identity MFMA) and o3 goes from the conv accumulators to the LFF's last two K-steps
// through registers only (round 4; before: wave-private LDS tiles).
// That kernel stages both precision planes per K-stage (141 KB of LDS) and its 8 waves hold 8 accumulator tiles + two
// planes of fragments (~200 VGPRs): ONE workgroup owns a whole CU, so its DMA prologue, its o3 / output epilogues and
// every barrier wait run with the matrix pipe idle, and no kernel of another stream can share the CU.
// Here: 4 waves (256 threads), tile 8 rows x 32 cols (wave w owns rows 2w, 2w+1: still R = 2, so every weight fragment
// feeds two MFMAs), K-stages split by precision plane as in binhip_conv_x3.hip:
//     hi:  conv  += Wlo*Xhi, Whi*Xhi      LFF += Llo*Xhi(centre), Lhi*Xhi(centre)   [acc += Xhi(centre), chunks 0-5]
//     lo:  conv  += Whi*Xlo               LFF += Lhi*Xlo(centre)                    [acc += Xlo(centre)]
// LDS: patch plane 11 KB double-buffered per sub-stage + (conv 9 + LFF 3 KB) x 2 planes double-buffered per chunk
// = 70 KB; registers <= 256 at one wave per SIMD.  Two such workgroups — or one of them beside a workgroup of the
// plane-split conv kernel from another stream — share a CU, each covering the other's waits.
#include "binhip_fused.h"
#include <type_traits>
#ifndef BINHIP_X3_PRIO
#define BINHIP_X3_PRIO 0      // bit 1: progress-ordered wave priority in this kernel (binhip_conv_x3.hip explains)
#endif
#ifndef BINHIP_EPI_PRIO
#define BINHIP_EPI_PRIO 1        // K loop at wave priority 2, epilogues at 0: see binhip_conv_x3.hip
#endif
#ifndef BINHIP_TAIL_RES_MFMA
#define BINHIP_TAIL_RES_MFMA 0   // side builds: 1 = the residual as an identity MFMA (rounds 1-3), for A/B runs
#endif

namespace {

struct TX {
    static constexpr int R = 2, NW = 4, TH = 8, PH = 10, PW = 34;
    static constexpr int PP = (PH * PW * 2 + 63) / 64;          // 11 one-KiB pieces per patch plane
    static constexpr int CWP = 9, LWP = 3;                      // conv / LFF weight pieces per plane and chunk
    static constexpr int PATCH_BYTES = PP * 1024;
    static constexpr int WPL = (CWP + LWP) * 1024;              // one weight plane of a chunk: conv taps, then LFF rows
    static constexpr int WBUF_BYTES = 2 * WPL;                  // hi plane, lo plane
    static constexpr int W_OFF = 2 * PATCH_BYTES;
    static constexpr int LDS_BYTES = W_OFF + 2 * WBUF_BYTES;    // 71 680 B
    static constexpr int NCHUNK = 12;
    static constexpr int NPJ = (PP + NW - 1) / NW;
    // after the K-loop: LFF weights of chunks 12/13 sit in weight buffer 0 ([hi: 2 x 3 KiB][lo: 2 x 3 KiB]); o3 itself
    // stays in registers (the swapped epilogue slots ARE the B fragments of those two K-steps)
    static constexpr int TAILW_OFF = W_OFF;
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

__device__ __forceinline__ half8 ld8(const char* p) { return *reinterpret_cast<const half8*>(p); }

__device__ __forceinline__ void tx_issue_patch(const TailKArgs& a, char* smem, int c, int pl, int wave, const unsigned* voff,
                                               long long plane_elems, unsigned plane_bytes) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((pl ? a.x_lo : a.x_hi) + (long long)c * plane_elems), 0, plane_bytes, 0x00020000);
    char* lds = smem + pl * TX::PATCH_BYTES;                   // hi planes live in ring slot 0, lo planes in slot 1
#pragma unroll
    for (int j = 0; j < TX::NPJ; ++j) {
        const int i = wave + TX::NW * j;
        if (i < TX::PP) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(lds + i * 1024), 16, voff[j], 0, 0, 0);
    }
}

// weights of chunk c (conv 9 KiB + LFF 3 KiB per plane) -> weight buffer `buf`: 24 pieces over 4 waves
__device__ __forceinline__ void tx_issue_weights(const TailKArgs& a, char* smem, int c, int buf, int wave, int lane) {
    __amdgpu_buffer_rsrc_t ch = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wc_hi + (long long)c * (9 * 32 * 16)), 0, 9 * 1024, 0x00020000);
    __amdgpu_buffer_rsrc_t cl = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wc_lo + (long long)c * (9 * 32 * 16)), 0, 9 * 1024, 0x00020000);
    __amdgpu_buffer_rsrc_t lh = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wl_hi + (long long)c * (96 * 16)), 0, 3 * 1024, 0x00020000);
    __amdgpu_buffer_rsrc_t ll = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wl_lo + (long long)c * (96 * 16)), 0, 3 * 1024, 0x00020000);
    char* lds = smem + TX::W_OFF + buf * TX::WBUF_BYTES;
    const unsigned v = (unsigned)(lane * 16);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int i = wave + TX::NW * j;                       // 0..23: [conv hi 0-8][LFF hi 9-11][conv lo 12-20][LFF lo 21-23]
        const int pl = i >= 12 ? 1 : 0;
        const int k = i - 12 * pl;                             // piece within the plane
        const bool lff = k >= TX::CWP;
        const int t = lff ? k - TX::CWP : k;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pl ? (lff ? ll : cl) : (lff ? lh : ch), (lds_void_t*)(lds + i * 1024), 16, v,
                                                 t * 1024, 0, 0);
    }
}

// LFF weights of the two o3 chunks (12, 13): [hi: chunk 12 (3 KiB), chunk 13][lo: ...] -> weight buffer 0
__device__ __forceinline__ void tx_issue_tailw(const TailKArgs& a, char* smem, int wave, int lane) {
    __amdgpu_buffer_rsrc_t lh = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wl_hi + (long long)12 * (96 * 16)), 0, 6 * 1024, 0x00020000);
    __amdgpu_buffer_rsrc_t ll = __builtin_amdgcn_make_buffer_rsrc((void*)(a.wl_lo + (long long)12 * (96 * 16)), 0, 6 * 1024, 0x00020000);
    char* lds = smem + TX::TAILW_OFF;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int i = wave + TX::NW * j;                       // 0..11: hi pieces 0-5, lo pieces 6-11
        const int pl = i >= 6 ? 1 : 0;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pl ? ll : lh, (lds_void_t*)(lds + i * 1024), 16, (unsigned)(lane * 16),
                                                 (i - 6 * pl) * 1024, 0, 0);
    }
}

// One sub-stage of chunk `st`.  HI: hi patch plane against both weight planes; !HI: lo patch plane against the hi weights.
// A flat list of 12 steps — conv taps of column 0, of column 1, the three LFF row blocks (their B operand is the centre
// tap's fragment: row r+1 of column 1), conv taps of column 2 — with the A fragments of step s+1 and the patch rows of the
// next column fetched before the MFMAs of step s.
// Residual (RDN.py:165, `+ x`): the block input IS the centre-tap B fragment of chunks 0-5 — lane (n, kg) holds channels
// 8kg..8kg+7 of pixel n, the accumulator tile wants channels 8g + 4kg + j.  One v_permlane32_swap per dword pair hands
// lanes n / n+32 each other's halves (the inverse of the store epilogue's swap), then 8 cvt + 8 fp32 adds per row — on
// the VALU, in the shadow of the matrix pipe, instead of 24 identity MFMAs per wave and tile (rounds 1-3).
// Row r of the residual of chunk RES (K = RES >> 1: accumulator tile, HF = RES & 1: its channel half).
template <int RES>
__device__ __forceinline__ void tx_add_residual_row(const half8& bc, floatx16 (&accl)[3][TX::R], int r) {
    constexpr int K = RES >> 1, HF = RES & 1;
    union H8 { half8 h; unsigned u[4]; };
    H8 v;
    v.h = bc;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        auto sw = __builtin_amdgcn_permlane32_swap(v.u[k], v.u[2 + k], false, false);
        v.u[k] = sw[0]; v.u[2 + k] = sw[1];
    }
#pragma unroll
    for (int ge = 0; ge < 2; ++ge)
#pragma unroll
        for (int j = 0; j < 4; ++j) accl[K][r][4 * (2 * HF + ge) + j] += (float)v.h[4 * ge + j];
}

// RES: chunk index 0..5 when the sub-stage BEFORE this one left a residual to add (its centre-column fragments, `carry`),
// else -1 — a compile-time constant (it names accumulator registers): the kernel peels chunks 0-6.  The ~16 VALU
// instructions per row go between the issue of this sub-stage's first fragment reads and its first MFMA, i.e. under the
// LDS latency the wave would otherwise sit out; every sub-stage leaves its own centre column in `carry`.
template <bool HI, int RES>
__device__ __forceinline__ void tx_compute(const char* pb, const char* wb, int a_lane_off, int b_lane_off,
                                           floatx16 (&accc)[TX::R], floatx16 (&accl)[3][TX::R], half8 (&carry)[TX::R]) {
    constexpr int R = TX::R;
    constexpr int NSTEP = 12;
    half8 B[2][R + 2];
    half8 Ah[2], Al[2];
    auto load_b = [&](int dx, half8 (&dst)[R + 2]) {
#pragma unroll
        for (int rr = 0; rr < R + 2; ++rr) dst[rr] = ld8(pb + b_lane_off + (rr * TX::PW + dx) * 16);
    };
    // step -> (is LFF, dx, dy or mt)
    auto step_kind = [](int s, bool& lff, int& dx, int& k) {
        if (s < 3) { lff = false; dx = 0; k = s; }
        else if (s < 6) { lff = false; dx = 1; k = s - 3; }
        else if (s < 9) { lff = true; dx = 1; k = s - 6; }
        else { lff = false; dx = 2; k = s - 9; }
    };
    auto load_a = [&](int s, half8& h, half8& l) {
        bool lff; int dx, k;
        step_kind(s, lff, dx, k);
        const int off = (lff ? TX::CWP * 1024 + (k * 32) * 32 : ((k * 3 + dx) * 32) * 32) + a_lane_off;
        h = ld8(wb + off);
        if constexpr (HI) l = ld8(wb + TX::WPL + off);
    };
#if BINHIP_TAIL_RES_MFMA
    // rounds 1-3: the residual as an identity MFMA (row m of A selects input channel k of the chunk when m == 16 HF + k)
    half8 ident;
    {
        const int n = threadIdx.x & 31, kg = (threadIdx.x >> 5) & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) ident[e] = (n == (RES & 1) * 16 + kg * 8 + e) ? (_Float16)1.0f : (_Float16)0.0f;
    }
#endif
    load_b(0, B[0]);
    load_a(0, Ah[0], Al[0]);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        bool lff; int dx, k;
        step_kind(s, lff, dx, k);
        if (s + 1 < NSTEP) load_a(s + 1, Ah[(s + 1) & 1], Al[(s + 1) & 1]);
        if (s == 0) load_b(1, B[1]);
        if (s == 3) load_b(2, B[0]);                           // column 0's rows are dead after step 2
        __builtin_amdgcn_sched_barrier(0);
#if !BINHIP_TAIL_RES_MFMA
        if constexpr (RES >= 0) {
            if (s == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) tx_add_residual_row<RES>(carry[r], accl, r);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        if (!lff) {
            const int set = dx & 1;
            if constexpr (HI) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    accc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[s & 1], B[set][r + k], accc[r], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
                accc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s & 1], B[set][r + k], accc[r], 0, 0, 0);
        } else {
#if BINHIP_TAIL_RES_MFMA
            if constexpr (RES >= 0) {
                if (k == (RES >> 1)) {
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        accl[k][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ident, B[1][r + 1], accl[k][r], 0, 0, 0);
                }
            }
#endif
            if constexpr (HI) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    accl[k][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[s & 1], B[1][r + 1], accl[k][r], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
                accl[k][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s & 1], B[1][r + 1], accl[k][r], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) carry[r] = B[1][r + 1];
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
rdb_tail_x3_kernel(const TailKArgs a, const float* __restrict__ bias_c, const float* __restrict__ bias_l) {
    // (the two biases again as the kernel's own restrict parameters: scalar loads in the epilogues, binhip_conv_common.h)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BH_TL_DECL;
    BH_TL_BEGIN();
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
    const int tx0 = tx * 32, ty0 = ty * TX::TH;
    const int H = a.H, W = a.W;
    const long long plane_elems = (long long)a.N * H * W * 16;
    const unsigned plane_bytes = (unsigned)(plane_elems * 2);

    unsigned voff[TX::NPJ];
#pragma unroll
    for (int j = 0; j < TX::NPJ; ++j) {
        const int i = wave + TX::NW * j;
        const int q = i * 64 + lane;                            // LDS image = [channel half][patch pixel][16 B]
        const int cg = q >= TX::PH * TX::PW ? 1 : 0;
        const int p = q - cg * (TX::PH * TX::PW);
        const int py = p / TX::PW, px = p - py * TX::PW;
        const int gy = ty0 + py - 1, gx = tx0 + px - 1;
        const bool ok = (p < TX::PH * TX::PW) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        voff[j] = ok ? (unsigned)((((long long)img * H + gy) * W + gx) * 32 + cg * 16) : 0x80000000u;
    }

    floatx16 accc[TX::R];
    floatx16 accl[3][TX::R];
#pragma unroll
    for (int r = 0; r < TX::R; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            accc[r][e] = 0.f;
            accl[0][r][e] = 0.f; accl[1][r][e] = 0.f; accl[2][r][e] = 0.f;
        }
    const int a_lane_off = n * 32 + ((kg ^ ((n >> 3) & 1)) << 4);
    const int b_lane_off = (kg * (TX::PH * TX::PW) + wave * TX::R * TX::PW + n) * 16;

#if BINHIP_EPI_PRIO
    __builtin_amdgcn_s_setprio(2);
#endif
    tx_issue_weights(a, smem, 0, 0, wave, lane);
    tx_issue_patch(a, smem, 0, 0, wave, voff, plane_elems, plane_bytes);
    // one 16-channel chunk = a hi and a lo sub-stage.  Chunks 0-5 (the block input) also carry the residual, with their index
    // as a compile-time constant (which accumulator registers it lands in): those six are peeled, 6-11 stay a loop.
    half8 carry[TX::R];
    auto chunk = [&](const int c, auto res_hi, auto res_lo) __attribute__((always_inline)) {
        constexpr int RH = decltype(res_hi)::value, RL = decltype(res_lo)::value;
        const char* wb = smem + TX::W_OFF + (c & 1) * TX::WBUF_BYTES;
#if BINHIP_X3_PRIO & 2
        {   // priority falls with progress (see binhip_conv_x3.hip): the workgroup that lags its CU neighbour issues first
            const int q = (4 * c) / TX::NCHUNK;
            if (q == 0) __builtin_amdgcn_s_setprio(3);
            else if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
        // ---- hi sub-stage; meanwhile the lo plane and the next chunk's weights (last chunk: the o3 LFF weights) land
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#if BINHIP_TIMELINE
        if (c == 0) BH_TL_STAMP(1);
#endif
        tx_issue_patch(a, smem, c, 1, wave, voff, plane_elems, plane_bytes);
        if (c + 1 < TX::NCHUNK) tx_issue_weights(a, smem, c + 1, (c + 1) & 1, wave, lane);
        else tx_issue_tailw(a, smem, wave, lane);                // weight buffer 0: chunk 10's, read for the last time in lo(10)
        tx_compute<true, RH>(smem, wb, a_lane_off, b_lane_off, accc, accl, carry);
        // ---- lo sub-stage; meanwhile the next chunk's hi plane lands
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < TX::NCHUNK) tx_issue_patch(a, smem, c + 1, 0, wave, voff, plane_elems, plane_bytes);
        tx_compute<false, RL>(smem + TX::PATCH_BYTES, wb, a_lane_off, b_lane_off, accc, accl, carry);
    };
    using std::integral_constant;
#if BINHIP_TAIL_RES_MFMA
    // (identity-MFMA form: a sub-stage adds its OWN chunk's plane)
    chunk(0, integral_constant<int, 0>{}, integral_constant<int, 0>{});
    chunk(1, integral_constant<int, 1>{}, integral_constant<int, 1>{});
    chunk(2, integral_constant<int, 2>{}, integral_constant<int, 2>{});
    chunk(3, integral_constant<int, 3>{}, integral_constant<int, 3>{});
    chunk(4, integral_constant<int, 4>{}, integral_constant<int, 4>{});
    chunk(5, integral_constant<int, 5>{}, integral_constant<int, 5>{});
    chunk(6, integral_constant<int, -1>{}, integral_constant<int, -1>{});
#else
    // hi(c) adds the lo plane of chunk c - 1, lo(c) the hi plane of chunk c: each one sub-stage after it was read
    chunk(0, integral_constant<int, -1>{}, integral_constant<int, 0>{});
    chunk(1, integral_constant<int, 0>{}, integral_constant<int, 1>{});
    chunk(2, integral_constant<int, 1>{}, integral_constant<int, 2>{});
    chunk(3, integral_constant<int, 2>{}, integral_constant<int, 3>{});
    chunk(4, integral_constant<int, 3>{}, integral_constant<int, 4>{});
    chunk(5, integral_constant<int, 4>{}, integral_constant<int, 5>{});
    chunk(6, integral_constant<int, 5>{}, integral_constant<int, -1>{});
#endif
    for (int c = 7; c < TX::NCHUNK; ++c) chunk(c, integral_constant<int, -1>{}, integral_constant<int, -1>{});
    // every wave has finished reading the patch ring and weight buffer 1 (the o3 staging area); the o3 LFF weights landed
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

#if BINHIP_EPI_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    // ---- conv #3 epilogue: bias + ReLU + hi/lo split; o3 never leaves the registers -------------------------------------
    // The accumulator tile gives lane (n, kg) channels 8g + 4kg + j; after the store epilogue's v_permlane32_swap of a
    // (g even, g odd) pair lane (n, kg) owns the full 16-byte slot kg of 16-channel group gp — channels 16gp + 8kg .. +7 of
    // pixel n — which is exactly the B fragment (k = 8kg .. 8kg+7 of chunk 12 + gp) the LFF's last two K-steps want.
    // (Round 2 staged o3 through wave-private LDS tiles with 8-byte stores: 258 048 bank conflicts per launch.)
    const int gx = tx0 + n;
    union H4 { half4 h; unsigned u[2]; };
    union H8 { half8 h; unsigned u[4]; };
    unsigned sat = 0;
    half8 Bh[2][TX::R], Bl[2][TX::R];
#pragma unroll
    for (int r = 0; r < TX::R; ++r) {
        const int gy = ty0 + wave * TX::R + r;
        const bool ok = (gy < H) && (gx < W);
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            H4 hv[2], lv[2];
#pragma unroll
            for (int ge = 0; ge < 2; ++ge) {
                const int g = 2 * gp + ge;
                float b8[8];                              // wave-uniform slot of 8 biases, the lane's half picked by kg
#pragma unroll
                for (int j = 0; j < 8; ++j) b8[j] = bias_c[8 * g + j];
                // (the ReLU rides on split_pair's clamp)
                const float v[4] = {accc[r][4 * g + 0] + (kg ? b8[4] : b8[0]), accc[r][4 * g + 1] + (kg ? b8[5] : b8[1]),
                                    accc[r][4 * g + 2] + (kg ? b8[6] : b8[2]), accc[r][4 * g + 3] + (kg ? b8[7] : b8[3])};
                split_pair(v[0], v[1], sat, hv[ge].u[0], lv[ge].u[0], true);
                split_pair(v[2], v[3], sat, hv[ge].u[1], lv[ge].u[1], true);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                auto sw = __builtin_amdgcn_permlane32_swap(hv[0].u[k], hv[1].u[k], false, false);
                hv[0].u[k] = sw[0]; hv[1].u[k] = sw[1];
                auto sl = __builtin_amdgcn_permlane32_swap(lv[0].u[k], lv[1].u[k], false, false);
                lv[0].u[k] = sl[0]; lv[1].u[k] = sl[1];
            }
            H8 fh, fl;
            fh.u[0] = hv[0].u[0]; fh.u[1] = hv[0].u[1]; fh.u[2] = hv[1].u[0]; fh.u[3] = hv[1].u[1];
            fl.u[0] = lv[0].u[0]; fl.u[1] = lv[0].u[1]; fl.u[2] = lv[1].u[0]; fl.u[3] = lv[1].u[1];
            Bh[gp][r] = fh.h;
            Bl[gp][r] = fl.h;
            if (a.o3_hi && ok) {      // training only: keep o3 for the backward pass (16-byte coalesced stores)
                const long long o = (long long)gp * plane_elems + ((((long long)img * H + gy) * W + gx) << 4) + kg * 8;
                *reinterpret_cast<uint4*>(a.o3_hi + o) = make_uint4(fh.u[0], fh.u[1], fh.u[2], fh.u[3]);
                *reinterpret_cast<uint4*>(a.o3_lo + o) = make_uint4(fl.u[0], fl.u[1], fl.u[2], fl.u[3]);
            }
        }
    }

    // ---- LFF K-steps 12, 13 on the o3 fragments (hi sub-stage products first, then the lo plane: same order as above)
    const char* tailw = smem + TX::TAILW_OFF;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            const int off = (t * 96 + mt * 32) * 32 + a_lane_off;
            const half8 Ah = ld8(tailw + off);
            const half8 Al = ld8(tailw + 6 * 1024 + off);
#pragma unroll
            for (int r = 0; r < TX::R; ++r) {
                accl[mt][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh[t][r], accl[mt][r], 0, 0, 0);
                accl[mt][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh[t][r], accl[mt][r], 0, 0, 0);
                accl[mt][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl[t][r], accl[mt][r], 0, 0, 0);
            }
        }
    }
    if (a.flags && __builtin_amdgcn_ballot_w64(sat != 0) != 0 && lane == 0) atomicOr(a.flags, BINHIP_FLAG_SATURATED);
    BH_TL_STAMP(2);

    // ---- LFF epilogue: bias (the block input = RDB residual is already in the accumulators) -> 6 output planes -------
    ConvKArgs e;
    e.bias = a.bl; e.y_hi = a.y_hi; e.y_lo = a.y_lo;
    e.r_hi = e.r_lo = e.r2_hi = e.r2_lo = e.m_hi = nullptr;
    e.flags = a.flags;
    e.N = a.N; e.H = H; e.W = W;
    e.relu = 0; e.has_res = 0; e.cout = 96; e.wt = a.wt;
    e.och_limit = 6; e.res_chunks = 0; e.mask_from = 0; e.y_cpg = 0; e.y_cpg_inv = 0; e.y_group_stride = 0; e.y_unshuf = 0;
    conv_epilogue<3, TX::R, 3, BINHIP_EPI_PLANES, false>(e, bias_l, accl, img, ty0 + wave * TX::R, tx0, 0, true, n, kg, plane_elems);
    BH_TL_FINISH(a, blockIdx.x);
}

}  // namespace

int bh_launch_tail_x3(const TailKArgs& a0, hipStream_t s) {
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = bh_set_max_lds(&rdb_tail_x3_kernel, TX::LDS_BYTES, lds_set)) return rc;
    TailKArgs a = a0;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + TX::TH - 1) / TX::TH;
#if BINHIP_TIMELINE
    a.tl = bh_tl_reserve((unsigned)(a.tiles_x * a.tiles_y * a.N), &a.tl_base);
    a.tl_launch = (2u << 24) | (g_bh_tl_serial.fetch_add(1) & 0xFFFFFFu);      // kind 2 = the fused tail
#endif
    rdb_tail_x3_kernel<<<dim3((unsigned)(a.tiles_x * a.tiles_y * a.N)), dim3(256), TX::LDS_BYTES, s>>>(a, a.bc, a.bl);
    BH_CHECK_LAUNCH();
    return 0;
}
