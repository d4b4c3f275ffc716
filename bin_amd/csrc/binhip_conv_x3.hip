// binhip_conv_x3.hip — the fp32-class (hi/lo split, three MFMA products) 3x3 / Cout-block-32 convolution with
// PLANE-SPLIT K-stages, sized so that TWO workgroups share a CU.
//
// Same math and data layouts as conv_mfma_kernel<..., NT = 3, ...> (binhip_conv.hip; reference RDN.py:141 RDB_Conv,
// :207 UPNet.2 and the gather-form dense-block backward-data convs of the same shape), different pipeline.  The
// generic kernel stages BOTH precision planes of a 16-channel chunk per K-stage (patch hi+lo 40 KB, weights hi+lo
// 18 KB, double-buffered = 117 KB) and holds two full fragment sets (157 VGPRs): one 8-wave workgroup per CU, so a
// launch of 504 tiles runs as two rounds whose DMA prologues and store epilogues overlap with nothing, and kernels
// of other streams cannot co-reside.  Here a chunk is two sub-stages,
//     hi:  acc += Wlo * Xhi ; acc += Whi * Xhi      (patch hi plane, both weight planes)
//     lo:  acc += Whi * Xlo                         (patch lo plane, hi weights again)
// with the 20 KB patch planes double-buffered per SUB-stage and the 18 KB weight slab double-buffered per CHUNK:
// 2 x 20 + 2 x 18 + 1 = 77 KB of LDS and <= 128 VGPRs (one B set per tap column, A fragments per tap) ->
// 2 workgroups / CU = 4 waves / SIMD.  One workgroup's prologue / epilogue / barrier waits are covered by the other's
// MFMAs (they drift apart on their own: the matrix pipe arbitrates by age), and tails of a launch are filled by the
// next stream's workgroups.
// Measured on MI355X, whole 720p forward, same box (profiles/r02_layer_variants.md): 81.5 ms with the generic kernel,
// 74.9 ms with this one.  A 16-wave / 32x32-tile variant (one workgroup per CU whose halves share one weight slab, 3-deep
// patch ring with counted vmcnt = prefetch distance of two sub-stages) was built and measured at 76.4 ms: the DMA
// prefetch distance is not what limits the loop, and a 1024-thread workgroup loses the phase drift.  Not kept.
// Neither was a 4-wave / 8x32-tile variant for grids that give each CU only one 16x32 workgroup (training crops,
// 8 x 128x128 = 256 tiles): training step 176.0 vs 173.1 ms, 720p window 76.0 vs 72.2 ms (twice the weight DMA per pixel).
#include "binhip_conv_common.h"
#ifndef BINHIP_ABLATE
#define BINHIP_ABLATE 0
#endif
// wave tile of the default instantiation: R output rows per wave, WN waves per workgroup (tile = R*WN x 32 pixels); side builds
// (tools/) override it to measure other shapes
#ifndef BINHIP_X3_R
#define BINHIP_X3_R 2
#endif
#ifndef BINHIP_X3_WN
#define BINHIP_X3_WN 8
#endif
// bit 0 (this kernel) / bit 1 (the fused tail, binhip_fused_x3.hip): wave priority falls with the tile's progress (s_setprio 3 .. 0
// per quarter of the K loop), so of the two workgroups sharing a CU the one that lags issues first and they finish together
// instead of one running alone at the end of the launch
#ifndef BINHIP_X3_PRIO
#define BINHIP_X3_PRIO 0
#endif
// 1 (product, round 6) = waves run their K loop at priority 2 and their epilogue at 0, so that a workgroup's epilogue VALU work does not
// take issue slots from its CU partner's MFMAs: window -0.46 % (6 of 6 alternating pairs), training step -0.15 % (5 of 6); 0 = side builds
#ifndef BINHIP_PROG_PRIO
#define BINHIP_PROG_PRIO 1     // 0 = side builds without the progress-ordered priority of one-round launches
#endif
#ifndef BINHIP_EPI_PRIO
#define BINHIP_EPI_PRIO 1
#endif

template <int KS, int R, int WN>
struct X3Cfg {
    static constexpr int PAD = KS / 2;
    static constexpr int TH = R * WN;
    static constexpr int PH = TH + KS - 1;
    static constexpr int PW = 32 + KS - 1;
    static constexpr int PP = (PH * PW * 2 + 63) / 64;     // 1-KiB DMA pieces of one patch plane
    static constexpr int WP = KS * KS;                     // 1-KiB pieces of one weight plane (32 rows x 32 B per tap)
    static constexpr int NW = WN;
    static constexpr int PATCH_BYTES = PP * 1024;
    static constexpr int WBUF_BYTES = 2 * WP * 1024;       // hi taps, then lo taps
    static constexpr int LDS_BYTES = 2 * PATCH_BYTES + 2 * WBUF_BYTES;
    static constexpr int NPJ = (PP + NW - 1) / NW;
    static constexpr int NWJ = (2 * WP + NW - 1) / NW;
    // 3x3: two workgroups per CU.  5x5 (SFENet1, round 4): the 25-tap weight slab alone is 50 KB per chunk, so ONE workgroup per CU
    // — still with both rings double-buffered (the generic kernel had to single-buffer its 96 KB stage: NBUF = 1, nothing overlapped)
    static_assert((KS == 3 ? 2 : 1) * LDS_BYTES <= 160 * 1024, "LDS budget");
};

// cache-policy bits of the patch DMA (side builds; 0 = default policy in the product): 2 = nt.  Measured in profiles/r06_experiments.md.
#ifndef BINHIP_X3_PATCH_AUX
#define BINHIP_X3_PATCH_AUX 0
#endif
template <class C>
__device__ __forceinline__ void x3_issue_patch(const ConvKArgs& a, char* smem, int c, int pl, int buf, int wave,
                                               const unsigned* voff, long long plane_elems, unsigned plane_bytes) {
    const _Float16* xb = pl ? a.x_lo : a.x_hi;
    const long long coff = (a.cpg > 0) ? (long long)(c / a.cpg) * a.group_stride + (long long)(c % a.cpg) * plane_elems
                                       : (long long)c * plane_elems;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xb + coff), 0, plane_bytes, 0x00020000);
    char* lds = smem + buf * C::PATCH_BYTES;
    // every sub-stage is waited for with vmcnt(0), so waves need not issue equal instruction counts: pieces beyond the
    // image are simply skipped (wave-uniform branch)
#pragma unroll
    for (int j = 0; j < C::NPJ; ++j) {
        const int i = wave + C::NW * j;
        if (BINHIP_ABLATE != 4 && ((C::PP % C::NW == 0) || (i < C::PP)))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(lds + i * 1024), 16, voff[j], 0, 0, BINHIP_X3_PATCH_AUX);
    }
}

// piece j of the patch DMA alone (BINHIP_X3_SPREAD)
template <class C>
__device__ __forceinline__ void x3_issue_patch_piece(const ConvKArgs& a, char* smem, int c, int pl, int buf, int wave, int j,
                                                     const unsigned* voff, long long plane_elems, unsigned plane_bytes) {
    const _Float16* xb = pl ? a.x_lo : a.x_hi;
    const long long coff = (a.cpg > 0) ? (long long)(c / a.cpg) * a.group_stride + (long long)(c % a.cpg) * plane_elems
                                       : (long long)c * plane_elems;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xb + coff), 0, plane_bytes, 0x00020000);
    const int i = wave + C::NW * j;
    if ((C::PP % C::NW == 0) || (i < C::PP))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + buf * C::PATCH_BYTES + i * 1024), 16, voff[j], 0, 0, 0);
}
template <class C, int KS>
__device__ __forceinline__ void x3_issue_weights_piece(const ConvKArgs& a, char* smem, int c, int buf, int wave, int lane, int z, int j) {
    const long long woff = ((long long)z * a.nchunks + c) * (KS * KS * 32 * 16);
    const int i = wave + C::NW * j;
    if (((2 * C::WP) % C::NW == 0) || (i < 2 * C::WP)) {
        const bool lo = i >= C::WP;
        const int t = lo ? i - C::WP : i;
        __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc((void*)((lo ? a.w_lo : a.w_hi) + woff), 0, KS * KS * 1024, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w, (lds_void_t*)(smem + 2 * C::PATCH_BYTES + buf * C::WBUF_BYTES + i * 1024), 16,
                                                 (unsigned)(lane * 16), t * 1024, 0, 0);
    }
}

template <class C, int KS>
__device__ __forceinline__ void x3_issue_weights(const ConvKArgs& a, char* smem, int c, int buf, int wave, int lane, int z) {
    const long long woff = ((long long)z * a.nchunks + c) * (KS * KS * 32 * 16);
    __amdgpu_buffer_rsrc_t wh = __builtin_amdgcn_make_buffer_rsrc((void*)(a.w_hi + woff), 0, KS * KS * 1024, 0x00020000);
    __amdgpu_buffer_rsrc_t wl = __builtin_amdgcn_make_buffer_rsrc((void*)(a.w_lo + woff), 0, KS * KS * 1024, 0x00020000);
    char* lds = smem + 2 * C::PATCH_BYTES + buf * C::WBUF_BYTES;
#pragma unroll
    for (int j = 0; j < C::NWJ; ++j) {
        const int i = wave + C::NW * j;                  // piece: 0..WP-1 hi taps, WP..2WP-1 lo taps
        if (BINHIP_ABLATE != 4 && (((2 * C::WP) % C::NW == 0) || (i < 2 * C::WP))) {
            const bool lo = i >= C::WP;
            const int t = lo ? i - C::WP : i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(lo ? wl : wh, (lds_void_t*)(lds + i * 1024), 16,
                                                     (unsigned)(lane * 16), t * 1024, 0, 0);
        }
    }
}

// BINHIP_ABLATE (side builds only, 0 in the product; results are garbage by construction): 1 = no MFMA (fragment loads kept
// alive), 2 = no LDS fragment loads (MFMA on undefined registers), 4 = no LDS-DMA — used with tools/power_probe.py to split the
// package power of the kernel into its matrix / LDS / DMA shares
__device__ __forceinline__ half8 x3_ld8(const char* p) {
#if BINHIP_ABLATE == 2
    half8 v;
    asm volatile("" : "=v"(v));
    return v;
#else
    return *reinterpret_cast<const half8*>(p);
#endif
}
__device__ __forceinline__ floatx16 x3_mfma(half8 a, half8 b, floatx16 c) {
#if BINHIP_ABLATE == 1
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

// One sub-stage out of LDS.  HI: both weight planes against the hi patch (2 products); !HI: hi weights against the lo
// patch.  Tap order dx-major: the R+KS-1 patch-row fragments of a tap column are fetched once and serve its KS taps;
// weight fragments are fetched one tap ahead, the next column's patch rows at the column's first tap.
// BINHIP_X3_SPREAD (side builds; 0 in the product): the DMA instructions of the next sub-stage go out one per tap BEHIND that tap's
// MFMAs (`hook(step)`) instead of as a burst at the top of the sub-stage — a wave executes in order and a burst of
// buffer_load ... lds sits at the head of its stream until the memory pipeline has taken all of it (binhip_wgrad.hip gained 5 %
// from the same change); measured for this kernel in profiles/r06_experiments.md.
#ifndef BINHIP_X3_SPREAD
#define BINHIP_X3_SPREAD 0
#endif
struct X3NoHook { __device__ __forceinline__ void operator()(int) const {} };

template <class C, int KS, int R, bool HI, class Hook = X3NoHook>
__device__ __forceinline__ void x3_compute(const char* pb, const char* wb, int a_lane_off, int b_lane_off,
                                           floatx16 (&acc)[R], Hook hook = Hook{}) {
    constexpr int NTAP = KS * KS;
    half8 B[2][R + KS - 1];
    half8 Ah[2], Al[2];
    auto load_b = [&](int dx, half8 (&dst)[R + KS - 1]) {
#pragma unroll
        for (int rr = 0; rr < R + KS - 1; ++rr) dst[rr] = x3_ld8(pb + b_lane_off + (rr * C::PW + dx) * 16);
    };
    auto load_a = [&](int s, half8& h, half8& l) {
        const int dx = s / KS, dy = s % KS;
        const int off = ((dy * KS + dx) * 32) * 32 + a_lane_off;
        h = x3_ld8(wb + off);
        if constexpr (HI) l = x3_ld8(wb + C::WP * 1024 + off);
    };
    load_b(0, B[0]);
    load_a(0, Ah[0], Al[0]);
#pragma unroll
    for (int s = 0; s < NTAP; ++s) {
        const int dx = s / KS, dy = s % KS;
        if (s + 1 < NTAP) load_a(s + 1, Ah[(s + 1) & 1], Al[(s + 1) & 1]);
        if (dy == 0 && dx + 1 < KS) load_b(dx + 1, B[(dx + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (HI) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                acc[r] = x3_mfma(Al[s & 1], B[dx & 1][r + dy], acc[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            acc[r] = x3_mfma(Ah[s & 1], B[dx & 1][r + dy], acc[r]);
        hook(s);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The LAST chunk of a layer whose upper 8 input channels are zero padding (SFENet1, RDN.py:187/245: 24 = 16 + 8 and 36 = 2 x 16 + 4
// channels), round 5: the K = 16 of an MFMA is spent on TWO taps of the lower channel half instead of one tap of both halves.
// Lanes 32-63 (k = 8..15) read the patch one row down and the weights of tap (dy + 1, dx), both from channel half 0, so one
// instruction covers taps (dy, dx) and (dy + 1, dx); the last row tap dy = KS - 1 has no partner and keeps the plain form (its
// k = 8..15 lanes see the zero half of patch and weights).  KS * KS -> KS * (KS + 1) / 2 K-steps for that chunk: 25 -> 15.
// Same LDS images, same weight layout — only the per-lane addresses differ.  (The row below the pair's second tap is inside the
// patch for every dy <= KS - 2; nothing is read beyond it.)
template <class C, int KS, int R, bool HI>
__device__ __forceinline__ void x3_compute_pair(const char* pb, const char* wb, int n, int kg, int wave, floatx16 (&acc)[R]) {
    static_assert(KS & 1, "odd kernels");
    constexpr int NP = KS / 2;                                          // paired steps per tap column (+ 1 plain step)
    const int s16 = ((n >> 3) & 1) << 4;
    const int a_pair = n * 32 + s16 + kg * (KS * 1024);                // half 0 (slot s) of tap (dy + kg, dx)
    const int a_plain = n * 32 + ((kg << 4) ^ s16);                    // half kg of tap (KS - 1, dx)
    const int b_pair = ((wave * R + kg) * C::PW + n) * 16;             // channel half 0, row + kg
    const int b_plain = (kg * (C::PH * C::PW) + wave * R * C::PW + n) * 16;
    half8 Bp[2][R + KS - 3], Bl[2][R];
    half8 Ah[2], Al[2];
    auto load_b = [&](int dx, half8 (&bp)[R + KS - 3], half8 (&bl)[R]) {
#pragma unroll
        for (int rr = 0; rr < R + KS - 3; ++rr) bp[rr] = x3_ld8(pb + b_pair + (rr * C::PW + dx) * 16);
#pragma unroll
        for (int r = 0; r < R; ++r) bl[r] = x3_ld8(pb + b_plain + ((r + KS - 1) * C::PW + dx) * 16);
    };
    auto load_a = [&](int s, half8& h, half8& l) {                     // step s = dx * (NP + 1) + j ; j < NP: pair (2j, 2j + 1), j == NP: plain
        const int dx = s / (NP + 1), j = s % (NP + 1);
        const int off = (j < NP) ? ((2 * j) * KS + dx) * 1024 + a_pair : ((KS - 1) * KS + dx) * 1024 + a_plain;
        h = x3_ld8(wb + off);
        if constexpr (HI) l = x3_ld8(wb + C::WP * 1024 + off);
    };
    constexpr int NS = KS * (NP + 1);
    load_b(0, Bp[0], Bl[0]);
    load_a(0, Ah[0], Al[0]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int dx = s / (NP + 1), j = s % (NP + 1);
        if (s + 1 < NS) load_a(s + 1, Ah[(s + 1) & 1], Al[(s + 1) & 1]);
        if (j == 0 && dx + 1 < KS) load_b(dx + 1, Bp[(dx + 1) & 1], Bl[(dx + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const half8 b = (j < NP) ? Bp[dx & 1][r + 2 * j] : Bl[dx & 1][r];
            if constexpr (HI) acc[r] = x3_mfma(Al[s & 1], b, acc[r]);
            acc[r] = x3_mfma(Ah[s & 1], b, acc[r]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Dependency gate of the one-launch dense block (conv_x3_rdbs_kernel below): the input chunks >= `chunk` of this tile (and of its
// halo) are produced by OTHER workgroups of the same launch; `flags[tile]` == `epoch` says a tile's share of them is visible.
// Chunks < `chunk` were checked by an earlier phase, so the K loop runs them ungated — only the last chunks wait.
struct X3Gate {
    const unsigned* flags;    // per-tile "previous phase visible" words (null: nothing to wait for)
    unsigned epoch;
    int chunk;                // first chunk that needs the neighbours' previous phase
    unsigned* status;         // device status word for the bounded-spin bit
};
constexpr unsigned RDB3_SPIN_LIMIT = 1u << 22;

// One output tile (TH x 32 pixels, 32 output channels of column z) from prologue DMA to epilogue stores.  Every wave of
// the workgroup calls it with the same arguments; LDS must be free of readers on entry.
template <int KS, int R, int WN, int EPI, bool XTRA, bool GATED = false>
__device__ __forceinline__ void x3_tile(const ConvKArgs& a, const float* __restrict__ bias, char* smem, int img, int ty, int tx,
                                        int z, const X3Gate gate = X3Gate{nullptr, 0u, 1 << 30, nullptr}) {
    using C = X3Cfg<KS, R, WN>;
    BH_TL_DECL;
    BH_TL_BEGIN();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31;     // pixel column (B/N index) and cout row (A/M index) of this lane
    const int kg = lane >> 5;    // which 8-channel half of the 16-channel chunk
    const int tx0 = tx * 32, ty0 = ty * C::TH;
    const int H = a.H, W = a.W;
    const long long plane_elems = (long long)a.N * H * W * 16;
    const unsigned plane_bytes = (unsigned)(plane_elems * 2);

    // per-lane source offsets of the patch DMA pieces (stage independent); LDS image = [channel half][patch pixel][16 B]
    unsigned voff[C::NPJ];
#pragma unroll
    for (int j = 0; j < C::NPJ; ++j) {
        const int i = wave + C::NW * j;
        const int q = i * 64 + lane;
        const int cg = q >= C::PH * C::PW ? 1 : 0;
        const int p = q - cg * (C::PH * C::PW);
        const int py = p / C::PW;
        const int px = p - py * C::PW;
        const int gy = ty0 + py - C::PAD;
        const int gx = tx0 + px - C::PAD;
        const bool ok = (p < C::PH * C::PW) && (gy >= 0) && (gy < H) && (gx >= 0) && (gx < W);
        voff[j] = ok ? (unsigned)((((long long)img * H + gy) * W + gx) * 32 + cg * 16) : 0x80000000u;
    }

    floatx16 acc[1][R];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][r][e] = 0.f;

    const int nchunks = a.nchunks;
    const int a_lane_off = n * 32 + ((kg ^ ((n >> 3) & 1)) << 4);
    const int b_lane_off = (kg * (C::PH * C::PW) + wave * R * C::PW + n) * 16;

#if BINHIP_EPI_PRIO
    __builtin_amdgcn_s_setprio(2);       // K loop above the CU partner's epilogue (profiles/r06_experiments.md)
#endif
    x3_issue_weights<C, KS>(a, smem, 0, 0, wave, lane, z);
    if constexpr (GATED) {
        if (gate.chunk <= 0 && gate.flags) {              // every input chunk is gated: wait before the first patch DMA
            if (tid < 9) {
                const int ny = ty + tid / 3 - 1, nx = tx + tid % 3 - 1;
                if (ny >= 0 && ny < a.tiles_y && nx >= 0 && nx < a.tiles_x) {
                    const unsigned* f = gate.flags + ((size_t)img * a.tiles_y + ny) * a.tiles_x + nx;
                    unsigned spins = 0;
                    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gate.epoch) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > RDB3_SPIN_LIMIT) {
                            if (gate.status) atomicOr(gate.status, BINHIP_STATUS_SYNC_TIMEOUT);
                            break;
                        }
                    }
                }
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    x3_issue_patch<C>(a, smem, 0, 0, 0, wave, voff, plane_elems, plane_bytes);
    for (int c = 0; c < nchunks; ++c) {
        const char* wb = smem + 2 * C::PATCH_BYTES + (c & 1) * C::WBUF_BYTES;
        // Progress-ordered priority (round 3 side build, round 6 product for ONE-ROUND grids): of the two workgroups sharing a CU the
        // one that lags issues first, so the pair finishes together instead of the younger one running alone at the end of the launch
        // (profiles/r06_wg_timeline.md).  Only when every workgroup of the launch is resident at once (`a.prog_prio`, set by the
        // launcher): in a multi-round grid a freshly started workgroup would outrank the ones about to free their slot — the
        // training step loses 0.6 % with it, the 720p window gains 0.7 %.
        if ((BINHIP_X3_PRIO & 1) || a.prog_prio) {
            const int q = (4 * c) / nchunks;
            if (q == 0) __builtin_amdgcn_s_setprio(3);
            else if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        // ---- hi sub-stage (the long one: 2 products): meanwhile the lo patch plane and the NEXT chunk's weights land
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#if BINHIP_TIMELINE
        if (c == 0) BH_TL_STAMP(1);
#endif
        constexpr bool SPREAD = BINHIP_X3_SPREAD && KS == 3;
        if constexpr (!SPREAD) {
            x3_issue_patch<C>(a, smem, c, 1, 1, wave, voff, plane_elems, plane_bytes);
            if (c + 1 < nchunks) x3_issue_weights<C, KS>(a, smem, c + 1, (c + 1) & 1, wave, lane, z);
        }
        // GATED: one sub-stage before the first DMA of a gated chunk (the hi plane of chunk c + 1, issued in the lo sub-stage
        // below), lanes 0-8 of wave 0 fetch the flags of the 3 x 3 tile neighbourhood; the loads ride under this sub-stage's MFMAs
        unsigned gate_seen = gate.epoch;
        const unsigned* gate_word = nullptr;
        if constexpr (GATED) {
            if (c + 1 == gate.chunk && gate.flags && tid < 9) {
                const int ny = ty + tid / 3 - 1, nx = tx + tid % 3 - 1;
                if (ny >= 0 && ny < a.tiles_y && nx >= 0 && nx < a.tiles_x) {
                    gate_word = gate.flags + ((size_t)img * a.tiles_y + ny) * a.tiles_x + nx;
                    gate_seen = __hip_atomic_load(gate_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // (5x5 only: the last chunk of a 24- / 36-channel layer on tap pairs, see x3_compute_pair)
        const bool pair = (KS == 5) && !XTRA && a.half_last && (c + 1 == nchunks);
        if constexpr (KS == 5 && !XTRA) {
            if (pair) x3_compute_pair<C, KS, R, true>(smem, wb, n, kg, wave, acc[0]);
            else x3_compute<C, KS, R, true>(smem, wb, a_lane_off, b_lane_off, acc[0]);
        } else if constexpr (SPREAD) {
            // steps 0 .. NPJ-1: the lo patch plane, then NWJ steps: the next chunk's weights (9 taps: 6 of them carry a piece)
            const bool more = c + 1 < nchunks;
            x3_compute<C, KS, R, true>(smem, wb, a_lane_off, b_lane_off, acc[0], [&](int s) {
                if (s < C::NPJ) x3_issue_patch_piece<C>(a, smem, c, 1, 1, wave, s, voff, plane_elems, plane_bytes);
                else if (s < C::NPJ + C::NWJ && more) x3_issue_weights_piece<C, KS>(a, smem, c + 1, (c + 1) & 1, wave, lane, z, s - C::NPJ);
            });
        } else {
            x3_compute<C, KS, R, true>(smem, wb, a_lane_off, b_lane_off, acc[0]);
        }
        // ---- lo sub-stage: meanwhile the next chunk's hi patch plane lands
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (GATED) {
            if (c + 1 == gate.chunk && gate.flags) {      // (workgroup-uniform)
                if (gate_word) {
                    unsigned spins = 0;
                    while (gate_seen != gate.epoch) {
                        __builtin_amdgcn_s_sleep(2);
                        gate_seen = __hip_atomic_load(gate_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (++spins > RDB3_SPIN_LIMIT) {
                            if (gate.status) atomicOr(gate.status, BINHIP_STATUS_SYNC_TIMEOUT);
                            break;
                        }
                    }
                }
                __builtin_amdgcn_s_barrier();             // every wave: the neighbourhood's previous phase is visible
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (!SPREAD) {
            if (c + 1 < nchunks) x3_issue_patch<C>(a, smem, c + 1, 0, 0, wave, voff, plane_elems, plane_bytes);
        }
        if constexpr (KS == 5 && !XTRA) {
            if (pair) x3_compute_pair<C, KS, R, false>(smem + C::PATCH_BYTES, wb, n, kg, wave, acc[0]);
            else x3_compute<C, KS, R, false>(smem + C::PATCH_BYTES, wb, a_lane_off, b_lane_off, acc[0]);
        } else if constexpr (SPREAD) {
            const bool more = c + 1 < nchunks;
            x3_compute<C, KS, R, false>(smem + C::PATCH_BYTES, wb, a_lane_off, b_lane_off, acc[0], [&](int s) {
                if (s < C::NPJ && more) x3_issue_patch_piece<C>(a, smem, c + 1, 0, 0, wave, s, voff, plane_elems, plane_bytes);
            });
        } else {
            x3_compute<C, KS, R, false>(smem + C::PATCH_BYTES, wb, a_lane_off, b_lane_off, acc[0]);
        }
    }
    BH_TL_STAMP(2);
#if BINHIP_EPI_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    conv_epilogue<1, R, 3, EPI, XTRA>(a, bias, acc, img, ty0 + wave * R, tx0, z * 32, z == 0, n, kg, plane_elems);
    BH_TL_FINISH(a, (((img * a.tiles_y + ty) * a.tiles_x + tx) * a.ncol + z));
}

// WIDE only names the instantiation: 0 = one 32-row output block (the dense-block convs, UPNet.2), 1 = several workgroup
// columns — same code, separate symbols, so per-kernel profiles keep the dominant dense-block conv apart from the wide layers
// XTRA: the epilogue also reads residual / accumulator / ReLU-mask planes (the backward-data convs); `bias`: the layer's bias
// as the kernel's own restrict parameter, for scalar loads (binhip_conv_common.h)
// (waves per SIMD: two workgroups per CU = WN / 2 -> at most 128 VGPRs; the 5x5 forward kernel runs ONE workgroup per CU (146 KB of
//  LDS), and with the tap-pair path of its last chunk compiled in it needs more than 128: its bound is WN / 4 = 256 VGPRs)
template <int KS, int R, int WN, int EPI, int WIDE, bool XTRA>
__global__ void __launch_bounds__(64 * WN)
    __attribute__((amdgpu_waves_per_eu((KS == 5 && !XTRA) ? WN / 4 : WN / 2, (KS == 5 && !XTRA) ? WN / 4 : WN / 2)))
conv_x3_kernel(const ConvKArgs a, const float* __restrict__ bias) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // 1-D grid of tiles x output columns, column fastest: after the XCD banding the `ncol` workgroups that share one input
    // patch are neighbours on ONE XCD, so the patch is fetched into that L2 once (a (tiles, ncol) grid ran each column as a
    // separate sweep: UPNet.0's 8 columns fetched 884 MB for 99 MB of input)
    int bid = blockIdx.x;
    if (a.xcd_remap) bid = xcd_band(bid, gridDim.x);
    const int ncol = a.ncol;
    const int z = bid % ncol;
    bid /= ncol;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    x3_tile<KS, R, WN, EPI, XTRA>(a, bias, smem, img, ty, tx, z);
}

// ---------------------------------------------------------------------------------------------------------------------
// The three Cout = 32 convolutions of a residual dense block (RDN.py:135-147: convs 0, 1, 2 of `RDB.convs`) as phases of
// ONE launch with STATIC tile ownership (round 6; rounds 2-5 handed (phase, tile) items out of an atomic work queue and lost
// 2-3 % to it).  What the per-workgroup timeline of the per-conv launches showed (profiles/r06_wg_timeline.md): the two
// workgroups of a CU do NOT share the matrix pipe evenly — the older one runs as if alone (30 us), the younger one gets the
// gaps and then runs 20 us by itself with its prologue, epilogue and barrier waits exposed; every launch then waits for the
// slowest CU (span 57 us for a mean pair time of 50) and the next launch starts 3.7 us later.  Here a workgroup keeps its
// tile(s) for all three convs:
//   * conv p+1 of a tile needs conv p's output of the tile and of its 8 neighbours (1-pixel halo) — and ONLY for its last
//     two input chunks (planes 6+2p, 7+2p of the block buffer); the chunks before them were complete before this phase
//     began.  The K loop runs in natural order, so the dependency comes last by itself: a workgroup that finishes conv p
//     early streams 6+2p of the 8+2p chunks of conv p+1 while its neighbours finish, and only then looks at their flags
//     (X3Gate) — the chip never drains between the convs, and the CU's other workgroup fills every gap.
//   * grid = min(tiles, 2 x CUs) workgroups, all co-resident (the launcher checks the occupancy: 76 KB of LDS and <= 128
//     VGPRs give two per CU); workgroup b owns tiles b, b + grid, ... in every phase, so nothing is handed out at run
//     time.  No deadlock: an item only waits for items of the PREVIOUS phase, and every workgroup finishes its phase-p
//     items before it starts a phase-(p+1) item.
//   * Visibility across CUs / XCDs (MI355X: per-XCD L2s are not coherent, a CU's L1 is never refreshed): the producer's
//     output stores are the write-through (sc1) 16-byte plane stores the kernel uses anyway; every wave drains them
//     (s_waitcnt vmcnt(0)), the workgroup barriers, ONE lane publishes the flag with an agent-scope (sc1) store.  The
//     consumer polls with agent-scope loads, barriers, and only then issues its LDS-DMA of the gated planes: they were
//     never read in this launch before their flags were set, so no L1 / L2 holds a stale line of them (kernel boundaries
//     invalidate both).
//   * A bounded spin (RDB3_SPIN_LIMIT polls with s_sleep) turns a protocol error into a status bit instead of a hang.
// Same arithmetic in the same order as the per-conv launches: bit-identical outputs (tests/test_gpu_conv.py).
struct Rdb3Args {
    ConvKArgs conv[3];
    unsigned* flags;          // [2][T] "phase p of tile t is visible" == epoch
    unsigned epoch;           // value that means "done" in this launch (flags are reused by later launches)
    int T;                    // tiles per phase = tiles_x * tiles_y * N
};

template <int R, int WN>
__global__ void __launch_bounds__(64 * WN) __attribute__((amdgpu_waves_per_eu(WN / 2, WN / 2)))
conv_x3_rdbs_kernel(const Rdb3Args a, const float* __restrict__ bias0, const float* __restrict__ bias1,
                    const float* __restrict__ bias2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_x = a.conv[0].tiles_x, tiles_y = a.conv[0].tiles_y;
    const int G = (int)gridDim.x;
    const int b = xcd_band((int)blockIdx.x, G);           // consecutive owners = neighbouring tiles on one XCD
#pragma unroll 1
    for (int p = 0; p < 3; ++p) {
        const ConvKArgs& ka = a.conv[p];
        const float* bias = p == 0 ? bias0 : (p == 1 ? bias1 : bias2);
        X3Gate gate;
        gate.flags = p > 0 ? a.flags + (size_t)(p - 1) * a.T : nullptr;
        gate.epoch = a.epoch;
        gate.chunk = ka.nchunks - 2;
        gate.status = ka.flags;
#pragma unroll 1
        for (int t0 = b; t0 < a.T; t0 += G) {
            int t = t0;
            const int tx = t % tiles_x;
            t /= tiles_x;
            const int ty = t % tiles_y;
            const int img = t / tiles_y;
            x3_tile<3, R, WN, BINHIP_EPI_PLANES, false, true>(ka, bias, smem, img, ty, tx, 0, gate);
            // publish: this wave's write-through stores have left (vmcnt(0)), then all waves', then the flag; the barrier
            // also frees LDS for the next tile's prologue
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0 && p < 2)
                __hip_atomic_store(a.flags + (size_t)p * a.T + t0, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int KS, int R, int WN, int EPI, int WIDE, bool XTRA>
static int launch_x3_x(const ConvKArgs& ka, int cout_pad, hipStream_t s) {
    using C = X3Cfg<KS, R, WN>;
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = bh_set_max_lds(&conv_x3_kernel<KS, R, WN, EPI, WIDE, XTRA>, C::LDS_BYTES, lds_set)) return rc;
    ConvKArgs a = ka;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + C::TH - 1) / C::TH;
    a.ncol = cout_pad / 32;
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.N * a.ncol));
    // progress-ordered wave priority: only for launches whose workgroups are all resident at once (two per CU), forward layers only
    a.prog_prio = (BINHIP_PROG_PRIO && KS == 3 && !XTRA && (long long)grid.x <= 2ll * binhip_device_cus()) ? 1 : 0;
#if BINHIP_TIMELINE
    // kind: 1 = the dense-block conv (3x3, one 32-row column, plane epilogue, no extras), 3 = every other instantiation
    a.tl = bh_tl_reserve(grid.x, &a.tl_base);
    a.tl_launch = ((KS == 3 && EPI == BINHIP_EPI_PLANES && WIDE == 0 && !XTRA ? 1u : 3u) << 24) | (g_bh_tl_serial.fetch_add(1) & 0xFFFFFFu);
#endif
    conv_x3_kernel<KS, R, WN, EPI, WIDE, XTRA><<<grid, dim3(64 * C::NW), C::LDS_BYTES, s>>>(a, a.bias);
    BH_CHECK_LAUNCH();
    return 0;
}
template <int KS, int R, int WN, int EPI, int WIDE>
static int launch_x3(const ConvKArgs& ka, int cout_pad, hipStream_t s) {
    if constexpr (EPI == BINHIP_EPI_PLANES) {
        if (ka.has_res || ka.r2_hi || ka.m_hi || ka.y_unshuf) return launch_x3_x<KS, R, WN, EPI, WIDE, true>(ka, cout_pad, s);
    }
    return launch_x3_x<KS, R, WN, EPI, WIDE, false>(ka, cout_pad, s);
}

// the one-launch dense block: `convs` are the fully prepared kernel arguments of convs 0, 1, 2 (same N, H, W; write-through
// stores required), `flags` = 2 * T device words whose values differ from `epoch` (the caller zeroes them once per RDN call and
// passes epoch = block index + 1).  Returns BINHIP_E_SHAPE when the kernel cannot keep two workgroups per CU resident.
int bh_launch_rdb3_x3(const ConvKArgs* convs, unsigned* flags, unsigned epoch, int cus, hipStream_t s) {
    using C = X3Cfg<3, BINHIP_X3_R, BINHIP_X3_WN>;
    static std::atomic<unsigned long long> lds_set{0};
    static std::atomic<int> per_cu{-1};
    if (int rc = bh_set_max_lds(&conv_x3_rdbs_kernel<BINHIP_X3_R, BINHIP_X3_WN>, C::LDS_BYTES, lds_set)) return rc;
    int occ = per_cu.load(std::memory_order_acquire);
    if (occ < 0) {             // co-residency of the whole grid is what makes the flag waits safe: ask the runtime once
        int n = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_x3_rdbs_kernel<BINHIP_X3_R, BINHIP_X3_WN>, 64 * C::NW,
                                                                    C::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        occ = n > 2 ? 2 : n;
        per_cu.store(occ, std::memory_order_release);
    }
    if (occ < 1) return BINHIP_E_SHAPE;
    Rdb3Args a;
    for (int p = 0; p < 3; ++p) {
        a.conv[p] = convs[p];
        a.conv[p].tiles_x = (convs[p].W + 31) / 32;
        a.conv[p].tiles_y = (convs[p].H + C::TH - 1) / C::TH;
        if (!a.conv[p].wt || a.conv[p].nchunks < 3) return BINHIP_E_SHAPE;
#if BINHIP_TIMELINE
        // kinds 4, 5, 6 = phases 0, 1, 2 of the one-launch dense block
        a.conv[p].tl = bh_tl_reserve((unsigned)(a.conv[p].tiles_x * a.conv[p].tiles_y * a.conv[p].N), &a.conv[p].tl_base);
        a.conv[p].tl_launch = ((4u + p) << 24) | (g_bh_tl_serial.load() & 0xFFFFFFu);
#endif
    }
#if BINHIP_TIMELINE
    g_bh_tl_serial.fetch_add(1);
#endif
    a.T = a.conv[0].tiles_x * a.conv[0].tiles_y * a.conv[0].N;
    a.flags = flags; a.epoch = epoch;
    const long long slots = (long long)occ * (cus > 0 ? cus : 256);
    const unsigned grid = (unsigned)(a.T < slots ? a.T : slots);
    for (int p = 0; p < 3; ++p) a.conv[p].prog_prio = (BINHIP_PROG_PRIO && a.T <= slots) ? 1 : 0;     // one tile per workgroup and phase
    conv_x3_rdbs_kernel<BINHIP_X3_R, BINHIP_X3_WN><<<dim3(grid), dim3(64 * C::NW), C::LDS_BYTES, s>>>(a, a.conv[0].bias, a.conv[1].bias,
                                                                                                   a.conv[2].bias);
    BH_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// UPNet.2 (reference RDN.py:207: conv3x3 64 -> 3 at FULL resolution, + mean of the input frames :221/279/333) without the
// matrix cores.  With three output channels a 32-row MFMA tile is 91 % padding (round 2: 110 us per launch at 720p, 2.6 % of
// the window, 0.04 of the MFMA peak); here every lane owns ONE output pixel and accumulates its three channels with
// v_dot2_f32_f16 (two fp16 products + fp32 add per lane and instruction): operands stay fp16 — the patch planes as they sit in
// LDS, the weights as wave-uniform scalar loads straight from the relayouted rows 0..2 (no swizzle below row 8) — and the
// hi / lo split keeps its three products (x_hi w_hi + x_hi w_lo + x_lo w_hi), each exact in fp32.
//   * tile 8 x 32 pixels, 4 waves (lane = pixel column, wave / lane half = row), patch planes DMA'd per 16-channel chunk and
//     per precision plane into a two-slot ring exactly like x3_tile (20 KB of LDS: many workgroups per CU);
//   * per chunk and lane: 72 ds_read_b128 (9 taps x 2 channel halves x 2 planes x ... ) and 648 dot2 (f16x3) / 216 (f16).
template <int NT>
__global__ void __launch_bounds__(256)
final_dot2_kernel(const ConvKArgs a, const unsigned* __restrict__ w_hi32, const unsigned* __restrict__ w_lo32) {
    // (the weight planes come as separate `const __restrict__` kernel parameters: only then does the compiler know that the
    //  kernel's own stores cannot clobber them and turns the wave-uniform loads into s_load_dwordx8)
    using C = X3Cfg<3, 2, 4>;                     // TH = 8, patch 10 x 34, 11 DMA pieces per plane, 4 waves
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
        const int cg = q >= C::PH * C::PW ? 1 : 0;
        const int p = q - cg * (C::PH * C::PW);
        const int py = p / C::PW;
        const int px = p - py * C::PW;
        const int gy = ty0 + py - 1;
        const int gx = tx0 + px - 1;
        const bool ok = (p < C::PH * C::PW) && (gy >= 0) && (gy < H) && (gx >= 0) && (gx < W);
        voff[j] = ok ? (unsigned)((((long long)img * H + gy) * W + gx) * 32 + cg * 16) : 0x80000000u;
    }
    const int px = lane & 31, py = 2 * wave + (lane >> 5);          // this lane's pixel inside the tile
    const int base_off = (py * C::PW + px) * 16;                      // patch pixel (py, px) = tap (0, 0) of the output pixel
    float acc[3] = {0.f, 0.f, 0.f};
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    union X8 { half8 h; half2v p[4]; };
    union W2 { unsigned u; half2v p; };

    const int nchunks = a.nchunks;
    auto compute = [&](const char* pb, int c, bool hi_plane) {
        // weights of chunk c: rows 0..2 of every tap are 3 x 32 B contiguous (row stride 32 B, no slot swizzle below row 8)
        const unsigned* wh = w_hi32 + (long long)c * (9 * 32 * 8);
        const unsigned* wl = w_lo32 + (long long)c * (9 * 32 * 8);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3, dx = t % 3;
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                X8 x;
                x.h = *reinterpret_cast<const half8*>(pb + cg * (C::PH * C::PW * 16) + base_off + (dy * C::PW + dx) * 16);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        W2 w;
                        w.u = wh[(t * 32 + r) * 8 + cg * 4 + k];
                        acc[r] = __builtin_amdgcn_fdot2(x.p[k], w.p, acc[r], false);
                        if (NT == 3 && hi_plane) {
                            W2 v;
                            v.u = wl[(t * 32 + r) * 8 + cg * 4 + k];
                            acc[r] = __builtin_amdgcn_fdot2(x.p[k], v.p, acc[r], false);
                        }
                    }
                }
            }
        }
    };

    x3_issue_patch<C>(a, smem, 0, 0, 0, wave, voff, plane_elems, plane_bytes);
    for (int c = 0; c < nchunks; ++c) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NT == 3) {
            x3_issue_patch<C>(a, smem, c, 1, 1, wave, voff, plane_elems, plane_bytes);
            compute(smem, c, true);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < nchunks) x3_issue_patch<C>(a, smem, c + 1, 0, 0, wave, voff, plane_elems, plane_bytes);
            compute(smem + C::PATCH_BYTES, c, false);
        } else {
            if (c + 1 < nchunks) x3_issue_patch<C>(a, smem, c + 1, 0, (c + 1) & 1, wave, voff, plane_elems, plane_bytes);
            compute(smem + (c & 1) * C::PATCH_BYTES, c, true);
        }
    }
    const int gy = ty0 + py, gx = tx0 + px;
    if (gy < H && gx < W) {
        // every load of the epilogue — the input frames AND the three biases (a.bias sits inside the by-value argument struct: per-lane
        // vector loads) — before the first store, pinned: a load behind a store can only be used once the store has drained (shared
        // vmcnt); the round-3 form fetched bias[j] between the stores: three serial round trips per pixel row (round 4)
        float sum[3], bj[3];
        const int nimg = a.nimg;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const long long idx = (((long long)img * a.cout + (j < a.cout ? j : 0)) * H + gy) * W + gx;
            float v[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) v[t] = (t < nimg) ? a.img[t][idx] : 0.f;
            bj[j] = a.bias[j < a.cout ? j : 0];
            float sj = v[0];
#pragma unroll
            for (int t = 1; t < 5; ++t)
                if (t < nimg) sj += v[t];
            sum[j] = nimg > 0 ? sj / (float)nimg : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(sum[j]), "+v"(bj[j]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j >= a.cout) break;
            const long long idx = (((long long)img * a.cout + j) * H + gy) * W + gx;
            a.out_f32[idx] = (acc[j] + bj[j]) + sum[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// UPNet.2 in the fp32-class mode on a 16-ROW matrix tile with a THREE-stage patch ring (round 4).
// Three output channels on the 32x32x16 tile of conv_x3_kernel are 91 % padding — but the layer was never bound by its matrix
// work: 110 us per launch at 720p = 2.5 TB/s of a 277 MB layer.  A sub-stage of x3_tile has ONE LDS-DMA batch in flight per workgroup
// (prefetch distance: one sub-stage); with a sub-stage's matrix work at ~0.3-0.6 us the loop runs at the DMA round trip, ~3.5 us per
// sub-stage.  (First form of this kernel, same ring as x3_tile, 1.8x fewer matrix cycles: 120 us.  profiles/r04_experiments.md)
// What changes here:
//   * the weights of the WHOLE layer stay in LDS for the tile's life: only rows 0-3 of every tap's 32-row slab matter (rows 0-2 are
//     the layer, row 3 is zero padding), 128 B per tap and plane = 10 KB instead of a 2 x 18 KB per-chunk ring;
//   * that frees a third 20 KB patch slot: prefetch distance TWO sub-stages with counted vmcnt (a wave issues 2 or 3 pieces per
//     stage: wave-uniform choice of the immediate), still 70 KB = two workgroups per CU;
//   * v_mfma_f32_16x16x32_f16: M = 16 rows, N = 16 pixels, K = 32 spent on TAP PAIRS: k = 8 kb + e (kb = lane >> 4) = channel
//     8 (kb & 1) + e of the chunk at tap 2p + (kb >> 1) of pair p — 5 instructions per 16 pixels, chunk and product.  An A fragment
//     is row min(lane & 15, 3) of its tap (D rows 3-15 are never read, so what multiplies into them is irrelevant; the odd tap of
//     pair 4 does not exist: its lanes read row 3 = zeros).  B fragment of lane (n = lane & 15, kb): channel half kb & 1 of the
//     patch pixel under the lane's tap, one ds_read_b128 with a per-lane address.  C/D: col = lane & 15 = pixel, row = 4 (lane >> 4)
//     + reg: the three channels of 16 pixels sit in lanes 0-15, registers 0-2.
struct M16 {
    using C = X3Cfg<3, 2, 8>;
    static constexpr int NSTAGE = 3;
    static constexpr int WPIECES = 5;                              // 36 taps x 128 B per plane, in 1-KiB DMA pieces
    static constexpr int W_OFF = NSTAGE * C::PATCH_BYTES;
    static constexpr int LDS_BYTES = W_OFF + 2 * WPIECES * 1024;    // 70 KB
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

template <bool HI>
__device__ __forceinline__ void m16_compute(const char* pb, const char* wb, int a_lane_off, int b_lane_off, int slot, int a_zero,
                                            floatx4 (&acc)[2][2]) {
    using C = X3Cfg<3, 2, 8>;
    half8 Ah[2], Al[2], B[2][2][2];
    auto tap_off = [](int t) { return ((t / 3) * C::PW + (t % 3)) * 16; };
    auto load = [&](int p, half8& ah, half8& al, half8 (&b)[2][2]) {
        // tap slab = 128 B (rows 0-3 x 32 B); pair 4's second tap: row 3 of tap 8 (zeros) for every lane
        const int aoff = (p < 4) ? (2 * p + slot) * 128 + a_lane_off : (slot ? 8 * 128 + a_zero : 8 * 128 + a_lane_off);
        ah = x3_ld8(wb + aoff);
        if constexpr (HI) al = x3_ld8(wb + M16::WPIECES * 1024 + aoff);
        const int boff = b_lane_off + tap_off(2 * p) + slot * (p < 4 ? tap_off(2 * p + 1) - tap_off(2 * p) : 0);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 2; ++g) b[r][g] = x3_ld8(pb + boff + (r * C::PW + g * 16) * 16);
    };
    load(0, Ah[0], Al[0], B[0]);
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        if (p + 1 < 5) load(p + 1, Ah[(p + 1) & 1], Al[(p + 1) & 1], B[(p + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if constexpr (HI) acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[p & 1], B[p & 1][r][g], acc[r][g], 0, 0, 0);
                acc[r][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[p & 1], B[p & 1][r][g], acc[r][g], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
final_m16_kernel(const ConvKArgs a, const float* __restrict__ bias) {
    using C = X3Cfg<3, 2, 8>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
        const int cg = q >= C::PH * C::PW ? 1 : 0;
        const int p = q - cg * (C::PH * C::PW);
        const int py = p / C::PW;
        const int px = p - py * C::PW;
        const int gy = ty0 + py - 1;
        const int gx = tx0 + px - 1;
        const bool ok = (p < C::PH * C::PW) && (gy >= 0) && (gy < H) && (gx >= 0) && (gx < W);
        voff[j] = ok ? (unsigned)((((long long)img * H + gy) * W + gx) * 32 + cg * 16) : 0x80000000u;
    }
    const int n16 = lane & 15, kb = lane >> 4;
    const int slot = kb >> 1;
    const int a_lane_off = ((n16 < 3 ? n16 : 3) * 2 + (kb & 1)) * 16;  // row min(n16, 3) of a tap's 4-row slab, channel half kb & 1
    const int a_zero = (3 * 2 + (kb & 1)) * 16;                        // row 3: zero padding
    const int b_lane_off = ((kb & 1) * (C::PH * C::PW) + wave * 2 * C::PW + n16) * 16;
    floatx4 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[r][g] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.nchunks;
    const int nstage = 2 * nchunks;                       // stage s = (chunk s / 2, plane s & 1), ring slot s % 3
    // ---- the layer's weights, rows 0-3 of every (chunk, tap) slab: lane l of piece i fetches 16 B no. l % 8 of tap 8 i + l / 8
    {
        const int ntaps = nchunks * 9;
        for (int i = wave; i < 2 * M16::WPIECES; i += C::NW) {
            const int pl = i >= M16::WPIECES ? 1 : 0;
            const int t = (i - pl * M16::WPIECES) * 8 + (lane >> 3);
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(pl ? a.w_lo : a.w_hi), 0, (unsigned)(ntaps * 1024), 0x00020000);
            const unsigned off = t < ntaps ? (unsigned)(t * 1024 + (lane & 7) * 16) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(smem + M16::W_OFF + i * 1024), 16, off, 0, 0, 0);
        }
    }
    x3_issue_patch<C>(a, smem, 0, 0, 0, wave, voff, plane_elems, plane_bytes);
    if (nstage > 1) x3_issue_patch<C>(a, smem, 0, 1, 1, wave, voff, plane_elems, plane_bytes);
    const bool three = wave < (C::PP - 2 * C::NW);        // this wave issues 3 patch pieces per stage (else 2)
    for (int s = 0; s < nstage; ++s) {
        // stage s landed: at most the pieces of stage s + 1 (issued one sub-stage ago) may still be in flight
        if (s + 1 < nstage) {
            if (three) wait_vmcnt<3>(); else wait_vmcnt<2>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();                      // every wave's pieces of stage s landed; stage s - 1's slot is free
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < nstage) x3_issue_patch<C>(a, smem, (s + 2) >> 1, (s + 2) & 1, (s + 2) % 3, wave, voff, plane_elems, plane_bytes);
        const char* pb = smem + (s % 3) * C::PATCH_BYTES;
        const char* wb = smem + M16::W_OFF + (s >> 1) * (9 * 128);
        if (s & 1) m16_compute<false>(pb, wb, a_lane_off, b_lane_off, slot, a_zero, acc);
        else m16_compute<true>(pb, wb, a_lane_off, b_lane_off, slot, a_zero, acc);
    }
    // ---- epilogue (BINHIP_EPI_FINAL, the arithmetic of conv_epilogue): + bias + mean of the input frames -> fp32 NCHW.
    // EVERY frame load of the wave first, the twelve means pinned in registers, then the stores: loads and stores share vmcnt on
    // gfx9, so a value loaded before a store can only be used after that store has drained.  Left alone, hipcc sinks each 16-pixel
    // group's adds behind the previous group's stores: four serial round trips per tile, and the kernel's time followed the number of
    // input frames (94 / 122 / 157 us for 2 / 3 / 5; now 84 / 89 / 109).  Fetching the frames BEFORE the K loop instead was measured
    // and is slower (93.7 vs 88.9 us: the values live across the loop's control flow and the compiler drains the whole prefetch
    // ring once per tile to be sure of them).
    if (kb == 0) {
        float sum[2][2][3];
        const int nimg = a.nimg;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int gy = ty0 + wave * 2 + r, gx = tx0 + g * 16 + n16;
                const bool ok = gy < H && gx < W;
                const int gyc = gy < H ? gy : H - 1, gxc = gx < W ? gx : W - 1;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const long long idx = (((long long)img * a.cout + (j < a.cout ? j : 0)) * H + gyc) * W + gxc;
                    float v[5];
#pragma unroll
                    for (int t = 0; t < 5; ++t) v[t] = (t < nimg) ? a.img[t][idx] : 0.f;
                    float sj = v[0];
#pragma unroll
                    for (int t = 1; t < 5; ++t)
                        if (t < nimg) sj += v[t];
                    sum[r][g][j] = (ok && nimg > 0) ? sj / (float)nimg : 0.f;
                }
            }
        // (the three biases too: `bias` is a scalar-load parameter, but a load of it BETWEEN the stores would still sit behind
        //  their drain — advisor r04; final_dot2_kernel hoists them the same way)
        float bj[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) bj[j] = bias[j < a.cout ? j : 0];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(sum[r][g][j]));
#pragma unroll
        for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(bj[j]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int gy = ty0 + wave * 2 + r, gx = tx0 + g * 16 + n16;
                if (gy >= H || gx >= W) continue;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (j >= a.cout) break;
                    const long long idx = (((long long)img * a.cout + j) * H + gy) * W + gx;
                    a.out_f32[idx] = (acc[r][g][j] + bj[j]) + sum[r][g][j];
                }
            }
    }
}

int bh_launch_final_m16(const ConvKArgs& ka, hipStream_t s) {
    using C = X3Cfg<3, 2, 8>;
    static std::atomic<unsigned long long> lds_set{0};
    if (ka.nchunks * 9 > M16::WPIECES * 8) return BINHIP_E_SHAPE;   // 5 weight pieces per plane = 40 taps = 4 chunks (UPNet.2: 64 inputs)
    if (int rc = bh_set_max_lds(&final_m16_kernel, M16::LDS_BYTES, lds_set)) return rc;
    ConvKArgs a = ka;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + C::TH - 1) / C::TH;
    a.ncol = 1;
    final_m16_kernel<<<dim3((unsigned)(a.tiles_x * a.tiles_y * a.N)), dim3(512), M16::LDS_BYTES, s>>>(a, a.bias);
    BH_CHECK_LAUNCH();
    return 0;
}

// FINAL epilogue with <= 3 output channels (UPNet.2), both precisions
int bh_launch_final_dot2(const ConvKArgs& ka, int nterms, hipStream_t s) {
    using C = X3Cfg<3, 2, 4>;
    ConvKArgs a = ka;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + C::TH - 1) / C::TH;
    a.ncol = 1;
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.N));
    const unsigned* wh = reinterpret_cast<const unsigned*>(a.w_hi);
    const unsigned* wl = reinterpret_cast<const unsigned*>(nterms == 3 ? a.w_lo : a.w_hi);
    if (nterms == 3) final_dot2_kernel<3><<<grid, dim3(256), 2 * C::PATCH_BYTES, s>>>(a, wh, wl);
    else final_dot2_kernel<1><<<grid, dim3(256), 2 * C::PATCH_BYTES, s>>>(a, wh, wl);
    BH_CHECK_LAUNCH();
    return 0;
}

// entry for the dispatcher in binhip_conv.hip: every 3x3 convolution of the nterms = 3 path, in 32-row output blocks
// (wider layers — 96 -> 96, UPNet.0's 96 -> 256, the 96-row backward-data convs — run as cout_pad / 32 workgroup
// columns over the same tiles: the input patch is re-read per column, from L2)
// SFENet1 (5x5, 12 n_inputs -> 96 as three 32-row columns) and its backward-data on the plane-split pipeline
int bh_launch_conv_x3_k5(const ConvKArgs& a, int cout_pad, hipStream_t s) {
    return launch_x3<5, BINHIP_X3_R, BINHIP_X3_WN, BINHIP_EPI_PLANES, 1>(a, cout_pad, s);
}

// the fused UPNet (BINHIP_PLAN_FUSED_UPNET): 5x5, G0 -> 12 sub-pixel channels at half resolution, fp32 NCHW full-resolution output
int bh_launch_conv_x3_k5_subpix(const ConvKArgs& a, hipStream_t s) {
    return launch_x3<5, BINHIP_X3_R, BINHIP_X3_WN, BINHIP_EPI_FINAL_SUBPIX, 0>(a, 32, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// The border ring of the fused UPNet.  UPNet.2 (RDN.py:207) zero-pads the 64-channel full-resolution tensor it reads, i.e. where
// its 3x3 window leaves the image a tap contributes nothing — the one-convolution form (W_eff over a zero-padded INPUT) instead
// sees "UPNet.0 of the padded input" there.  The two differ on the outermost full-resolution pixel ring only; for those
// 4 H + 4 W - 4 pixels (H, W: half resolution) the host supplies the exact operators (fp32 [9][12][25][cin], variant = 3 vy + vx,
// bin_amd/rdn_plan.py fused_upnet_weights) and this kernel recomputes them (fp32 FMA on hi + lo; ~30 MFLOP per call).
struct RingArgs {
    const _Float16* x_hi;
    const _Float16* x_lo;
    const float* wvar;
    const float* bvar;
    float* out;
    const float* img[5];
    int N, H, W, cin, nimg;
};
constexpr int RING_THREADS = 128;
// One workgroup per ring pixel.  Work item = (tap, group of 8 input channels): two 16-byte plane loads (hi, lo) and six float4
// weight loads feed 24 FMAs — 25 * cin / 8 items per pixel (300 at 96 channels) over 128 threads.  (Measured on the way, 720p call:
// one 2-byte load per product 26-30 us; eight pixels per workgroup sharing their weights 50 us — it is the number of load
// instructions, not the 29 KB of L2-resident weights per pixel, that this small kernel pays for.)
__global__ void __launch_bounds__(RING_THREADS) upnet_ring_kernel(const RingArgs a) {
    __shared__ float red[3][RING_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, W = a.W, H2 = 2 * H, W2 = 2 * W;
    const int ring = 2 * W2 + 2 * (H2 - 2);
    int id = blockIdx.x;
    const int img = id / ring;
    id -= img * ring;
    int Y, X;
    if (id < W2) { Y = 0; X = id; }
    else if (id < 2 * W2) { Y = H2 - 1; X = id - W2; }
    else { const int k = id - 2 * W2; Y = 1 + (k >> 1); X = (k & 1) ? W2 - 1 : 0; }
    const int y = Y >> 1, i = Y & 1, x = X >> 1, j = X & 1;
    const int vy = (Y == 0) ? 0 : (Y == H2 - 1 ? 2 : 1), vx = (X == 0) ? 0 : (X == W2 - 1 ? 2 : 1);
    const int var = 3 * vy + vx, sub = 2 * i + j;
    const int cin = a.cin, ng = cin >> 3, items = 25 * ng;
    const long long wstride = (long long)25 * cin;                       // one output channel
    const float* w = a.wvar + ((long long)var * 12 + sub) * wstride;       // colour c at + 4 c * wstride
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int it = tid; it < items; it += RING_THREADS) {
        const int tap = it / ng, ci = (it - tap * ng) << 3;
        const int yy = y + tap / 5 - 2, xx = x + tap % 5 - 2;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const long long o = (((((long long)(ci >> 4) * a.N + img) * H + yy) * W + xx) << 4) + (ci & 15);
        const half8 xh = *reinterpret_cast<const half8*>(a.x_hi + o);
        half8 xl;
        if (a.x_lo) xl = *reinterpret_cast<const half8*>(a.x_lo + o);
        const float* wq = w + (long long)tap * cin + ci;
        floatx4 wa[3][2];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            wa[c][0] = *reinterpret_cast<const floatx4*>(wq + 4 * c * wstride);
            wa[c][1] = *reinterpret_cast<const floatx4*>(wq + 4 * c * wstride + 4);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = (float)xh[e];
            if (a.x_lo) v += (float)xl[e];
            s0 = fmaf(wa[0][e >> 2][e & 3], v, s0);
            s1 = fmaf(wa[1][e >> 2][e & 3], v, s1);
            s2 = fmaf(wa[2][e >> 2][e & 3], v, s2);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off);
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
    }
    if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; red[2][wave] = s2; }
    __syncthreads();
    if (tid < 3) {
        const int c = tid;
        const float v = red[c][0] + red[c][1];
        const long long idx = (((long long)img * 3 + c) * H2 + Y) * W2 + X;
        float m = 0.f;
        if (a.nimg > 0) {
            m = a.img[0][idx];
            for (int t = 1; t < a.nimg; ++t) m += a.img[t][idx];
            m = m / (float)a.nimg;
        }
        a.out[idx] = (v + a.bvar[var * 12 + 4 * c + sub]) + m;
    }
}
int bh_launch_upnet_ring(const void* x_hi, const void* x_lo, const float* wvar, const float* bvar, float* out, const float* const* images,
                         int nimg, int N, int H, int W, int cin, hipStream_t s) {
    if (!x_hi || !wvar || !bvar || !out || N <= 0 || H <= 0 || W <= 0 || cin <= 0 || (cin & 15) || nimg < 0 || nimg > 5) return BINHIP_E_ARG;
    RingArgs a;
    a.x_hi = (const _Float16*)x_hi; a.x_lo = (const _Float16*)x_lo; a.wvar = wvar; a.bvar = bvar; a.out = out;
    for (int t = 0; t < 5; ++t) a.img[t] = (t < nimg) ? images[t] : nullptr;
    a.N = N; a.H = H; a.W = W; a.cin = cin; a.nimg = nimg;
    const long long ring = 4ll * W + 4ll * H - 4;
    upnet_ring_kernel<<<dim3((unsigned)(ring * N)), dim3(RING_THREADS), 0, s>>>(a);
    BH_CHECK_LAUNCH();
    return 0;
}

int bh_launch_conv_x3(const ConvKArgs& a, int cout_pad, int epilogue, hipStream_t s) {
    if (epilogue == BINHIP_EPI_PLANES)
        return cout_pad == 32 ? launch_x3<3, BINHIP_X3_R, BINHIP_X3_WN, BINHIP_EPI_PLANES, 0>(a, cout_pad, s)
                              : launch_x3<3, BINHIP_X3_R, BINHIP_X3_WN, BINHIP_EPI_PLANES, 1>(a, cout_pad, s);
    if (epilogue == BINHIP_EPI_SHUFFLE) return launch_x3<3, BINHIP_X3_R, BINHIP_X3_WN, BINHIP_EPI_SHUFFLE, 1>(a, cout_pad, s);
    if (epilogue == BINHIP_EPI_FINAL) return launch_x3<3, BINHIP_X3_R, BINHIP_X3_WN, BINHIP_EPI_FINAL, 0>(a, cout_pad, s);
    return BINHIP_E_SHAPE;
}
