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
#include <utility>
// 3x3 weight gradient, tile walk of a workgroup: 1 (default) = row-major tiles at stride PB; 0 = a contiguous range, DOWN a
// 32-pixel column first, so that two of a tile's ten patch rows are still in L2 from the tile before: round 3 measured 583 ->
// 494 MB of HBM-side traffic per launch and NO gain in time (140.9 vs 142.8 us, training step 132.5 vs 132.9 ms, same box) —
// like the XCD mapping of round 2, the bytes are not what this kernel waits for.
#ifndef BINHIP_WG3_STRIDE_WALK
#define BINHIP_WG3_STRIDE_WALK 1
#endif
// 3x3 weight gradient, BINHIP_TUNING side builds only: 1 = wave owns an X row (wgrad3x3_xrow_kernel, the product's only form),
// 0 = wave owns a gY row (wgrad3x3_db_kernel, round 2; tools/experiments/wgrad_experiments.inc)
#ifndef BINHIP_WG3_XROW
#define BINHIP_WG3_XROW 1
#endif
// ... and its prefetch DMA: 1 = spread over the multiply steps of a tile, 0 = one burst at the top of the tile (side builds)
#ifndef BINHIP_WG3_SPREAD_DMA
#define BINHIP_WG3_SPREAD_DMA 1
#endif
// ... the multiply steps (of 18) behind which the four (plane, chunk) groups go out: FIRST + k * STRIDE.  A group is needed at the
// next tile's step 0, so a later step leaves its round trip less cover (steps 1 / 5 / 9 / 13: 17 / 13 / 9 / 5 steps)
#ifndef BINHIP_WG3_DMA_FIRST
#define BINHIP_WG3_DMA_FIRST 1
#endif
#ifndef BINHIP_WG3_DMA_STRIDE
#define BINHIP_WG3_DMA_STRIDE 4
#endif

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
    int ppg;      // 1x1 kernel: input-channel pairs per workgroup column (blockIdx.y)
    int cgroups;  // rolling-row 3x3 kernel: channel-pair groups (blockIdx.y = cot * cgroups + group)
    int nz;       // generic / lean kernels: ncot * ndyg (1-D grid of PB * ncp * nz workgroups, see wg_block())
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

// The same fragment with the reads issued from inline asm.  The compiler knows nothing about the alias classes of the
// transpose-read builtin, so after an LDS-DMA (buffer_load ... lds) it protects every tr_frag() with s_waitcnt vmcnt(0) —
// which serialises "prefetch the next stage" and "compute this one".  Asm reads are invisible to that pass; the price is
// that the lgkmcnt wait is ours: tr_issue() ... tr_wait(...) on the SAME registers before their first use.
typedef __attribute__((address_space(3))) const char lds_cchar_t;
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(size_t)(lds_cchar_t*)p; }
// per-lane part of the address, valid for pixel offsets that are multiples of 16 (then (p >> 3) & 1 == lane >> 5)
__device__ __forceinline__ unsigned tr_lane_off(int chunk_bytes, int lane) {
    const int t = lane & 15, ch = (lane >> 4) & 1, kg = lane >> 5;
    return (unsigned)(ch * chunk_bytes + (kg * 8 + (t >> 2)) * 32 + ((((t & 3) >> 1) ^ kg) << 4) + ((t & 1) << 3));
}
struct TrFrag { short4_ a, b; };     // pixels +0..3 and +4..7 of the lane's channel
__device__ __forceinline__ void tr_issue(TrFrag& f, unsigned addr) {      // addr = lds_addr(chunk strip) + lane part + p0 * 32
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.a) : "v"(addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:128" : "=v"(f.b) : "v"(addr));
}
// (ordinary ds_read_b64 in place of the transpose reads, timing-only build: 209.8 vs 204.9 us — the transpose unit is free)
#define WG_TR_OP "ds_read_b64_tr_b16"
template <int OFF>
__device__ __forceinline__ void tr_issue_pair(TrFrag& f, unsigned oa, unsigned ob) {      // two pre-computed addresses + immediate
    asm volatile(WG_TR_OP " %0, %1 offset:%2" : "=v"(f.a) : "v"(oa), "n"(OFF));
    asm volatile(WG_TR_OP " %0, %1 offset:%2" : "=v"(f.b) : "v"(ob), "n"(OFF));
}
template <class F, int... S>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, S...>) {
    (f(std::integral_constant<int, S>{}), ...);
}
// general pixel offset (tap-shifted patches): both 4-pixel reads get their own swizzled address
__device__ __forceinline__ void tr_issue_at(TrFrag& f, const char* img, int chunk_bytes, int p0, int lane) {
    const int t = lane & 15, ch = (lane >> 4) & 1, kg = lane >> 5;
    const unsigned base = lds_addr(img + ch * chunk_bytes) + ((t & 1) << 3);
    const int pa = p0 + kg * 8 + (t >> 2), pb = pa + 4;
    const unsigned oa = base + pa * 32 + ((((t & 3) >> 1) ^ ((pa >> 3) & 1)) << 4);
    const unsigned ob = base + pb * 32 + ((((t & 3) >> 1) ^ ((pb >> 3) & 1)) << 4);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.a) : "v"(oa));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.b) : "v"(ob));
}
// after an `s_waitcnt lgkmcnt(0)` asm: re-defines the fragment's registers so that no use can be scheduled above the wait
__device__ __forceinline__ void tr_tie(TrFrag& f) { asm volatile("" : "+v"(f.a), "+v"(f.b)); }
__device__ __forceinline__ half8 tr_value(const TrFrag& f) {
    union { struct { short4_ a, b; } s; half8 h; } u;
    u.s.a = f.a; u.s.b = f.b;
    return u.h;
}

template <int KS, int TR, int NT, int NW = 4>
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
            for (int j = 0; j < (C::XP + NW - 1) / NW; ++j) {
                const int i = wave + NW * j;
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
            for (int j = 0; j < (C::GP + NW - 1) / NW; ++j) {
                const int i = wave + NW * j;
                if (C::GP % NW != 0 && i >= C::GP) break;
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

// Workgroup -> (pixel-block pb, channel pair cp, z = output tile / tap-row group) for the generic and lean kernels.  The
// ncp * nz workgroups of one pb walk the SAME pixel tiles (each re-reading the gY tile its siblings read, and for nz > 1 the X
// patch): the hardware deals consecutive workgroup ids round-robin to the 8 XCDs, so the plain (pb, cp, z) grid put the
// siblings on different XCDs = different L2s and every one of them fetched its gY from HBM (PMC: 811 MB per launch for
// 392 MB of operands in the dense-block layers).  With PB a multiple of 8 the id is unpacked as
//     xcd = id % 8, slot = id / 8, pb = (slot / G) * 8 + xcd, (cp, z) = slot % G        (G = ncp * nz)
// which keeps all siblings of a pb on ONE XCD, dispatched back to back.
__device__ __forceinline__ void wg_block(const WgradKArgs& a, int& pb, int& cp, int& z) {
    const int id = blockIdx.x, G = a.ncp * a.nz;
    int r;
    if ((a.PB & 7) == 0) {
        const int slot = id >> 3;
        pb = (slot / G) * 8 + (id & 7);
        r = slot % G;
    } else {
        pb = id / G;
        r = id % G;
    }
    cp = r % a.ncp;
    z = r / a.ncp;
}

template <int KS, int TR, int NT>
__global__ void __launch_bounds__(256)
wgrad_mfma_kernel(const WgradKArgs a) {
    using C = WgCfg<KS, TR, NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pb, cp, bz;
    wg_block(a, pb, cp, bz);
    const int cot = bz % a.ncot;
    const int dyg = bz / a.ncot;
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
        // The reads are issued from asm (tr_issue_at): the builtin would be fenced with vmcnt(0) against the prefetch above.
        TrFrag Bh[2], Bl[2], Ah[2][C::NTAP], Al[2][C::NTAP];
        auto load = [&](int s4, int q) {
            const int row = wave * 2 + (s4 >> 1), x0 = (s4 & 1) * 16;
            tr_issue_at(Bh[q], gb, C::GBYTES, row * 32 + x0, lane);
            if constexpr (NT == 3) tr_issue_at(Bl[q], gb + C::PLANE_BYTES, C::GBYTES, row * 32 + x0, lane);
#pragma unroll
            for (int t = 0; t < C::NTAP; ++t) {
                const int p0 = (row + t / KS) * C::PW + x0 + t % KS;
                tr_issue_at(Ah[q][t], xb, C::XBYTES, p0, lane);
                if constexpr (NT == 3) tr_issue_at(Al[q][t], xb + C::PLANE_BYTES, C::XBYTES, p0, lane);
            }
        };
        load(0, 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int q = s4 & 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tr_tie(Bh[q]);
            if constexpr (NT == 3) tr_tie(Bl[q]);
#pragma unroll
            for (int t = 0; t < C::NTAP; ++t) {
                tr_tie(Ah[q][t]);
                if constexpr (NT == 3) tr_tie(Al[q][t]);
            }
            if (s4 + 1 < 4) load(s4 + 1, q ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            const half8 bh = tr_value(Bh[q]);
            half8 bl;
            if constexpr (NT == 3) bl = tr_value(Bl[q]);
            if (do_bias) {
                accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, bh, accb, 0, 0, 0);
                if constexpr (NT == 3) accb = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, bl, accb, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < C::NTAP; ++t) {
                const half8 ah = tr_value(Ah[q][t]);
                if constexpr (NT == 3) {
                    const half8 al = tr_value(Al[q][t]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
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
    const long long blk = ((long long)bz * a.ncp + cp) * a.PB + pb;
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


constexpr int R3_SEG = 16;     // rolling-row kernel (side builds): pixel rows per column segment
#if BINHIP_TUNING
// round 1-3 kernels that the product does not dispatch (lean single-stage, wave = gY row, rolling rows): side builds only
#include "../../tools/experiments/wgrad_experiments.inc"
#endif

// 3x3 layers, X-ROW form of the eight-wave kernel (round 3; the product's 3x3 weight gradient).  Its round-2 predecessor
// (wgrad3x3_db_kernel, now in tools/experiments/wgrad_experiments.inc: eight waves, two LDS stages fed by LDS-DMA, transpose
// reads issued from asm, chunk pairs 128 B apart against bank conflicts) gave wave w gY row w of the tile; it read, per K-step,
// one gY fragment pair and NINE tap-shifted X fragments (rows w .. w + 2 of a 10-row halo patch): 20 fragments for 27 MFMAs —
// fragment reads + DMA writes keep the LDS ~87 % busy, and the DMA prefetch measurably does not overlap with the multiply
// (tools/bench_wgrad.py, 160 -> 32: MFMAs alone 104 us, + fragment reads 132, DMA alone 107, all together 210 = the SUM).
// Here the tile is 8 X rows and the halo moves to the operand that needs no column shifts:
//     dW[dy][dx] = sum_r sum_x X[r][x + dx - 1] * gY[r - dy + 1][x]
// wave w owns X row r0 + w: three column-shifted X fragments per K-step, each used against the THREE gY rows r - dy + 1
// (a 10-row gY patch, rows r0 - 1 .. r0 + 8, zero outside the image): 6 X + 6 gY = 12 fragments for the same 27 MFMAs
// (-40 % LDS reads), identical for every wave, and the X halo rows are no longer fetched twice (X 8 x 34, gY 10 x 32 pixels per
// tile and pair: the same 75 KB).  Every (X row, dy) product is counted in exactly one tile because the X rows are
// partitioned; rows outside the image arrive as zeros from the DMA range check.  Step s = (ks, dy, dx): the reads of step
// s + 2 are issued while step s multiplies, counted lgkmcnt as above; X fragments live in one buffer per dx (A[1][dx] is
// fetched two steps after the last use of A[0][dx]), gY fragments alternate between two.
template <int NT>
struct Wg3xCfg {
    static constexpr int NPL = (NT == 3) ? 2 : 1;
    static constexpr int XR = 8, GR = 10, PW = 34;
    static constexpr int XP = (XR * PW * 2 + 63) / 64;                   // 1-KiB DMA pieces of an X chunk patch (9)
    static constexpr int GP = GR;                                        // one piece = one 32-pixel gY row
    static constexpr int XS = XP * 1024 + 128, GS = GP * 1024 + 128;     // chunk strides (bank offset as in Wg3Cfg)
    static constexpr int G0 = 2 * XS;
    static constexpr int PLANE = 2 * XS + 2 * GS;
    static constexpr int STAGE = NPL * PLANE;
    static constexpr int LDS_BYTES = 2 * STAGE;
    static constexpr int NXJ = (XP + 7) / 8, NGJ = (GP + 7) / 8;
    static_assert(PLANE + 512 + 2 * 1024 + 128 < 65536, "lo plane / K-step / gY row reachable with the 16-bit DS offset");
    static_assert(LDS_BYTES <= 160 * 1024 && LDS_BYTES >= 8 * 4096, "LDS budget; the final reduction needs 32 KB");
};

// DMA of one tile, in two halves so that the caller can spread the instructions over the multiply steps of the tile before
// (see the kernel): wg3x_offsets() = the per-lane buffer offsets of the tile (VALU only), wg3x_issue_part(PL, H) = the
// buffer_load ... lds instructions of plane PL, chunk H of the pair (X pieces, then gY pieces).
template <int NT>
struct Wg3xTile {
    unsigned xvo[Wg3xCfg<NT>::NXJ], gvo[Wg3xCfg<NT>::NGJ];
};
template <int NT>
__device__ __forceinline__ void wg3x_offsets(const WgradKArgs& a, int tile, int wave, const int* x_py, const int* x_px,
                                             const int* x_src, int g_px, const int* g_src, Wg3xTile<NT>& o) {
    using G = Wg3xCfg<NT>;
    const int H = a.H, W = a.W;
    int b = tile;
#if BINHIP_WG3_STRIDE_WALK
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const int img = b / a.tiles_y;
#else
    const int ty = b % a.tiles_y; b /= a.tiles_y;
    const int tx = b % a.tiles_x;
    const int img = b / a.tiles_x;
#endif
    const int tx0 = tx * 32, ty0 = ty * G::XR;
    const int x0 = tx0 - 1;
    const long long row0 = (long long)img * H;
    const int xbase = (int)(((row0 + ty0) * W + x0) * 32);          // may be negative at the image border (then !ok)
    const int gbase = (int)(((row0 + ty0 - 1) * W + tx0) * 32);
#pragma unroll
    for (int j = 0; j < G::NXJ; ++j) {
        const bool ok = (unsigned)(ty0 + x_py[j]) < (unsigned)H && (unsigned)(x0 + x_px[j]) < (unsigned)W;
        o.xvo[j] = ok ? (unsigned)(xbase + x_src[j]) : 0x80000000u;
    }
#pragma unroll
    for (int j = 0; j < G::NGJ; ++j) {
        const bool ok = (wave + 8 * j < G::GP) && (unsigned)(ty0 - 1 + wave + 8 * j) < (unsigned)H && (tx0 + g_px < W);
        o.gvo[j] = ok ? (unsigned)(gbase + g_src[j]) : 0x80000000u;
    }
}
template <int NT>
__device__ __forceinline__ void wg3x_issue_part(const WgradKArgs& a, char* stage, const int PL, const int HH, int cp, int cot,
                                                int wave, const Wg3xTile<NT>& o, long long plane_elems, unsigned plane_bytes) {
    using G = Wg3xCfg<NT>;
    const int c = 2 * cp + HH;
    const _Float16* xb = PL ? a.x_lo : a.x_hi;
    const long long coff = (a.x_cpg > 0)
        ? (long long)(c / a.x_cpg) * a.x_group_stride + (long long)(c % a.x_cpg) * plane_elems
        : (long long)c * plane_elems;
    const bool have = c < a.cin_chunks;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(have ? xb + coff : xb), 0,
                                                                  have ? plane_bytes : 0u, 0x00020000);
    char* lds = stage + PL * G::PLANE + HH * G::XS;
#pragma unroll
    for (int j = 0; j < G::NXJ; ++j)
        if (wave + 8 * j < G::XP)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(lds + (wave + 8 * j) * 1024), 16, o.xvo[j], 0, 0, 0);
    const int gc = 2 * cot + HH;
    const bool haveg = gc < a.cout_chunks;
    const _Float16* gb = PL ? a.g_lo : a.g_hi;
    __amdgpu_buffer_rsrc_t gs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(haveg ? gb + (long long)gc * plane_elems : gb), 0, haveg ? plane_bytes : 0u, 0x00020000);
    char* ldg = stage + PL * G::PLANE + G::G0 + HH * G::GS;
#pragma unroll
    for (int j = 0; j < G::NGJ; ++j)
        if (wave + 8 * j < G::GP)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(gs, (lds_void_t*)(ldg + (wave + 8 * j) * 1024), 16, o.gvo[j], 0, 0, 0);
}
template <int NT>
__device__ __forceinline__ void wg3x_issue_all(const WgradKArgs& a, char* stage, int cp, int cot, int wave,
                                               const Wg3xTile<NT>& o, long long plane_elems, unsigned plane_bytes) {
    wg3x_issue_part<NT>(a, stage, 0, 0, cp, cot, wave, o, plane_elems, plane_bytes);
    wg3x_issue_part<NT>(a, stage, 0, 1, cp, cot, wave, o, plane_elems, plane_bytes);
    if constexpr (NT == 3) {
        wg3x_issue_part<NT>(a, stage, 1, 0, cp, cot, wave, o, plane_elems, plane_bytes);
        wg3x_issue_part<NT>(a, stage, 1, 1, cp, cot, wave, o, plane_elems, plane_bytes);
    }
}

template <int NT>
__global__ void __launch_bounds__(512)
wgrad3x3_xrow_kernel(const WgradKArgs a) {
    using G = Wg3xCfg<NT>;
    constexpr int NTAP = 9, NSTEP = 2 * NTAP;
    constexpr int NA = (NT == 3) ? 4 : 2;                                // read instructions of one fragment (pair x planes)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = X row of the 8-row tile
    int pb, cp, bz;
    wg_block(a, pb, cp, bz);
    const int cot = bz % a.ncot;
    const int W = a.W;
    const long long plane_elems = (long long)a.N * a.H * W * 16;
    const unsigned plane_bytes = (unsigned)(plane_elems * 2);
    const bool do_bias = (cp == 0);

    floatx16 acc[NTAP];
    float bsum = 0.f;
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // ---- tile-invariant per-lane state: the two 4-pixel reads of the X fragment of column shift dx (K-step 0, hi plane)
    unsigned xa[3], xb2[3];
    {
        const int tt = lane & 15, ch = (lane >> 4) & 1, kg = lane >> 5;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int p0 = wave * G::PW + dx;
            const int pa = p0 + kg * 8 + (tt >> 2), pb4 = pa + 4;
            const unsigned base = (unsigned)(ch * G::XS + ((tt & 1) << 3));
            xa[dx] = base + pa * 32 + ((((tt & 3) >> 1) ^ ((pa >> 3) & 1)) << 4);
            xb2[dx] = base + pb4 * 32 + ((((tt & 3) >> 1) ^ ((pb4 >> 3) & 1)) << 4);
        }
    }
    // gY patch row `wave` (= image row r - 2); tap dy multiplies patch row wave + 2 - dy (an immediate)
    const unsigned g_off = (unsigned)G::G0 + tr_lane_off(G::GS, lane) + wave * 32 * 32;
    int x_py[G::NXJ], x_px[G::NXJ], x_src[G::NXJ], g_src[G::NGJ];
#pragma unroll
    for (int j = 0; j < G::NXJ; ++j) {
        const int q = (wave + 8 * j) * 64 + lane;
        const int p = q >> 1, sh = q & 1;
        const bool in_patch = (wave + 8 * j < G::XP) && (p < G::XR * G::PW);
        x_py[j] = in_patch ? p / G::PW : -(1 << 20);                       // out-of-patch lanes fail the range check
        x_px[j] = p % G::PW;
        x_src[j] = (x_py[j] * W + x_px[j]) * 32 + ((sh ^ ((p >> 3) & 1)) << 4);
    }
    const int g_px = lane >> 1;
#pragma unroll
    for (int j = 0; j < G::NGJ; ++j)
        g_src[j] = ((wave + 8 * j) * W + g_px) * 32 + (((lane & 1) ^ ((g_px >> 3) & 1)) << 4);

#if BINHIP_WG3_STRIDE_WALK
    int tile = pb;
    const int tend = a.ntiles, tstep = a.PB;
#else
    int tile = (int)(((long long)pb * a.ntiles) / a.PB);
    const int tend = (int)(((long long)(pb + 1) * a.ntiles) / a.PB), tstep = 1;
#endif
    Wg3xTile<NT> to;
    if (tile < tend && !(a.dbg & 1)) {
        wg3x_offsets<NT>(a, tile, wave, x_py, x_px, x_src, g_px, g_src, to);
        wg3x_issue_all<NT>(a, smem, cp, cot, wave, to, plane_elems, plane_bytes);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (; tile < tend; tile += tstep) {
        const int nxt = tile + tstep;
        // The next tile's DMA is NOT issued in one burst here: a wave executes in order, and a burst of buffer_load ... lds
        // sits at the head of its instruction stream until the memory pipeline has accepted all of it — with one workgroup
        // per CU nobody multiplies meanwhile, and prefetch and multiply ran one after the other (tools/bench_wgrad.py: DMA
        // alone 110 us + MFMAs and reads alone 119 us = 214 us measured, 160 -> 32 channels).  One (plane, chunk) group of DMA
        // instructions goes out behind the MFMAs of steps 1, 5, 9 and 13 instead.
        const bool pre = nxt < tend && !(a.dbg & 1);
        if (pre) wg3x_offsets<NT>(a, nxt, wave, x_py, x_px, x_src, g_px, g_src, to);
        char* const stage_nxt = smem + (cur ^ 1) * G::STAGE;
#if BINHIP_WG3_SPREAD_DMA == 0
        if (pre) wg3x_issue_all<NT>(a, stage_nxt, cp, cot, wave, to, plane_elems, plane_bytes);
#endif
        if (!(a.dbg & 2)) {
            const unsigned st = lds_addr(smem + cur * G::STAGE);
            TrFrag Bh[2], Bl[2], Ah[3], Al[3];
            auto load = [&](auto SC) {
                constexpr int s = decltype(SC)::value, ks = s / NTAP, dy = (s % NTAP) / 3, dx = s % 3;
#if BINHIP_TUNING
                if (a.dbg & 8) return;                 // ablation: MFMAs on stale registers, no LDS fragment reads
#endif
                if constexpr (dx == 0) {
                    constexpr int bb = (ks * 3 + dy) & 1, off = ks * 512 + (2 - dy) * 1024;
                    tr_issue_pair<off>(Bh[bb], st + g_off, st + g_off + 128);
                    if constexpr (NT == 3) tr_issue_pair<off + G::PLANE>(Bl[bb], st + g_off, st + g_off + 128);
                }
                if constexpr (dy == 0) {
                    tr_issue_pair<ks * 512>(Ah[dx], st + xa[dx], st + xb2[dx]);
                    if constexpr (NT == 3) tr_issue_pair<ks * 512 + G::PLANE>(Al[dx], st + xa[dx], st + xb2[dx]);
                }
            };
            load(std::integral_constant<int, 0>{});
            load(std::integral_constant<int, 1>{});
            static_for([&](auto SC) {
                constexpr int s = decltype(SC)::value, ks = s / NTAP, dy = (s % NTAP) / 3, dx = s % 3, t = dy * 3 + dx;
                constexpr int bb = (ks * 3 + dy) & 1;
                // outstanding: the reads of steps s and s + 1, in issue order -> leave step s + 1's in flight
                constexpr int s1 = s + 1;
                constexpr int later = (s1 < NSTEP) ? NA * ((s1 % 3 == 0 ? 1 : 0) + ((s1 % NTAP) / 3 == 0 ? 1 : 0)) : 0;
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(later) : "memory");
                if constexpr (dy == 0) {
                    tr_tie(Ah[dx]);
                    if constexpr (NT == 3) tr_tie(Al[dx]);
                }
                if constexpr (dx == 0) {
                    tr_tie(Bh[bb]);
                    if constexpr (NT == 3) tr_tie(Bl[bb]);
                }
                if constexpr (s + 2 < NSTEP) load(std::integral_constant<int, s + 2>{});
                __builtin_amdgcn_sched_barrier(0);
                const half8 bh = tr_value(Bh[bb]);
                half8 bl;
                if constexpr (NT == 3) bl = tr_value(Bl[bb]);
                if (dy == 1 && dx == 0 && do_bias) {           // patch row wave + 1 = the wave's own image row: once per gY row
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        bsum += (float)bh[e];
                        if constexpr (NT == 3) bsum += (float)bl[e];
                    }
                }
                const half8 ah = tr_value(Ah[dx]);
                if constexpr (NT == 3) {
                    const half8 al = tr_value(Al[dx]);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
#if BINHIP_WG3_SPREAD_DMA
                if constexpr (s >= BINHIP_WG3_DMA_FIRST && (s - BINHIP_WG3_DMA_FIRST) % BINHIP_WG3_DMA_STRIDE == 0 &&
                              (s - BINHIP_WG3_DMA_FIRST) / BINHIP_WG3_DMA_STRIDE < (NT == 3 ? 4 : 2)) {
                    constexpr int grp = (s - BINHIP_WG3_DMA_FIRST) / BINHIP_WG3_DMA_STRIDE;           // (plane, chunk) = (0,0) (0,1) (1,0) (1,1)
                    if (pre) wg3x_issue_part<NT>(a, stage_nxt, grp / 2, grp % 2, cp, cot, wave, to, plane_elems, plane_bytes);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }, std::make_integer_sequence<int, NSTEP>{});
        } else if (BINHIP_WG3_SPREAD_DMA && pre) {
            wg3x_issue_all<NT>(a, stage_nxt, cp, cot, wave, to, plane_elems, plane_bytes);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    if (a.dbg & 4) { if (acc[0][0] == 12345.f) a.partial[0] = bsum; return; }
    // ---- the eight rows are summed through LDS in a fixed order: one partial per workgroup and tap
    float* red = reinterpret_cast<float*>(smem);          // [8 waves][32 m][32 n]
    const int n = lane & 31, hi = lane >> 5;
    const long long blk = ((long long)bz * a.ncp + cp) * a.PB + pb;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * hi) * 32 + n] = acc[t][e];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 512 * i;
            a.partial[(blk * NTAP + t) * 1024 + idx] =
                ((red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx])) +
                ((red[4096 + idx] + red[5120 + idx]) + (red[6144 + idx] + red[7168 + idx]));
        }
    }
    if (do_bias) {
        __syncthreads();
        red[tid] = bsum;                              // [wave][kg][co]
        __syncthreads();
        if (tid < 32) {
            float tsum = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) tsum += red[k * 32 + tid];
            a.partial_b[((long long)cot * a.PB + pb) * 32 + tid] = tsum;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// 1x1 convolutions (LFF 224->96, GFF.0 1152->96).  A 1x1 weight gradient does 2*Cin*Cout flops per pixel for
// (Cin + Cout) * 2 B of operands per plane: 77 flop/B in f16x3 — four times below the machine balance, so this is a
// STREAMING kernel and its design goal is bytes: every X and gY plane byte crosses HBM once per channel-pair group and
// enough of them are in flight per CU to cover the HBM latency.
//   * One workgroup (8 waves, one per CU: the LDS is all staging buffer) walks strips of TR x 32 pixels; a stage holds the
//     strip of ALL its operands — 6 gY chunks and `ppg` input-channel pairs, hi and lo planes — and is double buffered
//     (LFF, f16x3: 2 x 80 KB, so ~80 KB per CU are always in flight).
//   * Wave w owns input-channel pair(s) w*PPW .. of the group and all three 32-wide output tiles: its 3*PPW accumulator tiles
//     see every pixel of the strip, so there is no K-split across waves, no end-of-kernel cross-wave reduction, and the
//     accumulators are 48*PPW registers instead of the 192 a "one wave = one pixel row of all tiles" split needs (the round-1
//     kernel: 1 wave per SIMD, a barrier per 16 KB, gY read once per group of FOUR pairs — 1.09 GB for 0.84 GB of
//     operands on LFF at 40 x 128 x 128 — and 2.8 TB/s).
//   * ncp > 8 (GFF.0: 36 pairs): PPW = 2 and ceil(ncp / 16) balanced groups (3 x 12), gY re-read once per group.
// The last wave of group 0 also accumulates the bias gradient from the gY fragments it loads anyway.
constexpr int W1_NW = 8, W1_NCOT = 3;

struct W1Plan { int ppw, tr, cgroups, ppg; unsigned lds; };
static inline W1Plan w1_plan(int ncp) {
    W1Plan p;
    p.ppw = ncp <= W1_NW ? 1 : 2;
    const int cap = W1_NW * p.ppw;
    p.cgroups = (ncp + cap - 1) / cap;
    p.ppg = (ncp + p.cgroups - 1) / p.cgroups;
    const int slots = 2 * (2 * W1_NCOT + 2 * p.ppg);     // sized for two planes (f16x3); f16 uses half of it
    p.tr = (p.ppw == 1 && 2 * slots * 2 * 1024 <= 160 * 1024) ? 2 : 1;
    p.lds = 2u * slots * p.tr * 1024;
    return p;
}

template <int NT, int PPW, int TR>
__global__ void __launch_bounds__(64 * W1_NW)
wgrad1x1_kernel(const WgradKArgs a) {
    constexpr int NPL = (NT == 3) ? 2 : 1;
    constexpr int CH = TR * 1024;                         // one chunk strip: TR rows x 32 px x 32 B
    constexpr int GSLOTS = NPL * 2 * W1_NCOT;
    constexpr int PAIR_BYTES = NPL * 2 * CH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pb = blockIdx.x;
    const int cpg = blockIdx.y;
    const int ppg = a.ppg;
    const int cp0 = cpg * ppg;
    const long long plane_elems = (long long)a.N * a.H * a.W * 16;
    const unsigned plane_bytes = (unsigned)(plane_elems * 2);
    const int nslots = GSLOTS + ppg * NPL * 2;
    const int stage_bytes = nslots * CH;
    const bool bias_wave = (cpg == 0) && (wave == W1_NW - 1);

    bool valid[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) valid[i] = (wave * PPW + i < ppg) && (cp0 + wave * PPW + i < a.ncp);

    floatx16 acc[PPW][W1_NCOT];
    float bsum[W1_NCOT] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < PPW; ++i)
#pragma unroll
        for (int j = 0; j < W1_NCOT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // per-lane part of a DMA piece (one 32-pixel row of one chunk = 1 KiB): pixel lane/2, 16-byte half swizzled by pixel/8
    const int lp = lane >> 1;
    const unsigned lane_off = (unsigned)(lp * 32 + (((lane & 1) ^ ((lp >> 3) & 1)) << 4));
    const unsigned tr_off = tr_lane_off(CH, lane);

    auto issue = [&](int tile, int buf) {
        int b = tile;
        const int tx = b % a.tiles_x; b /= a.tiles_x;
        const int ty = b % a.tiles_y;
        const int img = b / a.tiles_y;
        const int tx0 = tx * 32, ty0 = ty * TR;
        const bool col_ok = tx0 + lp < a.W;
        char* stage = smem + buf * stage_bytes;
        for (int k = wave; k < nslots * TR; k += W1_NW) {
            const int slot = k / TR, r = k % TR;
            const _Float16* src;
            bool have;
            if (slot < GSLOTS) {
                const int pl = slot / (2 * W1_NCOT), c6 = slot % (2 * W1_NCOT);
                have = c6 < a.cout_chunks;
                src = (pl ? a.g_lo : a.g_hi) + (long long)c6 * plane_elems;
            } else {
                const int xs = slot - GSLOTS;
                const int pr = xs / (NPL * 2), rem = xs % (NPL * 2);
                const int pl = rem >> 1, c = 2 * (cp0 + pr) + (rem & 1);
                have = c < a.cin_chunks;
                const long long coff = (a.x_cpg > 0)
                    ? (long long)(c / a.x_cpg) * a.x_group_stride + (long long)(c % a.x_cpg) * plane_elems
                    : (long long)c * plane_elems;
                src = (pl ? a.x_lo : a.x_hi) + (have ? coff : 0);
            }
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, have ? plane_bytes : 0u, 0x00020000);
            const int gy = ty0 + r;
            const bool ok = col_ok && gy < a.H;
            const unsigned vo = ok ? (unsigned)((((long long)img * a.H + gy) * a.W + tx0) * 32) + lane_off : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(stage + slot * CH + r * 1024), 16, vo, 0, 0, 0);
        }
    };

    int tile = pb;
    int buf = 0;
    if (tile < a.ntiles && !(a.dbg & 1)) issue(tile, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (; tile < a.ntiles; tile += a.PB) {
        if (tile + a.PB < a.ntiles && !(a.dbg & 1)) issue(tile + a.PB, buf ^ 1);
        if (!(a.dbg & 2)) {
            const unsigned gst = lds_addr(smem + buf * stage_bytes) + tr_off;
            const unsigned xst = gst + GSLOTS * CH + wave * PPW * PAIR_BYTES;
            // fragments of pixel step ks + 1 are in flight while step ks multiplies
            TrFrag Bh[2][W1_NCOT], Bl[2][W1_NCOT], Ah[2][PPW], Al[2][PPW];
            auto load = [&](int ks, int q) {
#pragma unroll
                for (int j = 0; j < W1_NCOT; ++j) {
                    tr_issue(Bh[q][j], gst + 2 * j * CH + ks * 512);
                    if constexpr (NT == 3) tr_issue(Bl[q][j], gst + (2 * W1_NCOT + 2 * j) * CH + ks * 512);
                }
#pragma unroll
                for (int i = 0; i < PPW; ++i) {
                    tr_issue(Ah[q][i], xst + i * PAIR_BYTES + ks * 512);
                    if constexpr (NT == 3) tr_issue(Al[q][i], xst + i * PAIR_BYTES + 2 * CH + ks * 512);
                }
            };
            load(0, 0);
#pragma unroll
            for (int ks = 0; ks < 2 * TR; ++ks) {
                const int q = ks & 1;
                // everything outstanding belongs to step ks; tie its registers to the wait so no use moves above it
                if constexpr (NT == 3) {
                    if constexpr (PPW == 2)
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(Bh[q][0].a), "+v"(Bh[q][0].b), "+v"(Bh[q][1].a), "+v"(Bh[q][1].b), "+v"(Bh[q][2].a),
                                       "+v"(Bh[q][2].b), "+v"(Bl[q][0].a), "+v"(Bl[q][0].b), "+v"(Bl[q][1].a), "+v"(Bl[q][1].b),
                                       "+v"(Bl[q][2].a), "+v"(Bl[q][2].b), "+v"(Ah[q][0].a), "+v"(Ah[q][0].b), "+v"(Al[q][0].a),
                                       "+v"(Al[q][0].b), "+v"(Ah[q][PPW - 1].a), "+v"(Ah[q][PPW - 1].b), "+v"(Al[q][PPW - 1].a),
                                       "+v"(Al[q][PPW - 1].b));
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(Bh[q][0].a), "+v"(Bh[q][0].b), "+v"(Bh[q][1].a), "+v"(Bh[q][1].b), "+v"(Bh[q][2].a),
                                       "+v"(Bh[q][2].b), "+v"(Bl[q][0].a), "+v"(Bl[q][0].b), "+v"(Bl[q][1].a), "+v"(Bl[q][1].b),
                                       "+v"(Bl[q][2].a), "+v"(Bl[q][2].b), "+v"(Ah[q][0].a), "+v"(Ah[q][0].b), "+v"(Al[q][0].a),
                                       "+v"(Al[q][0].b));
                } else {
                    if constexpr (PPW == 2)
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(Bh[q][0].a), "+v"(Bh[q][0].b), "+v"(Bh[q][1].a), "+v"(Bh[q][1].b), "+v"(Bh[q][2].a),
                                       "+v"(Bh[q][2].b), "+v"(Ah[q][0].a), "+v"(Ah[q][0].b), "+v"(Ah[q][PPW - 1].a),
                                       "+v"(Ah[q][PPW - 1].b));
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(Bh[q][0].a), "+v"(Bh[q][0].b), "+v"(Bh[q][1].a), "+v"(Bh[q][1].b), "+v"(Bh[q][2].a),
                                       "+v"(Bh[q][2].b), "+v"(Ah[q][0].a), "+v"(Ah[q][0].b));
                }
                if (ks + 1 < 2 * TR) load(ks + 1, q ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                half8 bh[W1_NCOT], bl[W1_NCOT];
#pragma unroll
                for (int j = 0; j < W1_NCOT; ++j) {
                    bh[j] = tr_value(Bh[q][j]);
                    if constexpr (NT == 3) bl[j] = tr_value(Bl[q][j]);
                }
                if (bias_wave) {
#pragma unroll
                    for (int j = 0; j < W1_NCOT; ++j)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            bsum[j] += (float)bh[j][e];
                            if constexpr (NT == 3) bsum[j] += (float)bl[j][e];
                        }
                }
#pragma unroll
                for (int i = 0; i < PPW; ++i) {
                    if (!valid[i]) continue;
                    const half8 ah = tr_value(Ah[q][i]);
                    half8 al;
                    if constexpr (NT == 3) al = tr_value(Al[q][i]);
#pragma unroll
                    for (int j = 0; j < W1_NCOT; ++j) {
                        if constexpr (NT == 3) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
    }

    if (a.dbg & 4) { if (acc[0][0][0] == 12345.f) a.partial[0] = bsum[0]; return; }
    // ---- every accumulator tile is complete in its wave: straight to the partial buffer (layout of the generic kernel,
    // ntap = 1, z = co tile; 32 lanes = 32 consecutive floats)
    const int n = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        if (!valid[i]) continue;
        const int cp = cp0 + wave * PPW + i;
#pragma unroll
        for (int j = 0; j < W1_NCOT; ++j) {
            if (j >= a.ncot) continue;
            float* dst = a.partial + (((long long)j * a.ncp + cp) * a.PB + pb) * 1024 + n;
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2) + 4 * hi) * 32] = acc[i][j][e];
        }
    }
    if (bias_wave) {
        // a gY fragment lane holds 8 pixels of output channel lane & 31; lanes l and l + 32 hold the two pixel halves
#pragma unroll
        for (int j = 0; j < W1_NCOT; ++j) {
            const float t = bsum[j] + __shfl_xor(bsum[j], 32);
            if (lane < 32 && j < a.ncot) a.partial_b[((long long)j * a.PB + pb) * 32 + lane] = t;
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

#if BINHIP_TUNING
// side builds only: ablation switches (1 skip DMA, 2 skip MFMA, 4 skip reduce/store; 3x3 kernel choice: 16 = the generic
// double-buffered 4-wave kernel, 32 = the lean single-stage kernel at two workgroups per CU, bits 8..15 = its start stagger, 128 = the rolling-row kernel)
int g_wg_dbg = 0;
#define WG_DBG g_wg_dbg
#else
#define WG_DBG 0
#endif
#define WG3_LEAN ((WG_DBG & 32) != 0)
bool use_w1(int ksize, int cout) { return ksize == 1 && cout <= 32 * W1_NCOT; }

// rolling-row 3x3 kernel: channel pairs per workgroup (<= 6), tiles per wave, and whether its 32-bit chunk addressing fits
struct R3Plan { int cgroups, ppg, tpw; };
static inline R3Plan r3_plan(int ncp) {
    R3Plan p;
    p.cgroups = (ncp + 5) / 6;
    p.ppg = (ncp + p.cgroups - 1) / p.cgroups;
    p.tpw = (9 * p.ppg + 7) / 8;
    return p;
}
// two-rows-per-stage variant: <= 4 pairs per workgroup, balanced groups
static inline R3Plan r3_plan2(int ncp) {
    R3Plan p;
    p.cgroups = (ncp + 3) / 4;
    p.ppg = (ncp + p.cgroups - 1) / p.cgroups;
    p.tpw = (9 * p.ppg + 7) / 8;
    return p;
}
static inline bool r3_usable(int ksize, int N, int H, int W, int cin_chunks, int x_cpg) {
    return ksize == 3 && x_cpg == 0 && (unsigned long long)cin_chunks * N * H * W * 32ull < 0xfffffff0ull;
}

WgGeom wg_geom(int ksize, int N, int H, int W, int cin_chunks, int cout, int cus, int roll = 0) {
    WgGeom g;
    if (roll) {
        g.tr = 3; g.ndyg = 1; g.ntap = 9;
        g.ncp = (cin_chunks + 1) / 2;
        g.ncot = (cout + 31) / 32;
        const R3Plan rp = (roll == 2) ? r3_plan2(g.ncp) : r3_plan(g.ncp);
        g.tiles_x = (W + 31) / 32;
        g.tiles_y = (H + R3_SEG - 1) / R3_SEG;              // column segments per image
        g.ntiles = g.tiles_x * g.tiles_y * N;
        int pb = (cus > 0 ? cus : 256) / (rp.cgroups * g.ncot);
        if (pb < 1) pb = 1;
        if (pb > g.ntiles) pb = g.ntiles;
        g.PB = pb;
        g.partial_floats = (size_t)g.ncp * g.ncot * pb * 9 * 1024;
        g.bias_floats = (size_t)g.ncot * pb * 32;
        return g;
    }
    if (use_w1(ksize, cout)) {
        g.tr = 1; g.ndyg = 1; g.ntap = 1;
        g.ncp = (cin_chunks + 1) / 2;
        g.ncot = (cout + 31) / 32;
        const W1Plan wp = w1_plan(g.ncp);
        g.tiles_x = (W + 31) / 32;
        g.tiles_y = (H + wp.tr - 1) / wp.tr;
        g.ntiles = g.tiles_x * g.tiles_y * N;
        int pb = (cus > 0 ? cus : 256) / wp.cgroups;      // one workgroup per CU (the LDS is all staging buffer)
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
    // 3x3: one 8-wave workgroup per CU (two LDS stages); 5x5 / wide 1x1: the generic kernel, also one per CU
    int pb = ((WG3_LEAN && ksize == 3 ? 2 : 1) * (cus > 0 ? cus : 256)) / groups;   // floor: no straggler in an extra round
    if (pb < 1) pb = 1;
    if (pb > g.ntiles) pb = g.ntiles;
    if (pb >= 8) pb &= ~7;                              // wg_block(): siblings of a pixel block share an XCD
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
    dim3 grid((unsigned)(g.PB * g.ncp * g.ncot * g.ndyg));
    wgrad_mfma_kernel<KS, TR, NT><<<grid, dim3(256), C::LDS_BYTES, s>>>(a);
    BH_CHECK_LAUNCH();
    return 0;
}

#if BINHIP_TUNING
template <int KS, int TR, int NT>
int launch_wg_sb(const WgradKArgs& a, const WgGeom& g, hipStream_t s) {
    using C = WgCfg<KS, TR, NT>;
    constexpr int LDS = C::BUF_BYTES > 16384 ? C::BUF_BYTES : 16384;
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = bh_set_max_lds(&wgrad_mfma_sb_kernel<KS, TR, NT>, LDS, lds_set)) return rc;
    dim3 grid((unsigned)(g.PB * g.ncp * g.ncot * g.ndyg));
    wgrad_mfma_sb_kernel<KS, TR, NT><<<grid, dim3(256), LDS, s>>>(a);
    BH_CHECK_LAUNCH();
    return 0;
}
#endif

template <int NT>
int launch_wg3x(const WgradKArgs& a, const WgGeom& g, hipStream_t s) {
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = bh_set_max_lds(&wgrad3x3_xrow_kernel<NT>, Wg3xCfg<NT>::LDS_BYTES, lds_set)) return rc;
    wgrad3x3_xrow_kernel<NT><<<dim3((unsigned)(g.PB * g.ncp * g.ncot * g.ndyg)), dim3(512), Wg3xCfg<NT>::LDS_BYTES, s>>>(a);
    BH_CHECK_LAUNCH();
    return 0;
}

#if BINHIP_TUNING
template <int NT>
int launch_wg3(const WgradKArgs& a, const WgGeom& g, hipStream_t s) {
    using C = WgCfg<3, 3, NT>;
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = bh_set_max_lds(&wgrad3x3_db_kernel<NT>, Wg3Cfg<NT>::LDS_BYTES, lds_set)) return rc;
    wgrad3x3_db_kernel<NT><<<dim3((unsigned)(g.PB * g.ncp * g.ncot * g.ndyg)), dim3(512), Wg3Cfg<NT>::LDS_BYTES, s>>>(a);
    BH_CHECK_LAUNCH();
    return 0;
}
#endif

#if BINHIP_TUNING
template <int NT, int TPW>
int launch_r3_t(const WgradKArgs& a, const WgGeom& g, const R3Plan& rp, hipStream_t s) {
    using R = R3Cfg<NT, TPW>;
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = bh_set_max_lds(&wgrad3x3_roll_kernel<NT, TPW>, R::LDS_BYTES, lds_set)) return rc;
    wgrad3x3_roll_kernel<NT, TPW><<<dim3((unsigned)g.PB, (unsigned)(rp.cgroups * g.ncot)), dim3(512), R::LDS_BYTES, s>>>(a);
    BH_CHECK_LAUNCH();
    return 0;
}
template <int NT, int TPW>
int launch_r32_t(const WgradKArgs& a, const WgGeom& g, const R3Plan& rp, hipStream_t s) {
    using R = R3Cfg2<NT, TPW>;
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = bh_set_max_lds(&wgrad3x3_roll2_kernel<NT, TPW>, R::LDS_BYTES, lds_set)) return rc;
    wgrad3x3_roll2_kernel<NT, TPW><<<dim3((unsigned)g.PB, (unsigned)(rp.cgroups * g.ncot)), dim3(512), R::LDS_BYTES, s>>>(a);
    BH_CHECK_LAUNCH();
    return 0;
}
template <int NT>
int launch_r32(const WgradKArgs& a, const WgGeom& g, const R3Plan& rp, hipStream_t s) {
    switch (rp.tpw) {
        case 2: return launch_r32_t<NT, 2>(a, g, rp, s);
        case 3: return launch_r32_t<NT, 3>(a, g, rp, s);
        case 4: return launch_r32_t<NT, 4>(a, g, rp, s);
        case 5: return launch_r32_t<NT, 5>(a, g, rp, s);
    }
    return BINHIP_E_SHAPE;
}
template <int NT>
int launch_r3(const WgradKArgs& a, const WgGeom& g, const R3Plan& rp, hipStream_t s) {
    switch (rp.tpw) {
        case 2: return launch_r3_t<NT, 2>(a, g, rp, s);
        case 3: return launch_r3_t<NT, 3>(a, g, rp, s);
        case 4: return launch_r3_t<NT, 4>(a, g, rp, s);
        case 5: return launch_r3_t<NT, 5>(a, g, rp, s);
        case 6: return launch_r3_t<NT, 6>(a, g, rp, s);
        case 7: return launch_r3_t<NT, 7>(a, g, rp, s);
    }
    return BINHIP_E_SHAPE;
}
#endif

template <int NT, int PPW, int TR>
int launch_w1(const WgradKArgs& a, const WgGeom& g, const W1Plan& wp, hipStream_t s) {
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = bh_set_max_lds(&wgrad1x1_kernel<NT, PPW, TR>, 160 * 1024, lds_set)) return rc;
    const unsigned lds = (NT == 3) ? wp.lds : wp.lds / 2;
    wgrad1x1_kernel<NT, PPW, TR><<<dim3((unsigned)g.PB, (unsigned)wp.cgroups), dim3(64 * W1_NW), lds < 16384 ? 16384 : lds, s>>>(a);
    BH_CHECK_LAUNCH();
    return 0;
}

// CU count of the current device (sizes the pixel-block split); looked up per call — no cached global
int cus() {
    const int n = binhip_device_cus();
    return n > 0 ? n : 256;
}

}  // namespace

extern "C" {

#if BINHIP_TUNING
BINHIP_API int binhip_wgrad_set_debug(int flags) { g_wg_dbg = flags; return 0; }
#endif

size_t binhip_wgrad_workspace_bytes(int ksize, int N, int H, int W, int cin_chunks, int cout) {
    if (N <= 0 || H <= 0 || W <= 0 || cin_chunks <= 0 || cout <= 0) return 0;
    const WgGeom g = wg_geom(ksize, N, H, W, cin_chunks, cout, cus());
    size_t fl = g.partial_floats + g.bias_floats;
    if (BINHIP_TUNING && r3_usable(ksize, N, H, W, cin_chunks, 0)) {   // the rolling-row experiment keeps more partials
        for (int roll = 1; roll <= 2; ++roll) {
            const WgGeom r = wg_geom(ksize, N, H, W, cin_chunks, cout, cus(), roll);
            if (r.partial_floats + r.bias_floats > fl) fl = r.partial_floats + r.bias_floats;
        }
    }
    return fl * sizeof(float) + 256;
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
    const bool r3ok = r3_usable(d->ksize, d->N, d->H, d->W, d->cin_chunks, d->x_cpg);
    const int roll = ((WG_DBG & 64) && r3ok) ? 2 : ((WG_DBG & 128) && r3ok) ? 1 : 0;       // side builds only
    const WgGeom g = wg_geom(d->ksize, d->N, d->H, d->W, d->cin_chunks, d->cout, cus(), roll);
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
    a.PB = g.PB; a.ncp = g.ncp; a.ncot = g.ncot; a.ppg = 0; a.nz = g.ncot * g.ndyg; a.cgroups = 1;
    a.dbg = WG_DBG & 0xff0f;      // bits 8..15: start stagger of the lean kernel in units of s_sleep 16 (experiment)
    hipStream_t s = (hipStream_t)stream;
    int rc = BINHIP_E_SHAPE;
    if (use_w1(d->ksize, d->cout)) {
        const W1Plan wp = w1_plan(g.ncp);
        a.ppg = wp.ppg;
        rc = (d->nterms == 1)
            ? (wp.ppw == 2 ? launch_w1<1, 2, 1>(a, g, wp, s) : wp.tr == 2 ? launch_w1<1, 1, 2>(a, g, wp, s) : launch_w1<1, 1, 1>(a, g, wp, s))
            : (wp.ppw == 2 ? launch_w1<3, 2, 1>(a, g, wp, s) : wp.tr == 2 ? launch_w1<3, 1, 2>(a, g, wp, s) : launch_w1<3, 1, 1>(a, g, wp, s));
#if BINHIP_TUNING
    } else if (d->ksize == 3 && (WG_DBG & 16)) {    // side builds: the generic double-buffered 4-wave kernel for 3x3
        rc = (d->nterms == 1) ? launch_wg<3, 3, 1>(a, g, s) : launch_wg<3, 3, 3>(a, g, s);
    } else if (d->ksize == 3 && WG3_LEAN) {         // side builds: the lean single-stage kernel, two workgroups per CU
        rc = (d->nterms == 1) ? launch_wg_sb<3, 3, 1>(a, g, s) : launch_wg_sb<3, 3, 3>(a, g, s);
#endif
#if BINHIP_TUNING
    } else if (roll == 2) {                         // 3x3, rolling rows, two rows per stage (experiment)
        const R3Plan rp = r3_plan2(g.ncp);
        a.cgroups = rp.cgroups;
        a.ppg = rp.ppg;
        rc = (d->nterms == 1) ? launch_r32<1>(a, g, rp, s) : launch_r32<3>(a, g, rp, s);
    } else if (roll) {                              // 3x3, rolling rows (experiment)
        const R3Plan rp = r3_plan(g.ncp);
        a.cgroups = rp.cgroups;
        rc = (d->nterms == 1) ? launch_r3<1>(a, g, rp, s) : launch_r3<3>(a, g, rp, s);
#endif
#if BINHIP_TUNING
    } else if (d->ksize == 3 && ((BINHIP_WG3_XROW != 0) == (((WG_DBG >> 16) & 1) != 0))) {
        // side builds: the round-2 form, wave = gY row (debug bit 16, or -DBINHIP_WG3_XROW=0 to make it the side build's default)
        rc = (d->nterms == 1) ? launch_wg3<1>(a, g, s) : launch_wg3<3>(a, g, s);
#endif
    } else if (d->ksize == 3) {
        // eight waves, two LDS stages, one workgroup per CU; wave = X row
        rc = (d->nterms == 1) ? launch_wg3x<1>(a, g, s) : launch_wg3x<3>(a, g, s);
    } else if (d->ksize == 1) {                     // 1x1 with more than 96 outputs (not on the bin_stage4 path)
        rc = (d->nterms == 1) ? launch_wg<1, 1, 1>(a, g, s) : launch_wg<1, 1, 3>(a, g, s);
    } else {                                        // 5x5 (SFENet1)
        rc = (d->nterms == 1) ? launch_wg<5, 1, 1>(a, g, s) : launch_wg<5, 1, 3>(a, g, s);
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
