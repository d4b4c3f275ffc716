// Internal declarations shared by the libbinhip translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/binhip.h"

// BINHIP_TUNING (side builds for tools/: kernel-variant sweeps and ablations; 0 in the product): compiles the
// alternative tile configurations and the process-global switches that select them.  The product library has neither.
#ifndef BINHIP_TUNING
#define BINHIP_TUNING 0
#endif


// Chunk-plane helpers -------------------------------------------------------------------------
// CP tensor: fp16 [chunk][N][H][W][16]; plane_elems = N*H*W*16.
static inline int64_t bh_plane_elems(int N, int H, int W) { return (int64_t)N * H * W * 16; }
static inline int bh_chunks(int C) { return (C + 15) / 16; }
__device__ static inline int bh_chunks_dev(int C) { return (C + 15) / 16; }

#define BH_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)

// fp16 storage range.  The reference computes in fp32 (RDN.py:141, no AMP), so a value beyond +-65504 is legal there;
// here it cannot be stored in an fp16 hi plane.  Rather than let hi = inf, lo = -inf poison every later layer with NaN,
// the epilogue saturates hi to +-65504 (lo then carries what it can of the rest) and raises bit 0 of the status word,
// which the host checks (bin_amd/ops.py: RuntimeError "fp16 range exceeded").  Documented in include/binhip.h.
#define BINHIP_F16_MAX 65504.0f
#define BINHIP_FLAG_SATURATED 1u

// returns hi; sat |= (v left the range or is NaN).  The test is ONE integer compare on the magnitude bits (NaN and inf
// patterns are above 65504's), OR-ed without short-circuit: a `sat = sat || ...` chain made hipcc keep every converted
// value of the epilogue live (+86 VGPRs on the multi-tile kernels, one wave per SIMD less: GFF.0 253 -> 352 us)
__device__ __forceinline__ _Float16 split_hi(float v, unsigned& sat) {
    sat |= ((__float_as_uint(v) & 0x7fffffffu) > 0x477fe000u) ? 1u : 0u;
    // v_med3_f32 directly (round 4): fminf(fmaxf()) costs a canonicalising v_max per call on top of the med3 — two of the ~12 VALU
    // instructions per stored value, in epilogues that are instruction-issue-bound (GFF.0 backward-data: 3 583 instructions per pixel
    // row and wave for 48 stores).  Same results: in range the median is v; a NaN operand makes v_med3 return min3 = -65504, which is
    // what the fminf / fmaxf chain produced.
    return (_Float16)__builtin_amdgcn_fmed3f(v, -BINHIP_F16_MAX, BINHIP_F16_MAX);
}
// lo = v - hi, itself kept inside the fp16 range: in range it is |lo| <= ulp(hi)/2 and the clamp is the identity; after
// a saturated hi the excess can be anything (or NaN), and an unclamped conversion would store inf / NaN after all
__device__ __forceinline__ _Float16 split_lo(float v, _Float16 hi) {
    return (_Float16)__builtin_amdgcn_fmed3f(v - (float)hi, -BINHIP_F16_MAX, BINHIP_F16_MAX);
}

// Two values at once, packed (round 6).  The epilogues are VALU-bound — ~12 instructions per stored value, 192 values per lane in the
// fused tail's LFF epilogue (5.4 us per tile, profiles/r06_wg_timeline.md) — and most of them were this split: clamp, convert, convert
// back, subtract, clamp, convert, pack.  Here: one v_med3 per value (the clamp of hi ALSO bounds lo: |vc - hi| <= 32), one
// v_cvt_pk_f16_f32 per pair, and lo = vc - hi as ONE v_fma_mix{lo,hi}_f16 per value — an fp32 fma of (f16 hi) * -1.0 + vc whose exact
// result is rounded once to f16, straight into its half of the packed dword: the same bits as the subtract-then-convert form for every
// in-range value (tests/test_gpu_conv.py::test_packed_split_equals_the_scalar_split).  A saturated or NaN value gives lo = 0 instead of a
// clamped remainder; such outputs raise at the host either way.
#ifndef BINHIP_SPLIT_ASM
#define BINHIP_SPLIT_ASM 1
#endif
// RELU (wave-uniform): the layer's ReLU rides on the clamp — the lower bound of the v_med3 becomes 0 (fmaxf costs a canonicalising
// v_max on top of the v_max itself) — and a large NEGATIVE value is then not a saturation: the range test keeps the sign bit and
// compares signed.
__device__ __forceinline__ void split_pair(float a, float b, unsigned& sat, unsigned& hi2, unsigned& lo2, bool relu = false) {
#if BINHIP_SPLIT_ASM
    const unsigned smask = relu ? 0xffffffffu : 0x7fffffffu;
    const float lb = relu ? 0.f : -BINHIP_F16_MAX;
    sat |= ((int)(__float_as_uint(a) & smask) > 0x477fe000) ? 1u : 0u;
    sat |= ((int)(__float_as_uint(b) & smask) > 0x477fe000) ? 1u : 0u;
    const float ac = __builtin_amdgcn_fmed3f(a, lb, BINHIP_F16_MAX);
    const float bc = __builtin_amdgcn_fmed3f(b, lb, BINHIP_F16_MAX);
    unsigned h, l;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(ac), "v"(bc));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(ac));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(bc));
    hi2 = h; lo2 = l;
#else
    if (relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
    union { _Float16 h[2]; unsigned u; } H, L;
    H.h[0] = split_hi(a, sat); L.h[0] = split_lo(a, H.h[0]);
    H.h[1] = split_hi(b, sat); L.h[1] = split_lo(b, H.h[1]);
    hi2 = H.u; lo2 = L.u;
#endif
}

// internal launcher used by both the per-op ABI and the RDN plan
struct BhConvCall {
    BinConvDesc d;
    const void *x_hi, *x_lo, *w_hi, *w_lo;
    const float* bias;
    const void *r_hi, *r_lo;
    const void *r2_hi = nullptr, *r2_lo = nullptr;   // second residual (may alias y: in-place accumulate)
    const void* m_hi = nullptr;                      // ReLU mask source planes (hi)
    int res_chunks = 0;                              // 0: residual applies to every chunk
    int mask_from = 0;
    int y_cpg = 0;
    int64_t y_group_stride = 0;
    int y_unshuf = 0;                                // > 0: store through an inverse PixelShuffle(2), chunks per sub-position
    void *y_hi, *y_lo;
    float* y_f32;
    const float* images[5];
    void* status = nullptr;                          // device status word (BINHIP_STATUS_*), may be null
    BinhipProfiler* prof = nullptr;                  // optional live-timing handle (binhip_profiler_create)
};
int bh_launch_conv(const BhConvCall& c, hipStream_t s);
// live-timing handle (binhip_profiler_create): begin() records the start event and returns true when the launch matches
// the handle's (ksize, cout_pad, epilogue) class and a pair is free; end() records the stop event
bool bh_prof_begin(BinhipProfiler* pr, int ks, int cout_pad, int epi, hipStream_t s);
void bh_prof_end(BinhipProfiler* pr, hipStream_t s);
struct ConvKArgs;
int bh_prepare_conv(const BhConvCall& c, ConvKArgs* out);
// convs 0-2 of a residual dense block as three phases of one launch (binhip_conv_x3.hip); nterms = 3 only
int bh_launch_rdb3_x3(const ConvKArgs* convs, unsigned* flags, unsigned epoch, int cus, hipStream_t s);

// weight gradients in two phases, so a plan can reduce several layers' partials with one launch
#define BH_WGRAD_BATCH 8
struct BhWgradReduce {
    const float* partial;
    const float* partial_b;
    float* dw;
    float* db;
    long long nrows;
    int PB, ncp, ncot, ks, tr, cout, cin, shuffle;
};
int bh_wgrad_partials(const BinConvDesc* d, const void* x_hi, const void* x_lo, const void* gy_hi, const void* gy_lo,
                      void* workspace, size_t workspace_bytes, float* dw_oihw, float* dbias, int cin, int shuffle_perm,
                      BhWgradReduce* out, void* stream);
int bh_wgrad_reduce_batch(const BhWgradReduce* items, int n, const float* inv_scale, int accumulate, void* stream);
int bh_conv_cout_block(int ksize, int cout_pad, int nterms);

// per-device one-time hipFuncSetAttribute(MaxDynamicSharedMemorySize): the attribute is per device, so the cache is a
// bit per device ordinal (immutable once set; racing threads at worst set the attribute twice)
#include <atomic>
template <class K>
static inline int bh_set_max_lds(K kernel, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}
