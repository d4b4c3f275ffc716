// Argument block of the fused dense-block tail kernels (binhip_fused.hip, binhip_fused_x3.hip).  Internal.
#pragma once
#include "binhip_conv_common.h"

struct TailKArgs {
    const _Float16 *x_hi, *x_lo;       // dense block buffer, chunk 0
    const _Float16 *wc_hi, *wc_lo;     // conv #3 weights  [12][9][32][16]
    const _Float16 *wl_hi, *wl_lo;     // LFF weights      [14][1][96][16]
    const float *bc, *bl;              // biases (32 / 96 floats)
    _Float16 *y_hi, *y_lo;             // output planes (6 chunks)
    _Float16 *o3_hi, *o3_lo;           // optional: where to keep o3 (2 chunks) for the backward pass
    unsigned* flags;                   // device status word (BINHIP_STATUS_*), may be null
    int N, H, W, tiles_x, tiles_y, xcd_remap, wt;
#if BINHIP_TIMELINE
    void* tl;                          // side builds: see BhTl (binhip_conv_common.h)
    unsigned tl_launch, tl_base;
#endif
};

int bh_launch_tail_x3(const TailKArgs& a, hipStream_t s);     // binhip_fused_x3.hip (nterms = 3)
