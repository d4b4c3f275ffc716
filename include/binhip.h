/*
 * binhip.h — flat C ABI of libbinhip.so: the MI355X (gfx950) hot path of laomao0/BIN's
 * `bin_stage4` network (reference: /root/reference/models/archs/RDN.py).
 *
 * The reference has no native code; the interface this library replaces is the set of ATen
 * operator calls made by models/archs/RDN.py (F.conv2d / relu / cat / pixel_shuffle / sigmoid /
 * tanh) and models/loss.py:130-141 behind the nn.Module boundary `define_G(opt)`
 * (models/networks.py:5-14).  Each entry point cites the reference lines it stands in for.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller (PyTorch allocates); the library never
 *     allocates, frees or retains device memory.  `stream` is a hipStream_t passed as void*.
 *   - All work is enqueued asynchronously on `stream`; no host synchronisation inside.
 *   - Return value: 0 = ok, negative = argument/shape error (BINHIP_E_*), positive = hipError_t.
 *   - Activations between layers live in "chunk planes" (CP): fp16 [C/16][N][H][W][16]
 *     (channels-last inside 16-channel chunks, one contiguous plane per chunk, so a dense block's
 *     concat is just "the next plane").  A CP tensor has a `hi` plane set and, in split precision
 *     (nterms == 3), a `lo` plane set with x ~= hi + lo to ~22 mantissa bits.
 *   - nterms: 1 = fp16-input MFMA, fp32 accumulate (whole-net max-abs error ~3e-4 vs fp32);
 *             3 = fp16 hi/lo split, three MFMA products (error ~1e-6, fp32 class).
 *   - Dynamic range.  The reference computes in fp32 (RDN.py:141, no AMP); here every value stored
 *     between layers must fit the fp16 hi plane: |v| <= 65504.  Kernels SATURATE hi to +-65504
 *     (never inf/NaN from overflow; lo keeps what it can of the excess) and OR BINHIP_STATUS_SATURATED
 *     into the caller's device status word (`status`, may be NULL = not reported), so results that
 *     left the range are detectable instead of silently wrong; a NaN input raises the same bit.
 *     Small magnitudes: nterms = 3 represents |v| >= 2^-24 * 2^-11 relative steps down to the fp16
 *     subnormal floor 6e-8 (absolute error <= 3e-8 per stored value); nterms = 1 flushes nothing but
 *     rounds to 11 bits, absolute error <= 3e-8 below 6.1e-5.  Backward gradient planes carry a
 *     per-call power-of-two scale (binhip_grad_scale) chosen from amax(gout) so that the largest
 *     upstream gradient maps to 16; the status word covers their overflow as well.
 *   - The library holds NO mutable process-global state: every switch is an argument, the only
 *     caches are per-device "attribute already set" bits.  Entry points are re-entrant; calls for
 *     different devices / streams may run concurrently from different host threads.
 */
#ifndef BINHIP_H
#define BINHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the entry points declared here (BINHIP_API) are its ONLY dynamic
 * symbols (tests/test_cpu_host.py checks `nm -D` against this header). */
#define BINHIP_API __attribute__((visibility("default")))

/* Interface version = what binhip_version() of a matching library returns (100 x round + revision); a binder checks
 * `binhip_version() == BINHIP_VERSION` after dlopen.  BINHIP_ABI_EXPORTS = number of BINHIP_API entry points below
 * (tests/test_cpu_host.py keeps it equal to the declarations and to `nm -D`). */
#define BINHIP_VERSION 610
#define BINHIP_ABI_EXPORTS 45

#define BINHIP_E_ARG      (-1)   /* null pointer / bad enum */
#define BINHIP_E_SHAPE    (-2)   /* unsupported shape */
#define BINHIP_E_WORKSPACE (-3)  /* workspace too small */

#define BINHIP_STATUS_SATURATED 1u   /* bit 0 of a status word: a stored value was clamped to +-65504 (or was NaN) */
#define BINHIP_STATUS_SYNC_TIMEOUT 2u /* bit 1: a tile of the three-phase dense-block launch (BINHIP_PLAN_RDB3) gave up waiting for a
                                        neighbour's flag and computed from inputs that may not have been published: results invalid */

#define BINHIP_EPI_PLANES  0     /* y = [relu](conv + b [+ residual]) -> chunk planes            */
#define BINHIP_EPI_SHUFFLE 1     /* conv + b -> PixelShuffle(2) -> chunk planes at 2H x 2W       */
#define BINHIP_EPI_FINAL   2     /* conv + b + mean(images) -> fp32 NCHW [N,cout,H,W]            */
#define BINHIP_EPI_FINAL_SUBPIX 4 /* a half-resolution conv whose cout = 4 c' channels are the 2 x 2 sub-pixels of c' <= 3 colour
                                     channels (order c' * 4 + i * 2 + j): conv + b + mean(images) -> fp32 NCHW [N,c',2H,2W].  The
                                     epilogue of the FUSED UPNet (BINHIP_PLAN_FUSED_UPNET below); 5x5, cout_pad 32, both precision modes */

#define BINHIP_RDN_LAYERS 66     /* bin_stage4: SFE1, SFE2, 12 x (4 conv + LFF), GFF.0, GFF.1, UP.0, UP.2 */
/* Shape of an RDN sub-network (constructor arguments of RDN.py:168-186): G0 feature channels, D residual dense blocks of
 * C 3x3 convs growing by G channels each.  All zero = bin_stage4's (96, 12, 4, 32).  Supported: G0 and G multiples of 32,
 * 32 <= G0 <= 256, G <= 128, 1 <= C <= 7, 1 <= D <= 20.  Layer order / index: SFENet1 = 0, SFENet2 = 1, conv c of block d =
 * 2 + d (C + 1) + c, LFF of block d = 2 + d (C + 1) + C, then GFF.0, GFF.1, UPNet.0, UPNet.2; 2 + D (C + 1) + 4 layers.    */
typedef struct BinRdnShape { int32_t G0, D, C, G; } BinRdnShape;
#define BINHIP_RDN_MAX_LAYERS 192
#define BINHIP_RDN_MAX_CONVS 7

BINHIP_API int binhip_version(void);

/* Device properties the host needs for sizing (no allocation). */
BINHIP_API int binhip_device_cus(void);

/* ---- stride-1 "same" convolution on chunk planes (RDN.py:141,162,187-188,199-200,205-207) -----
 * Replaces F.conv2d(+bias)(+ReLU)(+cat)(+residual add)(+PixelShuffle)(+input mean).            */
typedef struct BinConvDesc {
    int32_t N, H, W;          /* input batch / height / width (output same, 2x for SHUFFLE)      */
    int32_t ksize;            /* 1, 3 or 5                                                       */
    int32_t cin_chunks;       /* number of 16-channel input chunks (zero-padded channels allowed) */
    int32_t cout;             /* real output channels                                            */
    int32_t cout_pad;         /* weight rows, multiple of 32 (32, 96, 256)                       */
    int32_t nterms;           /* 1 or 3                                                          */
    int32_t epilogue;         /* BINHIP_EPI_*                                                    */
    int32_t relu;             /* apply ReLU (PLANES only)                                        */
    int32_t x_cpg;            /* input chunks per group; chunk i lives at                        */
    int64_t x_group_stride;   /*   x + (i / x_cpg) * x_group_stride + (i % x_cpg) * N*H*W*16     */
                              /*   (elements). x_cpg <= 0: one group.                             */
    int32_t n_images;         /* FINAL: number of fp32 NCHW images averaged into the output      */
    int32_t reserved;         /* binhip_conv2d_bwd_data only: y_unshuf > 0 = store the result through an  */
                              /*   inverse PixelShuffle(2): full-resolution pixel (Y, X), chunk c goes to    */
                              /*   plane (2 (Y & 1) + (X & 1)) * y_unshuf + c at (Y / 2, X / 2) of a tensor  */
                              /*   of 4 * y_unshuf planes at H/2 x W/2 — what binhip_unshuffle_planes makes  */
                              /*   of the plain result (H, W even, y_cpg = 0, 16 * y_unshuf >= cout); a      */
                              /*   negative value is BINHIP_E_ARG.                                           */
                              /* binhip_conv2d_fwd: BINHIP_CONV_HALF_LAST_CHUNK (ksize 5 only) = the caller        */
                              /*   guarantees that input channels 8-15 of the LAST chunk are zero IN THE WEIGHTS   */
                              /*   (cin % 16 in 1..8: SFENet1's 24 / 36 inputs; the relayout pads them with zeros) */
                              /*   — the 5x5 fp32-class kernel then spends that chunk's K on tap pairs and never   */
                              /*   reads x's channels 8-15 of it; other kernels ignore the bit.  Any other bit, or */
                              /*   the bit with ksize != 5, is BINHIP_E_ARG.  0 everywhere else.                   */
    void*   status;           /* device uint32 status word (BINHIP_STATUS_*), OR-ed into; or NULL */
} BinConvDesc;

#define BINHIP_CONV_HALF_LAST_CHUNK 1

/* Rows per weight block for a (ksize, cout_pad, nterms) configuration (relayout needs it). */
BINHIP_API int binhip_conv_cout_block(int ksize, int cout_pad, int nterms);

/* OIHW fp32 -> kernel layout fp16 [cout_pad/cb][cin_chunks][k*k][cb][16] (+lo), 16-byte slots
 * XOR-swizzled for conflict-free ds_read_b128; input channels >= cin and rows >= cout are zero.
 * shuffle_perm != 0 reorders output rows co' = (co%4)*(cout/4) + co/4 so PixelShuffle becomes a
 * plain plane store.  bias_out: fp32 [cout_pad] (same row order, zero padded).  w_lo may be NULL. */
BINHIP_API int binhip_weights_relayout(const float* w_oihw, const float* bias, int cout, int cin, int ksize,
                            int cout_pad, int cin_chunks, int cout_block, int shuffle_perm,
                            void* w_hi, void* w_lo, float* bias_out, void* stream);
BINHIP_API size_t binhip_weights_bytes(int cout_pad, int cin_chunks, int ksize);   /* per plane (hi or lo) */
/* Several relayouts per launch (a training step re-lays-out all 66 layers of a weight set twice, forward and backward
 * layouts, after every optimizer update).  One item = the arguments of binhip_weights_relayout (kind FWD: rows_pad =
 * cout_pad, shuffle_or_group = shuffle_perm), of binhip_weights_relayout_dgrad (kind DGRAD) or of
 * binhip_weights_relayout_rdb_gather (kind RDB_GATHER: w[0..C-1] = the block's conv weights, shuffle_or_group = group,
 * rows_pad = G0 (group 0) / G, cin_chunks = (C - group) G / 16, ksize = 3; bias is ignored, bias_out zeroed; `shape` gives
 * G0, C, G for blocks other than bin_stage4's) — same results bit for bit.
 * `items` is a HOST array; nothing of it is referenced after the call returns.                                        */
#define BINHIP_RELAYOUT_FWD        0
#define BINHIP_RELAYOUT_DGRAD      1
#define BINHIP_RELAYOUT_RDB_GATHER 2
typedef struct BinRelayoutItem {
    const float* w[BINHIP_RDN_MAX_CONVS + 1]; /* OIHW fp32 sources (w[0] only, except RDB_GATHER: the block's C convs) */
    const float* bias;        /* FWD: bias or NULL                                                   */
    void* w_hi;
    void* w_lo;               /* NULL when nterms == 1                                               */
    float* bias_out;          /* rows_pad floats                                                     */
    int32_t kind, cout, cin, ksize, rows_pad, cin_chunks, cout_block, shuffle_or_group;
    BinRdnShape shape;        /* RDB_GATHER only: the dense block's (G0, -, C, G); all zero = (96, -, 4, 32)        */
} BinRelayoutItem;
BINHIP_API int binhip_weights_relayout_batch(const BinRelayoutItem* items, int n, void* stream);

BINHIP_API int binhip_conv2d_fwd(const BinConvDesc* d,
                      const void* x_hi, const void* x_lo,
                      const void* w_hi, const void* w_lo, const float* bias,
                      const void* res_hi, const void* res_lo,      /* PLANES residual or NULL   */
                      void* y_hi, void* y_lo,                      /* PLANES / SHUFFLE output   */
                      float* y_f32, const float* const* images,    /* FINAL output + host array */
                      void* stream);                               /*   of n_images device ptrs */

/* ---- backward-data (dgrad) of the same convolution (autograd of RDN.py's conv2d calls) -----------
 * dgrad(conv_W)(gy) = conv_{W'}(gy) with W'[ci][co][dy][dx] = W[co][ci][k-1-dy][k-1-dx]; it runs on the
 * forward kernel.  relayout_dgrad builds W' (rows = original input channels padded to rows_pad, input
 * chunks = original output channels; shuffle_perm: those are in UPNet.0's PixelShuffle-permuted order).
 * bwd_data: gx = [mask](conv_{W'}(gy) [+ res on chunks < res_chunks] [+ acc])
 *   res   — e.g. the RDB skip path gradient (RDN.py:165 `+ x`)
 *   acc   — running gradient of the dense block input being accumulated; may alias gx (in place)
 *   mask  — saved forward activation planes: output chunks >= mask_from are zeroed where it is <= 0
 *           (ReLU backward, RDN.py:142), applied when the last contribution to that chunk lands
 *   y_cpg / y_group_stride — output chunk grouping (GFF.0 dgrad scatters to the 12 block buffers)     */
BINHIP_API int binhip_dgrad_rows_pad(int ksize, int cin);   /* rows_pad the library expects for a layer's dgrad weights */
BINHIP_API int binhip_weights_relayout_dgrad(const float* w_oihw, int cout, int cin, int ksize, int rows_pad,
                                  int cin_chunks, int cout_block, int shuffle_perm, void* w_hi, void* w_lo,
                                  float* bias_out, void* stream);
/* Residual dense block (RDN.py:132-165) backward-data in GATHER form.  Concat group g of a block (g = 0: its 96 input
 * channels; g = 1..3: the 32 outputs of conv g-1) gets dgrad contributions from convs g..3 (0..3 for g = 0).  With
 * those convs' masked output gradients stacked along K (they are contiguous in the gradient concat buffer) the sum
 * is ONE forward-shaped conv:  Wg[r][32 (c - g) + co][dy][dx] = W_c[co][base_g + r][2-dy][2-dx]  (base_0 = 0,
 * base_g = 96 + 32 (g - 1)); rows = 96 (g = 0) or 32, input chunks = 2 (4 - g).  Every group is then written once
 * instead of being read-modified-written by each later conv.  w_oihw4 = HOST array of the block's four OIHW fp32
 * device pointers (entries < group may be NULL).                                                                  */
BINHIP_API int binhip_weights_relayout_rdb_gather(const float* const* w_oihw4, int group, int cout_block, void* w_hi, void* w_lo,
                                       float* bias_out, void* stream);
BINHIP_API int binhip_conv2d_bwd_data(const BinConvDesc* d, const void* gy_hi, const void* gy_lo,
                           const void* wt_hi, const void* wt_lo, const float* zero_bias,
                           const void* res_hi, const void* res_lo, int res_chunks,
                           const void* acc_hi, const void* acc_lo,
                           const void* mask_hi, int mask_from,
                           int y_cpg, int64_t y_group_stride,
                           void* gx_hi, void* gx_lo, void* stream);

/* ---- layout glue ------------------------------------------------------------------------------ */
/* fp32 NCHW [N,C,H,W] -> chunk planes (C padded with zeros to 16).                              */
BINHIP_API int binhip_nchw_to_planes(const float* x, int N, int C, int H, int W, void* y_hi, void* y_lo,
                          void* status, void* stream);
/* chunk planes -> fp32 NCHW (hi + lo when lo != NULL).                                           */
BINHIP_API int binhip_planes_to_nchw(const void* x_hi, const void* x_lo, int N, int C, int H, int W, float* y,
                          void* stream);
/* exact fp32 space-to-depth, the reference's standalone pixel_reshuffle (RDN.py:107-132)                */
BINHIP_API int binhip_pixel_unshuffle_f32(const float* x, int N, int C, int H, int W, int r, float* y, void* stream);
/* K1: pixel_reshuffle(cat(images), 2) (RDN.py:107-132, 211/269/323) fused into the CP writer:
 * `n_images` fp32 [N,3,H,W] -> CP [N, H/2, W/2, pad16(12*n_images)].                            */
BINHIP_API int binhip_pack_inputs(const float* const* images, int n_images, int N, int H, int W,
                       void* y_hi, void* y_lo, void* status, void* stream);

/* ---- harness glue: test.py's per-frame host work on the device --------------------
 * u8_to_frame: HWC BGR uint8 -> fp32 CHW RGB /255 (read_image, test.py:44-56) + ReplicationPad2d
 * (test.py:348-371) -> [3, H+pt+pb, W+pl+pr].   frame_to_u8: tensor2img (utils/util.py:113-137: clamp,
 * x255, round half to even, RGB->BGR) + the crop of test.py:394-402 -> HWC BGR uint8 [H, W, 3].      */
BINHIP_API int binhip_u8_to_frame(const unsigned char* bgr_hwc, int H, int W, int pad_left, int pad_right,
                       int pad_top, int pad_bottom, float* out_chw, void* stream);
BINHIP_API int binhip_frame_to_u8(const float* chw, int Hp, int Wp, int top, int left, int H, int W,
                       unsigned char* bgr_hwc, void* stream);

/* ---- ConvLSTM cell (RDN.py:50-95): gates = conv3x3(cat(x,h)) 6->12, i,j,f,o = chunk(4);
 * c' = c*sigmoid(f+forget_bias) + sigmoid(i)*tanh(j); h' = tanh(c')*sigmoid(o).  fp32 NCHW.
 * c_prev/h_prev may both be NULL (zero state, RDN.py:57-68).                                      */
BINHIP_API int binhip_convlstm_fwd(const float* x, const float* c_prev, const float* h_prev,
                        const float* w /*[12,6,3,3]*/, const float* b /*[12]*/, float forget_bias,
                        int N, int H, int W, float* c_new, float* h_new, void* stream);

/* Gate arithmetic of a ConvLSTM cell of ANY size (RDN.py:14-24 takes input_size / hidden_size; RDN.py:74-82): the gates
 * convolution of such a cell runs on binhip_conv2d_fwd / _bwd_data / _bwd_weight, these are the elementwise rest.
 * gates: fp32 [N, 4*hidden, H, W] in (i, j, f, o) order; c_prev may be NULL (zero state); backward: g_h / g_c = gradients of
 * h' / c' (either may be NULL), g_gates [N, 4*hidden, H, W], g_cprev may be NULL.                                      */
BINHIP_API int binhip_lstm_gates_fwd(const float* gates, const float* c_prev, float forget_bias, int N, int hidden, int H, int W,
                          float* c_new, float* h_new, void* stream);
BINHIP_API int binhip_lstm_gates_bwd(const float* gates, const float* c_prev, const float* g_h, const float* g_c,
                          float forget_bias, int N, int hidden, int H, int W, float* g_gates, float* g_cprev, void* stream);

/* ---- Charbonnier loss (loss.py:137-141): mean(sqrt((x-y)^2 + eps)).  Deterministic two-pass
 * reduction; `partials` is a caller workspace of binhip_charbonnier_partials() floats.           */
BINHIP_API int binhip_charbonnier_partials(int64_t numel);
BINHIP_API int binhip_charbonnier_fwd(const float* x, const float* y, int64_t numel, float eps,
                           float* partials, float* loss, void* stream);
/* gx = gloss * (x-y)/sqrt((x-y)^2+eps)/numel ; gy = -gx (either may be NULL).                    */
BINHIP_API int binhip_charbonnier_bwd(const float* x, const float* y, int64_t numel, float eps,
                           const float* gloss, float* gx, float* gy, void* stream);
/* The three pixel criteria bin_model.py:52-60 can select ('cb' | 'l1' | 'l2'), same reduction and workspace:
 * CHARBONNIER = mean(sqrt((x-y)^2 + eps)) (the two entry points above), L1_SUM = nn.L1Loss(reduction='sum') =
 * sum|x-y| (gradient sign(x-y), 0 at 0), L2_SUM = nn.MSELoss(reduction='sum') = sum (x-y)^2.                     */
#define BINHIP_LOSS_CHARBONNIER 0
#define BINHIP_LOSS_L1_SUM      1
#define BINHIP_LOSS_L2_SUM      2
BINHIP_API int binhip_pixel_loss_fwd(int kind, const float* x, const float* y, int64_t numel, float eps,
                          float* partials, float* loss, void* stream);
BINHIP_API int binhip_pixel_loss_bwd(int kind, const float* x, const float* y, int64_t numel, float eps,
                          const float* gloss, float* gx, float* gy, void* stream);

/* bin_model.get_loss (bin_model.py:395-425) as TWO launches forward and ONE backward (round 5; before: one pair of launches per
 * term, the 17-term sum / division and their autograd as ~100 scalar ATen kernels between forward and backward):
 *   terms[t] = criterion(x[t], y[t]) over `numel` floats each (same reduction, same bits as binhip_pixel_loss_fwd),
 *   loss     = (((terms[0] + terms[1]) + ...) + terms[T-1]) / T   in fp32, left to right = Python's sum(list) / len(list).
 * `partials`: n_terms * binhip_charbonnier_partials() floats.  A tensor may appear in several terms (the three cycle terms
 * pair two network outputs with each other).                                                                             */
#define BINHIP_LOSS_MAX_TERMS 24
typedef struct {
    const float* x[BINHIP_LOSS_MAX_TERMS];
    const float* y[BINHIP_LOSS_MAX_TERMS];
    int32_t n_terms;
} BinLossTerms;
BINHIP_API int binhip_multi_loss_fwd(int kind, const BinLossTerms* t, int64_t numel, float eps, float* partials, float* terms,
                          float* loss, void* stream);
/* d loss / d tensor for up to BINHIP_LOSS_MAX_TERMS distinct tensors in one launch: out[k] = gloss / T * sum over the (at most
 * two) terms that contain tensor k of sign * criterion'(x[term] - y[term]) (sign +1 where the tensor is the term's x, -1 where
 * it is its y; term_b[k] = -1: one term only).                                                                           */
typedef struct {
    float* out[BINHIP_LOSS_MAX_TERMS];
    int32_t term_a[BINHIP_LOSS_MAX_TERMS], term_b[BINHIP_LOSS_MAX_TERMS];
    float sign_a[BINHIP_LOSS_MAX_TERMS], sign_b[BINHIP_LOSS_MAX_TERMS];
    int32_t n_out;
} BinLossGrads;
BINHIP_API int binhip_multi_loss_bwd(int kind, const BinLossTerms* t, int64_t numel, float eps, const float* gloss,
                          const BinLossGrads* g, void* stream);

/* ---- fused tail of a residual dense block: o3 = relu(conv3x3(blk[0:192])), y = LFF(cat(blk[0:192], o3)) + blk[0:96]
 * (RDN.py:141-147 for conv #3, :162-165 for LFF + residual) in one kernel; blk = 14-chunk dense-block buffer,
 * wc/wl = binhip_weights_relayout outputs of conv #3 (cout_block 32) and LFF (cout_block 96); y = 6 planes.
 * store_o3 != 0 also writes o3 to blk planes 12, 13 (needed by the backward pass only).               */
BINHIP_API int binhip_rdb_tail_fwd(int N, int H, int W, int nterms, const void* blk_hi, const void* blk_lo,
                        const void* wc_hi, const void* wc_lo, const float* bias_c,
                        const void* wl_hi, const void* wl_lo, const float* bias_l,
                        void* y_hi, void* y_lo, int store_o3, void* status, void* stream);

/* ---- one whole RDN sub-network (RDN.py:210-222 / 268-280 / 322-334) ---------------------------
 * 66 convolutions launched back-to-back on `stream` from C (no Python between layers).           */
#define BINHIP_PLAN_KEEP_ACTS 1   /* training: keep every activation binhip_rdn_backward needs      */
#define BINHIP_PLAN_NO_FUSE   2   /* run conv #3 and LFF of each RDB as two kernels (A/B + tests)   */
#define BINHIP_PLAN_RDB3      4   /* nterms = 3: convs 0-2 of each RDB as three phases of ONE launch */
                                  /* (static tile ownership + per-tile neighbour flags, no grid       */
                                  /* barrier; per-conv launches when the grid cannot be co-resident)  */
#define BINHIP_PLAN_FUSED_UPNET 8 /* inference (not with KEEP_ACTS), both modes:  UPNet = conv3x3(G0 -> 256) -> PixelShuffle(2) ->   */
                                  /* conv3x3(64 -> 3) (RDN.py:203-207) has no activation in between, so it IS one linear map:     */
                                  /* a 5x5 convolution G0 -> 12 at half resolution (W_eff = W2 * shuffle * W0, 3.4 x fewer MACs,  */
                                  /* no 256-channel intermediate).  The caller supplies it: slot L = 2 + D (C + 1) + 4 of         */
                                  /* w_hi / w_lo / bias = the relayouted [12][G0][5][5] weights + 12 biases (interior form), slot  */
                                  /* L + 1: w_hi = fp32 [9][12][25][G0] border variants, bias = fp32 [9][12] (variant 3 vy + vx,   */
                                  /* v = 0 first row / column, 1 interior, 2 last): the one-pixel full-resolution border ring,     */
                                  /* where UPNet.2's zero padding of the INTERMEDIATE differs from padding the input, is           */
                                  /* recomputed exactly by a second small launch.  Same function as the two-layer form up to fp32  */
                                  /* summation order; bin_amd/rdn_plan.py builds the operands (fused_upnet_weights).               */
#define BINHIP_PLAN_FUSED_UPNET_TRAIN 16 /* with KEEP_ACTS: the fused UPNet in the TRAINING forward too (needs the slots of          */
                                        /* BINHIP_PLAN_FUSED_UPNET); its backward is BINHIP_BWD_FUSED_UPNET                          */
typedef struct BinRdnPlan {
    int32_t N, H, W;          /* full-resolution frame size (H, W even)                          */
    int32_t n_inputs;         /* 2, 3 or 5 input frames                                          */
    int32_t nterms;           /* 1 or 3                                                          */
    int32_t reserved;         /* BINHIP_PLAN_* flags                                             */
    BinRdnShape shape;        /* all zero = bin_stage4 (96, 12, 4, 32)                            */
    const void* w_hi[BINHIP_RDN_MAX_LAYERS];   /* relayouted weights per layer (first 2 + D (C + 1) + 4 used) */
    const void* w_lo[BINHIP_RDN_MAX_LAYERS];   /* NULL when nterms == 1                          */
    const float* bias[BINHIP_RDN_MAX_LAYERS];
    void* status;                          /* device uint32 status word (BINHIP_STATUS_*) or NULL */
    struct BinhipProfiler* profiler;       /* optional live kernel timing (below) or NULL         */
} BinRdnPlan;
/* The fused dense-block tail and the three-phase launch exist for the bin_stage4 shape only; other shapes run conv C-1 and
 * the LFF as two launches (what BINHIP_PLAN_NO_FUSE does for bin_stage4) — same values.                                 */

BINHIP_API size_t binhip_rdn_workspace_bytes(int N, int H, int W, int n_inputs, int nterms, const BinRdnShape* shape /* NULL = bin_stage4 */);
/* Where the activations of a call live inside its workspace (the caller owns it, and with BINHIP_PLAN_KEEP_ACTS it is
 * the saved state of the call): fp16-ELEMENT offsets from the workspace pointer rounded up to 256 B.  A tensor's hi
 * planes start at its offset, its lo planes (nterms == 3) at offset + size.  out[0..15] =
 *   P (elements of one half-resolution plane: N*H/2*W/2*16), PF (full resolution), kc0 (chunks of the packed input),
 *   x0, size | f1, size | blk, size (D + 1 blocks x B planes, B = (G0 + C G) / 16: block d input = planes [B d, B d + G0/16),
 *   conv c output = the G/16 planes from B d + (G0 + c G) / 16; bin_stage4: 13 x 14, conv c at 14d + 6 + 2c) | g0, size |
 *   g1, size | u, size (4 full-resolution planes) | has_lo.
 * Used by tools/fp16_headroom.py and the GPU tests to measure the stored dynamic range; no device work.            */
#define BINHIP_RDN_LAYOUT_WORDS 16
BINHIP_API int binhip_rdn_workspace_layout(int N, int H, int W, int n_inputs, int nterms, const BinRdnShape* shape,
                                int64_t* out, int n_out);
BINHIP_API int binhip_rdn_forward(const BinRdnPlan* plan, const float* const* inputs /* host array of
                       n_inputs device ptrs, fp32 [N,3,H,W] */, float* out /* fp32 [N,3,H,W] */,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---- weight / bias gradients (autograd of RDN.py's conv2d calls; bin_model.py:140) ---------------
 * dW[co][ci][dy][dx] = sum_px gY[co][px] * X[ci][px + tap], db[co] = sum_px gY[co][px], MFMA GEMM over the
 * pixel index with LDS transpose reads; deterministic two-stage reduction.  d describes the FORWARD conv
 * (N,H,W, ksize, cin_chunks, cout, nterms, x_cpg/x_group_stride).  The result is multiplied by
 * inv_scale[0] (device scalar, may be NULL) and written (or added, accumulate != 0) to OIHW fp32.
 * shuffle_perm != 0: gY's channels are in UPNet.0's PixelShuffle-permuted order.                      */
BINHIP_API size_t binhip_wgrad_workspace_bytes(int ksize, int N, int H, int W, int cin_chunks, int cout);
BINHIP_API int binhip_conv2d_bwd_weight(const BinConvDesc* d, const void* x_hi, const void* x_lo,
                             const void* gy_hi, const void* gy_lo, const float* inv_scale,
                             void* workspace, size_t workspace_bytes, float* dw_oihw, float* dbias,
                             int cin, int shuffle_perm, int accumulate, void* stream);

/* ---- backward glue -------------------------------------------------------------------------------- */
/* scale_out[0] = 2^floor(log2(target/amax|g|)), scale_out[1] = 1/scale_out[0] (fp16 gradient planes
 * are stored multiplied by scale_out[0]); partials: binhip_charbonnier_partials() floats.            */
BINHIP_API int binhip_grad_scale(const float* g, int64_t numel, float target, float* partials, float* scale_out,
                      void* stream);
BINHIP_API int binhip_nchw_to_planes_scaled(const float* x, int N, int C, int H, int W, const float* scale,
                                 void* y_hi, void* y_lo, void* status, void* stream);
/* inverse PixelShuffle(2) on planes: nchunks planes at 2H x 2W -> 4*nchunks planes at H x W           */
BINHIP_API int binhip_unshuffle_planes(const void* x_hi, const void* x_lo, int N, int H, int W, int nchunks,
                            void* y_hi, void* y_lo, void* stream);
/* input-frame gradients of one RDN: outs[i] = unshuffle(gX0)[frame i] * scale[1] + gout / n_images
 * (gx0 may be NULL: skip path only; outs[i] may be NULL: frame i needs no gradient)                   */
BINHIP_API int binhip_unpack_input_grads(const void* gx0_hi, const void* gx0_lo, const float* gout,
                              const float* scale, int n_images, int N, int H, int W,
                              float* const* outs, void* stream);

/* ---- ConvLSTM cell backward (autograd of RDN.py:74-82) ---------------------------------------------
 * Inputs as forward + g_h (grad of h', may be NULL) and g_c (grad of c', may be NULL); any output
 * pointer may be NULL.  dw [12,6,3,3], db [12] fp32 are overwritten.                                   */
BINHIP_API size_t binhip_convlstm_bwd_workspace_bytes(int N, int H, int W);
BINHIP_API int binhip_convlstm_bwd(const float* x, const float* c_prev, const float* h_prev, const float* w,
                        const float* b, float forget_bias, int N, int H, int W, const float* g_h,
                        const float* g_c, void* workspace, size_t workspace_bytes, float* gx,
                        float* g_hprev, float* g_cprev, float* dw, float* db, void* stream);

/* ---- backward of one whole RDN sub-network ----------------------------------------------------------
 * `saved` is the workspace binhip_rdn_forward filled (kept by the caller between forward and backward).
 * dw[i]/db[i]: OIHW fp32 gradients of layer i — overwritten, or accumulated into (`+=`, e.g. straight into the
 * parameters' .grad buffers when a weight set is shared by several calls) when `reserved` has
 * BINHIP_BWD_ACCUMULATE.  gin[i]: fp32 [N,3,H,W] or NULL.                                                */
#define BINHIP_BWD_ACCUMULATE 1
#define BINHIP_BWD_FUSED_UPNET 4  /* the forward ran UPNet as one 5x5 convolution (BINHIP_PLAN_FUSED_UPNET_TRAIN): with L = 2 + D (C + 1) + 4,
                                   * wt_hi / wt_lo[L] = binhip_weights_relayout_dgrad of the [12][G0][5][5] interior operator, wt_hi[L + 1] = the fp32
                                   * [9][12][25][G0] ring operators; results: dw[L] / db[L] = fp32 [12][G0][5][5] / [12] gradient of the interior
                                   * operator (always WRITTEN, never accumulated), dw[L + 1] / db[L + 1] = fp32 [N][9][12][25][G0] / [N][9][12]
                                   * per-image gradients of the ring operators: the caller ZERO-FILLS both, the kernel writes the entries that have ring pixels.  dw / db of UPNet.0 and UPNet.2 are NOT written:
                                   * the caller maps the operator gradients to them (autograd of rdn_plan.fused_upnet_weights).               */
#define BINHIP_BWD_SAVED_X3   2   /* `saved` has the nterms = 3 layout while this plan's nterms is 1: a single-product
                                   * backward behind the fp32-class forward (exact loss and ReLU masks, ~1e-3 relative
                                   * gradient error); wt_* are then the nterms = 1 backward-data weights              */
typedef struct BinRdnBwdPlan {
    int32_t N, H, W, n_inputs, nterms, reserved;   /* reserved = flags (BINHIP_BWD_*)                     */
    BinRdnShape shape;                      /* all zero = bin_stage4                                  */
    const void* wt_hi[BINHIP_RDN_MAX_LAYERS];   /* binhip_weights_relayout_dgrad outputs; for the slots */
                                            /* RDBs.d.convs.g: the gather-form weights of group g      */
    const void* wt_lo[BINHIP_RDN_MAX_LAYERS];
    const float* zero_bias;                 /* >= max(D G0, G0 + C G, 256) zero floats (bin_stage4: 1152) */
    float* dw[BINHIP_RDN_MAX_LAYERS];
    float* db[BINHIP_RDN_MAX_LAYERS];
    float* gin[5];
    void* status;                           /* device uint32 status word (BINHIP_STATUS_*) or NULL     */
    void* aux_stream;                       /* optional second hipStream_t: the weight-gradient kernels run on it,  */
                                            /* overlapping the backward-data chain (event-ordered inside the call;  */
                                            /* joined into `stream` before return).  NULL: everything on `stream`   */
    struct BinhipProfiler* profiler;        /* optional live timing of the weight-gradient launches (epilogue class  */
                                            /* BINHIP_PROF_WGRAD) or NULL                                            */
} BinRdnBwdPlan;
BINHIP_API size_t binhip_rdn_backward_workspace_bytes(int N, int H, int W, int n_inputs, int nterms, const BinRdnShape* shape);
/* Gradient planes inside the backward workspace after a call (same conventions as binhip_rdn_workspace_layout; all
 * stored multiplied by the call's power-of-two scale).  out[0..23] = P, PF, gx0_chunks,
 *   gout, size (1 full-res plane) | gu, size (4 full-res) | guu, size (16) | gg1, size | gg0, size | gf1, size (6 each) |
 *   gy, size ((D + 1) x G0/16: gradient of SFENet2's output and of every dense block's output) | gcat, size | gcat2, size
 *   ((G0 + C G)/16 planes each: gradient-concat buffers of the last even / odd dense block processed; gg1, gg0, gf1 have
 *   G0/16 planes) | gx0, size | byte offset of the float pair
 *   {scale, 1/scale}.                                                                                                */
#define BINHIP_RDN_BWD_LAYOUT_WORDS 24
BINHIP_API int binhip_rdn_backward_workspace_layout(int N, int H, int W, int n_inputs, int nterms, const BinRdnShape* shape,
                                         int64_t* out, int n_out);
BINHIP_API int binhip_rdn_backward(const BinRdnBwdPlan* plan, const void* saved, size_t saved_bytes,
                        const float* gout, void* workspace, size_t workspace_bytes, void* stream);

/* ---- live kernel timing (bench.py roofline leg) --------------------------------------------------
 * An explicit host-side handle: every conv launch of a plan that carries it and whose (ksize,
 * cout_pad, epilogue) matches is bracketed by a hipEvent pair recorded on the launch stream (at most
 * `max_launches` <= 16384 pairs); read() synchronises on the recorded pairs, returns their summed
 * kernel time and the launch count, and rewinds the handle.  Not thread-safe per handle (use one per
 * host thread); the library itself keeps no global timing state.
 * epilogue == BINHIP_PROF_WGRAD: time the weight-gradient partial kernels of binhip_rdn_backward whose
 * forward layer has this ksize and cout (= `cout_pad` argument), on the stream they are launched on.  */
#define BINHIP_PROF_WGRAD 16
typedef struct BinhipProfiler BinhipProfiler;
BINHIP_API int binhip_profiler_create(int ksize, int cout_pad, int epilogue, int max_launches, BinhipProfiler** out);
BINHIP_API int binhip_profiler_read(BinhipProfiler* p, double* total_ms, int* launches);
BINHIP_API void binhip_profiler_destroy(BinhipProfiler* p);

#ifdef __cplusplus
}
#endif
#endif /* BINHIP_H */
