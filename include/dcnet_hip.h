/*
 * dcnet_hip.h — C ABI of libdcnet_hip.so, the MI355X (gfx950) kernels behind DCNet's
 * dual-correspondence forward/backward hot path.
 *
 * The reference (mengcaopku/DCNet) has no native layer: its "operator API" for this
 * path is the set of torch.nn ops that model/DCNet_model.py and model/darknet.py call.
 * Each entry point below names the reference call site(s) it replaces (file:line in
 * the reference checkout).  A maintainer binds them with ctypes; our
 * own host side (dcnet_amd/) does exactly that.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to fp32 unless noted; buffers are caller-owned;
 *   - activations are NHWC ("channels-last": [N][H][W][C]); conv weights are OHWI
 *     ([Cout][kh][kw][Cin]) — the boundary kernels dcn_nchw_to_nhwc / dcn_nhwc_to_nchw
 *     and dcn_oihw_to_ohwi convert from/to the reference's NCHW / OIHW;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), does no
 *     allocation and no synchronisation, so it can be captured into a hipGraph;
 *   - return value 0 = success, negative = error (dcn_last_error() gives the text).
 */
#ifndef DCNET_HIP_H
#define DCNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCN_OK 0
#define DCN_ERR_ARG (-1)      /* bad shape / null pointer / unsupported size */
#define DCN_ERR_LAUNCH (-2)   /* hipLaunch failed */

#define DCN_AMAX_WORDS 64      /* words per abs-max vector (see dcn_absmax) */
#define DCN_SLAB_COUNTERS 4096 /* words of a split-K arrival-counter vector (see dcn_conv2d_bwd_weight) */

#define DCN_ACT_NONE 0
#define DCN_ACT_LEAKY 1       /* y = x > 0 ? x : slope*x   (slope 0 == ReLU) */

/* Bumped whenever an exported signature changes incompatibly (round 2 changed dcn_conv2d_*, dcn_scale_act, dcn_bn_act_bwd_apply,
 * dcn_l2norm_score_*, dcn_prof_collect; round 3 dcn_rmsprop_step).  dcn_version() returns the value the library was built with;
 * dcnet_amd/lib.py refuses a library whose version differs from the one its signature table was written for. */
#define DCN_ABI_VERSION 308

const char* dcn_last_error(void);
int dcn_version(void);

/* ---- layout boundary --------------------------------------------------------------- */
/* NCHW -> NHWC with the channel dim zero-padded to c_pad (>= c).  Replaces the implicit
 * layout of every nn.Conv2d input (model/darknet.py:399). */
int dcn_nchw_to_nhwc(const float* src, float* dst, int n, int c, int h, int w, int c_pad, void* stream);
/* NHWC (channel stride ld >= c) -> NCHW. */
int dcn_nhwc_to_nchw(const float* src, float* dst, int n, int c, int h, int w, int ld, void* stream);
/* OIHW -> OHWI with Cin zero-padded to ci_pad; and the adjoint (for weight gradients). */
int dcn_oihw_to_ohwi(const float* src, float* dst, int co, int ci, int kh, int kw, int ci_pad, void* stream);
int dcn_ohwi_to_oihw(const float* src, float* dst, int co, int ci, int kh, int kw, int ci_pad, void* stream);

/* ---- convolution (implicit GEMM on v_mfma_f32_32x32x2_f32) --------------------------- */
/* y[n,ho,wo,co] = act( scale[co] * (sum_{r,s,ci} x[n,ho*stride+r-pad,wo*stride+s-pad,ci] * w[co,r,s,ci])
 *                      + shift[co] ) + residual[n,ho,wo,co]
 * ksize in {1,3}, pad = (ksize-1)/2, stride in {1,2}; cin % 4 == 0 (cin % 32 == 0 unless cin == 4).
 * accumulate != 0: the current content of y is added to the raw sum first (before stats / scale / act) — the
 * fusion layer pre-fills y with its per-image language term and per-position coordinate term
 * (model/DCNet_model.py:491-499 tiles them as 520 extra input channels instead).
 * scale/shift/residual may be NULL.  y has pixel stride ldy >= cout (lets a layer write
 * straight into a channel slice of a route-concat buffer); residual has pixel stride ldr.
 * stats (optional, [grid_m][2][cout] floats, grid_m = dcn_conv2d_stats_rows(...)): partial
 * sums and sums-of-squares of the RAW conv result (before scale/shift/act) per output channel —
 * the batch statistics of train-mode BatchNorm.  All grid_m rows are written and their column sums
 * are the statistics; how the pixels are dealt to the rows is the kernel's business (one row per
 * 128/256-row tile; the persistent kernels of nconv.hip fill one row per workgroup and zero the rest).
 * Replaces nn.Conv2d (+ eval-mode BatchNorm2d + LeakyReLU/ReLU + shortcut add):
 * model/darknet.py:179-191,403-405 and ConvBatchNormReLU model/darknet.py:131-153. */
int dcn_conv2d_fwd(const float* x, const float* w, float* y,
                   int n, int h, int wd, int cin, int cout, int ksize, int stride,
                   const float* scale, const float* shift, int act, float slope,
                   const float* residual, int ldr, int ldy,
                   float* stats, int accumulate, const float* f8_scales,
                   const uint32_t* amax_x, const uint32_t* amax_w, uint32_t* amax_y, float* w_split, int w_split_ready,
                   void* stream);
/* w_split (optional scratch, cout*k*k*cin + 16 floats): with the f16 split, the filter bank is cut into its two f16 pieces
 * ONCE into this buffer (8 consecutive k -> [8 high | 8 low], same bytes) and the tiles copy it, instead of every M-tile
 * splitting the same weights again.  dcn_conv2d_bwd_data does the same in place on its `wt` scratch, which therefore
 * must hold 16 floats more than the bank when amax_w is given. */
/* amax_* (dcn_conv2d_fwd / bwd_data / bwd_weight; NULL = off): abs-max "words" of the tensors — each is a vector of
 * DCN_AMAX_WORDS (64) device words holding float bits of non-negative values whose maximum is max|tensor| (the waves of a
 * producing kernel spread their atomic maxima over the 64 words; readers take the maximum; a known bound is 64 copies).  With both
 * operand maxima given, the wide tiles run the fp32-accurate f16 two-piece split (the default "fp32" precision of this
 * library, dcn_set_tuning("precision", 4)): operands scaled by the power of two that brings the maximum below 2^14, cut
 * x = h + l into two f16 (11 + 11 significant bits), l*h + h*l + h*h on v_mfma_f32_32x32x16_f16 — three MFMAs per product
 * instead of six on the bf16 three-piece split they fall back to without maxima.  The words are maintained by the producing
 * kernels (amax_y here, dcn_scale_act, dcn_bn_act_bwd_apply) or by dcn_absmax, all with an order-independent atomic max on
 * a word the caller zeroed once. */
int dcn_absmax(const float* x, int64_t rows, int c, int ld, uint32_t* amax, void* stream);
int dcn_conv2d_stats_rows(int n, int h, int wd, int cout, int ksize, int stride);
/* f8_scales (dcn_conv2d_fwd / dcn_conv2d_bwd_data; NULL = off): device pointer to {s_activation, s_weight}, the
 * power-of-two operand scales of the builder-defined fp8 e4m3 conv path (BASELINE.json configs[4]): the wide tiles
 * round s*x to fp8, multiply with v_mfma_f32_32x32x16_fp8_fp8, accumulate in fp32 and rescale.  dcn_f8_scale writes
 * 2^floor(log2(448 / max|x|)) for a [rows][c] tensor (row stride ld) into *scale; ws = 4 bytes of device scratch. */
int dcn_f8_scale(const float* x, int64_t rows, int c, int ld, float* scale, void* ws, void* stream);

/* dx = conv_transpose(dy, w): gradient w.r.t. the conv input (autograd of nn.Conv2d,
 * reached from loss.backward() at train_DCNet.py:645).  dy NHWC (n,ho,wo,cout) with pixel
 * stride lddy, w OHWI as in the forward, wt = caller scratch of cout*k*k*cin floats
 * (the tap-flipped, channel-transposed weights are built there), dx NHWC (n,h,wd,cin).
 * accumulate != 0: dx += result (used where a tensor feeds two consumers). */
int dcn_conv2d_bwd_data(const float* dy, int lddy, const float* w, float* wt, float* dx,
                        int n, int h, int wd, int cin, int cout, int ksize, int stride,
                        int accumulate, const float* f8_scales, const uint32_t* amax_dy, const uint32_t* amax_w,
                        int wt_ready, const float* wt_split, void* stream);
/* dcn_conv2d_bwd_data with a TAP on the BatchNorm + activation in front of the convolution: dx is the gradient w.r.t.
 * act(bn(tap_y)) (+ a shortcut), so sum(g) and sum(g*xhat) per channel (g = dx*act'(.), the partial sums dcn_bn_act_bwd_reduce would
 * compute in a pass of its own over tap_y and dx) can be formed while dx is still in registers.  Only the register-bank kernels of
 * nconv.hip do it (the stride-2 layers of the 416x416 / 208x208 maps): *tap_rows > 0 on return means tap_stats[*tap_rows][2][cin] holds
 * the partials (feed them to dcn_bn_bwd_sums); 0 means the caller runs dcn_bn_act_bwd_reduce as before.  tap_stats_rows: rows the
 * buffer holds (dcn_conv2d_bwd_data_tap_rows(...) says how many a launch would need; 0 = it would not tap).  accumulate != 0: the
 * partials are those of the sum, so the caller must only tap the LAST contribution to dx.  tap_y dense (n, h, wd, cin). */
int dcn_conv2d_bwd_data_tap(const float* dy, int lddy, const float* w, float* wt, float* dx,
                            int n, int h, int wd, int cin, int cout, int ksize, int stride,
                            int accumulate, const float* f8_scales, const uint32_t* amax_dy, const uint32_t* amax_w,
                            int wt_ready, const float* wt_split,
                            const float* tap_y, const float* tap_mean, const float* tap_invstd, const float* tap_gamma,
                            const float* tap_beta, int tap_act, float tap_slope, float* tap_stats, int tap_stats_rows,
                            int* tap_rows, void* stream);
int dcn_conv2d_bwd_data_tap_rows(int n, int h, int wd, int cin, int cout, int ksize, int stride);
/* Convolution whose input x is the RAW output of the conv + train-mode BatchNorm layer in front: x' = act(pre_scale[c]*x + pre_shift[c])
 * (scale / shift of dcn_bn_finalize) is formed where the input is loaded — padding stays zero — so that layer's dcn_scale_act pass and
 * its activation tensor do not exist (the activation is recomputed from x by the weight gradient the same way).  Same result as
 * dcn_scale_act followed by dcn_conv2d_fwd(..., no epilogue) / dcn_conv2d_bwd_weight, bit for bit in the operand (the arithmetic of
 * dcn_scale_act), to summation order in the GEMM.  x dense (n,h,wd,cin); y raw outputs, pixel stride ldy; stats as dcn_conv2d_fwd.
 * amax_x: abs-max word of the ACTIVATION x' (dcn_bn_act_amax_bound gives a bound from the word of x); amax_w, amax_dy required.
 * Only the register-bank kernels do it: dcn_conv2d_pre_supported / dcn_conv2d_bwd_weight_pre_supported say whether a shape can
 * (3x3, 32 -> 64 channels, stride 1 | 2, large maps; default precision).  The two nn.Sequential blocks conv_0 / conv_1 of
 * model/yolov3.cfg as built at model/darknet.py:179-191. */
int dcn_conv2d_pre_supported(int n, int h, int wd, int cin, int cout, int ksize, int stride);
int dcn_conv2d_fwd_pre(const float* x, const float* w, float* y, int n, int h, int wd, int cin, int cout, int ksize, int stride,
                       const float* pre_scale, const float* pre_shift, int pre_act, float pre_slope,
                       int ldy, float* stats, const uint32_t* amax_x, const uint32_t* amax_w, void* stream);
int dcn_conv2d_bwd_weight_pre_supported(int n, int h, int wd, int cin, int cout, int ksize, int stride);
int dcn_conv2d_bwd_weight_pre(const float* x, int ldx, const float* dy, int lddy, float* dw, float* ws, uint32_t* counters,
                              int n, int h, int wd, int cin, int cout, int ksize, int stride,
                              const float* pre_scale, const float* pre_shift, int pre_act, float pre_slope,
                              const uint32_t* amax_x, const uint32_t* amax_dy, void* stream);
/* w_split_ready / wt_ready / wt_split: the banks were prepared for the whole network by dcn_prepare_filters (below) — w_split
 * holds the split OHWI bank, wt the transposed fp32 bank and wt_split its split form; nothing is converted per call.
 * Value 2 (bf16-operand mode, dcn_set_tuning("precision", 2)): w_split / wt_split point at the bank converted to bf16
 * (ohwi_b16 / t_b16 of dcn_prepare_filters, same [Co][k*k*Ci] layout, 2 bytes per element); the 3x3 stride-1 strip kernel
 * reads it, every other tile keeps rounding the fp32 bank itself.
 * cin == 4 (the stem): w_split is plain scratch of >= 27*32 floats — the vector-ALU stem kernel re-orders the [Co][64] bank
 * into [27][32] there; NULL keeps the stem on the implicit-GEMM tile.
 *
 * dcn_prepare_filters: every filter bank of a network in three launches (abs-max of each bank, then one LDS-tile pass that
 * writes, per job, any of: the OHWI bank, its f16-split form, the channel-transposed [Ci][T][Co] bank of the data gradient and
 * its split form; split banks carry their power-of-two scale in the float behind the bank, so they are numel + 16 floats).
 * jobs: device array of records {const float* src (OIHW); float* ohwi, *ohwi_split, *t, *t_split (NULL = not wanted);
 * uint32_t* amax; void* ohwi_b16, *t_b16 (the two banks in bf16, NULL = not wanted); int co, ci, T, blk0, ablk0, pad} of dcn_filter_job_bytes() bytes each; co and ci multiples of 32; blk0 /
 * ablk0 = ascending prefix sums of T*(co/32)*(ci/32) and ceil(co*ci*T/4096); amax_all/amax_words: the region that holds all
 * the jobs' abs-max words (zeroed here).  Replaces, per layer and step: the OIHW->OHWI transpose, dcn_absmax of the bank,
 * the pre-split and the data gradient's filter transpose (model/darknet.py:179-191 holds the parameters as OIHW). */
int dcn_filter_job_bytes(void);
int dcn_prepare_filters(const void* jobs, int njobs, int total_blocks, int total_amax_blocks,
                        uint32_t* amax_all, int64_t amax_words, void* stream);

/* Geometry table of a convolution (depends on n, h, wd, ksize, stride only; build once, reuse every step):
 * dcn_conv2d_geom_size entries of uint32, entry m = (index of the input pixel under the centre tap of
 * output pixel m) << 5 | edge flags.  The weight-gradient kernel walks K = output pixels and reads its
 * gather addresses from this table instead of dividing by Wo / Ho per pixel. */
int64_t dcn_conv2d_geom_size(int n, int h, int wd, int ksize, int stride);
int dcn_conv2d_geom(uint32_t* table, int n, int h, int wd, int ksize, int stride, void* stream);

/* dw[co,r,s,ci] = sum_{n,ho,wo} dy[n,ho,wo,co] * x[n,ho*stride+r-pad,wo*stride+s-pad,ci]
 * (OHWI).  x has pixel stride ldx.  ws = caller scratch of dcn_conv2d_bwd_weight_ws(...) floats
 * (split-K partial slabs, summed in slab order: bitwise repeatable).  geom = dcn_conv2d_geom table of this geometry.
 * counters (ABI 308; NULL = off): DCN_SLAB_COUNTERS device words, ZERO before the first call, left zero by every call, to be shared
 * only by weight-gradient calls that are ordered on one stream (the rule for ws).  With them AND dcn_set_tuning("Slabfold", KB) > 0 the
 * slabs of a launch whose tiles have at most KB of slabs each are summed by the last-arriving workgroup of each tile instead of by a
 * second launch — the same sum, bit for bit.  OFF by default (KB = 0): measured slower inside the training step (csrc/slabsum.h). */
int dcn_conv2d_bwd_weight(const float* x, int ldx, const float* dy, int lddy, float* dw, float* ws, uint32_t* counters,
                          const uint32_t* geom,
                          int n, int h, int wd, int cin, int cout, int ksize, int stride,
                          const uint32_t* amax_x, const uint32_t* amax_dy, void* stream);
int64_t dcn_conv2d_bwd_weight_ws(int n, int h, int wd, int cin, int cout, int ksize, int stride);
/* The stem's weight gradient with its BatchNorm + LeakyReLU backward applied on the fly (the stem has no data gradient, so its
 * dY is never written): x (n,h,wd,4) the padded image, y (n,h,wd,32) the raw output of dcn_conv2d_fwd, dout the gradient w.r.t.
 * act(bn(y)) with pixel stride lddo, mean / invstd of dcn_bn_finalize, sums [2][32] of dcn_bn_bwd_sums over dcn_bn_act_bwd_reduce
 * (= dbeta, dgamma), count = n*h*wd.  dw: [32][64], k = tap*4 + channel (the c4 layout dcn_conv2d_bwd_weight writes for cin == 4);
 * ws: dcn_stem_bwd_weight_bn_ws(n, h, wd) floats.  Same result as dcn_bn_act_bwd_apply followed by dcn_conv2d_bwd_weight, to
 * rounding.  Replaces autograd of nn.Conv2d(3, 32, 3) + nn.BatchNorm2d + LeakyReLU at model/darknet.py:179-191 (first block of
 * model/yolov3.cfg), reached from loss.backward() at train_DCNet.py:645. */
int dcn_stem_bwd_weight_bn(const float* x, const float* y, const float* dout, int lddo,
                           const float* mean, const float* invstd, const float* gamma, const float* beta,
                           int act, float slope, const float* sums, int64_t count,
                           int n, int h, int wd, int cout, float* dw, float* ws, void* stream);
int64_t dcn_stem_bwd_weight_bn_ws(int n, int h, int wd);
/* the same with dout as a bf16 tensor (bf16 storage, ABI 305: y stays fp32, the gradient that reaches the stem is bf16; lddo in elements) */
int dcn_stem_bwd_weight_bn_b16(const float* x, const float* y, const void* dout16, int lddo,
                               const float* mean, const float* invstd, const float* gamma, const float* beta,
                               int act, float slope, const float* sums, int64_t count,
                               int n, int h, int wd, int cout, float* dw, float* ws, void* stream);

/* ---- batched GEMM on pre-split operands (gemm3.hip) ----------------------------------- */
/* The torch.bmm products of the inter-frame co-attention (model/DCNet_model.py:449-459: affinity, the two attended features; and
 * their autograd) have two ACTIVATION operands: dcn_coattn_fwd / dcn_coattn_bwd run them through these.
 * dcn_gemm3_presplit: x [batch][rows][c] fp32 (row stride ld, batch stride bs, floats) -> its split form at dst (same shape and
 * byte size, strides ldd / bsd; dst == src allowed when the strides agree): every run of 8 consecutive elements of a row becomes
 * [8 f16 high | 8 f16 low] of s*x, s = the power of two that brings the abs-max word `amax` below 2^14 (see dcn_absmax).  c % 8 == 0.
 * dcn_gemm3: C[b] (+)= diag(row_scale[b]) . op(A[b]) . op(B[b])^T, fp32 result (three f16 MFMAs per product, as the conv engine):
 *   a_t == 0: A is [M][K] (row stride lda);  a_t != 0: A is [K][M] — likewise b_t for B ([N][K] | [K][N]).  (a_t, b_t) = (0,0) is
 *   torch.bmm(A, B.transpose(1, 2)), (0,1) torch.bmm(A, B), (1,1) torch.bmm(A.transpose(1, 2), B); (1,0) does not exist.
 *   A / B in split form with the abs-max words they were split with; operands with K along the row must hold zeros in columns
 *   [K, ceil16(K)) (lda / ldb >= ceil16(K)); rows / columns beyond a tensor read as zeros.  row_scale: optional [batch][M], batch
 *   stride rs_bs.  dcn_gemm3_supported says whether a shape runs here (large M; default precision). */
int dcn_gemm3_supported(int m, int n, int k, int batch);
int dcn_gemm3_presplit(const float* src, int ld, int64_t bs, float* dst, int ldd, int64_t bsd, int batch, int rows, int c,
                       const uint32_t* amax, void* stream);
int dcn_gemm3(const float* a, int lda, int64_t a_bs, int a_t, const float* b, int ldb, int64_t b_bs, int b_t,
              float* c, int ldc, int64_t c_bs, const float* row_scale, int64_t rs_bs,
              int m, int n, int k, int batch, int accumulate, const uint32_t* amax_a, const uint32_t* amax_b, void* stream);

/* ---- BatchNorm (train mode) + activation + shortcut ---------------------------------- */
/* Scratch (floats, 8-byte aligned) needed by dcn_bn_finalize / dcn_bn_bwd_sums for c channels. */
int64_t dcn_bn_ws(int c);
/* Reduce per-block partials [rows][2][c] (from dcn_conv2d_fwd's `stats` or dcn_channel_stats) to the
 * batch mean / biased var, update the running stats (unbiased var, momentum; pointers may be NULL)
 * and emit the per-channel affine scale = gamma*rsqrt(var+eps), shift = beta - mean*scale.
 * count = number of rows reduced per channel (n*ho*wo).  Replaces the statistics half of
 * F.batch_norm(training=True): nn.BatchNorm2d at model/darknet.py:189 (momentum 0.1) and
 * :145-147 (momentum 0.999), nn.BatchNorm1d at model/DCNet_model.py:258-259,270,274. */
int dcn_bn_finalize(const float* stats, int rows, int c, int64_t count,
                    const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var,
                    float* mean, float* invstd, float* scale, float* shift, float* ws, void* stream);
/* eval mode: scale = gamma*rsqrt(running_var+eps), shift = beta - running_mean*scale. */
int dcn_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                float eps, int c, float* scale, float* shift, void* stream);
/* Per-channel sum / sum-of-squares partials of an NHWC tensor [rows][c] (pixel stride ld) for
 * BatchNorm inputs not produced by dcn_conv2d_fwd.  stats: [dcn_channel_stats_rows(rows)][2][c]. */
int dcn_channel_stats(const float* x, int64_t rows, int c, int ld, float* stats, void* stream);
int dcn_channel_stats_rows(int64_t rows);
/* out = act(scale[c]*y + shift[c]) + residual  over [rows][c] (y, residual pixel stride c; out ldo); amax (optional):
 * running abs-max word of `out` (see dcn_absmax).
 * The normalise+LeakyReLU/ReLU(+shortcut) half of a train-mode block: model/darknet.py:189-191,403-405. */
int dcn_scale_act(const float* y, const float* scale, const float* shift, int act, float slope,
                  const float* residual, float* out, int64_t rows, int c, int ldo, uint32_t* amax, void* stream);
/* amax_out (all 64 words) = max_c(|scale[c]| * max|y| + |shift[c]|): a bound on the abs-max of act(scale*y + shift) from the abs-max
 * word of y, for consumers that form the activation themselves (dcn_conv2d_fwd_pre).  |slope| <= 1. */
int dcn_bn_act_amax_bound(const uint32_t* amax_y, const float* scale, const float* shift, int c, float slope,
                          uint32_t* amax_out, void* stream);
/* Backward of out = act(bn(y)) (a residual's gradient is the identity and handled by the caller):
 *   reduce: partials of g = dout*act'(.) and g*xhat per channel -> stats [dcn_channel_stats_rows(rows)][2][c]
 *   sums:   totals [2][c]  (sums[0] = dbeta, sums[1] = dgamma)
 *   apply:  dy = gamma*invstd*(g - sums[0]/count - xhat*sums[1]/count)                         */
int dcn_bn_act_bwd_reduce(const float* y, const float* dout, int lddo, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, int act, float slope,
                          int64_t rows, int c, float* stats, void* stream);
int dcn_bn_bwd_sums(const float* stats, int rows, int c, float* sums, float* ws, void* stream);
int dcn_bn_act_bwd_apply(const float* y, const float* dout, int lddo, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, int act, float slope,
                         const float* sums, int64_t count, int64_t rows, int c, float* dy, uint32_t* amax, void* stream);
/* Backward of out = act(z) alone: dy = dout * (out > 0 ? 1 : slope). */
int dcn_act_bwd(const float* out, const float* dout, int lddo, float slope, int64_t rows, int c, float* dy, void* stream);

/* ---- inter-frame co-attention ----------------------------------------------------------- */
/* f1,f2: [b][hw][c] NHWC (pixel stride ldf), unit L2 norm over c.  With A[i,j] = <f1_i, f2_j>:
 *   f1_attn[i,:] = sum_j softmax_j(t*A[i,j]) * f2[j,:]
 *   f2_attn[j,:] = sum_i softmax_i(t*A[i,j]) * f1[i,:]      (f2_attn may be NULL: inference model)
 * E (dcn_coattn_saved_size floats: the hw x ld_pad(hw) matrices of dcn_coattn_e_size and, behind them, the split forms of f1 and f2 where
 * the products run on gemm3.hip) receives exp(t*A - t), rinv/cinv ([b][hw]) the inverse row / column
 * sums; all three are kept for the backward.  E is OPAQUE to the caller: at the large scales (hw >= 512, c >= 256, default precision)
 * it holds the f16 two-piece split form the products on gemm3.hip read (dcn_gemm3_presplit's layout), otherwise fp32; dcn_coattn_bwd
 * decides the same way, so precision mode and the "Gemm3" knob must not change between a forward and its backward.
 * ws: dcn_coattn_fwd_ws floats of scratch.
 * Outputs have pixel stride ldo (they land in a channel slice of the concat [f, f_attn] buffer).
 * bsf / bso (and bsdo / bsdf in the backward) are the strides, in floats, between consecutive batch
 * items of the feature / output (gradient) tensors — frames of a pair are interleaved in the image
 * batch, so f1 = x[0::2], f2 = x[1::2] without a copy; <= 0 means densely packed.
 * Replaces the 3 bmm + 2 softmax of model/DCNet_model.py:449-459 / model/test_DCNet_model.py:259-274. */
int64_t dcn_coattn_e_size(int b, int hw);
int64_t dcn_coattn_saved_size(int b, int hw, int c);
int64_t dcn_coattn_fwd_ws(int b, int hw, int c);
int dcn_coattn_fwd(const float* f1, const float* f2, int ldf, int64_t bsf, float* f1_attn, float* f2_attn, int ldo,
                   int64_t bso, float* E, float* rinv, float* cinv, float* ws,
                   int b, int hw, int c, float temperature, void* stream);
/* d_f1, d_f2 (pixel stride lddf) (+)= gradient through both attention outputs; accumulate != 0 adds
 * to what the buffers hold (the direct [f, .] half of the concat contributes there as well). */
int64_t dcn_coattn_bwd_ws(int b, int hw, int c);
int dcn_coattn_bwd(const float* f1, const float* f2, int ldf, int64_t bsf,
                   const float* d_f1_attn, const float* d_f2_attn, int lddo, int64_t bsdo,
                   const float* f1_attn, const float* f2_attn, int ldo, int64_t bso,
                   const float* E, const float* rinv, const float* cinv,
                   float* d_f1, float* d_f2, int lddf, int64_t bsdf, int accumulate, float* ws,
                   int b, int hw, int c, float temperature, void* stream);

/* ---- cross-modal scoring (HBM-bound) --------------------------------------------------- */
/* One pass over x [rows][c] (pixel stride ldx): xn = x / max(||x||_2, 1e-12) over c
 * (F.normalize(dim=1), model/DCNet_model.py:359,469), out = (accumulate ? out : 0) + out_scale*xn, and, when q != NULL,
 * score[row] = <xn[row,:], q[img(row),:]> with img(row) = row / rows_per_image
 * (sim_score, model/DCNet_model.py:530-535).  score_flip (optional) = <xn[row,:], q[N-1-img(row),:]>: the caller-side
 * neg_sim_score (train_DCNet.py:623-627: flang_attn reversed along the batch) from the same pass.
 * norm [rows] keeps ||x|| for the backward.  out_scale / accumulate serve the n_frame model's mean over the T-1
 * correspondence features (model/test_DCNet_model.py:324-332). */
int dcn_l2norm_score_fwd(const float* x, int ldx, float* out, int ldo, float* norm,
                         const float* q, float* score, float* score_flip, int64_t rows, int rows_per_image, int c,
                         float out_scale, int accumulate, void* stream);
/* dx from (dout, dscore, dscore_flip) (each may be NULL); dq [n_img][c] = sum_rows dscore*out
 * + (rows of image N-1-img) dscore_flip*out (NULL to skip).  `out` is the UNSCALED normalised tensor. */
int dcn_l2norm_score_bwd(const float* out, int ldo, const float* norm, const float* dout, int lddo,
                         const float* q, const float* dscore, const float* dscore_flip, float* dx, int lddx, float* dq,
                         int64_t rows, int rows_per_image, int c, void* stream);
/* score[row] = <x[row,:], q[img(row),:]> without normalising x (flip != 0: q[N-1-img]): sim_score of the n_frame model on
 * the averaged correspondence feature (model/test_DCNet_model.py:386-391) and neg_sim_score on plain tensors. */
int dcn_rowdot_fwd(const float* x, int ldx, const float* q, int flip, float* score, int64_t rows, int rows_per_image, int c,
                   void* stream);
/* dx[row,:] = dscore[row]*q[img,:] (NULL to skip); dq[img,:] = sum over the rows scored against q[img] of dscore*x (NULL to skip). */
int dcn_rowdot_bwd(const float* x, int ldx, const float* q, int flip, const float* dscore, float* dx, int lddx, float* dq,
                   int64_t rows, int rows_per_image, int c, void* stream);

/* ---- phrase attention (language side of the cross-modal scores) --------------------------------- */
/* PhraseAttention (model/DCNet_model.py:190-219) for one or two heads sharing their inputs (sub_attn :525 and loc_attn
 * :556), with the F.normalize(p=2, dim=1) that follows each (:526,:557) when normalize != 0:
 *   attn[h][n][l] = softmax_l(<w_h, context[n,l,:]> + b_h) * (ids[n,l] != 0), renormalised;  out[h][n][:] = (normalised)
 *   sum_l attn*embedded[n,l,:].  context [n][l][d], embedded [n][l][e], ids int64 [n][l]; w1/b1 NULL = one head.
 * vnorm [heads][n] keeps ||sum|| for the backward. */
int dcn_phrase_attn_fwd(const float* context, const float* embedded, const int64_t* ids,
                        const float* w0, const float* b0, const float* w1, const float* b1,
                        float* attn, float* out, float* vnorm, int n, int l, int d, int e, int normalize, void* stream);
/* dout [heads][n][e] -> dcontext [n][l][d], dembedded [n][l][e] (sums over the heads), dw [heads*d + heads] = dw_h | db_h.
 * ws: dcn_phrase_attn_bwd_ws floats. */
int64_t dcn_phrase_attn_bwd_ws(int n, int d, int heads);
int dcn_phrase_attn_bwd(const float* context, const float* embedded, const float* w0, const float* w1,
                        const float* attn, const float* out, const float* vnorm, const float* dout,
                        float* dcontext, float* dembedded, float* dw, float* ws,
                        int n, int l, int d, int e, int normalize, void* stream);
/* out[c] = sum_r in[r*ld + c] in index order (small row counts: per-image partial sums of parameter gradients). */
int dcn_colsum(const float* in, int ld, int rows, int cols, float* out, void* stream);

/* ---- cross-scale head tail (model/DCNet_model.py:545-621 == model/test_DCNet_model.py:413-477) -------------- */
/* Arrays of three entries are per scale, coarsest first (the order of every list the reference returns); hw[s] = positions
 * of scale s, P = sum hw, Ppad = P rounded up to 32.  logits[s] is the NHWC output of fcn_out[s] (pixel stride ld[s] >= 15,
 * channels a*5+k), sim[s] [b][hw[s]] the similarity map.
 *
 * dcn_locemb_fwd: loc_embedding = Linear(8,8) + BatchNorm1d(8) + ReLU (:257-258,573) on the coordinate rows (identical
 *   for every image: coord is [p][8]; `count` = b*p is the row count of the reference's BatchNorm, used for the unbiased
 *   running variance), then F.normalize(dim=2) (:578) -> e8 [p][8].  xhat [p][8], stat [16] (mean | rstd) are kept for
 *   the backward, mom [72] (fp64) = sum_p e8[p] | sum_p e8[p] e8[p]^T feeds dcn_locbn_fwd.
 * dcn_locemb_bwd: grads [88] = dW (64) | db (8) | dgamma (8) | dbeta (8) from dE = de_a + de_b (+ the moment
 *   gradients dmom [72] of dcn_locbn_bwd); any of the three may be NULL. */
int dcn_locemb_fwd(const float* coord, const float* w, const float* b, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps, int training, int64_t count,
                   int p, float* e8, float* xhat, float* stat, double* mom, void* stream);
int dcn_locemb_bwd(const float* coord, const float* w, const float* gamma, const float* beta, const float* xhat,
                   const float* stat, const float* de_a, const float* de_b, const double* dmom, int training, int p,
                   float* grads, void* stream);
/* only_obj[s] = mean over the 3 anchors of the confidence logit (:551); obj = only_obj*sim (:550); obj_map [b][P] =
 * F.normalize(obj, dim=1) (:569), objn [b] its norm; x [(b*8)][Ppad] = e8[p,k]*obj_map[n,p], zero in the pad columns:
 * the left operand of  M[(n,k)][c] = sum_p x*W[c][p]  (dcn_gemm_nt against the Ppad-padded loc_text_embedding weight),
 * the rank-8 form of bmm(P x P) + Linear(P,512) (:581-585). */
int dcn_head_obj(const float* const* logits, const int* ld, const float* const* sim, const int* hw, const float* e8,
                 float* const* only_obj, float* obj_map, float* objn, float* x, int b, void* stream);
/* dst[r][0:cols) = src[r][0:cols), dst[r][cols:cols_out) = 0 for rows of any alignment (row strides lds / ldd): the
 * (512, P) <-> (512, Ppad) copies of the loc_text_embedding weight and its gradient. */
int dcn_pad_rows(const float* src, int lds, float* dst, int ldd, int rows, int cols, int cols_out, void* stream);
/* BatchNorm1d(512) of loc_text_embedding (:259,585) over the b*P rows  rel = e8_i . M[n] + bias  from the moments of e8
 * (fp64; no (b,P,512) tensor): mp = M*scale, bp = bias*scale + shift; train mode updates the running statistics
 * (count = b*P).  saved [3][512] (rstd | mean-without-bias | scale) for the backward. */
int dcn_locbn_fwd(const float* m, const double* mom, const float* bias, const float* gamma, const float* beta,
                  float* running_mean, float* running_var, float momentum, float eps, int training, int batch,
                  int64_t count, int c, float* mp, float* bp, float* saved, void* stream);
/* dmp [b][8][512] (image stride dmp_bs floats; <= 0: dense), dbp [512] -> dm, out [3][512] = dgamma | dbeta | dbias, dmom [72] (train mode; feed to dcn_locemb_bwd). */
int dcn_locbn_bwd(const float* m, const double* mom, const float* bias, const float* gamma, const float* running_mean,
                  const float* saved, const float* dmp, int64_t dmp_bs, const float* dbp, int training, int batch,
                  int64_t count, int c, float* dm, float* out, double* dmom, void* stream);
/* loc_score = (x - min)/(max - min + 1e-6) per image over the P positions of loc_map [b][P] (:597), split per scale
 * (:604-610); outbox[s] (NCHW [b][15][hw]) = logits with every confidence channel multiplied by sim*loc (:612-621).
 * minmax [b][4] keeps min, max and their positions (as int bits) for the backward. */
int dcn_head_final_fwd(const float* const* logits, const int* ld, const float* const* sim, const int* hw,
                       const float* loc_map, float* const* outbox, float* const* loc_score, float* minmax, int b, void* stream);
/* d(loc_map) [b][P] from d(outbox) / d(loc_score) (entries may be NULL). */
int dcn_head_dloc(const float* const* logits, const int* ld, const float* const* sim, const int* hw,
                  const float* const* loc_score, const float* const* d_outbox, const float* const* d_loc,
                  const float* minmax, float* dloc_map, int b, void* stream);
/* dx [(b*8)][ppad] (gradient of dcn_head_obj's x) -> dobj_map [b][p] and the e8 gradient de8 [p][8]. */
int dcn_head_fold(const float* dx, const float* e8, const float* obj_map, int b, int p, int ppad, float* dobj_map,
                  float* de8, void* stream);
/* d(logits) (NHWC, pad channels zeroed) and d(sim) from d(outbox), d(only_obj) (entries may be NULL) and dobj_map (may be NULL). */
int dcn_head_dlogits(const float* const* logits, const int* ld, const float* const* sim, const int* hw,
                     const float* const* loc_score, const float* const* only_obj, const float* const* d_outbox,
                     const float* const* d_only, const float* obj_map, const float* objn, const float* dobj_map,
                     float* const* dlogits, float* const* dsim, int b, void* stream);

/* ---- correspondence sampling heads (scale 0) ------------------------------------------------------------- */
/* K9, inter-frame (model/DCNet_model.py:381-430).  cmap [pairs][hw*hw] = <frame-1 position i, frame-2 position j> at
 * [i*hw + j] (:390, a batched dcn_gemm_nt), fv [2*pairs][hw][e] the normalised scale-0 features (frames of a pair adjacent),
 * raw_neg [pairs][top_k][neg_n] the list positions drawn by dcn_mt_sample_interframe(kpos = NULL).
 * index [pairs][top_k]: torch.topk(largest, sorted) of the row, ties by lowest flat index (:395); frame / corr
 * [pairs][top_k][e] = fv rows at index / hw (:407) and index % hw (:409); neg_idx, negf [pairs][top_k][neg_n]([e]) the
 * negatives after skipping the matched position (:411-418). */
int dcn_k9_fwd(const float* cmap, const float* fv, const int64_t* raw_neg, int pairs, int hw, int e, int top_k, int neg_n,
               int64_t* index, int64_t* neg_idx, float* frame, float* corr, float* negf, void* stream);
/* dfv [2*pairs][hw][e] = scatter of the three gradients (every row written; deterministic, no atomics). */
int dcn_k9_bwd(const int64_t* index, const int64_t* neg_idx, const float* d_frame, const float* d_corr, const float* d_neg,
               int pairs, int hw, int e, int top_k, int neg_n, float* dfv, void* stream);
/* K14, cross-modal (model/DCNet_model.py:625-637, Crossmodal_corrspondence :41-112):
 *   colnorm  vit = F.normalize(fvisu[0].flatten(-2), dim=2): over the POSITIONS of each channel (:629); v, vit [n][hw][e]
 *   lagnorm  lag = F.normalize(F.interpolate(context, 0.5), dim=1): even channels, over the WORDS (:631-632); lag [n][l][d/2]
 *   crossmap cols [n][hw] = arg-max over words of Conv1d(l,l,3,pad 1)(lag . vit^T) (:634-635,:48; the Softmax is monotone);
 *            lvmap [n][l][hw] (optional) receives the conv output
 *   gather   lag_pos [n][hw][e] = lag[n][cols] (:66-70), neg_cross [n][hw][neg_n][e] = vit[n-1 (LAST image)][neg] (:75-96)
 *   backward negscatter: extra [hw][e] = sum of d(neg_cross) rows per position of the last image (CSR from
 *            dcn_mt_sample_crossmodal_csr); colnorm_bwd: dv from dq (+ extra on image n_extra); dlag: d(context). */
int dcn_colnorm_fwd(const float* v, int n, int hw, int e, float* vit, float* cnorm, void* stream);
int dcn_colnorm_bwd(const float* vit, const float* cnorm, const float* dq, const float* extra, int n_extra,
                    int n, int hw, int e, float* dv, void* stream);
int dcn_lagnorm_fwd(const float* context, int n, int l, int d, float* lag, float* lnorm, void* stream);
int dcn_crossmap(const float* lag, const float* vit, const float* conv_w, const float* conv_b, int n, int l, int hw, int e,
                 int64_t* cols, float* lvmap, void* stream);
int dcn_k14_gather(const float* lag, const float* vit, const int64_t* cols, const int64_t* neg, int n, int l, int hw, int e,
                   int neg_n, float* lag_pos, float* neg_cross, void* stream);
int dcn_k14_negscatter(const float* d_neg, const int* csr_off, const int* csr_src, int hw, int e, float* extra, void* stream);
int dcn_k14_dlag(const float* lag, const float* lnorm, const int64_t* cols, const float* d_k, int n, int l, int hw, int e,
                 float* dcontext, void* stream);

/* ---- caller-side losses, targets and decode (train_DCNet.py) -------------------------------------------------- */
/* anchors [3][3][2]: the reference's reversed anchor table (train_DCNet.py:404-406) divided by (anchor_imsize / grid)
 * per scale, computed in double and rounded to float on the host (as the reference's Python does, :293-296,789-793).
 * build_target (:265-332), compact: target_i [n][4] = best anchor (0..8), gi, gj, cell index in the concatenated [P] axis;
 * target_f [n][4] = tx, ty, tw, th.  bbox [n][4] xyxy pixels (clamped to [0, size-1] here, :605). */
int dcn_build_target(const float* bbox, const float* anchors, int size, int n, int* target_i, float* target_f, void* stream);
/* the reference's dense tensors (pre-zeroed by the caller): bbox_list[s] [n][3][5][g][g], center_list[s] [n][5][g][g] (:322-323) */
int dcn_target_dense(const int* target_i, const float* target_f, int size, int n, float* const* bbox_list,
                     float* const* center_list, void* stream);
/* out [3] = yolo_loss (:45-72), rank_loss (:173-203, margin 0.1), loc_loss (:205-220) on outbox[s] [n][15][g][g] (NCHW),
 * sim[s] / negsim[s] / loc[s] [n][g][g].  vals [n][8], lse [n][2]: per-sample terms / log-sum-exps kept for the backward. */
int dcn_dense_loss_fwd(const float* const* outbox, const float* const* sim, const float* const* negsim, const float* const* loc,
                       const int* target_i, const float* target_f, int size, int n, float* vals, float* lse, float* out,
                       void* stream);
/* grad_out [3] = upstream gradients of the three losses (device); every element of the twelve outputs is written. */
int dcn_dense_loss_bwd(const float* const* outbox, const float* const* sim, const float* const* negsim, const float* const* loc,
                       const int* target_i, const float* target_f, const float* lse, const float* grad_out, int size, int n,
                       float* const* d_outbox, float* const* d_sim, float* const* d_negsim, float* const* d_loc, void* stream);
/* InfoNCE over `rows` rows of (q [e], pos [e], neg [m][e]): mean over rows of CE([cos(q,pos), cos(q,neg_*)]/T, 0) —
 * Interframe_contrastive_loss (:114-136) and Crossmodal_constrastive_loss with one positive per row (:140-166).
 * loss_rows [rows] scratch, loss [1]. */
int dcn_contrastive_fwd(const float* q, const float* pos, const float* neg, int64_t rows, int e, int m, float temperature,
                        float* loss_rows, float* loss, void* stream);
int dcn_contrastive_bwd(const float* q, const float* pos, const float* neg, int64_t rows, int e, int m, float temperature,
                        const float* grad_out, float* dq, float* dpos, float* dneg, void* stream);
/* evaluation decode (:764-810): per image the global arg-max of the 3 x 3 x g x g confidences (first maximum), box =
 * (sigmoid(tx)+gi, sigmoid(ty)+gj, exp(tw)*aw, exp(th)*ah)*stride as xyxy.  cellinfo [n][3] (optional) = anchor, gi, gj. */
int dcn_decode_boxes(const float* const* outbox, const float* anchors, int size, int n, float* boxes, int* cellinfo, void* stream);
/* utils/utils.py:76-104 (x1y1x2y2) row by row. */
int dcn_box_iou(const float* box1, const float* box2, int n, float* iou, void* stream);

/* ---- bf16 storage (BASELINE.json configs[2], ABI 305) -------------------------------------------------------------------
 * The conv stack on tensors that ARE bf16 in HBM: activations, raw conv outputs and their gradients are bf16 NHWC tensors,
 * filter banks the bf16 forms dcn_prepare_filters writes ([Cout][k*k*Cin] for the forward, [Cin][k*k*Cout] for the data
 * gradient); accumulators, BatchNorm statistics, weight gradients and master weights stay fp32.  Replaces, like their fp32
 * namesakes, nn.Conv2d / nn.BatchNorm2d / nn.LeakyReLU / the shortcut add / MyUpsample2 + route concat of
 * model/darknet.py:158-191,400-405 and their autograd (train_DCNet.py:645).  cin, cout multiples of 32 (the 3-channel stem
 * keeps fp32 kernels; dcn_scale_act_b16 with y_f32 = 1 turns its raw output into the first bf16 activation).  `*_f32` flags:
 * that tensor is fp32 instead (the boundary to fp32 consumers).  The reference has no bf16 semantics: parity
 * is defined against the exact model — the same arithmetic on the bf16 values — in tests/test_b16_gpu.py. */
int dcn_conv2d_stats_rows_b16(int n, int h, int wd, int cout, int ksize, int stride);
/* y = epilogue(conv(x, w16)): scale / shift / act / residual (bf16, pixel stride ldr) / accumulate as dcn_conv2d_fwd; stats
 * [dcn_conv2d_stats_rows_b16][2][cout] = sum / sum of squares of the raw result AS STORED (rounded to bf16 first). */
int dcn_conv2d_fwd_b16(const void* x, const void* w16, void* y, int y_f32, int n, int h, int wd, int cin, int cout, int ksize, int stride,
                       const float* scale, const float* shift, int act, float slope, const void* residual, int ldr, int ldy,
                       float* stats, int accumulate, void* stream);
/* dx (+)= conv^T(dy, wt16) (stride 1: flipped taps; stride 2: four parity classes).  tap_* (stride 1, optional): the BatchNorm +
 * activation whose output this convolution read, y in bf16 — tap_stats receives the partial sums dcn_bn_act_bwd_reduce_b16 would
 * form from the finished dx, *tap_rows the rows written (0: no tap taken). */
int dcn_conv2d_bwd_data_b16(const void* dy, int lddy, const void* wt16, void* dx, int dx_f32, int n, int h, int wd, int cin, int cout,
                            int ksize, int stride, int accumulate, const void* tap_y, const float* tap_mean, const float* tap_invstd,
                            const float* tap_gamma, const float* tap_beta, int tap_act, float tap_slope, float* tap_stats,
                            int tap_stats_rows, int* tap_rows, void* stream);
/* dw [cout][k][k][cin] fp32 = sum over pixels of dy (x) x, both bf16; ws: dcn_conv2d_bwd_weight_ws_b16 floats (split-K slabs,
 * summed in a fixed order); counters as dcn_conv2d_bwd_weight; geom: the table of dcn_conv2d_geom for this geometry. */
int64_t dcn_conv2d_bwd_weight_ws_b16(int n, int h, int wd, int cin, int cout, int ksize, int stride);
int dcn_conv2d_bwd_weight_b16(const void* x, int ldx, const void* dy, int lddy, float* dw, float* ws, uint32_t* counters, const uint32_t* geom,
                              int n, int h, int wd, int cin, int cout, int ksize, int stride, void* stream);
/* out (bf16, or fp32 with out_f32; pixel stride ldo) = act(scale * y + shift) + residual (bf16): BatchNorm apply + LeakyReLU + shortcut. */
int dcn_scale_act_b16(const void* y, int y_f32, const float* scale, const float* shift, int act, float slope, const void* residual,
                      int ldr, void* out, int out_f32, int64_t rows, int c, int ldo, void* stream);
/* BatchNorm + activation backward on bf16 y / dout / dy: the reduce pass (stats [dcn_bn_act_bwd_reduce_rows_b16(rows)][2][c], to be
 * summed by dcn_bn_bwd_sums) and the apply pass, arithmetic of dcn_bn_act_bwd_reduce / _apply term by term.  dout_f32: the incoming
 * gradient is an fp32 tensor (a head block whose consumer is an fp32 kernel). */
int dcn_bn_act_bwd_reduce_rows_b16(int64_t rows);
int dcn_bn_act_bwd_reduce_b16(const void* y, int y_f32, const void* dout, int dout_f32, int lddo, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, int act, float slope, int64_t rows, int c, float* stats,
                              void* stream);
int dcn_bn_act_bwd_apply_b16(const void* y, int y_f32, const void* dout, int dout_f32, int lddo, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, int act, float slope, const float* sums, int64_t count,
                             int64_t rows, int c, void* dy, void* stream);
/* dst[r][:c] (+)= src[r][:c], element types by flag (0 fp32, 1 bf16), element strides lds / ldd: casts at the fp32 boundary,
 * channel-slice copies of the route concat and their accumulating backward. */
int dcn_cast_rows(const void* src, int src_b16, int lds, void* dst, int dst_b16, int ldd, int64_t rows, int c, int accumulate,
                  void* stream);
int dcn_upsample2_nhwc_b16(const void* src, int lds, void* dst, int ldd, int n, int h, int w, int c, void* stream);
int dcn_upsample2_nhwc_bwd_b16(const void* ddst, int ldd, void* dsrc, int lds, int n, int h, int w, int c, int accumulate,
                               void* stream);

/* ---- fp8 storage (BASELINE.json configs[4], ABI 307) ----------------------------------------------------------------------
 * Forward and data gradient of the convolutions on OCP e4m3 operands that ARE 1-byte tensors in HBM, each row carrying ONE e8m0 scale
 * (the OCP MX scale, with the block = the row: a pixel's channel vector for activations and gradients, a filter's k*k*Cin coefficients
 * for the banks), multiplied by v_mfma_scale_f32_32x32x64_f8f6f4 (2 x the bf16 matrix rate, half the staged bytes per product), fp32
 * accumulate; results, shortcut, BatchNorm tap and statistics as in the bf16-storage functions above (bf16 tensors, fp32 sums).  Replaces
 * nn.Conv2d of model/darknet.py:172-191 and its input gradient under train_DCNet.py:645 for layers with cin % 64 == 0; the weight
 * gradient stays on bf16 tensors.  The reference has no fp8 semantics: parity is defined against the exact model —
 * the same convolution in fp64 on the dequantised operands — in tests/test_f8_gpu.py. */
/* q[r][:c] = e4m3(x[r][:c] * 2^-e_r) (round to nearest even, clamped to +-448), scales[r] = e_r + 127 with e_r = floor(log2(max|x[r]|)) - 8
 * (all-zero row: 127); x bf16, element strides ld / ldq, c % 8 == 0. */
int dcn_quant_rows_e4m3(const void* x, int ld, int64_t rows, int c, void* q, int ldq, void* scales, void* stream);
/* ABI 308: the same for MANY tensors in one launch (the filter banks of a step, once per refresh): jobs = device array of njobs records of
 * dcn_quant_job_bytes() bytes {const bf16* src [rows][c] dense; uint8* q; uint8* scales; int rows, c, first_block, 0}, first_block = running
 * sum of ceil(rows / 4), blocks = its total, elements = sum of rows * c.  Bitwise the result of dcn_quant_rows_e4m3 per tensor. */
int dcn_quant_job_bytes(void);
int dcn_quant_rows_e4m3_batched(const void* jobs, int njobs, int blocks, int64_t elements, void* stream);
/* The same bytes written by the pass that writes the bf16 tensor (c / 8 a power of two <= 64: dcn_quant_fusable): dcn_scale_act_b16 /
 * dcn_bn_act_bwd_apply_b16 on dense bf16 tensors, plus q8 [rows][c] and qs [rows] of their result. */
int dcn_quant_fusable(int c);
int dcn_scale_act_b16_q(const void* y, const float* scale, const float* shift, int act, float slope, const void* residual, int ldr, void* out,
                        int64_t rows, int c, void* q8, void* qs, void* stream);
int dcn_bn_act_bwd_apply_b16_q(const void* y, const void* dout, int lddo, const float* mean, const float* invstd, const float* gamma,
                               const float* beta, int act, float slope, const float* sums, int64_t count, int64_t rows, int c, void* dy,
                               void* q8, void* qs, void* stream);
int dcn_conv2d_stats_rows_f8(int n, int h, int wd, int cout, int ksize, int stride);
/* y (bf16 | fp32) = epilogue(conv(x8 * 2^(xs - 127), w8 * 2^(ws - 127))): x8 [n][h][wd][cin] e4m3, xs [n*h*wd] e8m0, w8 [cout][k*k*cin] e4m3,
 * ws [cout] e8m0; epilogue and stats as dcn_conv2d_fwd_b16 (rows: dcn_conv2d_stats_rows_f8). */
int dcn_conv2d_fwd_f8(const void* x8, const void* xs, const void* w8, const void* ws, void* y, int y_f32, int n, int h, int wd, int cin, int cout,
                      int ksize, int stride, const float* scale, const float* shift, int act, float slope, const void* residual, int ldr,
                      int ldy, float* stats, int accumulate, void* stream);
/* dx (bf16 | fp32) (+)= conv^T(dy8, wt8): dy8 [n][ho][wo][cout] e4m3 (dense) with dys per pixel, wt8 [cin][k*k*cout] e4m3 with wts per row;
 * tap_* as dcn_conv2d_bwd_data_b16 (y bf16). */
int dcn_conv2d_bwd_data_f8(const void* dy8, const void* dys, const void* wt8, const void* wts, void* dx, int dx_f32, int n, int h, int wd, int cin,
                           int cout, int ksize, int stride, int accumulate, const void* tap_y, const float* tap_mean, const float* tap_invstd,
                           const float* tap_gamma, const float* tap_beta, int tap_act, float tap_slope, float* tap_stats, int tap_stats_rows,
                           int* tap_rows, void* stream);

/* ---- top-k candidate cache + temporal post-processing of the inference path (ABI 305) ----------------------------- */
/* test_DCNet.py:587-643,662-705 (save_cache / get_topk_pred_bbox): per clip b of n, the top_k (<= 64) largest modulated
 * confidences over 3 scales x 3 anchors x g x g of outbox[s] [n][15][g][g] (contiguous; grids[s] = size / (32 >> s)), sorted
 * by value (equal values: lowest flat index first, the reference's "first exact match" :684); boxes [n][top_k][4] = the cell's
 * box (sigmoid / exp decode with anchors [3][3][2] as for dcn_decode_boxes, :670-681) un-letterboxed with ratio / dw / dh [n]
 * and clamped to [0, frame_hw[b] = (height, width)] (:615-633); scores [n][top_k]; feats [n][top_k][e] = the correspondence
 * feature of the winning cell, read from feat[s] through feat_strides[s][4] = (batch, channel, row, column) strides in floats
 * (the model hands out NHWC memory as a logical NCHW view); cells [n][top_k][4] = (scale, anchor, gj, gi).  Radix select,
 * no sort, no host synchronisation. */
int dcn_post_topk(const float* const* outbox, const float* const* feat, const int64_t* feat_strides, const int* grids,
                  const float* anchors, int size, int n, int e, int top_k, const float* ratio, const float* dw, const float* dh,
                  const int64_t* frame_hw, float* boxes, float* scores, float* feats, int64_t* cells, void* stream);
/* post_processing.py:246-278 for n windows at once: center [n][k][e], ref [n][r][k][e] (every frame of the window, centre
 * included), ref_score [n][r][k], valid [n][r] bytes or NULL (0 = neighbour missing: its weight is zeroed AFTER the softmax,
 * :266-269).  fused [n][k] = sum_r softmax_r(max_i <c, ref_r,i>) * ref_score[r][argmax_i], best [n] = first arg-max of fused. */
int dcn_post_fusion(const float* center, const float* ref, const float* ref_score, const unsigned char* valid, int n, int k, int r,
                    int e, float* fused, int64_t* best, void* stream);

/* ---- location module core (rank-8 form of model/DCNet_model.py:581-594) ------------------------- */
/* loc[n,i] = < normalize_c( relu( E[i,:8] . Mp[n,:8,:c] + bp[:c] ) ), q[n,:c] >   for i < p, c == 512.
 * Mp/bp already carry the BatchNorm1d affine (train-mode statistics follow from the 8x8 moments of E on the
 * host side).  Replaces bmm (P x P) + Linear(P,512) + BatchNorm1d + ReLU + F.normalize + the phrase dot. */
int dcn_locmod_fwd(const float* E, const float* Mp, const float* bp, const float* q, float* loc,
                   int n, int p, int c, void* stream);
/* Gradients for dloc [n][p]: dE_part [n][p][8] (sum over n = dE), dsum [n][10][c] with rows 0-7 = dMp[n],
 * row 8 = per-image part of dbp (sum over n), row 9 = dq[n].  ws: dcn_locmod_bwd_ws(n, p) floats. */
int64_t dcn_locmod_bwd_ws(int n, int p);
int dcn_locmod_bwd(const float* E, const float* Mp, const float* bp, const float* q, const float* dloc,
                   float* dE_part, float* dsum, float* ws, int n, int p, int c, void* stream);

/* ---- plain GEMMs on the conv engines and the LSTM cell (language branch) -------------------- */
/* C[M][N] (+)= act(A[M][K].B[N][K]^T + bias[N]) + residual     nn.Linear (model/DCNet_model.py:131,194,269,273)
 * and the BiLSTM input / recurrent projections (:134-137).  K % 32 == 0.  act: DCN_ACT_* with slope 0. */
int dcn_gemm_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                const float* bias, int act, const float* residual, int ldr, int accumulate, void* stream);
/* C[M][N] (+)= A[M][K].B[K][N] (rows k >= kvalid of B read as 0): input gradient of a Linear layer. */
int dcn_gemm_nn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, int kvalid,
                int accumulate, void* stream);
/* C[M][N] (+)= A[K][M]^T.B[K][N]: weight gradient of a Linear layer. */
int dcn_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                int accumulate, void* stream);
/* One LSTM step for n rows, gate order i,f,g,o like nn.LSTM: gates [n][4H] -> (c_out, h_out), y[n][ldy] = h
 * (0 and state carried over for rows with t >= lens[r]; lens may be NULL), act [n][5H] saved for the backward. */
int dcn_lstm_cell_fwd(const float* gates, const float* c_prev, const float* h_prev, const int64_t* lens, int t,
                      float* act, float* c_out, float* h_out, float* y, int ldy, int n, int hidden, void* stream);
/* Backward of one step: dgates [n][4H] (row stride ldg), dc_prev, dh_pass (gradient that bypasses a finished row). */
int dcn_lstm_cell_bwd(const float* dy, int lddy, const float* dh_rec, const float* dc_next, const float* act,
                      const float* c_prev, const int64_t* lens, int t, float* dgates, int ldg, float* dc_prev,
                      float* dh_pass, int n, int hidden, void* stream);

/* The whole recurrence of nn.LSTM(512, 512, 1, batch_first=True, bidirectional=True) over a packed batch
 * (model/DCNet_model.py:134-137,172-183) as ONE persistent launch per pass (csrc/lstm.hip): 64 workgroups per direction
 * keep their slice of W_hh in LDS and hand h_t to each other through global memory with agent-scope release/acquire.
 * xg [2][n][l][4H] = x.W_ih^T + b_ih per direction (dcn_gemm_nt); whh_* [4H][H], bhh_* [4H]; lens [n] int64 or NULL.
 * out [n][l][2H] (zeros where t >= len); hprev / cprev [2][n][l][H] = state BEFORE time t (hprev must arrive zeroed),
 * acts [2][n][l][5H] = i,f,g,o,tanh(c): kept for the backward.  sync: dcn_bilstm_sync_bytes() bytes of device memory,
 * zeroed on the stream by the call itself; word 8 is set if a bounded spin gave up.  hidden must be 512, n <= 512. */
int64_t dcn_bilstm_sync_bytes(void);
int dcn_bilstm_fwd(const float* xg, const float* whh_fwd, const float* whh_rev, const float* bhh_fwd, const float* bhh_rev,
                   const int64_t* lens, float* out, float* hprev, float* cprev, float* acts, void* sync,
                   int n, int l, int hidden, void* stream);
/* dout [n][l][2H] -> dxg [2][n][l][4H], the gradient of the gate pre-activations of every (row, time): the weight, input
 * and bias gradients are dcn_gemm_tn / dcn_gemm_nn / column sums over its n*l rows. */
int dcn_bilstm_bwd(const float* dout, const float* whh_fwd, const float* whh_rev, const float* acts, const float* cprev,
                   const int64_t* lens, float* dxg, void* sync, int n, int l, int hidden, void* stream);

/* ---- small data movers ------------------------------------------------------------------ */
/* nearest x2 upsample of NHWC src (n,h,w,c) into dst (n,2h,2w,·) pixel stride ldd
 * (MyUpsample2, model/darknet.py:158-160; fused with the route concat :400-402). */
int dcn_upsample2_nhwc(const float* src, int lds, float* dst, int ldd, int n, int h, int w, int c, void* stream);
/* adjoint: dsrc[n,y,x,c] (+)= sum of the 4 dst pixels. */
int dcn_upsample2_nhwc_bwd(const float* ddst, int ldd, float* dsrc, int lds, int n, int h, int w, int c,
                           int accumulate, void* stream);
/* strided copy / accumulate of a [rows][c] channel slice: dst (+)= src. */
int dcn_copy_slice(const float* src, int lds, float* dst, int ldd, int64_t rows, int c, int accumulate, void* stream);

/* Experiment knobs (in-process A/B runs: tools/bench_convs.py --ab, bench.py --schedule-tunes, the DCN_TUNE environment variable).
 * The key's first characters select the knob; the ones that switch a kernel family off and on (1 = default):
 *   "precision" 0..4 (see above) | "1x1dma" conv1.hip | "3x3strip" conv3.hip | "Nconv" nconv.hip (0 off, 2: 32 -> 64 stride-2 data
 *   gradient only, 3: dgrad2 only) | "9tap" wgrad9.hip | "u3row" wgrad3.hip | "jstem" the stem's direct forward | "merge" parity
 *   classes of a stride-2 data gradient in one launch | "bm" force the M tile (64 / 128, 0 = automatic).
 * Sizing knobs: "1stages", "9target", "v3target", "xwgtarget", "zwgsmall", "e2rpw" / "f2nt" (scoring pass), "dbnrev" (sweep
 * direction of the BatchNorm passes).  Unknown keys are ignored by design of the A/B tools; none changes results beyond rounding. */
int dcn_set_tuning(const char* key, int value);

/* ---- optimiser ---------------------------------------------------------------------------------
 * One fused RMSprop step over `count` fp32 tensors (host arrays of device pointers and element counts):
 * g += weight_decay*p;  v = alpha*v + (1-alpha)*g*g;  p -= lr*g/(sqrt(v)+eps)   — torch.optim.RMSprop with
 * momentum = 0, centered = False (train_DCNet.py:528-534), which the reference steps at train_DCNet.py:646. */
int dcn_rmsprop_step(float* const* params, const float* const* grads, float* const* square_avgs, const int64_t* numel,
                     int count, float lr, const float* lr_dev /* device scalar that overrides lr when non-NULL: a step captured into a
                     hipGraph then follows the caller's learning-rate schedule (train_DCNet.py:244-253) without re-capture */,
                     float alpha, float eps, float weight_decay, void* stream);
/* Keys: "precision" 4 (default): the wide tiles of the conv engine and of the weight-gradient GEMM run the f16 two-piece
 *         split (see dcn_absmax) wherever both operands carry their abs-max word, and as 1 otherwise;
 *         1: the bf16 matrix pipe with every fp32 operand cut into three bf16 pieces (exact) and the six cross terms
 *         >= 2^-16 accumulated in fp32 — measured error against fp64 is at or below that of v_mfma_f32_32x32x2_f32;
 *         0: v_mfma_f32_32x32x2_f32 everywhere; 2: bf16 operands; 3: fp8 operands (dcn_f8_scale).
 *       "bm", "k": tile / K-step overrides; "split" 16|32, "wsplit" 1: force the split pipe on every NT / TN tile. */

/* ---- the fusion layer's constant terms, and small language-branch ops ------------------------------------------ */
/* The first fcn_emb convolution (model/DCNet_model.py:491-505) convolves [corr | tile(flang) | coord] with W = [W1|W2|W3]; the
 * last two groups are constant over positions / images, so conv = W1.corr[n,p] + W2.flang[n] + W3.coord[p].
 * dcn_fusion_prefill: out[n][p][c] = A[n][c] + sum_k coord[p][k] * w3[c*ldw + k]  (A = flang.W2^T, e.g. from dcn_gemm_nt; w3 = the
 *   last 8 columns of the layer's [co][2e+8] filter matrix, ldw its row stride); dcn_conv2d_fwd then accumulates W1.corr onto it.
 * dcn_fusion_bwd: from dy[n][hw][co] (the gradient of that convolution's raw output), in one pass: d_img[n][c] = sum_p dy,
 *   dw3[c*ldd + k] = sum_{n,p} dy[n][p][c] * coord[p][k], then dw2[c*ldd + j] = sum_n d_img[n][c] * flang[n][j]; ws = dcn_fusion_bwd_ws
 *   floats.  Fixed summation order (no atomics).  These replace five torch.matmul and two .sum() of the round-2 host code. */
int dcn_fusion_prefill(const float* A, const float* coord, const float* w3, int ldw, float* out, int n, int hw, int co, void* stream);
int64_t dcn_fusion_bwd_ws(int n, int co);
int dcn_fusion_bwd(const float* dy, const float* coord, const float* flang, float* ws, float* d_img, float* dw2, float* dw3,
                   int ldd, int n, int hw, int co, int e, void* stream);
/* lengths[r] = #{l : ids[r][l] != 0}  (model/DCNet_model.py:150 `(input_labels != 0).sum(1)`). */
int dcn_row_lengths(const int64_t* ids, int n, int L, int64_t* out, void* stream);
/* nn.Embedding forward (row gather) and its backward as a deterministic per-vocabulary-row sum (model/DCNet_model.py:168; torch's
 * backward sorts the indices). */
int dcn_embedding_fwd(const int64_t* ids, const float* table, float* out, int tokens, int e, int vocab, void* stream);
int dcn_embedding_bwd(const int64_t* ids, const float* dout, float* dtable, int tokens, int e, int vocab, void* stream);

/* ---- streams with a dispatch priority ------------------------------------------------------- */
/* level -1 / 0 / +1 = highest / normal / lowest priority of the device.  Returns a hipStream_t (NULL on
 * error).  The host side runs the weight-gradient GEMMs on a lowest-priority stream so that they fill
 * the CUs the data-gradient chain leaves idle (replaces nothing in the reference: PyTorch autograd
 * serialises both on one stream, train_DCNet.py:645). */
void* dcn_stream_create(int level);
int dcn_stream_destroy(void* stream);
int dcn_stream_priority_range(int* least, int* greatest);

/* ---- optional kernel profiler (HIP events on the launch stream) -------------------------------- */
/* dcn_prof_enable(1) starts a recording window, (0) stops it; dcn_prof_collect waits for the events and
 * returns, per kernel tag (DCN_PROF_TAGS = 56 slots: size the arrays for 64; 24-27 = f16 two-piece split tiles: igemm 128x128 / wgrad / igemm 256x64 / NN; 28/29 = conv3 strip
 * kernel 256x128 / 128x128, 30 = reduce_slabs, 31 = dA, 32 = wgrad3; 15 = 128x128 NT tile with the 32-float K-step, 16 = split-bf16 128x128 NT
 * tile, 17 = split-bf16 128x128 weight-gradient / TN tile; 0-2 conv-engine NT tiles 128x128/128x64/256x32, 3-4 NN tiles,
 * 5 weight-gradient/TN GEMM, 6-7 64x128 tiles, 8/9 l2norm+score fwd/bwd, 10 scale_act, 11 BN backward,
 * 12 exp+sums, 13/14 small latency-bound GEMMs of the LSTM steps),
 * the launch count, the summed kernel milliseconds and the summed algorithmic work (FLOP or bytes). */
int dcn_prof_enable(int on);
int dcn_prof_collect(int64_t* counts, double* ms, double* work, double* bytes /* algorithmic HBM bytes of the FLOP-priced tags; may be NULL */);
/* The same window launch by launch (up to `max` records, launch order): returns the number written.  bench.py prices each launch
 * against the roofline that binds it: max(work / MFMA peak of its arithmetic, bytes / HBM peak). */
int dcn_prof_records(int32_t* tags, double* ms, double* work, double* bytes, int max);

/* ---- host-side negative sampling (CPU; bit-exact with Python's random.sample) -------------- */
/* state = the 625 uint32 of random.getstate()[1], advanced in place.
 * interframe: for each (pair, j) draws neg_n positions from range(hw) minus kpos[pair][j]
 *   (model/DCNet_model.py:411-413); out [pairs][top_k][neg_n] int64.  kpos == NULL returns the raw list
 *   positions p (the final index is p + (p >= kpos)), which lets the draw overlap with the GPU work.
 * crossmodal: the N*N*rows draws of Crossmodal_corrspondence (model/DCNet_model.py:62-96), keeping
 *   the index == N-1 draw of every (ii, jj); out [n][rows][neg_n] int64 (positions in image N-1). */
int dcn_mt_sample_interframe(uint32_t* state, const int64_t* kpos, int pairs, int top_k, int hw, int neg_n, int64_t* out);
int dcn_mt_sample_crossmodal(uint32_t* state, int n, int rows, int neg_n, int64_t* out);
/* The inverse of `out` for the backward of the gather: csr_off [rows+1], csr_src [n*rows*neg_n] = flat source indices
 * ((ii*rows + jj)*neg_n + m, ascending) of the negatives that point at each position of the last image. */
int dcn_mt_sample_crossmodal_csr(const int64_t* out, int n, int rows, int neg_n, int32_t* csr_off, int32_t* csr_src);
/* ABI 308: the three calls above for one training forward on n images (interframe with kpos = NULL, crossmodal, its csr) as ONE call;
 * *seconds (may be NULL) = CPU time of the calling thread spent in it. */
int dcn_mt_sample_step(uint32_t* state, int n, int top_k, int hw, int neg_n, int neg_c, int64_t* k9, int64_t* k14,
                       int32_t* csr_off, int32_t* csr_src, double* seconds);
/* ABI 308, opt-in (grounding_model.sampler = "device"; SURVEY H3 option (ii)): the same four tensors drawn ON THE DEVICE by a
 * counter-based generator (Philox-4x32-10) from state = {seed, step} (two device uint64; step is advanced by the call's last kernel, so
 * a replayed hipGraph draws new negatives every step with no host work).  Same distribution and exclusion rules as the random.sample
 * loops of model/DCNet_model.py:62-96 (of whose N*N*HW0 draws only the N*HW0 that are used are made) and :394-420; NOT the same
 * numbers as Python's MT19937 stream.  ws: dcn_device_sample_ws(hw) int32. */
int64_t dcn_device_sample_ws(int hw);
int dcn_device_sample(uint64_t* state, int n, int top_k, int hw, int neg_n, int neg_c, int64_t* k9, int64_t* k14,
                      int32_t* csr_off, int32_t* csr_src, int32_t* ws, void* stream);
/* C[b][M][N] = A[b][M][K] . B[b][N][K]^T for `batch` problems (strides in floats between problems; ldc may be any
 * value >= N): the HW0 x HW0 inter-frame affinity of K9 (model/DCNet_model.py:390). */
int dcn_gemm_nt_batched(const float* A, int lda, int64_t a_bs, const float* B, int ldb, int64_t b_bs, float* C, int ldc, int64_t c_bs,
                        int M, int N, int K, int batch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCNET_HIP_H */
