// C-ABI convolution entry points: forward and data gradient, both lowered onto the
// implicit-GEMM engine of igemm.hip.
#include "igemm.h"

namespace {

// [Co][T][Ci] -> [Ci][T][Co]: the data-gradient GEMM contracts over Cout, so its "weights"
// are the channel-transposed filters.  32x32 LDS tile transpose per tap, coalesced both ways.
__global__ __launch_bounds__(256) void transpose_filter_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                               int Co, int T, int Ci) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int co = co0 + ty + 8 * k, ci = ci0 + tx;
    tile[ty + 8 * k][tx] = (co < Co && ci < Ci) ? w[((size_t)co * T + t) * Ci + ci] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ci = ci0 + ty + 8 * k, co = co0 + tx;
    if (ci < Ci && co < Co) wt[((size_t)ci * T + t) * Co + co] = tile[tx][ty + 8 * k];
  }
}

// x*s = h + l (two f16, round to nearest) for every element; 8 consecutive k of a row become [8 x h | 8 x l] in the same
// 32 bytes, so the pre-split bank has the size, row stride and tap offsets of the fp32 one.  s (the power of two that maps
// the bank's abs-max below 2^14) goes to scale_out[0].  In place (src == dst) is fine: a thread rewrites its own 32 bytes.
__global__ __launch_bounds__(256) void presplit_kernel(const float* src, float* dst, int64_t groups,
                                                       const unsigned* __restrict__ amax, float* __restrict__ scale_out) {
  const unsigned bits = amax_read(amax);
  const int be = (int)((bits >> 23) & 0xFF);
  int e = (be == 0 || be == 255) ? 0 : 14 - (be - 126);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  const float s = __uint_as_float((unsigned)(e + 127) << 23);
  if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = s;
  typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + g * 8), b = *reinterpret_cast<const f32x4*>(src + g * 8 + 4);
    f16x8_t h, l;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float ta = a[k] * s, tb = b[k] * s;
      h[k] = (_Float16)ta; h[4 + k] = (_Float16)tb;
      l[k] = (_Float16)(ta - (float)h[k]); l[4 + k] = (_Float16)(tb - (float)h[4 + k]);
    }
    *reinterpret_cast<f16x8_t*>(dst + g * 8) = h;
    *reinterpret_cast<f16x8_t*>(dst + g * 8 + 4) = l;
  }
}

int launch_presplit(const float* src, float* dst, int64_t numel, const uint32_t* amax, float* scale_out, hipStream_t stream) {
  const int64_t groups = numel / 8;
  const int blocks = (int)((groups + 255) / 256 < 2048 ? (groups + 255) / 256 : 2048);
  hipLaunchKernelGGL(presplit_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, groups, (const unsigned*)amax, scale_out);
  DCN_CHECK_LAUNCH("presplit_f16");
  return DCN_OK;
}

int g_merge_classes = 1;    // dcn_set_tuning("merge", 0): stride-2 data gradients as four launches again

void base_params(IgemmParams& p) {
  p = IgemmParams{};
  p.osy = p.osx = 1; p.isy = p.isx = 1; p.dense_out = 1;
}

}  // namespace

void conv_set_merge(int v) { g_merge_classes = v; }
int g_n1_b16 = 1;           // dcn_set_tuning("Nb16", 0): bf16-storage 32 <-> 64 3x3 layers back on the gathered tiles of conv1.hip
void conv_set_n1_b16(int v) { g_n1_b16 = v; }
int g_d2_b16 = 1;           // dcn_set_tuning("Db16", 0): bf16-storage stride-2 data gradients of the narrow layers back on the gathered parity classes
void conv_set_d2_b16(int v) { g_d2_b16 = v; }

extern "C" int dcn_conv2d_stats_rows(int n, int h, int wd, int cout, int ksize, int stride) {
  const int pad = (ksize - 1) / 2;
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (wd + 2 * pad - ksize) / stride + 1;
  return igemm_grid_m(n * ho * wo, cout, ksize * ksize);
}

// Forward of a convolution whose input is the RAW output of the conv + BatchNorm layer in front: that layer's per-channel scale / shift
// and activation are applied where the input is loaded, so its scale_act pass and its activation tensor do not exist.
extern "C" int dcn_conv2d_pre_supported(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
  return igemm_precision() == 4 && nconv1_applicable(0, n, h, wd, cin, cout, ksize, stride) ? 1 : 0;
}

extern "C" int dcn_conv2d_fwd_pre(const float* x, const float* w, float* y, int n, int h, int wd, int cin, int cout, int ksize, int stride,
                                  const float* pre_scale, const float* pre_shift, int pre_act, float pre_slope,
                                  int ldy, float* stats, const uint32_t* amax_x, const uint32_t* amax_w, void* stream) {
  DCN_CHECK_ARG(x && w && y && pre_scale && pre_shift && amax_x && amax_w, "conv2d_fwd_pre: null pointer (the abs-max words are required)");
  DCN_CHECK_ARG(dcn_conv2d_pre_supported(n, h, wd, cin, cout, ksize, stride),
                "conv2d_fwd_pre: no loader-side activation for this shape / precision (ask dcn_conv2d_pre_supported first)");
  DCN_CHECK_ARG(pre_act == DCN_ACT_NONE || pre_act == DCN_ACT_LEAKY, "conv2d_fwd_pre: pre_act=%d", pre_act);
  if (ldy <= 0) ldy = cout;
  DCN_CHECK_ARG(ldy >= cout, "conv2d_fwd_pre: ldy=%d < cout=%d", ldy, cout);
  const DcnPreAct pre{pre_scale, pre_shift, pre_act, pre_slope};
  return nconv1_launch(0, x, cin, w, y, ldy, stats, stats ? dcn_conv2d_stats_rows(n, h, wd, cout, ksize, stride) : 0,
                       n, h, wd, stride, amax_x, amax_w, &pre, (hipStream_t)stream);
}

extern "C" int dcn_conv2d_fwd(const float* x, const float* w, float* y,
                              int n, int h, int wd, int cin, int cout, int ksize, int stride,
                              const float* scale, const float* shift, int act, float slope,
                              const float* residual, int ldr, int ldy,
                              float* stats, int accumulate, const float* f8_scales,
                              const uint32_t* amax_x, const uint32_t* amax_w, uint32_t* amax_y, float* w_split, int w_split_ready,
                              void* stream) {
  DCN_CHECK_ARG(ksize == 1 || ksize == 3, "conv2d_fwd: ksize=%d (1 or 3)", ksize);
  DCN_CHECK_ARG(stride == 1 || stride == 2, "conv2d_fwd: stride=%d (1 or 2)", stride);
  DCN_CHECK_ARG(n > 0 && h > 0 && wd > 0 && cin > 0 && cout > 0, "conv2d_fwd: bad shape");
  DCN_CHECK_ARG(cin == 4 || cin % 32 == 0, "conv2d_fwd: cin=%d must be 4 or a multiple of 32 (pad channels)", cin);
  DCN_CHECK_ARG(cin != 4 || (ksize == 3 && stride == 1), "conv2d_fwd: cin=4 path is the 3x3 stride-1 stem only");
  const int pad = (ksize - 1) / 2;
  IgemmParams p; base_params(p);
  p.amax_a = cin == 4 ? nullptr : amax_x; p.amax_b = cin == 4 ? nullptr : amax_w; p.amax_out = amax_y;
  p.in = x; p.wt = w; p.f8 = cin == 4 ? nullptr : f8_scales; p.out = y; p.scale = scale; p.shift = shift; p.residual = residual; p.stats = stats;
  p.N = n; p.Hi = h; p.Wi = wd; p.Ci = cin; p.ldi = cin;
  p.Ho = (h + 2 * pad - ksize) / stride + 1; p.Wo = (wd + 2 * pad - ksize) / stride + 1;
  p.Hs = p.Ho; p.Ws = p.Wo; p.isy = p.isx = stride;
  p.Co = cout; p.ldo = ldy > 0 ? ldy : cout; p.ldr = ldr > 0 ? ldr : cout;
  DCN_CHECK_ARG(p.ldo >= cout, "conv2d_fwd: ldy=%d < cout=%d", ldy, cout);
  p.M = n * p.Ho * p.Wo;
  p.ntaps = ksize * ksize;
  p.act = act; p.slope = slope; p.accumulate = accumulate;
  if (cin == 4) {
    // weights are [Co][64]: 9 taps x 4 channels then zero padding (host prepares them)
    p.c4 = 1; p.ldw = 64;
  } else {
    p.ldw = p.ntaps * cin; 
    for (int r = 0; r < ksize; ++r)
      for (int s = 0; s < ksize; ++s) {
        const int t = r * ksize + s;
        p.tap_dy[t] = r - pad; p.tap_dx[t] = s - pad; p.tap_w[t] = t * cin;
      }
  }
  if (p.c4 && stem_applicable(p, w_split)) return stem_launch(p, w_split, (hipStream_t)stream);      // (w_split: scratch of >= 27*32 floats)
  // the 32 -> 64 3x3 layers of the 416x416 / 208x208 maps, raw result + BatchNorm partial sums: filter bank in registers (nconv.hip)
  if (amax_x && amax_w && !amax_y && !f8_scales && !scale && !shift && !residual && act == DCN_ACT_NONE && !accumulate &&
      w_split_ready != 2 && igemm_precision() == 4 && nconv1_applicable(0, n, h, wd, cin, cout, ksize, stride))
    return nconv1_launch(0, x, cin, w, y, p.ldo, stats, stats ? dcn_conv2d_stats_rows(n, h, wd, cout, ksize, stride) : 0,
                         n, h, wd, stride, amax_x, amax_w, nullptr, (hipStream_t)stream);
  if (w_split && w_split_ready == 2) {
    p.wt16 = w_split;             // bf16-operand mode: the bank converted to bf16 (dcn_prepare_filters); read by the strip kernel only
  } else
  if (w_split && amax_x && amax_w && !p.c4 && igemm_will_presplit(p.M, cout, p.ntaps, cin)) {
    // split the filter bank once (into the caller's scratch: cout*k*k*cin + 16 floats), not once per M-tile
    const int64_t numel = (int64_t)cout * p.ntaps * cin;
    if (!w_split_ready) {         // (ready: dcn_prepare_filters already wrote the split bank and its scale there)
      int rc = launch_presplit(w, w_split, numel, amax_w, w_split + numel, (hipStream_t)stream);
      if (rc != DCN_OK) return rc;
    }
    p.wt = w_split; p.b_scale = w_split + numel;
  }
  return igemm_launch(p, (hipStream_t)stream);
}

namespace {
int bwd_data_impl(const float* dy, int lddy, const float* w, float* wt, float* dx,
                  int n, int h, int wd, int cin, int cout, int ksize, int stride,
                  int accumulate, const float* f8_scales, const uint32_t* amax_dy, const uint32_t* amax_w,
                  int wt_ready, const float* wt_split, const DcnBnTap* tap, int* tap_rows, hipStream_t stream);
}

extern "C" int dcn_conv2d_bwd_data(const float* dy, int lddy, const float* w, float* wt, float* dx,
                                   int n, int h, int wd, int cin, int cout, int ksize, int stride,
                                   int accumulate, const float* f8_scales, const uint32_t* amax_dy, const uint32_t* amax_w,
                                   int wt_ready, const float* wt_split, void* stream_) {
  return bwd_data_impl(dy, lddy, w, wt, dx, n, h, wd, cin, cout, ksize, stride, accumulate, f8_scales, amax_dy, amax_w, wt_ready, wt_split,
                       nullptr, nullptr, (hipStream_t)stream_);
}

extern "C" int dcn_conv2d_bwd_data_tap(const float* dy, int lddy, const float* w, float* wt, float* dx,
                                       int n, int h, int wd, int cin, int cout, int ksize, int stride,
                                       int accumulate, const float* f8_scales, const uint32_t* amax_dy, const uint32_t* amax_w,
                                       int wt_ready, const float* wt_split,
                                       const float* tap_y, const float* tap_mean, const float* tap_invstd, const float* tap_gamma,
                                       const float* tap_beta, int tap_act, float tap_slope, float* tap_stats, int tap_stats_rows,
                                       int* tap_rows, void* stream_) {
  DCN_CHECK_ARG(tap_rows, "conv2d_bwd_data_tap: tap_rows is where the number of statistics rows written comes back");
  *tap_rows = 0;
  DcnBnTap tap{tap_y, tap_mean, tap_invstd, tap_gamma, tap_beta, tap_act, tap_slope, tap_stats, tap_stats_rows};
  return bwd_data_impl(dy, lddy, w, wt, dx, n, h, wd, cin, cout, ksize, stride, accumulate, f8_scales, amax_dy, amax_w, wt_ready, wt_split,
                       (tap_y && tap_stats) ? &tap : nullptr, tap_rows, (hipStream_t)stream_);
}

extern "C" int dcn_conv2d_bwd_data_tap_rows(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
  if (cin == 32 && dgrad2_applicable(n, h, wd, cin, cout, ksize, stride, 0)) return dgrad2_grid(n, h, wd, cin);
  // stride-1 layers on conv1.hip / conv3.hip (an upper bound here: whether a launch taps is reported by the launch itself)
  if (stride == 1 && igemm_precision() == 4 && cin % 32 == 0 && cin >= 64) return igemm_grid_m(n * h * wd, cin, ksize * ksize);
  return 0;
}

namespace {
int bwd_data_impl(const float* dy, int lddy, const float* w, float* wt, float* dx,
                  int n, int h, int wd, int cin, int cout, int ksize, int stride,
                  int accumulate, const float* f8_scales, const uint32_t* amax_dy, const uint32_t* amax_w,
                  int wt_ready, const float* wt_split, const DcnBnTap* tap, int* tap_rows, hipStream_t stream) {
  DCN_CHECK_ARG(ksize == 1 || ksize == 3, "conv2d_bwd_data: ksize=%d", ksize);
  DCN_CHECK_ARG(stride == 1 || stride == 2, "conv2d_bwd_data: stride=%d", stride);
  DCN_CHECK_ARG(cout % 32 == 0, "conv2d_bwd_data: cout=%d must be a multiple of 32 (pad the filter bank)", cout);
  DCN_CHECK_ARG(dy && w && wt && dx, "conv2d_bwd_data: null pointer");
  const int pad = (ksize - 1) / 2, T = ksize * ksize;
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (wd + 2 * pad - ksize) / stride + 1;
  if (lddy <= 0) lddy = cout;
  // wt_ready: wt already holds the transposed bank (dcn_prepare_filters) and, when given, wt_split its split form with the
  // scale behind it; nothing is written to either here
  DCN_CHECK_ARG(wt_ready || !wt_split, "conv2d_bwd_data: wt_split needs wt_ready");
  if (!wt_ready) {
    hipLaunchKernelGGL(transpose_filter_kernel, dim3(cdiv(cin, 32), cdiv(cout, 32), T), dim3(256), 0, stream,
                       w, wt, cout, T, cin);
    DCN_CHECK_LAUNCH("transpose_filter");
  }
  // the stride-2 layers of the 416x416 / 208x208 maps: all four parity classes from one pass over dY, filter bank in registers (nconv.hip)
  if (amax_dy && amax_w && !f8_scales && wt_ready != 2 && igemm_precision() == 4 &&
      dgrad2_applicable(n, h, wd, cin, cout, ksize, stride, accumulate)) {
    if (tap && (cin != 32 || tap->stats_rows < dgrad2_grid(n, h, wd, cin))) tap = nullptr;      // (64-channel form / too few rows: the caller reduces by itself)
    const int rc = dgrad2_launch(dy, lddy, wt, dx, n, h, wd, cin, accumulate, amax_dy, amax_w, tap, stream);
    if (rc == DCN_OK && tap && tap_rows) *tap_rows = dgrad2_grid(n, h, wd, cin);
    return rc;
  }
  if (amax_dy && amax_w && !f8_scales && wt_ready != 2 && !accumulate && igemm_precision() == 4 &&
      nconv1_applicable(1, n, h, wd, cin, cout, ksize, stride))
    return nconv1_launch(1, dy, lddy, wt, dx, cin, nullptr, 0, n, h, wd, 1, amax_dy, amax_w, nullptr, stream);
  IgemmParams p; base_params(p);
  p.in = dy; p.wt = wt; p.f8 = f8_scales; p.out = dx; p.amax_a = amax_dy; p.amax_b = amax_w;
  p.N = n; p.Hi = ho; p.Wi = wo; p.Ci = cout; p.ldi = lddy;
  p.Ho = h; p.Wo = wd; p.Co = cin; p.ldo = cin; p.ldr = cin; p.ldw = T * cout;
  p.accumulate = accumulate;
  if (wt_ready == 2) {            // bf16-operand mode: wt = transposed fp32 bank, wt_split = the same bank in bf16 (strip kernel only)
    p.wt16 = wt_split; wt_split = nullptr;
  }
  // four parity classes in one launch: measured (tools/bench_convs.py --ab merge=0) -8..-9 % on the 52/26-wide maps, whose
  // classes fill 0.7-2.6 rounds of the chip each, and -7..+20 % on the larger ones (their classes are long grids already and
  // the kernels there are issue-bound, not waiting for dY): only where a class has at most 64 K rows (g_merge_classes = 2: always)
  const bool merged = stride == 2 && ksize == 3 && h >= 2 && wd >= 2 &&
                      (g_merge_classes == 2 || (g_merge_classes == 1 && (long long)n * ((h + 1) / 2) * ((wd + 1) / 2) <= 65536));
  if (amax_dy && amax_w && igemm_will_presplit((long long)n * (stride == 1 ? h * wd : ((h + 1) / 2) * ((wd + 1) / 2)), cin,
                                               stride == 1 ? T : (merged ? 4 : 1), cout)) {
    // (stride 2: the four parity classes share one bank; as separate launches the smallest class, with 1 tap, decides for all
    //  of them; merged, the launch is sized by its largest class: 4 taps)
    const int64_t numel = (int64_t)cin * T * cout;
    if (wt_ready) {
      if (wt_split) { p.wt = wt_split; p.b_scale = wt_split + numel; }       // else: the tiles split the fp32 bank themselves
    } else {
      int rc = launch_presplit(wt, wt, numel, amax_w, wt + numel, stream);   // in place; the scale lands behind the bank
      if (rc != DCN_OK) return rc;
      p.b_scale = wt + numel;
    }
  }
  if (stride == 1) {
    // dx[hi,wi] = sum_{r,s} dy[hi+pad-r, wi+pad-s] . w[:,r,s,:]
    p.Hs = h; p.Ws = wd; p.M = n * h * wd; p.ntaps = T;
    for (int r = 0; r < ksize; ++r)
      for (int s = 0; s < ksize; ++s) {
        const int t = r * ksize + s;
        p.tap_dy[t] = pad - r; p.tap_dx[t] = pad - s; p.tap_w[t] = t * cout;
      }
    if (tap) {
      // a stride-1 data gradient IS a forward convolution on the transposed bank: the BatchNorm partial-sum epilogues of conv1.hip /
      // conv3.hip take the tap (one partial row per M-tile, as dcn_conv2d_stats_rows counts them for the mirrored forward)
      const int rows = igemm_grid_m(p.M, cin, T);
      p.stats = tap->stats;
      if (tap->stats_rows >= rows && igemm_tap_capable(p)) {
        p.bt_y = tap->y; p.bt_mean = tap->mean; p.bt_invstd = tap->invstd; p.bt_gamma = tap->gamma; p.bt_beta = tap->beta;
        p.bt_act = tap->act; p.bt_slope = tap->slope;
        const int rc = igemm_launch(p, stream);
        if (rc == DCN_OK && tap_rows) *tap_rows = rows;
        return rc;
      }
      p.stats = nullptr;
    }
    return igemm_launch(p, stream);
  }
  // 1x1 stride 2: only the even-even pixels receive a gradient; the other three parity classes have no tap and are
  // never visited below, so they must be zeroed here (not a DCNet layer shape, but the entry point is general).
  if (ksize == 1 && !accumulate &&
      hipMemsetAsync(dx, 0, (size_t)n * h * wd * cin * sizeof(float), stream) != hipSuccess) {
    dcn_set_error("conv2d_bwd_data: memset failed"); return DCN_ERR_LAUNCH;
  }
  // stride 2: output pixels of parity class (a,b) only see taps with (a+pad-r), (b+pad-s) even.
  // Four dense sub-problems with 1/2/2/4 taps (3x3) instead of one 9-tap problem that is 3/4 zeros.
  if (merged) {
    // ... all in ONE launch (igemm.h ncls): block order (M-tile, class, N-tile) keeps the four classes of a region on one XCD
    // at the same time, so dY is fetched from HBM once instead of once per tap (nine times a tensor that does not fit the
    // Infinity Cache on the 416/208 maps), and the 13/26-wide maps fill the chip with one grid instead of four short ones.
    IgemmParams q = p;
    q.dense_out = 0; q.osy = q.osx = 2; q.ncls = 4; q.M = 0; q.ntaps = 0;
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        const int c = a * 2 + b;
        q.cls_oy0[c] = a; q.cls_ox0[c] = b;
        q.cls_Hs[c] = (h - a + 1) / 2; q.cls_Ws[c] = (wd - b + 1) / 2;
        q.cls_M[c] = n * q.cls_Hs[c] * q.cls_Ws[c];
        int nt = 0;
        for (int r = 0; r < ksize; ++r)
          for (int s = 0; s < ksize; ++s) {
            if (((a + pad - r) & 1) || ((b + pad - s) & 1)) continue;
            q.tap_dy[4 * c + nt] = (a + pad - r) / 2; q.tap_dx[4 * c + nt] = (b + pad - s) / 2;
            q.tap_w[4 * c + nt] = (r * ksize + s) * cout;
            ++nt;
          }
        q.cls_ntaps[c] = nt;
        if (q.cls_M[c] > q.M) { q.M = q.cls_M[c]; q.Hs = q.cls_Hs[c]; q.Ws = q.cls_Ws[c]; }
        if (nt > q.ntaps) q.ntaps = nt;
      }
    q.oy0 = q.ox0 = 0;
    return igemm_launch(q, stream);
  }
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      IgemmParams q = p;
      q.dense_out = 0; q.oy0 = a; q.ox0 = b; q.osy = q.osx = 2;
      q.Hs = (h - a + 1) / 2; q.Ws = (wd - b + 1) / 2;
      if (q.Hs <= 0 || q.Ws <= 0) continue;
      q.M = n * q.Hs * q.Ws;
      q.ntaps = 0;
      for (int r = 0; r < ksize; ++r)
        for (int s = 0; s < ksize; ++s) {
          if (((a + pad - r) & 1) || ((b + pad - s) & 1)) continue;
          // hi = 2i+a  ->  ho = (2i + a + pad - r)/2 = i + (a+pad-r)/2
          q.tap_dy[q.ntaps] = (a + pad - r) / 2; q.tap_dx[q.ntaps] = (b + pad - s) / 2;
          q.tap_w[q.ntaps] = (r * ksize + s) * cout;
          ++q.ntaps;
        }
      if (q.ntaps == 0) {
        // 1x1 stride 2: odd pixels get no gradient.  (Not used by DCNet; keep semantics right.)
        continue;
      }
      int rc = igemm_launch(q, stream);
      if (rc != DCN_OK) return rc;
    }
  return DCN_OK;
}
}  // namespace


// ---- bf16 storage (BASELINE.json configs[2]): activations, gradients and filter banks are bf16 tensors in HBM -----------------------
// x / dy / residual / tap_y: bf16 NHWC; w16: the bank as dcn_prepare_filters writes it (b16: [Co][k*k*Ci], tb16: [Ci][k*k*Co]); y / dx:
// bf16, or fp32 when *_f32 is set (the boundary to fp32 consumers).  fp32 accumulate, fp32 BatchNorm partial sums (of the values as
// stored).  All launches run on conv1.hip's ring kernel in its bf16 form; cin must be a multiple of 32 (the 3-channel stem keeps its
// fp32 input and kernel).
extern "C" int dcn_conv2d_stats_rows_b16(int n, int h, int wd, int cout, int ksize, int stride) {
  const int pad = (ksize - 1) / 2;
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (wd + 2 * pad - ksize) / stride + 1;
  return conv1b_grid_m(n * ho * wo, cout, ksize * ksize, (ksize == 3 && stride == 1) ? wd : 0);
}

extern "C" int dcn_conv2d_fwd_b16(const void* x, const void* w16, void* y, int y_f32, int n, int h, int wd, int cin, int cout, int ksize,
                                  int stride, const float* scale, const float* shift, int act, float slope, const void* residual, int ldr,
                                  int ldy, float* stats, int accumulate, void* stream) {
  DCN_CHECK_ARG(x && w16 && y, "conv2d_fwd_b16: null pointer");
  DCN_CHECK_ARG((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "conv2d_fwd_b16: ksize=%d stride=%d", ksize, stride);
  DCN_CHECK_ARG(n > 0 && h > 0 && wd > 0 && cin > 0 && cin % 32 == 0 && cout > 0 && cout % 32 == 0,
                "conv2d_fwd_b16: cin=%d / cout=%d must be multiples of 32", cin, cout);
  const int pad = (ksize - 1) / 2;
  IgemmParams p; base_params(p);
  p.in = (const float*)x; p.wt = (const float*)w16; p.out = (float*)y; p.scale = scale; p.shift = shift; p.residual = (const float*)residual;
  p.stats = stats;
  p.N = n; p.Hi = h; p.Wi = wd; p.Ci = cin; p.ldi = cin;
  p.Ho = (h + 2 * pad - ksize) / stride + 1; p.Wo = (wd + 2 * pad - ksize) / stride + 1;
  p.Hs = p.Ho; p.Ws = p.Wo; p.isy = p.isx = stride;
  p.Co = cout; p.ldo = ldy > 0 ? ldy : cout; p.ldr = ldr > 0 ? ldr : cout;
  DCN_CHECK_ARG(p.ldo >= cout, "conv2d_fwd_b16: ldy=%d < cout=%d", ldy, cout);
  p.M = n * p.Ho * p.Wo; p.ntaps = ksize * ksize; p.ldw = p.ntaps * cin;
  p.act = act; p.slope = slope; p.accumulate = accumulate;
  for (int r = 0; r < ksize; ++r)
    for (int s = 0; s < ksize; ++s) {
      const int t = r * ksize + s;
      p.tap_dy[t] = r - pad; p.tap_dx[t] = s - pad; p.tap_w[t] = t * cin;
    }
  // the raw forward of the 32 -> 64 layers on the 416 / 208-wide maps: the register-bank kernel (nconv.hip), x read once instead of nine times
  if (g_n1_b16 && !y_f32 && !scale && !shift && act == DCN_ACT_NONE && !residual && !accumulate && p.ldo == cout &&
      nconv1_applicable(0, n, h, wd, cin, cout, ksize, stride))
    return nconv1_launch_b16(0, x, cin, w16, y, cout, stats, stats ? dcn_conv2d_stats_rows_b16(n, h, wd, cout, ksize, stride) : 0, n, h, wd, stride,
                             (hipStream_t)stream);
  return conv1b_launch(p, y_f32, (hipStream_t)stream);
}

extern "C" int dcn_conv2d_bwd_data_b16(const void* dy, int lddy, const void* wt16, void* dx, int dx_f32, int n, int h, int wd, int cin,
                                       int cout, int ksize, int stride, int accumulate, const void* tap_y, const float* tap_mean,
                                       const float* tap_invstd, const float* tap_gamma, const float* tap_beta, int tap_act, float tap_slope,
                                       float* tap_stats, int tap_stats_rows, int* tap_rows, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DCN_CHECK_ARG(dy && wt16 && dx, "conv2d_bwd_data_b16: null pointer");
  DCN_CHECK_ARG((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "conv2d_bwd_data_b16: ksize=%d stride=%d", ksize, stride);
  DCN_CHECK_ARG(cout % 32 == 0 && cin % 32 == 0, "conv2d_bwd_data_b16: cin=%d / cout=%d must be multiples of 32", cin, cout);
  if (tap_rows) *tap_rows = 0;
  const int pad = (ksize - 1) / 2, T = ksize * ksize;
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (wd + 2 * pad - ksize) / stride + 1;
  if (lddy <= 0) lddy = cout;
  IgemmParams p; base_params(p);
  p.in = (const float*)dy; p.wt = (const float*)wt16; p.out = (float*)dx;
  p.N = n; p.Hi = ho; p.Wi = wo; p.Ci = cout; p.ldi = lddy;
  p.Ho = h; p.Wo = wd; p.Co = cin; p.ldo = cin; p.ldr = cin; p.ldw = T * cout;
  p.accumulate = accumulate;
  // the data gradient 64 -> 32 of the stride-1 layer on the 208-wide map: the register-bank kernel (no tap there: the caller reduces)
  if (g_n1_b16 && stride == 1 && !dx_f32 && !accumulate && lddy == cout && nconv1_applicable(1, n, h, wd, cin, cout, ksize, stride))
    return nconv1_launch_b16(1, dy, lddy, wt16, dx, cin, nullptr, 0, n, h, wd, 1, stream);
  if (stride == 1) {
    p.Hs = h; p.Ws = wd; p.M = n * h * wd; p.ntaps = T;
    for (int r = 0; r < ksize; ++r)
      for (int s = 0; s < ksize; ++s) {
        const int t = r * ksize + s;
        p.tap_dy[t] = pad - r; p.tap_dx[t] = pad - s; p.tap_w[t] = t * cout;
      }
    if (tap_y && tap_stats && tap_mean && tap_invstd) {
      const int rows = conv1b_grid_m(p.M, cin, T, ksize == 3 ? wd : 0);
      if (rows > 0 && tap_stats_rows >= rows) {
        p.stats = tap_stats; p.bt_y = (const float*)tap_y; p.bt_mean = tap_mean; p.bt_invstd = tap_invstd; p.bt_gamma = tap_gamma;
        p.bt_beta = tap_beta; p.bt_act = tap_act; p.bt_slope = tap_slope;
        const int rc = conv1b_launch(p, dx_f32, stream);
        if (rc == DCN_OK && tap_rows) *tap_rows = rows;
        return rc;
      }
    }
    return conv1b_launch(p, dx_f32, stream);
  }
  // the 32 <- 64 / 64 <- 128 stride-2 layers on the 416 / 208-wide maps: the register-bank kernel (nconv.hip), all four parity classes of a
  // position stored by the workgroup that computed them (the four gathered launches below: 1.14 ms against 0.2 ms of HBM traffic)
  if (g_d2_b16 && !dx_f32 && lddy == cout && dgrad2_applicable(n, h, wd, cin, cout, ksize, stride, accumulate))
    return dgrad2_launch_b16(dy, lddy, wt16, dx, n, h, wd, cin, accumulate, stream);
  if (ksize == 1 && !accumulate &&
      hipMemsetAsync(dx, 0, (size_t)n * h * wd * cin * (dx_f32 ? 4 : 2), stream) != hipSuccess) {
    dcn_set_error("conv2d_bwd_data_b16: memset failed"); return DCN_ERR_LAUNCH;
  }
  // stride 2: four dense parity classes with 1 / 2 / 2 / 4 taps (see bwd_data_impl)
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      IgemmParams q = p;
      q.dense_out = 0; q.oy0 = a; q.ox0 = b; q.osy = q.osx = 2;
      q.Hs = (h - a + 1) / 2; q.Ws = (wd - b + 1) / 2;
      if (q.Hs <= 0 || q.Ws <= 0) continue;
      q.M = n * q.Hs * q.Ws;
      q.ntaps = 0;
      for (int r = 0; r < ksize; ++r)
        for (int s = 0; s < ksize; ++s) {
          if (((a + pad - r) & 1) || ((b + pad - s) & 1)) continue;
          q.tap_dy[q.ntaps] = (a + pad - r) / 2; q.tap_dx[q.ntaps] = (b + pad - s) / 2;
          q.tap_w[q.ntaps] = (r * ksize + s) * cout;
          ++q.ntaps;
        }
      if (q.ntaps == 0) continue;
      const int rc = conv1b_launch(q, dx_f32, stream);
      if (rc != DCN_OK) return rc;
    }
  return DCN_OK;
}

// ---- fp8 storage (BASELINE.json configs[4]): forward / data gradient on e4m3 operands with one e8m0 scale per pixel / per filter ---------
// x8 / dy8: e4m3 bytes NHWC (dense), xs / dys: e8m0 [pixels] (dcn_quant_rows_e4m3); w8: the bank [Cout][k*k*Cin] (forward) or the
// transposed bank [Cin][k*k*Cout] (data gradient) as e4m3 bytes with ws: e8m0 per bank row.  y / dx, residual, the BatchNorm tap: bf16
// tensors exactly as in dcn_conv2d_*_b16 (fp32 accumulate, fp32 statistics of the values as stored).  cin, cout multiples of 64 / 32.
extern "C" int dcn_conv2d_stats_rows_f8(int n, int h, int wd, int cout, int ksize, int stride) {
  const int pad = (ksize - 1) / 2;
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (wd + 2 * pad - ksize) / stride + 1;
  return conv1q_grid_m(n * ho * wo, cout);
}

extern "C" int dcn_conv2d_fwd_f8(const void* x8, const void* xs, const void* w8, const void* ws, void* y, int y_f32, int n, int h, int wd, int cin,
                                 int cout, int ksize, int stride, const float* scale, const float* shift, int act, float slope,
                                 const void* residual, int ldr, int ldy, float* stats, int accumulate, void* stream) {
  DCN_CHECK_ARG(x8 && xs && w8 && ws && y, "conv2d_fwd_f8: null pointer");
  DCN_CHECK_ARG((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "conv2d_fwd_f8: ksize=%d stride=%d", ksize, stride);
  DCN_CHECK_ARG(n > 0 && h > 0 && wd > 0 && cin > 0 && cin % 64 == 0 && cout > 0 && cout % 32 == 0,
                "conv2d_fwd_f8: cin=%d must be a multiple of 64, cout=%d of 32", cin, cout);
  const int pad = (ksize - 1) / 2;
  IgemmParams p; base_params(p);
  p.in = (const float*)x8; p.wt = (const float*)w8; p.out = (float*)y; p.scale = scale; p.shift = shift; p.residual = (const float*)residual;
  p.a_scale8 = (const unsigned char*)xs; p.b_scale8 = (const unsigned char*)ws;
  p.stats = stats;
  p.N = n; p.Hi = h; p.Wi = wd; p.Ci = cin; p.ldi = cin;
  p.Ho = (h + 2 * pad - ksize) / stride + 1; p.Wo = (wd + 2 * pad - ksize) / stride + 1;
  p.Hs = p.Ho; p.Ws = p.Wo; p.isy = p.isx = stride;
  p.Co = cout; p.ldo = ldy > 0 ? ldy : cout; p.ldr = ldr > 0 ? ldr : cout;
  DCN_CHECK_ARG(p.ldo >= cout, "conv2d_fwd_f8: ldy=%d < cout=%d", ldy, cout);
  p.M = n * p.Ho * p.Wo; p.ntaps = ksize * ksize; p.ldw = p.ntaps * cin;
  p.act = act; p.slope = slope; p.accumulate = accumulate;
  for (int r = 0; r < ksize; ++r)
    for (int s = 0; s < ksize; ++s) {
      const int t = r * ksize + s;
      p.tap_dy[t] = r - pad; p.tap_dx[t] = s - pad; p.tap_w[t] = t * cin;
    }
  return conv1q_launch(p, y_f32, (hipStream_t)stream);
}

extern "C" int dcn_conv2d_bwd_data_f8(const void* dy8, const void* dys, const void* wt8, const void* wts, void* dx, int dx_f32, int n, int h, int wd,
                                      int cin, int cout, int ksize, int stride, int accumulate, const void* tap_y, const float* tap_mean,
                                      const float* tap_invstd, const float* tap_gamma, const float* tap_beta, int tap_act, float tap_slope,
                                      float* tap_stats, int tap_stats_rows, int* tap_rows, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DCN_CHECK_ARG(dy8 && dys && wt8 && wts && dx, "conv2d_bwd_data_f8: null pointer");
  DCN_CHECK_ARG((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "conv2d_bwd_data_f8: ksize=%d stride=%d", ksize, stride);
  DCN_CHECK_ARG(cout % 64 == 0 && cin % 32 == 0, "conv2d_bwd_data_f8: cout=%d must be a multiple of 64, cin=%d of 32", cout, cin);
  if (tap_rows) *tap_rows = 0;
  const int pad = (ksize - 1) / 2, T = ksize * ksize;
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (wd + 2 * pad - ksize) / stride + 1;
  IgemmParams p; base_params(p);
  p.in = (const float*)dy8; p.wt = (const float*)wt8; p.out = (float*)dx;
  p.a_scale8 = (const unsigned char*)dys; p.b_scale8 = (const unsigned char*)wts;
  p.N = n; p.Hi = ho; p.Wi = wo; p.Ci = cout; p.ldi = cout;
  p.Ho = h; p.Wo = wd; p.Co = cin; p.ldo = cin; p.ldr = cin; p.ldw = T * cout;
  p.accumulate = accumulate;
  if (stride == 1) {
    p.Hs = h; p.Ws = wd; p.M = n * h * wd; p.ntaps = T;
    for (int r = 0; r < ksize; ++r)
      for (int s = 0; s < ksize; ++s) {
        const int t = r * ksize + s;
        p.tap_dy[t] = pad - r; p.tap_dx[t] = pad - s; p.tap_w[t] = t * cout;
      }
    if (tap_y && tap_stats && tap_mean && tap_invstd) {
      const int rows = conv1q_grid_m(p.M, cin);
      if (rows > 0 && tap_stats_rows >= rows) {
        p.stats = tap_stats; p.bt_y = (const float*)tap_y; p.bt_mean = tap_mean; p.bt_invstd = tap_invstd; p.bt_gamma = tap_gamma;
        p.bt_beta = tap_beta; p.bt_act = tap_act; p.bt_slope = tap_slope;
        const int rc = conv1q_launch(p, dx_f32, stream);
        if (rc == DCN_OK && tap_rows) *tap_rows = rows;
        return rc;
      }
    }
    return conv1q_launch(p, dx_f32, stream);
  }
  if (ksize == 1 && !accumulate &&
      hipMemsetAsync(dx, 0, (size_t)n * h * wd * cin * (dx_f32 ? 4 : 2), stream) != hipSuccess) {
    dcn_set_error("conv2d_bwd_data_f8: memset failed"); return DCN_ERR_LAUNCH;
  }
  // stride 2: four dense parity classes with 1 / 2 / 2 / 4 taps (see dcn_conv2d_bwd_data_b16)
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      IgemmParams q = p;
      q.dense_out = 0; q.oy0 = a; q.ox0 = b; q.osy = q.osx = 2;
      q.Hs = (h - a + 1) / 2; q.Ws = (wd - b + 1) / 2;
      if (q.Hs <= 0 || q.Ws <= 0) continue;
      q.M = n * q.Hs * q.Ws;
      q.ntaps = 0;
      for (int r = 0; r < ksize; ++r)
        for (int s = 0; s < ksize; ++s) {
          if (((a + pad - r) & 1) || ((b + pad - s) & 1)) continue;
          q.tap_dy[q.ntaps] = (a + pad - r) / 2; q.tap_dx[q.ntaps] = (b + pad - s) / 2;
          q.tap_w[q.ntaps] = (r * ksize + s) * cout;
          ++q.ntaps;
        }
      if (q.ntaps == 0) continue;
      const int rc = conv1q_launch(q, dx_f32, stream);
      if (rc != DCN_OK) return rc;
    }
  return DCN_OK;
}
