// Layout boundary (reference NCHW / OIHW  <->  internal NHWC / OHWI) and the small data movers
// (nearest x2 upsample fused with the route concat, channel-slice copy/accumulate).
// All HBM-bound; transposes go through a padded 32x33 LDS tile so both sides stay coalesced.
#include "common.h"

namespace {

// src viewed as [B][R][C] -> dst [B][C][R_pad-less]: generic batched 2-D transpose with independent
// leading strides; dst rows may be padded (ldd >= R) and extra dst columns zero-filled up to Rpad.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        int R, int C, int lds_, int ldd, int Rpad,
                                                        int64_t sb, int64_t db) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* s = src + (int64_t)b * sb;
  float* d = dst + (int64_t)b * db;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < R && c < C) ? s[(int64_t)r * lds_ + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < C && r < Rpad) d[(int64_t)c * ldd + r] = tile[tx][ty + 8 * k];
  }
}

__global__ __launch_bounds__(256) void upsample2_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd,
                                                        int n, int h, int w, int c) {
  const int c4 = c >> 2;
  const int64_t total = (int64_t)n * 2 * h * 2 * w * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c4) * 4; int64_t pix = i / c4;
    const int x = (int)(pix % (2 * w)); pix /= 2 * w;
    const int y = (int)(pix % (2 * h)); const int b = (int)(pix / (2 * h));
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (((int64_t)b * h + (y >> 1)) * w + (x >> 1)) * lds_ + ch);
    *reinterpret_cast<f32x4*>(dst + (((int64_t)b * 2 * h + y) * 2 * w + x) * ldd + ch) = v;
  }
}

__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ ddst, int ldd, float* __restrict__ dsrc, int lds_,
                                                            int n, int h, int w, int c, int accumulate) {
  const int c4 = c >> 2;
  const int64_t total = (int64_t)n * h * w * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c4) * 4; int64_t pix = i / c4;
    const int x = (int)(pix % w); pix /= w;
    const int y = (int)(pix % h); const int b = (int)(pix / h);
    const float* p = ddst + (((int64_t)b * 2 * h + 2 * y) * 2 * w + 2 * x) * ldd + ch;
    f32x4 v = *reinterpret_cast<const f32x4*>(p) + *reinterpret_cast<const f32x4*>(p + ldd) +
              *reinterpret_cast<const f32x4*>(p + (int64_t)2 * w * ldd) + *reinterpret_cast<const f32x4*>(p + (int64_t)(2 * w + 1) * ldd);
    float* o = dsrc + (((int64_t)b * h + y) * w + x) * lds_ + ch;
    if (accumulate) v += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

__global__ __launch_bounds__(256) void copy_slice_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd,
                                                         int64_t rows, int c, int accumulate) {
  const int c4 = c >> 2;
  const int64_t total = rows * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / c4; const int ch = (int)(i - r * c4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(src + r * lds_ + ch);
    float* o = dst + r * ldd + ch;
    if (accumulate) v += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = v;
  }
}

inline int stream_grid(int64_t work_items) {
  int64_t b = (work_items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

namespace {
// The image: [C <= 4][HW] planes -> [HW][4] pixels (missing channels zero).  One pixel per thread: plane reads and 16-byte pixel writes
// are both coalesced (the 32 x 32 tile transpose below moves 32 channels per pixel tile for the 3 that exist: 0.32 ms at 64 x 416 x 416).
__global__ __launch_bounds__(256) void image_to_nhwc4_kernel(const float* __restrict__ src, float* __restrict__ dst, int c, int64_t hw, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int64_t n = i / hw, p = i - n * hw;
  const float* s = src + n * c * hw + p;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  v[0] = s[0];
  if (c > 1) v[1] = s[hw];
  if (c > 2) v[2] = s[2 * hw];
  if (c > 3) v[3] = s[3 * hw];
  *reinterpret_cast<f32x4*>(dst + i * 4) = v;
}
}  // namespace

extern "C" int dcn_nchw_to_nhwc(const float* src, float* dst, int n, int c, int h, int w, int c_pad, void* stream) {
  DCN_CHECK_ARG(src && dst && n > 0 && c > 0 && h > 0 && w > 0 && c_pad >= c, "nchw_to_nhwc: bad argument");
  if (c_pad == 4 && ((uintptr_t)dst & 15) == 0) {
    const int64_t total = (int64_t)n * h * w;
    hipLaunchKernelGGL(image_to_nhwc4_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, c, (int64_t)h * w, total);
    DCN_CHECK_LAUNCH("image_to_nhwc4");
    return DCN_OK;
  }
  // per image: [C][HW] -> [HW][c_pad]
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(h * w, 32), cdiv(c_pad, 32), n), dim3(256), 0, (hipStream_t)stream,
                     src, dst, c, h * w, h * w, c_pad, c_pad, (int64_t)c * h * w, (int64_t)h * w * c_pad);
  DCN_CHECK_LAUNCH("nchw_to_nhwc");
  return DCN_OK;
}

extern "C" int dcn_nhwc_to_nchw(const float* src, float* dst, int n, int c, int h, int w, int ld, void* stream) {
  if (ld <= 0) ld = c;
  DCN_CHECK_ARG(src && dst && n > 0 && c > 0 && h > 0 && w > 0 && ld >= c, "nhwc_to_nchw: bad argument");
  // per image: [HW][ld] (first c columns) -> [C][HW]
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(c, 32), cdiv(h * w, 32), n), dim3(256), 0, (hipStream_t)stream,
                     src, dst, h * w, c, ld, h * w, h * w, (int64_t)h * w * ld, (int64_t)c * h * w);
  DCN_CHECK_LAUNCH("nhwc_to_nchw");
  return DCN_OK;
}

// ---- fp8 path: per-tensor power-of-two scale ------------------------------------------------------------------
// scale = 2^floor(log2(448 / max|x|)) (1 for an all-zero tensor): x*scale fits the finite e4m3 range and the scaling is
// exact.  max is order-independent, so the atomic (on the bits of a non-negative float) keeps results reproducible.
// spread != 0: `out` is a DCN_AMAX_WORDS-word abs-max vector (common.h); 0: a single word (the fp8 path's scratch)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t rows, int c4, int ld, unsigned* __restrict__ out,
                                                     int spread) {
  const int64_t total = rows * c4;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / c4; const int c = (int)(i - r * c4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ld + c);
    m = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), m);
  }
  m = wave_max(m);
  __shared__ float red[4];
  m = wave_max(m);            // (already a wave maximum: cheap)
  if (spread) amax_update_block(out, m, red);
  else if ((threadIdx.x & 63) == 0) amax_update(out, m, 0u);
}
__global__ void f8_scale_finish_kernel(unsigned* bits, float* scale) {
  const float amax = __uint_as_float(*bits);
  float s = 1.f;
  if (amax > 0.f && amax < 3.0e38f) {
    int e; (void)frexpf(448.f / amax, &e);          // 448/amax = f * 2^e, f in [0.5, 1)  ->  floor(log2) = e - 1
    e = e - 1; e = e > 100 ? 100 : (e < -100 ? -100 : e);
    s = ldexpf(1.f, e);
  }
  *scale = s;
}

extern "C" int dcn_f8_scale(const float* x, int64_t rows, int c, int ld, float* scale, void* ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DCN_CHECK_ARG(x && scale && ws && rows > 0 && c > 0 && c % 4 == 0 && ld % 4 == 0 && ld >= c, "f8_scale: bad argument (c=%d ld=%d)", c, ld);
  DCN_CHECK_ARG(((uintptr_t)x & 15) == 0, "f8_scale: x must be 16-byte aligned");
  if (hipMemsetAsync(ws, 0, 4, stream) != hipSuccess) { dcn_set_error("f8_scale: memset failed"); return DCN_ERR_LAUNCH; }
  const int64_t total = rows * (c / 4);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, stream, x, rows, c / 4, ld, (unsigned*)ws, 0);
  hipLaunchKernelGGL(f8_scale_finish_kernel, dim3(1), dim3(1), 0, stream, (unsigned*)ws, scale);
  DCN_CHECK_LAUNCH("f8_scale");
  return DCN_OK;
}

// amax (float bits of a non-negative maximum) = max(amax, max|x|): the word the f16 two-piece split of the GEMM engines
// derives its power-of-two operand scale from.  Accumulating: call on every tensor that makes up one operand.
extern "C" int dcn_absmax(const float* x, int64_t rows, int c, int ld, uint32_t* amax, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (ld <= 0) ld = c;
  DCN_CHECK_ARG(x && amax && rows > 0 && c > 0 && c % 4 == 0 && ld % 4 == 0 && ld >= c, "absmax: bad argument (c=%d ld=%d)", c, ld);
  DCN_CHECK_ARG(((uintptr_t)x & 15) == 0, "absmax: x must be 16-byte aligned");
  const int64_t items = rows * (c / 4);
  const int blocks = (int)((items + 255) / 256 < 1024 ? (items + 255) / 256 : 1024);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, stream, x, rows, c / 4, ld, (unsigned*)amax, 1);
  DCN_CHECK_LAUNCH("absmax");
  return DCN_OK;
}

extern "C" int dcn_oihw_to_ohwi(const float* src, float* dst, int co, int ci, int kh, int kw, int ci_pad, void* stream) {
  DCN_CHECK_ARG(src && dst && co > 0 && ci > 0 && kh > 0 && kw > 0 && ci_pad >= ci, "oihw_to_ohwi: bad argument");
  // per output channel: [Ci][T] -> [T][ci_pad]
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(kh * kw, 32), cdiv(ci_pad, 32), co), dim3(256), 0, (hipStream_t)stream,
                     src, dst, ci, kh * kw, kh * kw, ci_pad, ci_pad, (int64_t)ci * kh * kw, (int64_t)kh * kw * ci_pad);
  DCN_CHECK_LAUNCH("oihw_to_ohwi");
  return DCN_OK;
}

extern "C" int dcn_ohwi_to_oihw(const float* src, float* dst, int co, int ci, int kh, int kw, int ci_pad, void* stream) {
  DCN_CHECK_ARG(src && dst && co > 0 && ci > 0 && kh > 0 && kw > 0 && ci_pad >= ci, "ohwi_to_oihw: bad argument");
  // per output channel: [T][ci_pad] (first ci columns) -> [Ci][T]
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(ci, 32), cdiv(kh * kw, 32), co), dim3(256), 0, (hipStream_t)stream,
                     src, dst, kh * kw, ci, ci_pad, kh * kw, kh * kw, (int64_t)kh * kw * ci_pad, (int64_t)ci * kh * kw);
  DCN_CHECK_LAUNCH("ohwi_to_oihw");
  return DCN_OK;
}

extern "C" int dcn_upsample2_nhwc(const float* src, int lds_, float* dst, int ldd, int n, int h, int w, int c, void* stream) {
  DCN_CHECK_ARG(src && dst && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && lds_ % 4 == 0 && ldd % 4 == 0, "upsample2: bad argument");
  hipLaunchKernelGGL(upsample2_kernel, dim3(stream_grid((int64_t)n * 4 * h * w * (c / 4))), dim3(256), 0, (hipStream_t)stream,
                     src, lds_, dst, ldd, n, h, w, c);
  DCN_CHECK_LAUNCH("upsample2");
  return DCN_OK;
}

extern "C" int dcn_upsample2_nhwc_bwd(const float* ddst, int ldd, float* dsrc, int lds_, int n, int h, int w, int c,
                                      int accumulate, void* stream) {
  DCN_CHECK_ARG(ddst && dsrc && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && lds_ % 4 == 0 && ldd % 4 == 0, "upsample2_bwd: bad argument");
  hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(stream_grid((int64_t)n * h * w * (c / 4))), dim3(256), 0, (hipStream_t)stream,
                     ddst, ldd, dsrc, lds_, n, h, w, c, accumulate);
  DCN_CHECK_LAUNCH("upsample2_bwd");
  return DCN_OK;
}

extern "C" int dcn_copy_slice(const float* src, int lds_, float* dst, int ldd, int64_t rows, int c, int accumulate, void* stream) {
  DCN_CHECK_ARG(src && dst && rows > 0 && c > 0 && c % 4 == 0 && lds_ % 4 == 0 && ldd % 4 == 0, "copy_slice: bad argument");
  hipLaunchKernelGGL(copy_slice_kernel, dim3(stream_grid(rows * (c / 4))), dim3(256), 0, (hipStream_t)stream,
                     src, lds_, dst, ldd, rows, c, accumulate);
  DCN_CHECK_LAUNCH("copy_slice");
  return DCN_OK;
}
