// bf16 storage (BASELINE.json configs[2]): the streaming passes around the bf16 convolutions — BatchNorm apply / backward, the
// route concat, casts at the boundary to fp32 consumers — on tensors that ARE bf16 in HBM.
//
// Same arithmetic as bn.hip / layout.hip, term by term, in fp32 registers; what changes is the traffic: 2 bytes per element
// read and written.  A thread handles 8 consecutive channels of a pixel (one 16-byte access on bf16, two on fp32), per-channel
// parameters (scale / shift, mean / invstd / gamma / beta, the backward sums) stay fp32.  Reference sites: nn.BatchNorm2d +
// LeakyReLU + shortcut of model/darknet.py:179-191,403-405 and their autograd; MyUpsample2 + route concat (:158-160,400-402).
// Roofline: HBM — scale_act 4 B/element (6 with a shortcut), backward apply 6 B/element, reduce 4 B/element.
#include "common.h"
#include "prof.h"

int bn_pc_enabled();         // bn.hip: dcn_set_tuning("Bpc", 0) turns the per-thread-channel apply passes off

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
struct F8 { float v[8]; };

template <typename T> __device__ __forceinline__ F8 ld8(const T* p);
template <> __device__ __forceinline__ F8 ld8<float>(const float* p) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  return F8{{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}};
}
template <> __device__ __forceinline__ F8 ld8<__bf16>(const __bf16* p) {
  const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(p);
  F8 r;
#pragma unroll
  for (int k = 0; k < 8; ++k) r.v[k] = (float)a[k];
  return r;
}
template <typename T> __device__ __forceinline__ void st8(T* p, const F8& x);
template <> __device__ __forceinline__ void st8<float>(float* p, const F8& x) {
  *reinterpret_cast<f32x4*>(p) = f32x4{x.v[0], x.v[1], x.v[2], x.v[3]};
  *reinterpret_cast<f32x4*>(p + 4) = f32x4{x.v[4], x.v[5], x.v[6], x.v[7]};
}
template <> __device__ __forceinline__ void st8<__bf16>(__bf16* p, const F8& x) {
  bf16x8_t a;
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = (__bf16)x.v[k];               // round to nearest even
  *reinterpret_cast<bf16x8_t*>(p) = a;
}

inline int grid_for(int64_t items) {
  int64_t b = (items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ---- dst[r][0..c) (+)= src[r][0..c)   (cast / strided copy / accumulate; rows of `c` elements, c % 8 == 0) ---------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_rows_kernel(const TS* __restrict__ src, int lds_, TD* __restrict__ dst, int ldd, int64_t rows, int c,
                                                        int accumulate) {
  const int c8 = c >> 3;
  const int64_t total = rows * c8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / c8; const int ch = (int)(i - r * c8) * 8;
    F8 v = ld8<TS>(src + r * lds_ + ch);
    if (accumulate) {
      const F8 o = ld8<TD>(dst + r * ldd + ch);
#pragma unroll
      for (int k = 0; k < 8; ++k) v.v[k] += o.v[k];
    }
    st8<TD>(dst + r * ldd + ch, v);
  }
}

// fp8 storage: the e4m3 copy of a row while it is written (the row = c8 consecutive lanes of one wave, 8 channels each; c8 a power of two
// <= 64): the arithmetic of quant_rows_e4m3_kernel below on the values AS STORED in bf16, so the copy equals that pass on the stored tensor.
__device__ __forceinline__ void quant8_row(const float* vals /* 8, already rounded to bf16 */, int c8, bool row_leader,
                                           unsigned char* __restrict__ qdst, unsigned char* __restrict__ sdst) {
  float amax = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) amax = fmaxf(amax, fabsf(vals[k]));
  for (int d = 1; d < c8; d <<= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
  int e = 0;
  const unsigned ab = __float_as_uint(amax);
  if (amax > 0.f && (ab >> 23) != 0xFFu) e = (int)(ab >> 23) - 127 - 8;
  e = e < -126 ? -126 : (e > 126 ? 126 : e);
  const float inv = __uint_as_float((unsigned)(127 - e) << 23);
  if (row_leader) *sdst = (unsigned char)(e + 127);
  float f[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) f[k] = fminf(fmaxf(vals[k] * inv, -448.f), 448.f);
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  *reinterpret_cast<uint2*>(qdst) = uint2{(unsigned)lo, (unsigned)hi};
}

// ---- out = act(scale * y + shift) + residual   (y: raw conv output, fp32 for the stem, bf16 elsewhere; out, residual: bf16) ------
template <typename TY, typename TO, bool Q = false>
__global__ __launch_bounds__(256) void scale_act16_kernel(const TY* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int act, float slope,
                                                          const __bf16* __restrict__ residual, int ldr, TO* __restrict__ out,
                                                          int64_t rows, int c, int ldo, unsigned char* __restrict__ q8 = nullptr,
                                                          unsigned char* __restrict__ qs = nullptr) {
  const int c8 = c >> 3;
  const int64_t total = rows * c8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / c8; const int ch = (int)(i - r * c8) * 8;
    F8 v = ld8<TY>(y + r * c + ch);
    F8 sc, sh;
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc.v[k] = 1.f; sh.v[k] = 0.f; }
    if (scale) sc = ld8<float>(scale + ch);
    if (shift) sh = ld8<float>(shift + ch);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t = __builtin_fmaf(v.v[k], sc.v[k], sh.v[k]);      // (spelled out: the same rounding in every instantiation)
      if (act == DCN_ACT_LEAKY) t = t > 0.f ? t : t * slope;
      v.v[k] = t;
    }
    if (residual) {
      const F8 rr = ld8<__bf16>(residual + r * ldr + ch);
#pragma unroll
      for (int k = 0; k < 8; ++k) v.v[k] += rr.v[k];
    }
    st8<TO>(out + r * ldo + ch, v);
    if constexpr (Q) {
      float vr[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) vr[k] = (float)(__bf16)v.v[k];
      quant8_row(vr, c8, ch == 0, q8 + r * c + ch, qs + r);
    }
  }
}

// ---- dy = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dout * act'(bn(y))   (bn.hip bn_act_bwd_apply_kernel) --------
template <typename TY, typename TD, bool Q = false>
__global__ __launch_bounds__(256) void bn_bwd_apply16_kernel(const TY* __restrict__ y, const TD* __restrict__ dout, int lddo,
                                                             const float* mean, const float* invstd, const float* gamma, const float* beta,
                                                             int act, float slope, const float* sums, float inv_count, int64_t rows, int c,
                                                             __bf16* __restrict__ dy, unsigned char* __restrict__ q8 = nullptr,
                                                             unsigned char* __restrict__ qs = nullptr) {
  const int c8 = c >> 3;
  const int64_t total = rows * c8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / c8; const int ch = (int)(i - r * c8) * 8;
    const F8 v = ld8<TY>(y + r * c + ch);
    const F8 d = ld8<TD>(dout + r * lddo + ch);
    const F8 mu = ld8<float>(mean + ch), is = ld8<float>(invstd + ch);
    F8 g, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { g.v[k] = 1.f; b.v[k] = 0.f; }
    if (gamma) g = ld8<float>(gamma + ch);
    if (beta) b = ld8<float>(beta + ch);
    const F8 sg = ld8<float>(sums + ch), sgx = ld8<float>(sums + c + ch);
    F8 o;
    {
      // (no FMA contraction: the Q build of this kernel must round exactly like the plain one — with contraction left to the optimiser the two
      //  instantiations differed in the last bit of one element in ten thousand)
#pragma clang fp contract(off)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (v.v[k] - mu.v[k]) * is.v[k];
        float dd = d.v[k];
        if (act == DCN_ACT_LEAKY && (g.v[k] * xh + b.v[k]) <= 0.f) dd *= slope;
        o.v[k] = g.v[k] * is.v[k] * (dd - sg.v[k] * inv_count - xh * sgx.v[k] * inv_count);
      }
    }
    st8<__bf16>(dy + r * c + ch, o);
    if constexpr (Q) {
      float vr[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) vr[k] = (float)(__bf16)o.v[k];
      quant8_row(vr, c8, ch == 0, q8 + r * c + ch, qs + r);
    }
  }
}

// ---- the two apply passes with the channel fixed per thread (round 5; bn.hip scale_act_pc_kernel has the reasoning) -----------------------
// c / 8 a power of two <= 256: thread t of a 256-thread group always lies in channel group t % (c / 8), so scale / shift (mean, invstd,
// gamma, beta, the two sums) are read ONCE per thread instead of once per 16-byte store, the row of an element group is a shift, and a
// workgroup takes chunks of four 256-group slices with four independent loads per tensor in flight.  Same arithmetic per element as the
// grid-stride kernels above (and the same fused e4m3 copy with Q).
constexpr int PC16_UNR = 4;
template <typename TY, typename TO, bool Q>
__global__ __launch_bounds__(256) void scale_act16_pc_kernel(const TY* __restrict__ y, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int act, float slope,
                                                             const __bf16* __restrict__ residual, int ldr, TO* __restrict__ out,
                                                             int64_t total, int lg_c8, int ldo, unsigned char* __restrict__ q8,
                                                             unsigned char* __restrict__ qs) {
  const int c8 = 1 << lg_c8, c = c8 * 8;
  const int ch = (threadIdx.x & (c8 - 1)) * 8;
  F8 sc, sh;
#pragma unroll
  for (int k = 0; k < 8; ++k) { sc.v[k] = 1.f; sh.v[k] = 0.f; }
  if (scale) sc = ld8<float>(scale + ch);
  if (shift) sh = ld8<float>(shift + ch);
  const int64_t nchunk = (total + 256 * PC16_UNR - 1) / (256 * PC16_UNR);
  for (int64_t cidx = blockIdx.x; cidx < nchunk; cidx += gridDim.x) {
    const int64_t i0 = cidx * (256 * PC16_UNR) + threadIdx.x;
    F8 v[PC16_UNR], rr[PC16_UNR];
#pragma unroll
    for (int u = 0; u < PC16_UNR; ++u) {
      const int64_t i = i0 + u * 256;
      if (i < total) {
        v[u] = ld8<TY>(y + (i >> lg_c8) * c + ch);
        if (residual) rr[u] = ld8<__bf16>(residual + (i >> lg_c8) * ldr + ch);
      }
    }
#pragma unroll
    for (int u = 0; u < PC16_UNR; ++u) {
      const int64_t i = i0 + u * 256;
      if (i >= total) continue;
      const int64_t r = i >> lg_c8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float t = __builtin_fmaf(v[u].v[k], sc.v[k], sh.v[k]);
        if (act == DCN_ACT_LEAKY) t = t > 0.f ? t : t * slope;
        v[u].v[k] = t;
      }
      if (residual) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[u].v[k] += rr[u].v[k];
      }
      st8<TO>(out + r * ldo + ch, v[u]);
      if constexpr (Q) {
        float vr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) vr[k] = (float)(__bf16)v[u].v[k];
        quant8_row(vr, c8 > 64 ? 64 : c8, ch == 0, q8 + r * c + ch, qs + r);
      }
    }
  }
}

template <typename TY, typename TD, bool Q>
__global__ __launch_bounds__(256) void bn_bwd_apply16_pc_kernel(const TY* __restrict__ y, const TD* __restrict__ dout, int lddo,
                                                                const float* mean, const float* invstd, const float* gamma, const float* beta,
                                                                int act, float slope, const float* sums, float inv_count, int64_t total, int lg_c8,
                                                                __bf16* __restrict__ dy, unsigned char* __restrict__ q8,
                                                                unsigned char* __restrict__ qs) {
  const int c8 = 1 << lg_c8, c = c8 * 8;
  const int ch = (threadIdx.x & (c8 - 1)) * 8;
  const F8 mu = ld8<float>(mean + ch), is = ld8<float>(invstd + ch);
  F8 g, b;
#pragma unroll
  for (int k = 0; k < 8; ++k) { g.v[k] = 1.f; b.v[k] = 0.f; }
  if (gamma) g = ld8<float>(gamma + ch);
  if (beta) b = ld8<float>(beta + ch);
  const F8 sg = ld8<float>(sums + ch), sgx = ld8<float>(sums + c + ch);
  const int64_t nchunk = (total + 256 * PC16_UNR - 1) / (256 * PC16_UNR);
  for (int64_t cidx = blockIdx.x; cidx < nchunk; cidx += gridDim.x) {
    const int64_t i0 = cidx * (256 * PC16_UNR) + threadIdx.x;
    F8 v[PC16_UNR], d[PC16_UNR];
#pragma unroll
    for (int u = 0; u < PC16_UNR; ++u) {
      const int64_t i = i0 + u * 256;
      if (i < total) { v[u] = ld8<TY>(y + (i >> lg_c8) * c + ch); d[u] = ld8<TD>(dout + (i >> lg_c8) * lddo + ch); }
    }
#pragma unroll
    for (int u = 0; u < PC16_UNR; ++u) {
      const int64_t i = i0 + u * 256;
      if (i >= total) continue;
      const int64_t r = i >> lg_c8;
      F8 o;
      {
#pragma clang fp contract(off)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float xh = (v[u].v[k] - mu.v[k]) * is.v[k];
          float dd = d[u].v[k];
          if (act == DCN_ACT_LEAKY && (g.v[k] * xh + b.v[k]) <= 0.f) dd *= slope;
          o.v[k] = g.v[k] * is.v[k] * (dd - sg.v[k] * inv_count - xh * sgx.v[k] * inv_count);
        }
      }
      st8<__bf16>(dy + r * c + ch, o);
      if constexpr (Q) {
        float vr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) vr[k] = (float)(__bf16)o.v[k];
        quant8_row(vr, c8 > 64 ? 64 : c8, ch == 0, q8 + r * c + ch, qs + r);
      }
    }
  }
}

// log2(c / 8) when the per-thread-channel kernels take the width (c / 8 a power of two <= 256), else -1; dcn_set_tuning("Bpc", 0) turns them off
inline int pc16_lg(int c) {
  const int c8 = c / 8;
  if (!bn_pc_enabled() || c % 8 || c8 < 1 || c8 > 256 || (c8 & (c8 - 1))) return -1;
  int lg = 0;
  while ((1 << lg) < c8) ++lg;
  return lg;
}
inline int pc16_grid(int64_t total) {
  const int64_t nchunk = (total + 256 * PC16_UNR - 1) / (256 * PC16_UNR);
  return (int)(nchunk < 1 ? 1 : (nchunk > 4096 ? 4096 : nchunk));
}

// ---- per-channel partial sums of g and g * xhat over 128-row blocks (bn.hip channel_partials_kernel<1>) ----------------------------
// block (row block, 128 channels): thread = (8 channels, row phase of 16); stats [row blocks][2][c]
template <typename TY, typename TD>
__global__ __launch_bounds__(256) void partials16_kernel(const TY* __restrict__ y, const TD* __restrict__ dout, int lddo, const float* mean,
                                                         const float* invstd, const float* gamma, const float* beta, int act, float slope,
                                                         int64_t rows, int c, float* __restrict__ stats) {
  __shared__ float red[2][16][128];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int ch = blockIdx.y * 128 + tx * 8;
  const int64_t r0 = (int64_t)blockIdx.x * 128;
  F8 s, ss;
#pragma unroll
  for (int k = 0; k < 8; ++k) { s.v[k] = 0.f; ss.v[k] = 0.f; }
  if (ch < c) {
    const F8 mu = ld8<float>(mean + ch), is = ld8<float>(invstd + ch);
    F8 g, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { g.v[k] = 1.f; b.v[k] = 0.f; }
    if (gamma) g = ld8<float>(gamma + ch);
    if (beta) b = ld8<float>(beta + ch);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t r = r0 + ty + 16 * j;
      if (r >= rows) continue;
      const F8 v = ld8<TY>(y + r * c + ch);
      const F8 d = ld8<TD>(dout + r * lddo + ch);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (v.v[k] - mu.v[k]) * is.v[k];
        float dd = d.v[k];
        if (act == DCN_ACT_LEAKY && g.v[k] * xh + b.v[k] <= 0.f) dd *= slope;
        s.v[k] += dd; ss.v[k] += dd * xh;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][ty][tx * 8 + k] = s.v[k]; red[1][ty][tx * 8 + k] = ss.v[k]; }
  __syncthreads();
  {
    const int which = threadIdx.x >> 7, t = threadIdx.x & 127, cc = blockIdx.y * 128 + t;
    if (cc < c) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc += red[which][k][t];
      stats[((size_t)blockIdx.x * 2 + which) * c + cc] = acc;
    }
  }
}

// ---- nearest x2 into a channel slice; its backward (sum of the 2 x 2 children) ----------------------------------------------------------
__global__ __launch_bounds__(256) void upsample2_16_kernel(const __bf16* __restrict__ src, int lds_, __bf16* __restrict__ dst, int ldd, int n,
                                                           int h, int w, int c) {
  const int c8 = c >> 3;
  const int64_t total = (int64_t)n * 2 * h * 2 * w * c8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c8) * 8;
    int64_t pix = i / c8;
    const int x = (int)(pix % (2 * w)); pix /= 2 * w;
    const int yy = (int)(pix % (2 * h)); const int64_t b = pix / (2 * h);
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(src + ((b * h + (yy >> 1)) * w + (x >> 1)) * lds_ + ch);
    *reinterpret_cast<bf16x8_t*>(dst + ((b * 2 * h + yy) * 2 * w + x) * ldd + ch) = v;
  }
}
__global__ __launch_bounds__(256) void upsample2_bwd16_kernel(const __bf16* __restrict__ ddst, int ldd, __bf16* __restrict__ dsrc, int lds_,
                                                              int n, int h, int w, int c, int accumulate) {
  const int c8 = c >> 3;
  const int64_t total = (int64_t)n * h * w * c8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % c8) * 8;
    int64_t pix = i / c8;
    const int x = (int)(pix % w); pix /= w;
    const int yy = (int)(pix % h); const int64_t b = pix / h;
    F8 s;
#pragma unroll
    for (int k = 0; k < 8; ++k) s.v[k] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const F8 v = ld8<__bf16>(ddst + ((b * 2 * h + 2 * yy + dy) * 2 * w + 2 * x + dx) * ldd + ch);
#pragma unroll
        for (int k = 0; k < 8; ++k) s.v[k] += v.v[k];
      }
    __bf16* o = dsrc + ((b * h + yy) * w + x) * lds_ + ch;
    if (accumulate) {
      const F8 v = ld8<__bf16>(o);
#pragma unroll
      for (int k = 0; k < 8; ++k) s.v[k] += v.v[k];
    }
    st8<__bf16>(o, s);
  }
}

}  // namespace

// src / dst element types: 0 = fp32, 1 = bf16.  dst[r][:c] (+)= src[r][:c] for `rows` rows with element strides lds / ldd.
extern "C" int dcn_cast_rows(const void* src, int src_b16, int lds_, void* dst, int dst_b16, int ldd, int64_t rows, int c, int accumulate,
                             void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (lds_ <= 0) lds_ = c;
  if (ldd <= 0) ldd = c;
  DCN_CHECK_ARG(src && dst && rows > 0 && c > 0 && c % 8 == 0 && lds_ % 8 == 0 && ldd % 8 == 0 && lds_ >= c && ldd >= c,
                "cast_rows: bad argument (c=%d lds=%d ldd=%d must be multiples of 8)", c, lds_, ldd);
  DCN_CHECK_ARG((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "cast_rows: pointers must be 16-byte aligned");
  const dim3 g(grid_for(rows * (c / 8)));
  if (src_b16 && dst_b16)
    hipLaunchKernelGGL((cast_rows_kernel<__bf16, __bf16>), g, dim3(256), 0, stream, (const __bf16*)src, lds_, (__bf16*)dst, ldd, rows, c, accumulate);
  else if (src_b16)
    hipLaunchKernelGGL((cast_rows_kernel<__bf16, float>), g, dim3(256), 0, stream, (const __bf16*)src, lds_, (float*)dst, ldd, rows, c, accumulate);
  else if (dst_b16)
    hipLaunchKernelGGL((cast_rows_kernel<float, __bf16>), g, dim3(256), 0, stream, (const float*)src, lds_, (__bf16*)dst, ldd, rows, c, accumulate);
  else
    hipLaunchKernelGGL((cast_rows_kernel<float, float>), g, dim3(256), 0, stream, (const float*)src, lds_, (float*)dst, ldd, rows, c, accumulate);
  DCN_CHECK_LAUNCH("cast_rows");
  return DCN_OK;
}

extern "C" int dcn_scale_act_b16(const void* y, int y_f32, const float* scale, const float* shift, int act, float slope, const void* residual,
                                 int ldr, void* out, int out_f32, int64_t rows, int c, int ldo, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (ldo <= 0) ldo = c;
  if (ldr <= 0) ldr = c;
  DCN_CHECK_ARG(y && out && rows > 0 && c > 0 && c % 8 == 0 && ldo % 8 == 0 && ldr % 8 == 0, "scale_act_b16: bad argument (c=%d)", c);
  const int pid = prof_begin(43, (double)rows * c * ((y_f32 ? 4.0 : 2.0) + (out_f32 ? 4.0 : 2.0) + (residual ? 2.0 : 0.0)), stream);
  const dim3 g(grid_for(rows * (c / 8)));
  const __bf16* res = (const __bf16*)residual;
  const int lg = pc16_lg(c);
  const int64_t total = rows * (c / 8);
#define DCN_SA(TY, TO) do { if (lg >= 0) hipLaunchKernelGGL((scale_act16_pc_kernel<TY, TO, false>), dim3(pc16_grid(total)), dim3(256), 0, stream, (const TY*)y, scale, shift, act, slope, res, ldr, (TO*)out, total, lg, ldo, nullptr, nullptr); \
    else hipLaunchKernelGGL((scale_act16_kernel<TY, TO>), g, dim3(256), 0, stream, (const TY*)y, scale, shift, act, slope, res, ldr, (TO*)out, rows, c, ldo); } while (0)
  if (y_f32) { if (out_f32) DCN_SA(float, float); else DCN_SA(float, __bf16); }
  else { if (out_f32) DCN_SA(__bf16, float); else DCN_SA(__bf16, __bf16); }
#undef DCN_SA
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("scale_act_b16");
  return DCN_OK;
}

// fp8 storage: can the e4m3 copy of a [rows][c] bf16 tensor be written by the pass that writes the tensor?  (a row inside one wave)
extern "C" int dcn_quant_fusable(int c) { return (c % 8 == 0 && c / 8 <= 64 && ((c / 8) & (c / 8 - 1)) == 0) ? 1 : 0; }

// dcn_scale_act_b16 on bf16 y / out (dense), writing the e4m3 copy of `out` (q8 [rows][c], qs [rows]: dcn_quant_rows_e4m3's result) as well
extern "C" int dcn_scale_act_b16_q(const void* y, const float* scale, const float* shift, int act, float slope, const void* residual, int ldr,
                                   void* out, int64_t rows, int c, void* q8, void* qs, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (ldr <= 0) ldr = c;
  DCN_CHECK_ARG(y && out && q8 && qs && rows > 0 && dcn_quant_fusable(c) && ldr % 8 == 0, "scale_act_b16_q: bad argument (c=%d)", c);
  const int pid = prof_begin(43, (double)rows * c * (2.0 + 2.0 + 1.0 + (residual ? 2.0 : 0.0)), stream);
  const int lg = pc16_lg(c);
  if (lg >= 0)
    hipLaunchKernelGGL((scale_act16_pc_kernel<__bf16, __bf16, true>), dim3(pc16_grid(rows * (c / 8))), dim3(256), 0, stream, (const __bf16*)y, scale, shift,
                       act, slope, (const __bf16*)residual, ldr, (__bf16*)out, rows * (c / 8), lg, c, (unsigned char*)q8, (unsigned char*)qs);
  else
    hipLaunchKernelGGL((scale_act16_kernel<__bf16, __bf16, true>), dim3(grid_for(rows * (c / 8))), dim3(256), 0, stream, (const __bf16*)y, scale, shift, act,
                       slope, (const __bf16*)residual, ldr, (__bf16*)out, rows, c, c, (unsigned char*)q8, (unsigned char*)qs);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("scale_act_b16_q");
  return DCN_OK;
}

// dcn_bn_act_bwd_apply_b16 on bf16 y / dout, writing the e4m3 copy of dy as well
extern "C" int dcn_bn_act_bwd_apply_b16_q(const void* y, const void* dout, int lddo, const float* mean, const float* invstd, const float* gamma,
                                          const float* beta, int act, float slope, const float* sums, int64_t count, int64_t rows, int c, void* dy,
                                          void* q8, void* qs, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (lddo <= 0) lddo = c;
  DCN_CHECK_ARG(y && dout && mean && invstd && sums && dy && q8 && qs && rows > 0 && dcn_quant_fusable(c) && lddo % 8 == 0 && count > 0,
                "bn_act_bwd_apply_b16_q: bad argument (c=%d)", c);
  const int pid = prof_begin(45, (double)rows * c * 7.0, stream);
  const int lg = pc16_lg(c);
  if (lg >= 0)
    hipLaunchKernelGGL((bn_bwd_apply16_pc_kernel<__bf16, __bf16, true>), dim3(pc16_grid(rows * (c / 8))), dim3(256), 0, stream, (const __bf16*)y,
                       (const __bf16*)dout, lddo, mean, invstd, gamma, beta, act, slope, sums, 1.f / (float)count, rows * (c / 8), lg, (__bf16*)dy,
                       (unsigned char*)q8, (unsigned char*)qs);
  else
    hipLaunchKernelGGL((bn_bwd_apply16_kernel<__bf16, __bf16, true>), dim3(grid_for(rows * (c / 8))), dim3(256), 0, stream, (const __bf16*)y,
                       (const __bf16*)dout, lddo, mean, invstd, gamma, beta, act, slope, sums, 1.f / (float)count, rows, c, (__bf16*)dy,
                       (unsigned char*)q8, (unsigned char*)qs);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("bn_act_bwd_apply_b16_q");
  return DCN_OK;
}

extern "C" int dcn_bn_act_bwd_reduce_rows_b16(int64_t rows) { return cdiv(rows, 128); }

extern "C" int dcn_bn_act_bwd_reduce_b16(const void* y, int y_f32, const void* dout, int dout_f32, int lddo, const float* mean, const float* invstd,
                                         const float* gamma, const float* beta, int act, float slope, int64_t rows, int c, float* stats,
                                         void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (lddo <= 0) lddo = c;
  DCN_CHECK_ARG(y && dout && mean && invstd && stats && rows > 0 && c > 0 && c % 8 == 0 && lddo % 8 == 0, "bn_act_bwd_reduce_b16: bad argument");
  const int pid = prof_begin(44, (double)rows * c * ((y_f32 ? 4.0 : 2.0) + (dout_f32 ? 4.0 : 2.0)), stream);
  const dim3 g(cdiv(rows, 128), cdiv(c, 128));
#define DCN_PT(TY, TD) hipLaunchKernelGGL((partials16_kernel<TY, TD>), g, dim3(256), 0, stream, (const TY*)y, (const TD*)dout, lddo, mean, invstd, gamma, beta, act, slope, rows, c, stats)
  if (y_f32) { if (dout_f32) DCN_PT(float, float); else DCN_PT(float, __bf16); }
  else { if (dout_f32) DCN_PT(__bf16, float); else DCN_PT(__bf16, __bf16); }
#undef DCN_PT
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("bn_act_bwd_reduce_b16");
  return DCN_OK;
}

extern "C" int dcn_bn_act_bwd_apply_b16(const void* y, int y_f32, const void* dout, int dout_f32, int lddo, const float* mean, const float* invstd,
                                        const float* gamma, const float* beta, int act, float slope, const float* sums, int64_t count,
                                        int64_t rows, int c, void* dy, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (lddo <= 0) lddo = c;
  DCN_CHECK_ARG(y && dout && mean && invstd && sums && dy && rows > 0 && c > 0 && c % 8 == 0 && lddo % 8 == 0 && count > 0,
                "bn_act_bwd_apply_b16: bad argument");
  const int pid = prof_begin(45, (double)rows * c * ((y_f32 ? 4.0 : 2.0) + (dout_f32 ? 4.0 : 2.0) + 2.0), stream);
  const dim3 g(grid_for(rows * (c / 8)));
  const int lg = pc16_lg(c);
  const int64_t total = rows * (c / 8);
#define DCN_AP(TY, TD) do { if (lg >= 0) hipLaunchKernelGGL((bn_bwd_apply16_pc_kernel<TY, TD, false>), dim3(pc16_grid(total)), dim3(256), 0, stream, (const TY*)y, (const TD*)dout, lddo, mean, invstd, gamma, beta, act, slope, sums, 1.f / (float)count, total, lg, (__bf16*)dy, nullptr, nullptr); \
    else hipLaunchKernelGGL((bn_bwd_apply16_kernel<TY, TD>), g, dim3(256), 0, stream, (const TY*)y, (const TD*)dout, lddo, mean, invstd, gamma, beta, act, slope, sums, 1.f / (float)count, rows, c, (__bf16*)dy); } while (0)
  if (y_f32) { if (dout_f32) DCN_AP(float, float); else DCN_AP(float, __bf16); }
  else { if (dout_f32) DCN_AP(__bf16, float); else DCN_AP(__bf16, __bf16); }
#undef DCN_AP
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("bn_act_bwd_apply_b16");
  return DCN_OK;
}

extern "C" int dcn_upsample2_nhwc_b16(const void* src, int lds_, void* dst, int ldd, int n, int h, int w, int c, void* stream) {
  DCN_CHECK_ARG(src && dst && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && lds_ % 8 == 0 && ldd % 8 == 0, "upsample2_b16: bad argument");
  hipLaunchKernelGGL(upsample2_16_kernel, dim3(grid_for((int64_t)n * 4 * h * w * (c / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)src, lds_, (__bf16*)dst, ldd, n, h, w, c);
  DCN_CHECK_LAUNCH("upsample2_b16");
  return DCN_OK;
}

extern "C" int dcn_upsample2_nhwc_bwd_b16(const void* ddst, int ldd, void* dsrc, int lds_, int n, int h, int w, int c, int accumulate,
                                          void* stream) {
  DCN_CHECK_ARG(ddst && dsrc && n > 0 && h > 0 && w > 0 && c > 0 && c % 8 == 0 && lds_ % 8 == 0 && ldd % 8 == 0, "upsample2_bwd_b16: bad argument");
  hipLaunchKernelGGL(upsample2_bwd16_kernel, dim3(grid_for((int64_t)n * h * w * (c / 8))), dim3(256), 0, (hipStream_t)stream,
                     (const __bf16*)ddst, ldd, (__bf16*)dsrc, lds_, n, h, w, c, accumulate);
  DCN_CHECK_LAUNCH("upsample2_bwd_b16");
  return DCN_OK;
}

// ---- fp8 storage (BASELINE.json configs[4]): bf16 rows -> OCP e4m3 bytes + ONE e8m0 scale per row ---------------------------------------
// The operand format of conv1.hip's conv1b_kernel<..., F8>: a row is a pixel's channel vector (activations, gradients) or a filter's
// k*k*Cin coefficients (banks).  Scale as in the OCP MX formats, with the block = the row: e = floor(log2(max|row|)) - 8 (e4m3's largest
// binade), q = e4m3(x * 2^-e) rounded to nearest even and clamped to +-448, scale byte = e + 127; an all-zero row gets byte 127.
namespace {
__global__ __launch_bounds__(256) void quant_rows_e4m3_kernel(const __bf16* __restrict__ x, int ld, int64_t rows, int c, int lanes,
                                                              unsigned char* __restrict__ q, int ldq, unsigned char* __restrict__ scales) {
  // `lanes` (a power of two <= 64) lanes per row, 8 channels per lane and pass
  const int lane = threadIdx.x & 63, sub = lane & (lanes - 1);
  const int64_t rows_per_wave = 64 / lanes;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = (int64_t)gridDim.x * 4;
  for (int64_t r0 = wave * rows_per_wave; r0 < rows; r0 += nwaves * rows_per_wave) {
    const int64_t r = r0 + lane / lanes;
    const bool ok = r < rows;
    const __bf16* xr = x + (ok ? r : 0) * ld;
    float amax = 0.f;
    for (int ch = sub * 8; ch < c; ch += lanes * 8) {
      const uint4 v = ok ? *reinterpret_cast<const uint4*>(xr + ch) : uint4{0u, 0u, 0u, 0u};
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        amax = fmaxf(amax, fabsf(__uint_as_float(w[k] << 16)));
        amax = fmaxf(amax, fabsf(__uint_as_float(w[k] & 0xFFFF0000u)));
      }
    }
    for (int d = 1; d < lanes; d <<= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
    int e = 0;
    const unsigned ab = __float_as_uint(amax);
    if (amax > 0.f && (ab >> 23) != 0xFFu) e = (int)(ab >> 23) - 127 - 8;         // floor(log2(amax)) - 8 (bf16 inputs are never fp32-subnormal unless zero)
    e = e < -126 ? -126 : (e > 126 ? 126 : e);
    const float inv = __uint_as_float((unsigned)(127 - e) << 23);                 // 2^-e, exact
    if (ok && sub == 0) scales[r] = (unsigned char)(e + 127);
    unsigned char* qr = q + (ok ? r : 0) * ldq;
    for (int ch = sub * 8; ch < c; ch += lanes * 8) {
      if (!ok) continue;
      const uint4 v = *reinterpret_cast<const uint4*>(xr + ch);
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
      float f[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f[2 * k] = fminf(fmaxf(__uint_as_float(w[k] << 16) * inv, -448.f), 448.f);
        f[2 * k + 1] = fminf(fmaxf(__uint_as_float(w[k] & 0xFFFF0000u) * inv, -448.f), 448.f);
      }
      int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
      int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
      *reinterpret_cast<uint2*>(qr + ch) = uint2{(unsigned)lo, (unsigned)hi};
    }
  }
}
}  // namespace

// x: bf16 [rows][c] (row stride ld elements) -> q: e4m3 bytes [rows][c] (row stride ldq), scales: e8m0 [rows]
extern "C" int dcn_quant_rows_e4m3(const void* x, int ld, int64_t rows, int c, void* q, int ldq, void* scales, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (ld <= 0) ld = c;
  if (ldq <= 0) ldq = c;
  DCN_CHECK_ARG(x && q && scales && rows > 0 && c > 0 && c % 8 == 0 && ld % 8 == 0 && ldq % 8 == 0 && ld >= c && ldq >= c,
                "quant_rows_e4m3: bad argument (c=%d ld=%d ldq=%d must be multiples of 8)", c, ld, ldq);
  DCN_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)q & 7) == 0, "quant_rows_e4m3: x must be 16-byte, q 8-byte aligned");
  int lanes = 1;
  while (lanes < 64 && lanes * 8 < c) lanes <<= 1;
  const int64_t rows_per_block = 4 * (64 / lanes);
  int64_t g = (rows + rows_per_block - 1) / rows_per_block;
  if (g > 8192) g = 8192;
  const int pid = prof_begin(49, (double)rows * c * 3.0, stream);
  hipLaunchKernelGGL(quant_rows_e4m3_kernel, dim3((int)g), dim3(256), 0, stream, (const __bf16*)x, ld, rows, c, lanes, (unsigned char*)q, ldq,
                     (unsigned char*)scales);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("quant_rows_e4m3");
  return DCN_OK;
}

// ---- fp8 storage: every bank of the step in ONE launch (round 6; round-5 advice) ---------------------------------------------------------
// The e4m3 forms of the 3x3 layers' banks (forward [Cout][k*k*Cin] and transposed [Cin][k*k*Cout], one e8m0 scale per row) used to be made
// by a quant_rows_e4m3 launch per layer in front of its forward convolution and another in front of its data gradient: 77 launches of
// ~10 us on the critical chain of the step (0.75 ms of 53).  The banks change once per step (FilterBanks.refresh): one job table, one launch,
// one wave per row — the arithmetic of quant_rows_e4m3_kernel with 64 lanes per row, bit for bit (tests/test_f8_gpu.py).
namespace {
struct QuantJob { const __bf16* src; unsigned char* q; unsigned char* scales; int rows, c, first_block, pad; };
__global__ __launch_bounds__(256) void quant_banks_kernel(const QuantJob* __restrict__ jobs, int njobs) {
  int j = 0;
  while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].first_block) ++j;           // (<= ~80 jobs, wave-uniform)
  const QuantJob jb = jobs[j];
  const int lane = threadIdx.x & 63;
  const int r = ((int)blockIdx.x - jb.first_block) * 4 + (threadIdx.x >> 6);
  if (r >= jb.rows) return;
  const __bf16* xr = jb.src + (size_t)r * jb.c;
  float amax = 0.f;
  for (int ch = lane * 8; ch < jb.c; ch += 512) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + ch);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      amax = fmaxf(amax, fabsf(__uint_as_float(w[k] << 16)));
      amax = fmaxf(amax, fabsf(__uint_as_float(w[k] & 0xFFFF0000u)));
    }
  }
  for (int d = 1; d < 64; d <<= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
  int e = 0;
  const unsigned ab = __float_as_uint(amax);
  if (amax > 0.f && (ab >> 23) != 0xFFu) e = (int)(ab >> 23) - 127 - 8;
  e = e < -126 ? -126 : (e > 126 ? 126 : e);
  const float inv = __uint_as_float((unsigned)(127 - e) << 23);
  if (lane == 0) jb.scales[r] = (unsigned char)(e + 127);
  unsigned char* qr = jb.q + (size_t)r * jb.c;
  for (int ch = lane * 8; ch < jb.c; ch += 512) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + ch);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    float f[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f[2 * k] = fminf(fmaxf(__uint_as_float(w[k] << 16) * inv, -448.f), 448.f);
      f[2 * k + 1] = fminf(fmaxf(__uint_as_float(w[k] & 0xFFFF0000u) * inv, -448.f), 448.f);
    }
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    *reinterpret_cast<uint2*>(qr + ch) = uint2{(unsigned)lo, (unsigned)hi};
  }
}
}  // namespace

extern "C" int dcn_quant_job_bytes() { return (int)sizeof(QuantJob); }
// jobs: device array of njobs records {src bf16 [rows][c] dense, q e4m3 [rows][c], scales [rows], rows, c (a multiple of 8), first_block, 0}
// with first_block = the running sum of ceil(rows / 4); blocks = that sum over all jobs; elements = sum of rows * c (for the profiler)
extern "C" int dcn_quant_rows_e4m3_batched(const void* jobs, int njobs, int blocks, int64_t elements, void* stream_) {
  DCN_CHECK_ARG(jobs && njobs > 0 && blocks > 0, "quant_rows_e4m3_batched: bad argument");
  const int pid = prof_begin(49, (double)elements * 3.0, (hipStream_t)stream_);
  hipLaunchKernelGGL(quant_banks_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream_, (const QuantJob*)jobs, njobs);
  prof_end(pid, (hipStream_t)stream_);
  DCN_CHECK_LAUNCH("quant_rows_e4m3_batched");
  return DCN_OK;
}
