// BatchNorm statistics / apply / backward and the fused elementwise passes around the convs.
// All of these are HBM-bound streaming kernels over NHWC [rows][c] tensors (roofline: HBM,
// 4 B read + 4 B written per element for the apply passes); 16-B accesses per lane, channel =
// fastest dim so per-channel parameters are read once per thread and stay in registers.
#include "common.h"
#include "prof.h"

namespace {

constexpr int RS_MAX = 64;   // row slices of the two-stage column sum

// ---- two-stage column sum of a [rows][cols] fp32 matrix into double ---------------------------
// stage 1: grid (cols/32, RS); 256 threads = 8 row lanes x 32 columns
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ in, int rows, int cols, int rs,
                                                     double* __restrict__ ws) {
  __shared__ double red[8][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + tx;
  const int per = (rows + rs - 1) / rs;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  double s = 0.0;
  if (col < cols)
    for (int r = r0 + ty; r < r1; r += 8) s += (double)in[(size_t)r * cols + col];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && col < cols) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][tx];
    ws[(size_t)blockIdx.y * cols + col] = t;
  }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ ws, int rs, int c, double inv_count,
                                                          double unbias, const float* gamma, const float* beta, float eps,
                                                          float momentum, float* running_mean, float* running_var,
                                                          float* mean, float* invstd, float* scale, float* shift) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= c) return;
  double s = 0.0, ss = 0.0;
  for (int k = 0; k < rs; ++k) { s += ws[(size_t)k * 2 * c + ch]; ss += ws[(size_t)k * 2 * c + c + ch]; }
  const double m = s * inv_count;
  double var = ss * inv_count - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[ch] = (float)m; invstd[ch] = is;
  const float g = gamma ? gamma[ch] : 1.f, b = beta ? beta[ch] : 0.f;
  const float sc = g * is;
  scale[ch] = sc; shift[ch] = b - (float)m * sc;
  if (running_mean) {
    running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)m;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)(var * unbias);
  }
}

__global__ __launch_bounds__(256) void sums_finalize_kernel(const double* __restrict__ ws, int rs, int cols, float* out) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= cols) return;
  double s = 0.0;
  for (int k = 0; k < rs; ++k) s += ws[(size_t)k * cols + col];
  out[col] = (float)s;
}

// ---- one-launch reduce + finalize for short partial lists -------------------------------------------------------
// Most layers hand over <= 4096 partial rows (one per 128-row tile): one block per 32 channels sums them in a fixed
// order (1024 threads = 32 row lanes x 32 channels, doubles) and finalises its channels itself — one launch instead of
// the two-stage pair, on the critical path between every conv and its BatchNorm apply.  MODE 0: BatchNorm statistics
// (same arithmetic as bn_finalize_kernel); MODE 1: the two backward sums.
constexpr int FUSED_ROWS_MAX = 4096;
// Block = 256 threads = 32 row lanes x RF_CH channels (round 3; was 1024 threads = 32 x 32): under the replayed step the kernel's few
// 16-wave blocks waited for a CU with 16 free wave slots beside the weight-gradient stream (42.8 us per launch on the critical
// chain of the backward instead of 9-13); 4-wave blocks find a slot at once.  Rows per lane and the order of the 32 lane sums
// are unchanged: bitwise the same result.
constexpr int RF_CH = 8;
template <int MODE>
__global__ __launch_bounds__(256) void reduce_finalize_kernel(const float* __restrict__ stats, int rows, int c, double inv_count,
                                                              double unbias, const float* gamma, const float* beta, float eps,
                                                              float momentum, float* running_mean, float* running_var,
                                                              float* mean, float* invstd, float* scale, float* shift,
                                                              float* sums) {
  __shared__ double red[2][32][RF_CH + 1];
  const int tx = threadIdx.x & (RF_CH - 1), ty = threadIdx.x / RF_CH;
  const int ch = blockIdx.x * RF_CH + tx;
  double s = 0.0, ss = 0.0;
  if (ch < c) {
    // eight rows (sixteen independent loads) in flight per thread, summed in the same order as the plain loop: the kernel sits on
    // the critical path between every convolution and its BatchNorm apply and is pure load latency (42 dependent-free iterations
    // of two loads each took 12.7 us on the 52x52 layers)
    int r = ty;
    for (; r + 7 * 32 < rows; r += 8 * 32) {
      float a[8], b[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a[k] = stats[((size_t)(r + 32 * k) * 2 + 0) * c + ch];
        b[k] = stats[((size_t)(r + 32 * k) * 2 + 1) * c + ch];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) { s += (double)a[k]; ss += (double)b[k]; }
    }
    for (; r < rows; r += 32) {
      s += (double)stats[((size_t)r * 2 + 0) * c + ch];
      ss += (double)stats[((size_t)r * 2 + 1) * c + ch];
    }
  }
  red[0][ty][tx] = s; red[1][ty][tx] = ss;
  __syncthreads();
  if (ty != 0 || ch >= c) return;
  s = 0.0; ss = 0.0;
#pragma unroll
  for (int k = 0; k < 32; ++k) { s += red[0][k][tx]; ss += red[1][k][tx]; }
  if (MODE == 1) { sums[ch] = (float)s; sums[c + ch] = (float)ss; return; }
  const double m = s * inv_count;
  double var = ss * inv_count - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[ch] = (float)m; invstd[ch] = is;
  const float g = gamma ? gamma[ch] : 1.f, b = beta ? beta[ch] : 0.f;
  const float sc = g * is;
  scale[ch] = sc; shift[ch] = b - (float)m * sc;
  if (running_mean) {
    running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)m;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)(var * unbias);
  }
}

__global__ __launch_bounds__(256) void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                                      float eps, int c, float* scale, float* shift) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= c) return;
  const float sc = gamma[ch] / sqrtf(rv[ch] + eps);
  scale[ch] = sc; shift[ch] = beta[ch] - rm[ch] * sc;
}

// ---- per-channel partial sums of an NHWC tensor -------------------------------------------------
// grid (rows/128, c/64): 256 threads = 16 row lanes x 16 channel quads, 8 rows per thread, every load a 16-B vector
// (the one-float-per-lane form of this kernel ran at 2.4 TB/s; HBM-bound streaming wants dwordx4).
// MODE 0: sum(x), sum(x^2).   MODE 1 (BN+act backward): sum(g), sum(g*xhat), g = dout*act'(bn(y)).
// Requires c % 4 == 0 and 16-B aligned rows (checked by the callers).
template <int MODE>
__global__ __launch_bounds__(256) void channel_partials_kernel(const float* __restrict__ x, int ld, const float* __restrict__ dout,
                                                               int lddo, const float* mean, const float* invstd,
                                                               const float* gamma, const float* beta, int act, float slope,
                                                               int64_t rows, int c, float* __restrict__ stats, int rev) {
  __shared__ float red[2][16][64];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int ch = blockIdx.y * 64 + tx * 4;
  const int bx = rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;      // (sweep direction: see g_bn_rev)
  const int64_t r0 = (int64_t)bx * 128;
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
  if (ch < c) {
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {1.f, 1.f, 1.f, 1.f}, g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
      mu = *reinterpret_cast<const f32x4*>(mean + ch); is = *reinterpret_cast<const f32x4*>(invstd + ch);
      if (gamma) g = *reinterpret_cast<const f32x4*>(gamma + ch);
      if (beta) b = *reinterpret_cast<const f32x4*>(beta + ch);
    }
    f32x4 v[8], d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t r = r0 + ty + 16 * k;
      const bool ok = r < rows;
      v[k] = ok ? *reinterpret_cast<const f32x4*>(x + r * ld + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (MODE == 1) d[k] = ok ? *reinterpret_cast<const f32x4*>(dout + r * lddo + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 0) { s += v[k]; ss += v[k] * v[k]; }
      else {
        const bool ok = r0 + ty + 16 * k < rows;
        const f32x4 xh = (v[k] - mu) * is;
        f32x4 dd = d[k];
        if (act == DCN_ACT_LEAKY) {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (g[e] * xh[e] + b[e] <= 0.f) dd[e] *= slope;
        }
        if (ok) { s += dd; ss += dd * xh; }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][ty][tx * 4 + e] = s[e]; red[1][ty][tx * 4 + e] = ss[e]; }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int which = threadIdx.x >> 6, t = threadIdx.x & 63, cc = blockIdx.y * 64 + t;
    if (cc < c) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) acc += red[which][k][t];
      stats[((size_t)bx * 2 + which) * c + cc] = acc;
    }
  }
}

// ---- out = act(scale*y + shift) + residual ---------------------------------------------------------
__global__ __launch_bounds__(256) void scale_act_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int act, float slope,
                                                        const float* __restrict__ residual, float* __restrict__ out,
                                                        int64_t rows, int c, int ldo, unsigned* __restrict__ amax, int rev) {
  const int c4 = c >> 2;
  const int64_t total = rows * c4;
  float vmax = 0.f;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < total; j += (int64_t)gridDim.x * 256) {
    const int64_t i = rev ? total - 1 - j : j;
    const int64_t r = i / c4; const int ch = (int)(i - r * c4) * 4;
    f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y + r * c + ch));      // (not read again before the backward)
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (scale) sc = *reinterpret_cast<const f32x4*>(scale + ch);
    if (shift) sh = *reinterpret_cast<const f32x4*>(shift + ch);
    v = v * sc + sh;
    if (act == DCN_ACT_LEAKY) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : v[k] * slope;
    }
    if (residual) v += *reinterpret_cast<const f32x4*>(residual + r * c + ch);
    *reinterpret_cast<f32x4*>(out + r * ldo + ch) = v;
    vmax = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), vmax);
  }
  if (amax) {                             // abs-max of the tensor just written (order-independent: reproducible)
    __shared__ float red[4];
    amax_update_block(amax, vmax, red);
  }
}

// ---- dy = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)) ---------------------------------------------
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ dout, int lddo,
                                                               const float* mean, const float* invstd, const float* gamma,
                                                               const float* beta, int act, float slope, const float* sums,
                                                               float inv_count, int64_t rows, int c, float* __restrict__ dy,
                                                               unsigned* __restrict__ amax, int rev) {
  const int c4 = c >> 2;
  const int64_t total = rows * c4;
  float vmax = 0.f;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < total; j += (int64_t)gridDim.x * 256) {
    const int64_t i = rev ? total - 1 - j : j;
    const int64_t r = i / c4; const int ch = (int)(i - r * c4) * 4;
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y + r * c + ch));      // last reads of both tensors
    f32x4 d = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dout + r * lddo + ch));
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + ch), is = *reinterpret_cast<const f32x4*>(invstd + ch);
    f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (gamma) g = *reinterpret_cast<const f32x4*>(gamma + ch);
    if (beta) b = *reinterpret_cast<const f32x4*>(beta + ch);
    const f32x4 sg = *reinterpret_cast<const f32x4*>(sums + ch), sgx = *reinterpret_cast<const f32x4*>(sums + c + ch);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (v[k] - mu[k]) * is[k];
      float dd = d[k];
      if (act == DCN_ACT_LEAKY && (g[k] * xh + b[k]) <= 0.f) dd *= slope;
      o[k] = g[k] * is[k] * (dd - sg[k] * inv_count - xh * sgx[k] * inv_count);
    }
    *reinterpret_cast<f32x4*>(dy + r * c + ch) = o;
    vmax = fmaxf(fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))), vmax);
  }
  if (amax) {
    __shared__ float red[4];
    amax_update_block(amax, vmax, red);
  }
}

// ---- the two passes above with the channel fixed per thread (round 5) ---------------------------------------------------------------
// c / 4 a power of two <= 256 (every BatchNorm width of the network but the 1056-channel fusion tensor): quad t of a 256-quad group lies
// in channel quad t % (c / 4) whatever the group, so a thread's per-channel vectors (scale / shift; mean, invstd, gamma, beta and the two
// sums) are read ONCE instead of once per element quad — the grid-stride form issued 2 + 6 loads per 16-byte store in the backward pass, the
// L1 busy with parameter re-reads —, the 64-bit division per element is a shift, and a workgroup takes 16-KB chunks (PC_UNR groups) with
// PC_UNR independent streaming loads per tensor in flight per thread (the grid-stride loop had one: 16-32 KB per CU against the ~64 KB that
// 8 TB/s x 2 us ask for).  Same arithmetic per element, order-independent abs-max: bitwise the same tensors and words.
constexpr int PC_UNR = 4;
__global__ __launch_bounds__(256) void scale_act_pc_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int act, float slope,
                                                           const float* __restrict__ residual, float* __restrict__ out,
                                                           int64_t total, int lg_c4, int ldo, unsigned* __restrict__ amax) {
  const int c4 = 1 << lg_c4;
  const int ch = (threadIdx.x & (c4 - 1)) * 4;
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
  if (scale) sc = *reinterpret_cast<const f32x4*>(scale + ch);
  if (shift) sh = *reinterpret_cast<const f32x4*>(shift + ch);
  float vmax = 0.f;
  const int64_t nchunk = (total + 256 * PC_UNR - 1) / (256 * PC_UNR);
  for (int64_t cidx = blockIdx.x; cidx < nchunk; cidx += gridDim.x) {
    const int64_t i0 = cidx * (256 * PC_UNR) + threadIdx.x;
    f32x4 v[PC_UNR], rs[PC_UNR];
#pragma unroll
    for (int k = 0; k < PC_UNR; ++k) {
      const int64_t i = i0 + k * 256;
      const bool ok = i < total;
      v[k] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y + i * 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (residual) rs[k] = ok ? *reinterpret_cast<const f32x4*>(residual + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < PC_UNR; ++k) {
      const int64_t i = i0 + k * 256;
      if (i >= total) continue;
      f32x4 t = v[k] * sc + sh;
      if (act == DCN_ACT_LEAKY) {
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] = t[e] > 0.f ? t[e] : t[e] * slope;
      }
      if (residual) t += rs[k];
      *reinterpret_cast<f32x4*>(out + (i >> lg_c4) * ldo + ch) = t;
      vmax = fmaxf(fmaxf(fmaxf(fabsf(t[0]), fabsf(t[1])), fmaxf(fabsf(t[2]), fabsf(t[3]))), vmax);
    }
  }
  if (amax) {
    __shared__ float red[4];
    amax_update_block(amax, vmax, red);
  }
}

__global__ __launch_bounds__(256) void bn_act_bwd_apply_pc_kernel(const float* __restrict__ y, const float* __restrict__ dout, int lddo,
                                                                  const float* mean, const float* invstd, const float* gamma,
                                                                  const float* beta, int act, float slope, const float* sums,
                                                                  float inv_count, int64_t total, int lg_c4, float* __restrict__ dy,
                                                                  unsigned* __restrict__ amax) {
  const int c4 = 1 << lg_c4, c = c4 * 4;
  const int ch = (threadIdx.x & (c4 - 1)) * 4;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + ch), is = *reinterpret_cast<const f32x4*>(invstd + ch);
  f32x4 g = {1.f, 1.f, 1.f, 1.f}, b = {0.f, 0.f, 0.f, 0.f};
  if (gamma) g = *reinterpret_cast<const f32x4*>(gamma + ch);
  if (beta) b = *reinterpret_cast<const f32x4*>(beta + ch);
  const f32x4 sg = *reinterpret_cast<const f32x4*>(sums + ch), sgx = *reinterpret_cast<const f32x4*>(sums + c + ch);
  float vmax = 0.f;
  const int64_t nchunk = (total + 256 * PC_UNR - 1) / (256 * PC_UNR);
  for (int64_t cidx = blockIdx.x; cidx < nchunk; cidx += gridDim.x) {
    const int64_t i0 = cidx * (256 * PC_UNR) + threadIdx.x;
    f32x4 v[PC_UNR], d[PC_UNR];
#pragma unroll
    for (int k = 0; k < PC_UNR; ++k) {
      const int64_t i = i0 + k * 256;
      const bool ok = i < total;
      v[k] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y + i * 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
      d[k] = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dout + (i >> lg_c4) * lddo + ch)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < PC_UNR; ++k) {
      const int64_t i = i0 + k * 256;
      if (i >= total) continue;
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (v[k][e] - mu[e]) * is[e];
        float dd = d[k][e];
        if (act == DCN_ACT_LEAKY && (g[e] * xh + b[e]) <= 0.f) dd *= slope;
        o[e] = g[e] * is[e] * (dd - sg[e] * inv_count - xh * sgx[e] * inv_count);
      }
      *reinterpret_cast<f32x4*>(dy + i * 4) = o;
      vmax = fmaxf(fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))), vmax);
    }
  }
  if (amax) {
    __shared__ float red[4];
    amax_update_block(amax, vmax, red);
  }
}

// plain activation backward (eval-style affine, no batch statistics): dy = dout*act'(scale*y+shift)*scale
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ outv, const float* __restrict__ dout, int lddo,
                                                      float slope, int64_t rows, int c, float* __restrict__ dy) {
  const int c4 = c >> 2;
  const int64_t total = rows * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / c4; const int ch = (int)(i - r * c4) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(outv + r * c + ch);
    f32x4 d = *reinterpret_cast<const f32x4*>(dout + r * lddo + ch);
#pragma unroll
    for (int k = 0; k < 4; ++k) if (v[k] <= 0.f) d[k] *= slope;
    *reinterpret_cast<f32x4*>(dy + r * c + ch) = d;
  }
}

// dcn_set_tuning("dbnrev", mask): sweep direction of the streaming passes (1 scale_act, 2 backward partials, 4 backward apply run
// from the last row to the first): a pass that starts where its producer stopped finds the producer's last lines in the
// memory-side cache.
int g_bn_rev = 0;
int g_bn_pc = 1;          // dcn_set_tuning("Bpc", 0): the apply passes back on their grid-stride forms (A/B switch)
// log2(c / 4) when the per-thread-channel kernels take the width (c / 4 a power of two <= 256), else -1
inline int pc_lg(int c) {
  const int c4 = c / 4;
  if (c % 4 || c4 < 1 || c4 > 256 || (c4 & (c4 - 1))) return -1;
  int lg = 0;
  while ((1 << lg) < c4) ++lg;
  return lg;
}
inline int stream_grid(int64_t work_items) {
  int64_t b = (work_items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
inline int row_slices(int rows) { int rs = (rows + 63) / 64; return rs < 1 ? 1 : (rs > RS_MAX ? RS_MAX : rs); }

}  // namespace

void bn_set_tuning(int v) { g_bn_rev = v; }
void bn_set_pc(int v) { g_bn_pc = v; }
int bn_pc_enabled() { return g_bn_pc; }

extern "C" int64_t dcn_bn_ws(int c) { return (int64_t)RS_MAX * 2 * c * 2; }   // doubles stored in a float-typed scratch

extern "C" int dcn_bn_finalize(const float* stats, int rows, int c, int64_t count,
                               const float* gamma, const float* beta, float eps, float momentum,
                               float* running_mean, float* running_var,
                               float* mean, float* invstd, float* scale, float* shift, float* ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DCN_CHECK_ARG(stats && mean && invstd && scale && shift && ws, "bn_finalize: null pointer");
  DCN_CHECK_ARG(rows > 0 && c > 0 && count > 0, "bn_finalize: bad shape");
  DCN_CHECK_ARG(((uintptr_t)ws & 7) == 0, "bn_finalize: ws must be 8-byte aligned");
  const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
  if (rows <= FUSED_ROWS_MAX) {
    hipLaunchKernelGGL((reduce_finalize_kernel<0>), dim3(cdiv(c, RF_CH)), dim3(256), 0, stream, stats, rows, c, 1.0 / (double)count,
                       unbias, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift, (float*)nullptr);
    DCN_CHECK_LAUNCH("bn_reduce_finalize");
    return DCN_OK;
  }
  const int rs = row_slices(rows);
  hipLaunchKernelGGL(colsum_stage1, dim3(cdiv(2 * c, 32), rs), dim3(256), 0, stream, stats, rows, 2 * c, rs, (double*)ws);
  DCN_CHECK_LAUNCH("colsum_stage1");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(c, 256)), dim3(256), 0, stream, (const double*)ws, rs, c,
                     1.0 / (double)count, unbias, gamma, beta, eps, momentum, running_mean, running_var,
                     mean, invstd, scale, shift);
  DCN_CHECK_LAUNCH("bn_finalize");
  return DCN_OK;
}

extern "C" int dcn_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                           float eps, int c, float* scale, float* shift, void* stream) {
  DCN_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && c > 0, "bn_fold: bad argument");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                     running_var, eps, c, scale, shift);
  DCN_CHECK_LAUNCH("bn_fold");
  return DCN_OK;
}

extern "C" int dcn_channel_stats_rows(int64_t rows) { return cdiv(rows, 128); }

extern "C" int dcn_channel_stats(const float* x, int64_t rows, int c, int ld, float* stats, void* stream) {
  DCN_CHECK_ARG(x && stats && rows > 0 && c > 0, "channel_stats: bad argument");
  DCN_CHECK_ARG(c % 4 == 0 && (ld <= 0 || ld % 4 == 0) && ((uintptr_t)x & 15) == 0, "channel_stats: c=%d / ld=%d must be multiples of 4 floats, x 16-byte aligned", c, ld);
  hipLaunchKernelGGL((channel_partials_kernel<0>), dim3(cdiv(rows, 128), cdiv(c, 64)), dim3(256), 0, (hipStream_t)stream,
                     x, ld > 0 ? ld : c, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0.f, rows, c, stats, 0);
  DCN_CHECK_LAUNCH("channel_stats");
  return DCN_OK;
}

// abs-max word of act(scale[c]*y + shift[c]) from the abs-max word of y, without the tensor: max_c(|scale[c]| * max|y| + |shift[c]|) bounds
// it (LeakyReLU with |slope| <= 1 only shrinks values), and the f16 split needs a bound, not the maximum (a loose bound costs the
// smallest elements a few of their low bits: the pieces keep 22 bits down to 2^-18 of the bound)
__global__ __launch_bounds__(64) void amax_bound_kernel(const unsigned* __restrict__ amax_y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int c, unsigned* __restrict__ amax_out) {
  const float a = __uint_as_float(amax_read(amax_y));
  float b = 0.f;
  for (int i = threadIdx.x; i < c; i += 64) b = fmaxf(b, __builtin_fmaf(fabsf(scale[i]), a, fabsf(shift[i])));
  b = wave_max(b);
  amax_out[threadIdx.x & (DCN_AMAX_WORDS - 1)] = __float_as_uint(b);
}

extern "C" int dcn_bn_act_amax_bound(const uint32_t* amax_y, const float* scale, const float* shift, int c, float slope,
                                     uint32_t* amax_out, void* stream) {
  DCN_CHECK_ARG(amax_y && scale && shift && amax_out && c > 0, "bn_act_amax_bound: bad argument");
  DCN_CHECK_ARG(slope >= -1.f && slope <= 1.f, "bn_act_amax_bound: |slope|=%g > 1 would stretch negative values past the bound", slope);
  hipLaunchKernelGGL(amax_bound_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, amax_y, scale, shift, c, amax_out);
  DCN_CHECK_LAUNCH("bn_act_amax_bound");
  return DCN_OK;
}

extern "C" int dcn_scale_act(const float* y, const float* scale, const float* shift, int act, float slope,
                             const float* residual, float* out, int64_t rows, int c, int ldo, uint32_t* amax, void* stream) {
  DCN_CHECK_ARG(y && out && rows > 0 && c > 0 && c % 4 == 0, "scale_act: bad argument (c=%d must be a multiple of 4)", c);
  if (ldo <= 0) ldo = c;
  DCN_CHECK_ARG(ldo % 4 == 0, "scale_act: ldo=%d must be a multiple of 4", ldo);
  const int pid = prof_begin(10, (double)rows * c * 4.0 * (residual ? 3 : 2), (hipStream_t)stream);
  // (with an abs-max word: at most 1024 workgroups, i.e. 1024 atomics over 64 words; they all finish together)
  const int want = amax ? min(stream_grid(rows * (c / 4)), 1024) : stream_grid(rows * (c / 4));
  const int lg = (g_bn_pc && !(g_bn_rev & 1)) ? pc_lg(c) : -1;
  if (lg >= 0) {
    const int64_t total = rows * (c / 4), nchunk = (total + 256 * PC_UNR - 1) / (256 * PC_UNR);
    hipLaunchKernelGGL(scale_act_pc_kernel, dim3((int)(nchunk < want ? nchunk : want)), dim3(256), 0, (hipStream_t)stream,
                       y, scale, shift, act, slope, residual, out, total, lg, ldo, amax);
  } else
    hipLaunchKernelGGL(scale_act_kernel, dim3(want), dim3(256), 0, (hipStream_t)stream,
                       y, scale, shift, act, slope, residual, out, rows, c, ldo, amax, g_bn_rev & 1);
  prof_end(pid, (hipStream_t)stream);
  DCN_CHECK_LAUNCH("scale_act");
  return DCN_OK;
}

extern "C" int dcn_bn_act_bwd_reduce(const float* y, const float* dout, int lddo, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, int act, float slope,
                                     int64_t rows, int c, float* stats, void* stream) {
  DCN_CHECK_ARG(y && dout && mean && invstd && stats && rows > 0 && c > 0, "bn_act_bwd_reduce: bad argument");
  DCN_CHECK_ARG(c % 4 == 0 && (lddo <= 0 || lddo % 4 == 0) && (((uintptr_t)y | (uintptr_t)dout | (uintptr_t)mean | (uintptr_t)invstd) & 15) == 0,
                "bn_act_bwd_reduce: c=%d / lddo=%d must be multiples of 4 floats, pointers 16-byte aligned", c, lddo);
  const int pid = prof_begin(22, 8.0 * (double)rows * c, (hipStream_t)stream);
  hipLaunchKernelGGL((channel_partials_kernel<1>), dim3(cdiv(rows, 128), cdiv(c, 64)), dim3(256), 0, (hipStream_t)stream,
                     y, c, dout, lddo > 0 ? lddo : c, mean, invstd, gamma, beta, act, slope, rows, c, stats, (g_bn_rev >> 1) & 1);
  prof_end(pid, (hipStream_t)stream);
  DCN_CHECK_LAUNCH("bn_act_bwd_reduce");
  return DCN_OK;
}

extern "C" int dcn_bn_bwd_sums(const float* stats, int rows, int c, float* sums, float* ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DCN_CHECK_ARG(stats && sums && ws && rows > 0 && c > 0, "bn_bwd_sums: bad argument");
  if (rows <= FUSED_ROWS_MAX) {
    hipLaunchKernelGGL((reduce_finalize_kernel<1>), dim3(cdiv(c, RF_CH)), dim3(256), 0, stream, stats, rows, c, 0.0, 0.0,
                       (const float*)nullptr, (const float*)nullptr, 0.f, 0.f, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, sums);
    DCN_CHECK_LAUNCH("bn_reduce_sums");
    return DCN_OK;
  }
  const int rs = row_slices(rows);
  hipLaunchKernelGGL(colsum_stage1, dim3(cdiv(2 * c, 32), rs), dim3(256), 0, stream, stats, rows, 2 * c, rs, (double*)ws);
  DCN_CHECK_LAUNCH("colsum_stage1");
  hipLaunchKernelGGL(sums_finalize_kernel, dim3(cdiv(2 * c, 256)), dim3(256), 0, stream, (const double*)ws, rs, 2 * c, sums);
  DCN_CHECK_LAUNCH("sums_finalize");
  return DCN_OK;
}

extern "C" int dcn_bn_act_bwd_apply(const float* y, const float* dout, int lddo, const float* mean, const float* invstd,
                                    const float* gamma, const float* beta, int act, float slope,
                                    const float* sums, int64_t count, int64_t rows, int c, float* dy, uint32_t* amax, void* stream) {
  DCN_CHECK_ARG(y && dout && mean && invstd && sums && dy && rows > 0 && c > 0 && c % 4 == 0, "bn_act_bwd_apply: bad argument");
  if (lddo <= 0) lddo = c;
  const int pid = prof_begin(11, (double)rows * c * 4.0 * 3, (hipStream_t)stream);
  const int want = amax ? min(stream_grid(rows * (c / 4)), 1024) : stream_grid(rows * (c / 4));
  const int lg = (g_bn_pc && !((g_bn_rev >> 2) & 1) && (((uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)sums | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0)
                     ? pc_lg(c) : -1;
  if (lg >= 0) {
    const int64_t total = rows * (c / 4), nchunk = (total + 256 * PC_UNR - 1) / (256 * PC_UNR);
    hipLaunchKernelGGL(bn_act_bwd_apply_pc_kernel, dim3((int)(nchunk < want ? nchunk : want)), dim3(256), 0, (hipStream_t)stream,
                       y, dout, lddo, mean, invstd, gamma, beta, act, slope, sums, 1.f / (float)count, total, lg, dy, amax);
  }
  else
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(want), dim3(256), 0, (hipStream_t)stream,
                       y, dout, lddo, mean, invstd, gamma, beta, act, slope, sums, 1.f / (float)count, rows, c, dy, amax, (g_bn_rev >> 2) & 1);
  prof_end(pid, (hipStream_t)stream);
  DCN_CHECK_LAUNCH("bn_act_bwd_apply");
  return DCN_OK;
}

extern "C" int dcn_act_bwd(const float* out, const float* dout, int lddo, float slope, int64_t rows, int c, float* dy, void* stream) {
  DCN_CHECK_ARG(out && dout && dy && rows > 0 && c > 0 && c % 4 == 0, "act_bwd: bad argument");
  if (lddo <= 0) lddo = c;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(stream_grid(rows * (c / 4))), dim3(256), 0, (hipStream_t)stream,
                     out, dout, lddo, slope, rows, c, dy);
  DCN_CHECK_LAUNCH("act_bwd");
  return DCN_OK;
}
