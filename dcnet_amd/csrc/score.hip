// Cross-modal scoring (HBM-bound): L2-normalise a feature map over channels and, in the same pass,
// score every position against the per-image language vector.
//   out[row,:] = x[row,:] / max(||x[row,:]||, 1e-12)        F.normalize(dim=1)   DCNet_model.py:359,469
//   score[row] = <out[row,:], q[img(row),:]>                 sim_score            DCNet_model.py:530-535
// Algorithmic bytes per row: c*4 read + c*4 written (+8); one wave per row, 16-B accesses per lane.
#include "common.h"
#include "prof.h"

namespace {

constexpr int MAX_V4 = 4;   // c <= 1024

// Each wave normalises RPW rows at once: all their 16-B loads are issued before the first reduction, which keeps RPW * c * 4
// bytes in flight per wave (tools/bench_score.py, 64 x 52 x 52 x 512: one or two rows 5.1 TB/s, four 4.9, eight 4.4).
// V4 = 16-B loads per lane and row (c <= 256 V4): registers are sized for the launch's channel count, not for the maximum.
// NT: x is read with non-temporal loads (it is not read again before the backward; `out` is, by the kernels that follow):
// 5.1 -> 5.6 TB/s on the same shape, a device copy of the same bytes runs at 5.5.
template <int V4, int RPW, bool NT>
__global__ __launch_bounds__(256) void l2norm_score_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out, int ldo,
                                                               float* __restrict__ norm, const float* __restrict__ q,
                                                               float* __restrict__ score, float* __restrict__ score_flip,
                                                               int64_t rows, int rpi, int c, float out_scale, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int nimg = (int)(rows / rpi);
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (row0 >= rows) return;
  f32x4 v[RPW][V4];
  float ss[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int64_t row = row0 + r < rows ? row0 + r : rows - 1;       // tail: recompute the last row, store is guarded
#pragma unroll
    for (int k = 0; k < V4; ++k) {
      const int ch = (lane + 64 * k) * 4;
      v[r][k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ch < c) {
        const f32x4* src = reinterpret_cast<const f32x4*>(x + row * ldx + ch);
        v[r][k] = NT ? __builtin_nontemporal_load(src) : *src;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < V4; ++k)           // explicit fma chain: every instantiation rounds alike
#pragma unroll
      for (int j = 0; j < 4; ++j) s = fmaf(v[r][k][j], v[r][k][j], s);
    ss[r] = wave_sum(s);
  }
  // image of the wave's first row: one division per wave, then a compare per row (rows of a wave are consecutive)
  int img = (int)(row0 / rpi);
  int64_t img_end = (int64_t)(img + 1) * rpi;
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int64_t row = row0 + r;
    if (row >= rows) break;
    if (row >= img_end) { ++img; img_end += rpi; }
    const float nrm = sqrtf(ss[r]);
    const float inv = 1.f / fmaxf(nrm, 1e-12f);
    const float* qq = q ? q + (int64_t)img * c : nullptr;
    const float* qf = (q && score_flip) ? q + (int64_t)(nimg - 1 - img) * c : nullptr;     // the language vector of image N-1-n
    float dot = 0.f, dotf = 0.f;
#pragma unroll
    for (int k = 0; k < V4; ++k) {
      const int ch = (lane + 64 * k) * 4;
      if (ch < c) {
        const f32x4 o = v[r][k] * inv;
        f32x4 st = o * out_scale;
        if (accumulate) st += *reinterpret_cast<const f32x4*>(out + row * ldo + ch);
        *reinterpret_cast<f32x4*>(out + row * ldo + ch) = st;
        if (qq) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(qq + ch);
#pragma unroll
          for (int j = 0; j < 4; ++j) dot = fmaf(o[j], w[j], dot);
        }
        if (qf) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(qf + ch);
#pragma unroll
          for (int j = 0; j < 4; ++j) dotf = fmaf(o[j], w[j], dotf);
        }
      }
    }
    if (qq) dot = wave_sum(dot);
    if (qf) dotf = wave_sum(dotf);
    if (lane == 0) {
      if (norm) norm[row] = nrm;
      if (qq) score[row] = dot;
      if (qf) score_flip[row] = dotf;
    }
  }
}

int g_l2_rpw = 2;      // dcn_set_tuning("e2rpw", 1|2|4|8): rows per wave of l2norm_score_fwd (tools/bench_score.py)
int g_l2_nt = 1;       // dcn_set_tuning("f2nt", 0|1): non-temporal loads of x

template <int V4>
void launch_l2fwd(const float* x, int ldx, float* out, int ldo, float* norm, const float* q, float* score, float* score_flip,
                  int64_t rows, int rpi, int c, float out_scale, int accumulate, hipStream_t stream) {
#define L2F(R, N) hipLaunchKernelGGL((l2norm_score_fwd_kernel<V4, R, N>), dim3(cdiv(rows, 4 * R)), dim3(256), 0, stream, \
                                     x, ldx, out, ldo, norm, q, score, score_flip, rows, rpi, c, out_scale, accumulate)
  const int r = (g_l2_rpw == 8 && V4 > 2) ? 4 : g_l2_rpw;          // (8 rows x 1024 channels: 128 data registers)
  if (g_l2_nt) { if (r == 1) L2F(1, true); else if (r == 2) L2F(2, true); else if (r == 8) L2F(8, true); else L2F(4, true); }
  else { if (r == 1) L2F(1, false); else if (r == 2) L2F(2, false); else if (r == 8) L2F(8, false); else L2F(4, false); }
#undef L2F
}

// g = dout + dscore*q ;  dx = (g - out*<g,out>) / max(norm, eps)
__global__ __launch_bounds__(256) void l2norm_score_bwd_kernel(const float* __restrict__ out, int ldo, const float* __restrict__ norm,
                                                               const float* __restrict__ dout, int lddo, const float* __restrict__ q,
                                                               const float* __restrict__ dscore, const float* __restrict__ dscore_flip,
                                                               float* __restrict__ dx, int lddx, int64_t rows, int rpi, int c) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float ds = (q && dscore) ? dscore[row] : 0.f;
  const float* qq = (q && dscore) ? q + (row / rpi) * c : nullptr;
  const float dsf = (q && dscore_flip) ? dscore_flip[row] : 0.f;
  const float* qf = (q && dscore_flip) ? q + (rows / rpi - 1 - row / rpi) * c : nullptr;
  f32x4 g[MAX_V4], o[MAX_V4];
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < MAX_V4; ++k) {
    const int ch = (lane + 64 * k) * 4;
    g[k] = f32x4{0.f, 0.f, 0.f, 0.f}; o[k] = g[k];
    if (ch < c) {
      o[k] = *reinterpret_cast<const f32x4*>(out + row * ldo + ch);
      if (dout) g[k] = *reinterpret_cast<const f32x4*>(dout + row * lddo + ch);
      if (qq) g[k] += *reinterpret_cast<const f32x4*>(qq + ch) * ds;
      if (qf) g[k] += *reinterpret_cast<const f32x4*>(qf + ch) * dsf;
      dot += g[k][0] * o[k][0] + g[k][1] * o[k][1] + g[k][2] * o[k][2] + g[k][3] * o[k][3];
    }
  }
  dot = wave_sum(dot);
  const float inv = 1.f / fmaxf(norm[row], 1e-12f);
#pragma unroll
  for (int k = 0; k < MAX_V4; ++k) {
    const int ch = (lane + 64 * k) * 4;
    if (ch < c) *reinterpret_cast<f32x4*>(dx + row * lddx + ch) = (g[k] - o[k] * dot) * inv;
  }
}

// dq[img][ch] = sum_{rows of img} dscore[row] * out[row][ch];  grid (c/64, n_img), 4 row lanes x 64 channels
// (+ the flipped score: rows of image N-1-img weighted by dscore_flip)
__global__ __launch_bounds__(256) void score_dq_kernel(const float* __restrict__ out, int ldo, const float* __restrict__ dscore,
                                                       const float* __restrict__ dscore_flip, float* __restrict__ dq, int rpi, int c) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int ch = blockIdx.x * 64 + tx, img = blockIdx.y;
  float s = 0.f;
  // eight rows of loads in flight per thread, added in row order (one load per iteration ran at the latency of a load: 0.56 ms
  // for the 354 MB of the 52 x 52 scale)
  auto sweep = [&](const float* w, int64_t row0) {
    int r = ty;
    for (; r + 28 < rpi; r += 32) {
      float o[8], d[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int64_t row = row0 + r + 4 * k; o[k] = out[row * ldo + ch]; d[k] = w[row]; }
#pragma unroll
      for (int k = 0; k < 8; ++k) s += d[k] * o[k];
    }
    for (; r < rpi; r += 4) { const int64_t row = row0 + r; s += w[row] * out[row * ldo + ch]; }
  };
  if (ch < c) {
    if (dscore) sweep(dscore, (int64_t)img * rpi);
    if (dscore_flip) sweep(dscore_flip, (int64_t)(gridDim.y - 1 - img) * rpi);
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && ch < c) dq[(size_t)img * c + ch] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

// score[row] = <x[row,:], q[img(row) (or N-1-img when flip)]>   — sim_score of the n_frame model (model/test_DCNet_model.py:386-391)
// and neg_sim_score of the training harness (train_DCNet.py:623-627) on tensors that are not normalised here
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ q, int flip,
                                                         float* __restrict__ score, int64_t rows, int rpi, int c) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t img = flip ? rows / rpi - 1 - row / rpi : row / rpi;
  float dot = 0.f;
  for (int ch = lane * 4; ch < c; ch += 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + row * ldx + ch), w = *reinterpret_cast<const f32x4*>(q + img * c + ch);
    dot += a[0] * w[0] + a[1] * w[1] + a[2] * w[2] + a[3] * w[3];
  }
  dot = wave_sum(dot);
  if (lane == 0) score[row] = dot;
}

__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ q, int flip, const float* __restrict__ dscore,
                                                         float* __restrict__ dx, int lddx, int64_t rows, int rpi, int c) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t img = flip ? rows / rpi - 1 - row / rpi : row / rpi;
  const float ds = dscore[row];
  for (int ch = lane * 4; ch < c; ch += 256)
    *reinterpret_cast<f32x4*>(dx + row * lddx + ch) = *reinterpret_cast<const f32x4*>(q + img * c + ch) * ds;
}

}  // namespace

extern "C" int dcn_l2norm_score_fwd(const float* x, int ldx, float* out, int ldo, float* norm,
                                    const float* q, float* score, float* score_flip, int64_t rows, int rows_per_image, int c,
                                    float out_scale, int accumulate, void* stream) {
  DCN_CHECK_ARG(x && out && rows > 0 && c > 0 && c % 4 == 0 && c <= 256 * MAX_V4, "l2norm_score_fwd: bad argument (c=%d)", c);
  DCN_CHECK_ARG(!q || (score && rows_per_image > 0 && rows % rows_per_image == 0), "l2norm_score_fwd: q given without score/rows_per_image");
  DCN_CHECK_ARG(!score_flip || q, "l2norm_score_fwd: score_flip needs q");
  if (ldx <= 0) ldx = c;
  if (ldo <= 0) ldo = c;
  // algorithmic bytes: read x, write out (+ norm, score)
  const int pid = prof_begin(8, (double)rows * (2.0 * c * 4 + 8), (hipStream_t)stream);
  const int rpi_ = rows_per_image > 0 ? rows_per_image : 1;
  if (c <= 256) launch_l2fwd<1>(x, ldx, out, ldo, norm, q, score, score_flip, rows, rpi_, c, out_scale, accumulate, (hipStream_t)stream);
  else if (c <= 512) launch_l2fwd<2>(x, ldx, out, ldo, norm, q, score, score_flip, rows, rpi_, c, out_scale, accumulate, (hipStream_t)stream);
  else launch_l2fwd<4>(x, ldx, out, ldo, norm, q, score, score_flip, rows, rpi_, c, out_scale, accumulate, (hipStream_t)stream);
  prof_end(pid, (hipStream_t)stream);
  DCN_CHECK_LAUNCH("l2norm_score_fwd");
  return DCN_OK;
}

extern "C" int dcn_l2norm_score_bwd(const float* out, int ldo, const float* norm, const float* dout, int lddo,
                                    const float* q, const float* dscore, const float* dscore_flip, float* dx, int lddx, float* dq,
                                    int64_t rows, int rows_per_image, int c, void* stream) {
  DCN_CHECK_ARG(out && norm && dx && rows > 0 && c > 0 && c % 4 == 0 && c <= 256 * MAX_V4, "l2norm_score_bwd: bad argument");
  DCN_CHECK_ARG(dout || (q && (dscore || dscore_flip)), "l2norm_score_bwd: no upstream gradient");
  if (ldo <= 0) ldo = c;
  if (lddo <= 0) lddo = c;
  if (lddx <= 0) lddx = c;
  const int pid = prof_begin(9, (double)rows * ((dout ? 3.0 : 2.0) * c * 4 + 8), (hipStream_t)stream);
  hipLaunchKernelGGL(l2norm_score_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                     out, ldo, norm, dout, lddo, q, dscore, dscore_flip, dx, lddx, rows, rows_per_image > 0 ? rows_per_image : 1, c);
  prof_end(pid, (hipStream_t)stream);
  DCN_CHECK_LAUNCH("l2norm_score_bwd");
  if (dq && q && (dscore || dscore_flip)) {
    DCN_CHECK_ARG(rows_per_image > 0 && rows % rows_per_image == 0, "l2norm_score_bwd: rows %% rows_per_image != 0");
    hipLaunchKernelGGL(score_dq_kernel, dim3(cdiv(c, 64), (int)(rows / rows_per_image)), dim3(256), 0, (hipStream_t)stream,
                       out, ldo, dscore, dscore_flip, dq, rows_per_image, c);
    DCN_CHECK_LAUNCH("score_dq");
  }
  return DCN_OK;
}

extern "C" int dcn_rowdot_fwd(const float* x, int ldx, const float* q, int flip, float* score, int64_t rows, int rows_per_image, int c,
                              void* stream) {
  DCN_CHECK_ARG(x && q && score && rows > 0 && rows_per_image > 0 && rows % rows_per_image == 0 && c > 0 && c % 4 == 0 && ldx % 4 == 0,
                "rowdot_fwd: bad argument");
  hipLaunchKernelGGL(rowdot_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, q, flip, score, rows, rows_per_image, c);
  DCN_CHECK_LAUNCH("rowdot_fwd");
  return DCN_OK;
}

extern "C" int dcn_rowdot_bwd(const float* x, int ldx, const float* q, int flip, const float* dscore, float* dx, int lddx, float* dq,
                              int64_t rows, int rows_per_image, int c, void* stream) {
  DCN_CHECK_ARG(x && q && dscore && rows > 0 && rows_per_image > 0 && rows % rows_per_image == 0 && c > 0 && c % 4 == 0 && ldx % 4 == 0,
                "rowdot_bwd: bad argument");
  if (dx) {
    DCN_CHECK_ARG(lddx % 4 == 0, "rowdot_bwd: lddx");
    hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, q, flip, dscore, dx, lddx, rows, rows_per_image, c);
    DCN_CHECK_LAUNCH("rowdot_bwd");
  }
  if (dq) {
    hipLaunchKernelGGL(score_dq_kernel, dim3(cdiv(c, 64), (int)(rows / rows_per_image)), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, flip ? nullptr : dscore, flip ? dscore : nullptr, dq, rows_per_image, c);
    DCN_CHECK_LAUNCH("rowdot dq");
  }
  return DCN_OK;
}

void score_set_tuning(int key, int value) { if (key == 0) g_l2_rpw = value; else g_l2_nt = value; }
