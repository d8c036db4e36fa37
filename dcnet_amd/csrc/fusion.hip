// Small kernels that replace what was left of stock torch ops on the step's path (round-2 trace: five hipBLASLt GEMMs and two
// reduce kernels of the fusion layer's pre-fill term, F.normalize of the language vector, the reduce behind `(ids != 0).sum(1)`,
// the sort-based backward of nn.Embedding).  All of them are latency-sized; none uses float atomics (fixed summation order).
//
//   fusion layer (model/DCNet_model.py:491-505): the first fcn_emb convolution sees [corr | tile(flang) | coord]; the last two
//   groups are constant over positions / images, so   conv = W1.corr[n,p] + (W2.flang[n]) + (W3.coord[p]).
//   prefill:   out[n,p,c] = A[n,c] + sum_k coord[p,k] * W3[c,k]          (A = flang . W2^T from dcn_gemm_nt)
//   backward:  d_img[n,c] = sum_p dy[n,p,c],  dW3[c,k] = sum_{n,p} dy[n,p,c] * coord[p,k]   — ONE pass over dy,
//              dW2[c,j]   = sum_n d_img[n,c] * flang[n,j]
#include "common.h"

namespace {

constexpr int FCH = 16;         // row chunks per image of the backward reduce

// grid (ceil(hw / rows_per_block), n), 256 threads = (256 / (co/4)) row lanes x co/4 channel quads (co <= 1024)
__global__ __launch_bounds__(256) void fusion_prefill_kernel(const float* __restrict__ A, const float* __restrict__ coord,
                                                             const float* __restrict__ w3, int ldw, float* __restrict__ out,
                                                             int hw, int co, int rows_per_block) {
  const int n = blockIdx.y;
  const int c4n = co >> 2, lanes = 256 / c4n;
  const int lane_r = threadIdx.x / c4n, cq = threadIdx.x - lane_r * c4n;
  if (lane_r >= lanes) return;
  const int p0 = blockIdx.x * rows_per_block, p1 = min(hw, p0 + rows_per_block);
  const int c = cq * 4;
  const f32x4 a = *reinterpret_cast<const f32x4*>(A + (size_t)n * co + c);
  float w[4][8];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int k = 0; k < 8; ++k) w[e][k] = w3[(size_t)(c + e) * ldw + k];
  for (int p = p0 + lane_r; p < p1; p += lanes) {
    const float* cr = coord + (size_t)p * 8;
    f32x4 v = a;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float ck = cr[k];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += ck * w[e][k];
    }
    *reinterpret_cast<f32x4*>(out + ((size_t)n * hw + p) * co + c) = v;
  }
}

// grid (FCH, n), 256 threads = 2 row lanes x 128 channel quads (co <= 512 per pass; loops for wider).  partial[(n*FCH+chunk)][9][co]
__global__ __launch_bounds__(256) void fusion_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ coord,
                                                                float* __restrict__ partial, int hw, int co) {
  __shared__ float red[9][512];
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int per = (hw + FCH - 1) / FCH;
  const int r0 = chunk * per, r1 = min(hw, r0 + per);
  const int lane_r = threadIdx.x >> 7, q = threadIdx.x & 127;
  for (int cb = 0; cb < co; cb += 512) {
    const int c = cb + q * 4;
    float acc[9][4];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[k][e] = 0.f;
    if (c < co) {
      for (int p = r0 + lane_r; p < r1; p += 2) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(dy + ((size_t)n * hw + p) * co + c);
        const float* cr = coord + (size_t)p * 8;        // uniform per wave (a wave = one row lane)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[0][e] += v[e];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float ck = cr[k];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[1 + k][e] += v[e] * ck;
        }
      }
    }
    if (lane_r == 1) {
#pragma unroll
      for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[k][q * 4 + e] = acc[k][e];
    }
    __syncthreads();
    if (lane_r == 0 && c < co) {
      float* dst = partial + ((size_t)(n * FCH + chunk) * 9) * co;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[k][e] + red[k][q * 4 + e];
        *reinterpret_cast<f32x4*>(dst + (size_t)k * co + c) = o;
      }
    }
    __syncthreads();
  }
}

// stage 2: per image, the chunk partials -> d_img[n][c] (k = 0) and tmp[n][k-1][c] (k = 1..8)
__global__ __launch_bounds__(256) void fusion_bwd_finish1_kernel(const float* __restrict__ partial, float* __restrict__ d_img,
                                                                 float* __restrict__ tmp, int n, int co) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * 9 * co) return;
  const int i = idx / (9 * co), r = idx - i * 9 * co, k = r / co, c = r - k * co;
  float s = 0.f;
#pragma unroll
  for (int ch = 0; ch < FCH; ++ch) s += partial[((size_t)(i * FCH + ch) * 9 + k) * co + c];
  if (k == 0) d_img[(size_t)i * co + c] = s;
  else tmp[((size_t)i * 8 + (k - 1)) * co + c] = s;
}

// stage 3: dw3[c][k] = sum_n tmp[n][k][c]   (dw3 row stride ldd)
__global__ __launch_bounds__(256) void fusion_bwd_finish2_kernel(const float* __restrict__ tmp, float* __restrict__ dw3, int ldd, int n, int co) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 8 * co) return;
  const int k = idx / co, c = idx - k * co;
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += tmp[((size_t)i * 8 + k) * co + c];
  dw3[(size_t)c * ldd + k] = s;
}

// out[c][j] = sum_n a[n][c] * b[n][j]    (a (n,co), b (n,e); out row stride ldo)
__global__ __launch_bounds__(256) void outer_sum_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                        int ldo, int n, int co, int e) {
  const int j = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if (j >= e) return;
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += a[(size_t)i * co + c] * b[(size_t)i * e + j];
  out[(size_t)c * ldo + j] = s;
}

// lengths[r] = number of non-zero ids of row r (model/DCNet_model.py:150)
__global__ __launch_bounds__(256) void row_lengths_kernel(const int64_t* __restrict__ ids, int n, int L, int64_t* __restrict__ out) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  int cnt = 0;
  for (int l = 0; l < L; ++l) cnt += ids[(size_t)r * L + l] != 0;
  out[r] = cnt;
}

// out[t][:] = table[ids[t]][:] * keep   (nn.Embedding forward; e % 4 == 0)
__global__ __launch_bounds__(128) void embedding_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                            float* __restrict__ out, int e, int vocab) {
  const int t = blockIdx.x;
  const int64_t v = ids[t];
  for (int c = threadIdx.x * 4; c < e; c += 128 * 4) {
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (v >= 0 && v < vocab) x = *reinterpret_cast<const f32x4*>(table + (size_t)v * e + c);
    *reinterpret_cast<f32x4*>(out + (size_t)t * e + c) = x;
  }
}

// dtable[v][:] = sum over tokens t with ids[t] == v of dout[t][:], tokens in ascending order (deterministic; no sort, no atomics):
// one workgroup per vocabulary row scans the (few thousand) ids.
__global__ __launch_bounds__(128) void embedding_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dout,
                                                            float* __restrict__ dtable, int tokens, int e) {
  const int v = blockIdx.x;
  for (int c = threadIdx.x * 4; c < e; c += 128 * 4) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < tokens; ++t)
      if (ids[t] == v) s += *reinterpret_cast<const f32x4*>(dout + (size_t)t * e + c);
    *reinterpret_cast<f32x4*>(dtable + (size_t)v * e + c) = s;
  }
}

}  // namespace

extern "C" int dcn_fusion_prefill(const float* A, const float* coord, const float* w3, int ldw, float* out, int n, int hw, int co,
                                  void* stream) {
  DCN_CHECK_ARG(A && coord && w3 && out && n > 0 && hw > 0 && co >= 4 && co % 4 == 0 && co <= 1024 && ldw >= 8, "fusion_prefill: bad argument");
  const int rpb = 32;
  hipLaunchKernelGGL(fusion_prefill_kernel, dim3(cdiv(hw, rpb), n), dim3(256), 0, (hipStream_t)stream, A, coord, w3, ldw, out, hw, co, rpb);
  DCN_CHECK_LAUNCH("fusion_prefill");
  return DCN_OK;
}

extern "C" int64_t dcn_fusion_bwd_ws(int n, int co) { return (int64_t)n * FCH * 9 * co + (int64_t)n * 8 * co; }

extern "C" int dcn_fusion_bwd(const float* dy, const float* coord, const float* flang, float* ws, float* d_img, float* dw2, float* dw3,
                              int ldd, int n, int hw, int co, int e, void* stream) {
  DCN_CHECK_ARG(dy && coord && flang && ws && d_img && dw2 && dw3 && n > 0 && hw > 0 && co > 0 && co % 4 == 0 && e > 0 && ldd >= 8 && co <= 512,
                "fusion_bwd: bad argument (co <= 512)");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(fusion_bwd_reduce_kernel, dim3(FCH, n), dim3(256), 0, s, dy, coord, ws, hw, co);
  DCN_CHECK_LAUNCH("fusion_bwd_reduce");
  float* tmp = ws + (size_t)n * FCH * 9 * co;
  hipLaunchKernelGGL(fusion_bwd_finish1_kernel, dim3(cdiv((int64_t)n * 9 * co, 256)), dim3(256), 0, s, ws, d_img, tmp, n, co);
  DCN_CHECK_LAUNCH("fusion_bwd_finish1");
  hipLaunchKernelGGL(fusion_bwd_finish2_kernel, dim3(cdiv(8 * co, 256)), dim3(256), 0, s, tmp, dw3, ldd, n, co);
  DCN_CHECK_LAUNCH("fusion_bwd_finish2");
  hipLaunchKernelGGL(outer_sum_kernel, dim3(cdiv(e, 256), co), dim3(256), 0, s, d_img, flang, dw2, ldd, n, co, e);
  DCN_CHECK_LAUNCH("fusion_bwd_dw2");
  return DCN_OK;
}

extern "C" int dcn_row_lengths(const int64_t* ids, int n, int L, int64_t* out, void* stream) {
  DCN_CHECK_ARG(ids && out && n > 0 && L > 0, "row_lengths: bad argument");
  hipLaunchKernelGGL(row_lengths_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, ids, n, L, out);
  DCN_CHECK_LAUNCH("row_lengths");
  return DCN_OK;
}

extern "C" int dcn_embedding_fwd(const int64_t* ids, const float* table, float* out, int tokens, int e, int vocab, void* stream) {
  DCN_CHECK_ARG(ids && table && out && tokens > 0 && e > 0 && e % 4 == 0 && vocab > 0, "embedding_fwd: bad argument");
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(tokens), dim3(128), 0, (hipStream_t)stream, ids, table, out, e, vocab);
  DCN_CHECK_LAUNCH("embedding_fwd");
  return DCN_OK;
}

extern "C" int dcn_embedding_bwd(const int64_t* ids, const float* dout, float* dtable, int tokens, int e, int vocab, void* stream) {
  DCN_CHECK_ARG(ids && dout && dtable && tokens > 0 && e > 0 && e % 4 == 0 && vocab > 0, "embedding_bwd: bad argument");
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3(vocab), dim3(128), 0, (hipStream_t)stream, ids, dout, dtable, tokens, e);
  DCN_CHECK_LAUNCH("embedding_bwd");
  return DCN_OK;
}
