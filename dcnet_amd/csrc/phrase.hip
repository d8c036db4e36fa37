// PhraseAttention (model/DCNet_model.py:190-219) for up to two heads that share their inputs (sub_attn :525 and
// loc_attn :556 read the same context / embedded), fused with the F.normalize(p=2, dim=1) that follows each (:526,:557):
//     s_l    = <w_h, context[n,l,:]> + b_h
//     a      = softmax_l(s) * (ids != 0);  a /= sum(a)          (:207-212)
//     v      = sum_l a_l * embedded[n,l,:]                       (:215-216, the bmm)
//     out    = v / max(||v||, 1e-12)                              (when normalize != 0)
// One workgroup per (image, head): a wave per word for the 1024-wide score GEMV, then 512 channels over 256 threads.
// Roofline: latency (N*L*(D+E)*4 B = 7.9 MB per step at N = 64); the point is one launch instead of ~15 per head.
#include "common.h"

namespace {

constexpr int PH_MAXL = 32;      // words per query (the reference is built for 20)
constexpr int PH_MAXE = 512;     // embedded width handled per thread pair (2 channels per thread)

__device__ __forceinline__ float block_sum256(float v, float* red /* [4] */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void phrase_fwd_kernel(const float* __restrict__ ctx, const float* __restrict__ emb,
                                                         const int64_t* __restrict__ ids, const float* __restrict__ w0,
                                                         const float* __restrict__ b0, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, float* __restrict__ attn,
                                                         float* __restrict__ out, float* __restrict__ vnorm,
                                                         int N, int L, int D, int E, int normalize) {
  __shared__ float s[PH_MAXL], a[PH_MAXL], red[4];
  const int n = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* w = h == 0 ? w0 : w1;
  const float bias = (h == 0 ? b0 : b1)[0];
  for (int l = wave; l < L; l += 4) {
    const float* row = ctx + ((size_t)n * L + l) * D;
    float acc = 0.f;
    for (int i = lane * 4; i < D; i += 256) {
      const f32x4 x = *reinterpret_cast<const f32x4*>(row + i), ww = *reinterpret_cast<const f32x4*>(w + i);
      acc += x[0] * ww[0] + x[1] * ww[1] + x[2] * ww[2] + x[3] * ww[3];
    }
    acc = wave_sum(acc);
    if (lane == 0) s[l] = acc + bias;
  }
  __syncthreads();
  if (tid < 64) {                      // one wave: softmax over all L words, mask, renormalise
    const float sv = tid < L ? s[tid] : -INFINITY;
    const float mx = wave_max(sv);
    const float e = tid < L ? expf(sv - mx) : 0.f;
    const float z = wave_sum(e);
    const float p = (tid < L && ids[(size_t)n * L + tid] != 0) ? e / z : 0.f;
    const float zm = wave_sum(p);
    if (tid < L) {
      const float av = p / zm;
      a[tid] = av;
      attn[((size_t)h * N + n) * L + tid] = av;
    }
  }
  __syncthreads();
  float v[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = tid + 256 * j;
    if (c < E) {
      float acc = 0.f;
      for (int l = 0; l < L; ++l) acc = fmaf(a[l], emb[((size_t)n * L + l) * E + c], acc);
      v[j] = acc;
    }
  }
  float nrm = 1.f;
  if (normalize) {
    nrm = sqrtf(block_sum256(v[0] * v[0] + v[1] * v[1], red));
    if (tid == 0) vnorm[(size_t)h * N + n] = nrm;
    nrm = fmaxf(nrm, 1e-12f);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = tid + 256 * j;
    if (c < E) out[((size_t)h * N + n) * E + c] = v[j] / nrm;
  }
}

// One workgroup per image, both heads: dctx [N][L][D], demb [N][L][E] (sums over the heads),
// part [N][H*D + H] = per-image slices of dw_h (H*D) and db_h (H); the caller column-sums them over n.
__global__ __launch_bounds__(256) void phrase_bwd_kernel(const float* __restrict__ ctx, const float* __restrict__ emb,
                                                         const float* __restrict__ w0, const float* __restrict__ w1,
                                                         const float* __restrict__ attn, const float* __restrict__ out,
                                                         const float* __restrict__ vnorm, const float* __restrict__ dout,
                                                         float* __restrict__ dctx, float* __restrict__ demb,
                                                         float* __restrict__ part, int N, int L, int D, int E, int H, int normalize) {
  __shared__ float dv[PH_MAXE], da[PH_MAXL], ds[2][PH_MAXL], red[4];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float de[PH_MAXL][2];
#pragma unroll
  for (int l = 0; l < PH_MAXL; ++l) de[l][0] = de[l][1] = 0.f;
  for (int h = 0; h < H; ++h) {
    const float* at = attn + ((size_t)h * N + n) * L;
    float g[2], o[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = tid + 256 * j;
      g[j] = c < E ? dout[((size_t)h * N + n) * E + c] : 0.f;
      o[j] = c < E ? out[((size_t)h * N + n) * E + c] : 0.f;
    }
    if (normalize) {
      const float dot = block_sum256(g[0] * o[0] + g[1] * o[1], red);
      const float inv = 1.f / fmaxf(vnorm[(size_t)h * N + n], 1e-12f);
      g[0] = (g[0] - o[0] * dot) * inv; g[1] = (g[1] - o[1] * dot) * inv;
    }
    __syncthreads();                     // previous head's readers of dv / da are done
#pragma unroll
    for (int j = 0; j < 2; ++j) if (tid + 256 * j < E) dv[tid + 256 * j] = g[j];
    __syncthreads();
    for (int l = wave; l < L; l += 4) {  // da_l = <dv, embedded[n,l,:]>
      const float* row = emb + ((size_t)n * L + l) * E;
      float acc = 0.f;
      for (int c = lane; c < E; c += 64) acc = fmaf(dv[c], row[c], acc);
      acc = wave_sum(acc);
      if (lane == 0) da[l] = acc;
    }
    __syncthreads();
    float sad = 0.f;
    for (int l = 0; l < L; ++l) sad = fmaf(at[l], da[l], sad);
    if (tid < L) ds[h][tid] = at[tid] * (da[tid] - sad);     // masked softmax Jacobian (a_l = 0 on padded words)
#pragma unroll
    for (int l = 0; l < PH_MAXL; ++l)
      if (l < L) { const float al = at[l]; de[l][0] = fmaf(al, g[0], de[l][0]); de[l][1] = fmaf(al, g[1], de[l][1]); }
  }
  __syncthreads();
#pragma unroll
  for (int l = 0; l < PH_MAXL; ++l)
    if (l < L) {
#pragma unroll
      for (int j = 0; j < 2; ++j) if (tid + 256 * j < E) demb[((size_t)n * L + l) * E + tid + 256 * j] = de[l][j];
    }
  float* prt = part + (size_t)n * (H * D + H);
  for (int i = tid * 4; i < D; i += 1024) {
    const f32x4 wa = *reinterpret_cast<const f32x4*>(w0 + i);
    const f32x4 wb = H > 1 ? *reinterpret_cast<const f32x4*>(w1 + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 dwa = {0.f, 0.f, 0.f, 0.f}, dwb = dwa;
    for (int l = 0; l < L; ++l) {
      const float sa = ds[0][l], sb = H > 1 ? ds[1][l] : 0.f;
      const f32x4 x = *reinterpret_cast<const f32x4*>(ctx + ((size_t)n * L + l) * D + i);
      *reinterpret_cast<f32x4*>(dctx + ((size_t)n * L + l) * D + i) = wa * sa + wb * sb;
      dwa += x * sa; dwb += x * sb;
    }
    *reinterpret_cast<f32x4*>(prt + i) = dwa;
    if (H > 1) *reinterpret_cast<f32x4*>(prt + D + i) = dwb;
  }
  if (tid < H) {
    float sb = 0.f;
    for (int l = 0; l < L; ++l) sb += ds[tid][l];
    prt[H * D + tid] = sb;
  }
}

// out[c] = sum_r in[r][c]  (fixed order: deterministic); rows is small (a batch), cols up to a few thousand
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, int ld, int rows, int cols, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  int r = 0;
  for (; r + 8 <= rows; r += 8) {               // eight loads in flight, added in the same order (one dependent load per row ran at load latency)
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = in[(size_t)(r + k) * ld + c];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
  }
  for (; r < rows; ++r) s += in[(size_t)r * ld + c];
  out[c] = s;
}

}  // namespace

extern "C" int dcn_colsum(const float* in, int ld, int rows, int cols, float* out, void* stream) {
  DCN_CHECK_ARG(in && out && rows > 0 && cols > 0 && ld >= cols, "colsum: bad argument");
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, (hipStream_t)stream, in, ld, rows, cols, out);
  DCN_CHECK_LAUNCH("colsum");
  return DCN_OK;
}

extern "C" int dcn_phrase_attn_fwd(const float* context, const float* embedded, const int64_t* ids,
                                   const float* w0, const float* b0, const float* w1, const float* b1,
                                   float* attn, float* out, float* vnorm, int n, int l, int d, int e, int normalize, void* stream) {
  DCN_CHECK_ARG(context && embedded && ids && w0 && b0 && attn && out, "phrase_attn_fwd: null pointer");
  DCN_CHECK_ARG(!normalize || vnorm, "phrase_attn_fwd: normalize needs vnorm");
  DCN_CHECK_ARG(n > 0 && l > 0 && l <= PH_MAXL && d % 4 == 0 && e > 0 && e <= PH_MAXE, "phrase_attn_fwd: bad shape (L=%d D=%d E=%d)", l, d, e);
  DCN_CHECK_ARG((w1 == nullptr) == (b1 == nullptr), "phrase_attn_fwd: w1/b1 must be given together");
  const int heads = w1 ? 2 : 1;
  hipLaunchKernelGGL(phrase_fwd_kernel, dim3(n, heads), dim3(256), 0, (hipStream_t)stream, context, embedded, ids, w0, b0, w1, b1,
                     attn, out, vnorm, n, l, d, e, normalize);
  DCN_CHECK_LAUNCH("phrase_attn_fwd");
  return DCN_OK;
}

extern "C" int64_t dcn_phrase_attn_bwd_ws(int n, int d, int heads) { return (int64_t)n * (heads * d + heads); }

extern "C" int dcn_phrase_attn_bwd(const float* context, const float* embedded, const float* w0, const float* w1,
                                   const float* attn, const float* out, const float* vnorm, const float* dout,
                                   float* dcontext, float* dembedded, float* dw, float* ws,
                                   int n, int l, int d, int e, int normalize, void* stream) {
  DCN_CHECK_ARG(context && embedded && w0 && attn && out && dout && dcontext && dembedded && dw && ws, "phrase_attn_bwd: null pointer");
  DCN_CHECK_ARG(!normalize || vnorm, "phrase_attn_bwd: normalize needs vnorm");
  DCN_CHECK_ARG(n > 0 && l > 0 && l <= PH_MAXL && d % 4 == 0 && e > 0 && e <= PH_MAXE, "phrase_attn_bwd: bad shape (L=%d D=%d E=%d)", l, d, e);
  const int heads = w1 ? 2 : 1;
  hipLaunchKernelGGL(phrase_bwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, context, embedded, w0, w1, attn, out, vnorm, dout,
                     dcontext, dembedded, ws, n, l, d, e, heads, normalize);
  DCN_CHECK_LAUNCH("phrase_attn_bwd");
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(heads * d + heads, 256)), dim3(256), 0, (hipStream_t)stream, ws, heads * d + heads, n, heads * d + heads, dw);
  DCN_CHECK_LAUNCH("phrase_attn_bwd colsum");
  return DCN_OK;
}
