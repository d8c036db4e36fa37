// Location module core (model/DCNet_model.py:581-594), rank-8 form, fully fused.
//
// The reference builds, per image, a P x P relation matrix  G = (E E^T) diag(obj)  (E: P x 8 normalised
// coordinate embedding), pushes every row through Linear(P -> 512) + BatchNorm1d + ReLU, L2-normalises
// over the 512 channels and scores against the phrase vector:  50 MB / image and 12.9 GFLOP / image at
// 416x416.  Since rank(E E^T) <= 8,
//       Linear(G)[i, :] = E_i . (E^T diag(obj) W^T) + b = E_i . M_n + b ,     M_n : 8 x 512
// and with the BatchNorm statistics expressed through the 8x8 moments of E (host side, tiny tensors), the
// whole chain per position is
//       loc[n,i] = < normalize( relu( E_i . M'_n + b' ) ), q_n >
// with M' = M * bn_scale, b' = b * bn_scale + bn_shift.  This file computes exactly that, and its gradient
// with respect to E, M', b', q, in one pass each: no (N, P, 512) tensor is ever materialised.
// Roofline: VALU/latency (N*P*512*8 FMA = 0.9 GFLOP at C2); bytes are negligible.
#include "common.h"

namespace {

constexpr int LC = 512;          // channels (embdim)
constexpr int CPL = LC / 64;     // channels per lane
constexpr int ROWS_PER_WAVE = 16;

struct RowOut { float inv_norm, dotq; };

// per-lane channel c = lane + 64*j  (coalesced 256-B segments per j)
__device__ __forceinline__ void load_image(const float* __restrict__ Mn, const float* __restrict__ bp,
                                           const float* __restrict__ qn, int lane, float (&m)[8][CPL], float (&b)[CPL], float (&q)[CPL]) {
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = lane + 64 * j;
    b[j] = bp[c]; q[j] = qn[c];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k][j] = Mn[k * LC + c];
  }
}

__global__ __launch_bounds__(256) void locmod_fwd_kernel(const float* __restrict__ E, const float* __restrict__ Mp,
                                                         const float* __restrict__ bp, const float* __restrict__ q,
                                                         float* __restrict__ loc, int P) {
  const int n = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m[8][CPL], b[CPL], qq[CPL];
  load_image(Mp + (size_t)n * 8 * LC, bp, q + (size_t)n * LC, lane, m, b, qq);
  const int i0 = (blockIdx.x * 4 + wave) * ROWS_PER_WAVE;
  for (int r = 0; r < ROWS_PER_WAVE; ++r) {
    const int i = i0 + r;
    if (i >= P) break;
    float e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = E[(size_t)i * 8 + k];
    float ss = 0.f, dq = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      float z = b[j];
#pragma unroll
      for (int k = 0; k < 8; ++k) z = fmaf(e[k], m[k][j], z);
      z = fmaxf(z, 0.f);
      ss = fmaf(z, z, ss); dq = fmaf(z, qq[j], dq);
    }
    ss = wave_sum(ss); dq = wave_sum(dq);
    if (lane == 0) loc[(size_t)n * P + i] = dq / fmaxf(sqrtf(ss), 1e-12f);
  }
}

// partial[n][chunk][10][LC]: rows 0..7 = dM', 8 = db', 9 = dq;  dE_part[n][i][8]
__global__ __launch_bounds__(256) void locmod_bwd_kernel(const float* __restrict__ E, const float* __restrict__ Mp,
                                                         const float* __restrict__ bp, const float* __restrict__ q,
                                                         const float* __restrict__ dloc, float* __restrict__ partial,
                                                         float* __restrict__ dE_part, int P) {
  __shared__ float red[3][10][LC];
  const int n = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m[8][CPL], b[CPL], qq[CPL];
  load_image(Mp + (size_t)n * 8 * LC, bp, q + (size_t)n * LC, lane, m, b, qq);
  float acc[10][CPL];
#pragma unroll
  for (int a = 0; a < 10; ++a)
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[a][j] = 0.f;
  const int i0 = (blockIdx.x * 4 + wave) * ROWS_PER_WAVE;
  for (int r = 0; r < ROWS_PER_WAVE; ++r) {
    const int i = i0 + r;
    if (i >= P) break;
    float e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = E[(size_t)i * 8 + k];
    float z[CPL], ss = 0.f, dq = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      float v = b[j];
#pragma unroll
      for (int k = 0; k < 8; ++k) v = fmaf(e[k], m[k][j], v);
      z[j] = fmaxf(v, 0.f);
      ss = fmaf(z[j], z[j], ss); dq = fmaf(z[j], qq[j], dq);
    }
    ss = wave_sum(ss); dq = wave_sum(dq);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f), inv = 1.f / nrm;
    const float g = dloc[(size_t)n * P + i];
    // zhat = z*inv ; loc = <zhat, q> ; dz = g*inv*(q - zhat*loc) on active channels
    const float lv = dq * inv;
    float de[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) de[k] = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const float zh = z[j] * inv;
      acc[9][j] = fmaf(g, zh, acc[9][j]);                       // dq
      const float dz = z[j] > 0.f ? g * inv * (qq[j] - zh * lv) : 0.f;
      acc[8][j] += dz;                                          // db'
#pragma unroll
      for (int k = 0; k < 8; ++k) { acc[k][j] = fmaf(e[k], dz, acc[k][j]); de[k] = fmaf(m[k][j], dz, de[k]); }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) de[k] = wave_sum(de[k]);
    if (lane < 8) {
      float v = de[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) v = lane == k ? de[k] : v;
      dE_part[((size_t)n * P + i) * 8 + lane] = v;
    }
  }
  // combine the 4 waves' partial sums and write this block's slab
  if (wave > 0) {
#pragma unroll
    for (int a = 0; a < 10; ++a)
#pragma unroll
      for (int j = 0; j < CPL; ++j) red[wave - 1][a][lane + 64 * j] = acc[a][j];
  }
  __syncthreads();
  if (wave == 0) {
    float* out = partial + ((size_t)n * gridDim.x + blockIdx.x) * 10 * LC;
#pragma unroll
    for (int a = 0; a < 10; ++a)
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const int c = lane + 64 * j;
        out[a * LC + c] = acc[a][j] + red[0][a][c] + red[1][a][c] + red[2][a][c];
      }
  }
}

// out[n][a][c] = sum_chunk partial[n][chunk][a][c]
__global__ __launch_bounds__(256) void locmod_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int chunks) {
  const int n = blockIdx.y;
  const int idx = blockIdx.x * 256 + threadIdx.x;           // over 10*LC
  if (idx >= 10 * LC) return;
  float s = 0.f;
  for (int ch = 0; ch < chunks; ++ch) s += partial[((size_t)n * chunks + ch) * 10 * LC + idx];
  out[(size_t)n * 10 * LC + idx] = s;
}

inline int loc_chunks(int P) { return cdiv(P, 4 * ROWS_PER_WAVE); }

}  // namespace

extern "C" int64_t dcn_locmod_bwd_ws(int n, int p) { return (int64_t)n * loc_chunks(p) * 10 * LC; }

extern "C" int dcn_locmod_fwd(const float* E, const float* Mp, const float* bp, const float* q, float* loc,
                              int n, int p, int c, void* stream) {
  DCN_CHECK_ARG(E && Mp && bp && q && loc && n > 0 && p > 0, "locmod_fwd: bad argument");
  DCN_CHECK_ARG(c == LC, "locmod_fwd: channel count %d (built for %d)", c, LC);
  hipLaunchKernelGGL(locmod_fwd_kernel, dim3(loc_chunks(p), n), dim3(256), 0, (hipStream_t)stream, E, Mp, bp, q, loc, p);
  DCN_CHECK_LAUNCH("locmod_fwd");
  return DCN_OK;
}

extern "C" int dcn_locmod_bwd(const float* E, const float* Mp, const float* bp, const float* q, const float* dloc,
                              float* dE_part, float* dsum, float* ws, int n, int p, int c, void* stream) {
  DCN_CHECK_ARG(E && Mp && bp && q && dloc && dE_part && dsum && ws && n > 0 && p > 0, "locmod_bwd: bad argument");
  DCN_CHECK_ARG(c == LC, "locmod_bwd: channel count %d (built for %d)", c, LC);
  const int chunks = loc_chunks(p);
  hipLaunchKernelGGL(locmod_bwd_kernel, dim3(chunks, n), dim3(256), 0, (hipStream_t)stream, E, Mp, bp, q, dloc, ws, dE_part, p);
  DCN_CHECK_LAUNCH("locmod_bwd");
  hipLaunchKernelGGL(locmod_reduce_kernel, dim3(cdiv(10 * LC, 256), n), dim3(256), 0, (hipStream_t)stream, ws, dsum, chunks);
  DCN_CHECK_LAUNCH("locmod_reduce");
  return DCN_OK;
}
