// The cross-scale tail of grounding_model.forward (model/DCNet_model.py:545-621 == model/test_DCNet_model.py:413-477):
// objectness x similarity, the location module around csrc/locmod.hip, min-max normalisation, confidence modulation
// and the NHWC -> NCHW store of `outbox`, forward and backward, as a handful of kernels:
//
//   locemb_fwd    coord (P,8) -> Linear(8,8) -> BatchNorm1d(8) -> ReLU -> L2-normalise = E8 (P,8)     (:572-578)
//                 + the 8 / 8x8 moments of E8 that stand in for the (B*P,512) BatchNorm statistics
//   head_obj      only_obj = mean_anchor(conf) (:551), obj = only_obj*sim (:550), obj_map = normalize_P(obj) (:569),
//                 X[(n,k),p] = E8[p,k]*obj_map[n,p]  (K-padded operand of the rank-8 GEMM  M = X . W^T, :581-585)
//   locbn         BatchNorm1d(512) over the B*P rows of rel = E8_i.M_n + b from the moments (fp64), folded into
//                 M' = M*scale, b' = b*scale + shift                                                   (:585)
//   [dcn_locmod_fwd: relu, normalise over channels, dot with the phrase vector                         (:585-594)]
//   head_final    loc = (x - min)/(max - min + 1e-6) per image (:597), conf *= sim*loc (:612-621), outbox as NCHW
//
// and their gradients (head_dloc, locbn_bwd, head_fold, head_dlogits, locemb_bwd).  Everything here is a few MB per
// step: the roofline is launch latency, which is why the ~150 torch launches this replaces are ~13.
// Scales are concatenated along P in the order 0,1,2 (coarsest first), as the reference does (:565-568,604-610).
#include "common.h"

namespace {

struct Geom3 {
  int hw[3];        // positions per scale
  int off[3];       // first column of the scale in the concatenated [P] axis
  int ld[3];        // pixel stride (floats) of the NHWC logits of the scale (>= 15)
  int P, Ppad;
};
struct CPtr3 { const float* p[3]; };
struct Ptr3 { float* p[3]; };

__device__ __forceinline__ int scale_of(const Geom3& g, int p) { return p < g.off[1] ? 0 : (p < g.off[2] ? 1 : 2); }

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Sum K doubles per thread over the whole workgroup (NW waves); result in every thread.  lds: [NW + 1][K] doubles.
template <int K, int NW>
__device__ __forceinline__ void block_sum_d(double (&v)[K], double* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum_d(v[k]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) lds[wave * K + k] = v[k];
  }
  __syncthreads();
  if ((int)threadIdx.x < K) {
    double s = 0.0;
    for (int w = 0; w < NW; ++w) s += lds[w * K + threadIdx.x];
    lds[NW * K + threadIdx.x] = s;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = lds[NW * K + k];
}

__device__ __forceinline__ float block_sum_f(float v, float* red, int nw) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < nw; ++w) s += red[w];
  return s;
}

// ---- location embedding --------------------------------------------------------------------------------------
constexpr int LE_T = 256;                     // one workgroup; rows are walked with a stride of 256

__device__ __forceinline__ void le_linear(const float* __restrict__ coord, int p, const float* w_s, const float* b_s, float (&ce)[8]) {
  float c[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) c[j] = coord[(size_t)p * 8 + j];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float v = b_s[k];
#pragma unroll
    for (int j = 0; j < 8; ++j) v = fmaf(c[j], w_s[k * 8 + j], v);
    ce[k] = v;
  }
}

__global__ __launch_bounds__(LE_T) void locemb_fwd_kernel(const float* __restrict__ coord, const float* __restrict__ W,
                                                          const float* __restrict__ b, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, float momentum, float eps, int training,
                                                          long long count, int P, float* __restrict__ E8, float* __restrict__ xh,
                                                          float* __restrict__ stat, double* __restrict__ mom) {
  __shared__ double lds[(LE_T / 64 + 1) * 72];
  __shared__ float w_s[64], b_s[8], mean_s[8], rstd_s[8], g_s[8], be_s[8];
  const int tid = threadIdx.x;
  if (tid < 64) w_s[tid] = W[tid];
  if (tid < 8) { b_s[tid] = b[tid]; g_s[tid] = gamma[tid]; be_s[tid] = beta[tid]; }
  __syncthreads();
  if (training) {
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0;
    for (int p = tid; p < P; p += LE_T) {
      float ce[8];
      le_linear(coord, p, w_s, b_s, ce);
#pragma unroll
      for (int k = 0; k < 8; ++k) { acc[k] += (double)ce[k]; acc[8 + k] += (double)ce[k] * (double)ce[k]; }
    }
    block_sum_d<16, LE_T / 64>(acc, lds);
    if (tid < 8) {
      double a1 = acc[0], a2 = acc[8];
#pragma unroll
      for (int k = 1; k < 8; ++k) { a1 = tid == k ? acc[k] : a1; a2 = tid == k ? acc[8 + k] : a2; }
      const double mean = a1 / P;
      double var = a2 / P - mean * mean;
      var = var > 0.0 ? var : 0.0;
      mean_s[tid] = (float)mean; rstd_s[tid] = (float)(1.0 / sqrt(var + (double)eps));
      if (rmean) {        // every row is repeated for each of the B images: same mean / biased var, count = B*P rows
        rmean[tid] = (1.f - momentum) * rmean[tid] + momentum * (float)mean;
        rvar[tid] = (1.f - momentum) * rvar[tid] + momentum * (float)(var * (double)count / (double)(count - 1));
      }
    }
  } else if (tid < 8) {
    mean_s[tid] = rmean[tid]; rstd_s[tid] = 1.f / sqrtf(rvar[tid] + eps);
  }
  __syncthreads();
  if (tid < 8) { stat[tid] = mean_s[tid]; stat[8 + tid] = rstd_s[tid]; }
  double m[72];
#pragma unroll
  for (int k = 0; k < 72; ++k) m[k] = 0.0;
  for (int p = tid; p < P; p += LE_T) {
    float ce[8], z[8], ss = 0.f;
    le_linear(coord, p, w_s, b_s, ce);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float x = (ce[k] - mean_s[k]) * rstd_s[k];
      xh[(size_t)p * 8 + k] = x;
      z[k] = fmaxf(fmaf(g_s[k], x, be_s[k]), 0.f);
      ss = fmaf(z[k], z[k], ss);
    }
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int k = 0; k < 8; ++k) { z[k] *= inv; E8[(size_t)p * 8 + k] = z[k]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      m[k] += (double)z[k];
#pragma unroll
      for (int j = 0; j < 8; ++j) m[8 + k * 8 + j] += (double)z[k] * (double)z[j];
    }
  }
  block_sum_d<72, LE_T / 64>(m, lds);
  if (tid < 72) {
    double v = m[0];
#pragma unroll
    for (int k = 1; k < 72; ++k) v = tid == k ? m[k] : v;
    mom[tid] = v;
  }
}

// gradient of one row through normalise / relu: dy[8] (w.r.t. the BatchNorm output), given x-hat of the row
__device__ __forceinline__ void le_row_bwd(const float* __restrict__ xh, const float* __restrict__ dEa, const float* __restrict__ dEb,
                                           const double* dm_s, const float* g_s, const float* be_s, int p, float (&x)[8], float (&dy)[8]) {
  float y[8], e[8], ss = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    x[k] = xh[(size_t)p * 8 + k];
    y[k] = fmaf(g_s[k], x[k], be_s[k]);
    e[k] = fmaxf(y[k], 0.f);
    ss = fmaf(e[k], e[k], ss);
  }
  const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
  float dE[8], dot = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] *= inv;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    double v = dm_s[k];
#pragma unroll
    for (int j = 0; j < 8; ++j) v += (dm_s[8 + k * 8 + j] + dm_s[8 + j * 8 + k]) * (double)e[j];
    dE[k] = (float)v + (dEa ? dEa[(size_t)p * 8 + k] : 0.f) + (dEb ? dEb[(size_t)p * 8 + k] : 0.f);
    dot = fmaf(dE[k], e[k], dot);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) dy[k] = y[k] > 0.f ? (dE[k] - e[k] * dot) * inv : 0.f;
}

// dE = dEa + dEb + d(moments);  grads [88] = dW (64) | db (8) | dgamma (8) | dbeta (8)
__global__ __launch_bounds__(LE_T) void locemb_bwd_kernel(const float* __restrict__ coord, const float* __restrict__ W,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ xh, const float* __restrict__ stat,
                                                          const float* __restrict__ dEa, const float* __restrict__ dEb,
                                                          const double* __restrict__ dmom, int training, int P,
                                                          float* __restrict__ grads) {
  __shared__ double lds[(LE_T / 64 + 1) * 72];
  __shared__ float g_s[8], be_s[8], rstd_s[8];
  __shared__ double dm_s[72];
  const int tid = threadIdx.x;
  if (tid < 8) { g_s[tid] = gamma[tid]; be_s[tid] = beta[tid]; rstd_s[tid] = stat[8 + tid]; }
  if (tid < 72) dm_s[tid] = dmom ? dmom[tid] : 0.0;
  __syncthreads();
  double acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.0;
  for (int p = tid; p < P; p += LE_T) {
    float x[8], dy[8];
    le_row_bwd(xh, dEa, dEb, dm_s, g_s, be_s, p, x, dy);
#pragma unroll
    for (int k = 0; k < 8; ++k) { acc[k] += (double)dy[k] * (double)x[k]; acc[8 + k] += (double)dy[k]; }   // dgamma, dbeta
  }
  block_sum_d<16, LE_T / 64>(acc, lds);
  double w[72];
#pragma unroll
  for (int k = 0; k < 72; ++k) w[k] = 0.0;
  for (int p = tid; p < P; p += LE_T) {
    float x[8], dy[8], c[8];
    le_row_bwd(xh, dEa, dEb, dm_s, g_s, be_s, p, x, dy);
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = coord[(size_t)p * 8 + j];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float d = dy[k];
      if (training) d = d - (float)(acc[8 + k] / P) - x[k] * (float)(acc[k] / P);
      d *= g_s[k] * rstd_s[k];
      w[64 + k] += (double)d;
#pragma unroll
      for (int j = 0; j < 8; ++j) w[k * 8 + j] += (double)d * (double)c[j];
    }
  }
  block_sum_d<72, LE_T / 64>(w, lds);
  if (tid < 72) {
    double v = w[0];
#pragma unroll
    for (int k = 1; k < 72; ++k) v = tid == k ? w[k] : v;
    grads[tid] = (float)v;
  }
  if (tid < 16) {
    double v = acc[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) v = tid == k ? acc[k] : v;
    grads[72 + tid] = (float)v;
  }
}

// ---- objectness map and the rank-8 GEMM operand ---------------------------------------------------------------
constexpr int HO_T = 256, HO_R = 32;          // one workgroup per image, up to 8192 positions

__global__ __launch_bounds__(HO_T) void head_obj_kernel(const CPtr3 logits, const CPtr3 sim, const float* __restrict__ E8,
                                                        const Geom3 g, const Ptr3 only_obj, float* __restrict__ obj_map,
                                                        float* __restrict__ objn, float* __restrict__ X) {
  __shared__ float red[HO_T / 64];
  const int n = blockIdx.x, tid = threadIdx.x;
  float obj[HO_R];
  float ss = 0.f;
#pragma unroll
  for (int r = 0; r < HO_R; ++r) {
    const int p = tid + HO_T * r;
    obj[r] = 0.f;
    if (p < g.P) {
      const int s = scale_of(g, p), q = p - g.off[s];
      const float* l = logits.p[s] + ((size_t)n * g.hw[s] + q) * g.ld[s];
      const float oo = ((l[4] + l[9]) + l[14]) / 3.f;                           // :551
      only_obj.p[s][(size_t)n * g.hw[s] + q] = oo;
      obj[r] = oo * sim.p[s][(size_t)n * g.hw[s] + q];                          // :550
      ss = fmaf(obj[r], obj[r], ss);
    }
  }
  const float nrm = sqrtf(block_sum_f(ss, red, HO_T / 64));
  if (tid == 0) objn[n] = nrm;
  const float inv = 1.f / fmaxf(nrm, 1e-12f);                                   // :569
#pragma unroll
  for (int r = 0; r < HO_R; ++r) {
    const int p = tid + HO_T * r;
    if (p < g.Ppad) {
      const float om = p < g.P ? obj[r] * inv : 0.f;
      if (p < g.P) obj_map[(size_t)n * g.P + p] = om;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        X[((size_t)n * 8 + k) * g.Ppad + p] = p < g.P ? E8[(size_t)p * 8 + k] * om : 0.f;
    }
  }
}

// ---- BatchNorm1d(512) of the relation embedding, from the moments of E8 -----------------------------------------
// rel[n,i,c] = E8_i . M[n,:,c] + b[c].  With s1 = sum_i E8_i and S2 = sum_i E8_i E8_i^T (mom[0:8], mom[8:72]):
//   mu'[c] = sum_n s1.M[n,:,c] / cnt,   E[(rel-b)^2] = sum_n M[n,:,c]^T S2 M[n,:,c] / cnt,   var = that - mu'^2
// (the bias b cancels in train mode).  Output: M' = M*scale, b' = beta - scale*mu' (train) / b*scale + shift (eval).
constexpr int LC = 512;

__global__ __launch_bounds__(256) void locbn_fwd_kernel(const float* __restrict__ M, const double* __restrict__ mom,
                                                        const float* __restrict__ b, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ rmean,
                                                        float* __restrict__ rvar, float momentum, float eps, int training,
                                                        int B, long long count, float* __restrict__ Mp, float* __restrict__ bp,
                                                        float* __restrict__ saved /* [3][LC]: r, mu', scale */) {
  __shared__ double mo[72];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (threadIdx.x < 72) mo[threadIdx.x] = mom[threadIdx.x];
  __syncthreads();
  float scale, shift_b;
  if (training) {
    double t1 = 0.0, q2 = 0.0;
    for (int n = 0; n < B; ++n) {
      double m[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] = (double)M[((size_t)n * 8 + k) * LC + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        t1 += mo[k] * m[k];
        double u = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) u += mo[8 + k * 8 + j] * m[j];
        q2 += m[k] * u;
      }
    }
    const double cnt = (double)count;
    const double mu = t1 / cnt;
    double var = q2 / cnt - mu * mu;
    var = var > 0.0 ? var : 0.0;
    const double r = 1.0 / sqrt(var + (double)eps);
    scale = (float)((double)gamma[c] * r);
    shift_b = (float)((double)beta[c] - (double)gamma[c] * r * mu);
    saved[c] = (float)r; saved[LC + c] = (float)mu; saved[2 * LC + c] = scale;
    if (rmean) {
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)(mu + (double)b[c]);
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * cnt / (cnt - 1.0));
    }
  } else {
    const float r = 1.f / sqrtf(rvar[c] + eps);
    scale = gamma[c] * r;
    shift_b = b[c] * scale + (beta[c] - rmean[c] * scale);
    saved[c] = r; saved[LC + c] = 0.f; saved[2 * LC + c] = scale;
  }
  bp[c] = shift_b;
  for (int n = 0; n < B; ++n)
#pragma unroll
    for (int k = 0; k < 8; ++k) Mp[((size_t)n * 8 + k) * LC + c] = M[((size_t)n * 8 + k) * LC + c] * scale;
}

// One workgroup of 512 threads (a channel each).  out [3][LC] = dgamma | dbeta | db;  dmom [72] (train mode only).
__global__ __launch_bounds__(LC) void locbn_bwd_kernel(const float* __restrict__ M, const double* __restrict__ mom,
                                                       const float* __restrict__ b, const float* __restrict__ gamma,
                                                       const float* __restrict__ rmean, const float* __restrict__ saved,
                                                       const float* __restrict__ dMp, long long dmp_bs, const float* __restrict__ dbp,
                                                       int training, int B, long long count, float* __restrict__ dM,
                                                       float* __restrict__ out, double* __restrict__ dmom) {
  __shared__ double lds[(LC / 64 + 1) * 72];
  __shared__ double mo[72];
  const int c = threadIdx.x;
  if (c < 72) mo[c] = mom[c];
  __syncthreads();
  const float r = saved[c], scale = saved[2 * LC + c];
  const float gb = dbp[c];
  if (!training) {
    double A = 0.0;
    for (int n = 0; n < B; ++n)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const size_t i = ((size_t)n * 8 + k) * LC + c;
        const float g = dMp[(size_t)n * dmp_bs + k * LC + c];
        A += (double)g * (double)M[i];
        dM[i] = g * scale;
      }
    out[c] = (float)((A + (double)gb * ((double)b[c] - (double)rmean[c])) * (double)r);
    out[LC + c] = gb;
    out[2 * LC + c] = gb * scale;
    return;
  }
  const double mu = (double)saved[LC + c], cnt = (double)count, gam = (double)gamma[c];
  double A = 0.0, sm[8], smm[36];
#pragma unroll
  for (int k = 0; k < 8; ++k) sm[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 36; ++k) smm[k] = 0.0;
  for (int n = 0; n < B; ++n) {
    double m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t i = ((size_t)n * 8 + k) * LC + c;
      m[k] = (double)M[i];
      A += (double)dMp[(size_t)n * dmp_bs + k * LC + c] * m[k];
      sm[k] += m[k];
    }
    int t = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int j = k; j < 8; ++j) smm[t++] += m[k] * m[j];
  }
  const double core = A - (double)gb * mu;              // d(loss)/d(gamma*r) .. see DESIGN: M' = M*gamma*r, b' = beta - gamma*r*mu
  const double dgamma = (double)r * core;
  const double dr = gam * core;
  const double dvar = -0.5 * (double)r * (double)r * (double)r * dr;
  const double dmu = -gam * (double)r * (double)gb - 2.0 * mu * dvar;
  const double dq2 = dvar / cnt, dt1 = dmu / cnt;
  out[c] = (float)dgamma; out[LC + c] = gb; out[2 * LC + c] = 0.f;      // the Linear bias cancels under batch statistics
  for (int n = 0; n < B; ++n) {
    double m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = (double)M[((size_t)n * 8 + k) * LC + c];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double u = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) u += (mo[8 + k * 8 + j] + mo[8 + j * 8 + k]) * m[j];
      const size_t i = ((size_t)n * 8 + k) * LC + c;
      dM[i] = (float)((double)dMp[(size_t)n * dmp_bs + k * LC + c] * (double)scale + dt1 * mo[k] + dq2 * u);
    }
  }
  // moments: ds1[k] = sum_c dt1*sum_n M[n,k,c];  dS2[k][j] = sum_c dq2*sum_n M[n,k,c] M[n,j,c]
  double v[72];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = dt1 * sm[k];
  {
    int t = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int j = k; j < 8; ++j) { v[8 + k * 8 + j] = dq2 * smm[t]; v[8 + j * 8 + k] = dq2 * smm[t]; ++t; }
  }
  block_sum_d<72, LC / 64>(v, lds);
  if (c < 72) {
    double x = v[0];
#pragma unroll
    for (int k = 1; k < 72; ++k) x = c == k ? v[k] : x;
    dmom[c] = x;
  }
}

// ---- min-max normalisation, confidence modulation, NCHW store ----------------------------------------------------
constexpr int HF_T = 1024;

struct MinMax { float mn, mx; int imn, imx; };

__device__ __forceinline__ MinMax block_minmax(const float* __restrict__ x, int P, float* sv, int* si) {
  // first occurrence wins on ties (lowest index), like the index torch's min/max(dim) report
  float mn = INFINITY, mx = -INFINITY; int imn = 0x7fffffff, imx = 0x7fffffff;
  for (int p = threadIdx.x; p < P; p += HF_T) {
    const float v = x[p];
    if (v < mn) { mn = v; imn = p; }
    if (v > mx) { mx = v; imx = p; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float omn = __shfl_xor(mn, o), omx = __shfl_xor(mx, o);
    const int oimn = __shfl_xor(imn, o), oimx = __shfl_xor(imx, o);
    if (omn < mn || (omn == mn && oimn < imn)) { mn = omn; imn = oimn; }
    if (omx > mx || (omx == mx && oimx < imx)) { mx = omx; imx = oimx; }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { sv[wave] = mn; sv[16 + wave] = mx; si[wave] = imn; si[16 + wave] = imx; }
  __syncthreads();
  MinMax r{sv[0], sv[16], si[0], si[16]};
  for (int w = 1; w < HF_T / 64; ++w) {
    if (sv[w] < r.mn || (sv[w] == r.mn && si[w] < r.imn)) { r.mn = sv[w]; r.imn = si[w]; }
    if (sv[16 + w] > r.mx || (sv[16 + w] == r.mx && si[16 + w] < r.imx)) { r.mx = sv[16 + w]; r.imx = si[16 + w]; }
  }
  return r;
}

__global__ __launch_bounds__(HF_T) void head_final_fwd_kernel(const CPtr3 logits, const CPtr3 sim, const float* __restrict__ loc_map,
                                                              const Geom3 g, const Ptr3 outbox, const Ptr3 loc_score,
                                                              float* __restrict__ mm /* [B][4]: min, max, argmin, argmax */) {
  __shared__ float sv[32]; __shared__ int si[32];
  const int n = blockIdx.x;
  const float* x = loc_map + (size_t)n * g.P;
  const MinMax r = block_minmax(x, g.P, sv, si);
  if (threadIdx.x == 0) {
    mm[n * 4 + 0] = r.mn; mm[n * 4 + 1] = r.mx;
    mm[n * 4 + 2] = __int_as_float(r.imn); mm[n * 4 + 3] = __int_as_float(r.imx);
  }
  const float den = r.mx - r.mn + 1e-6f;                                        // :597
  for (int p = threadIdx.x; p < g.P; p += HF_T) {
    const int s = scale_of(g, p), q = p - g.off[s], hw = g.hw[s];
    const float lc = (x[p] - r.mn) / den;
    loc_score.p[s][(size_t)n * hw + q] = lc;
    const float* l = logits.p[s] + ((size_t)n * hw + q) * g.ld[s];
    float* o = outbox.p[s] + (size_t)n * 15 * hw + q;
#pragma unroll
    for (int ch = 0; ch < 15; ++ch) {
      const float v = l[ch];
      o[(size_t)ch * hw] = (ch % 5 == 4) ? v * sim.p[s][(size_t)n * hw + q] * lc : v;   // :618 (conf * sim * loc, in that order)
    }
  }
}

// dloc_map[n][p] from d(loc_score) and d(outbox): through conf*sim*loc and the min-max normalisation
__global__ __launch_bounds__(HF_T) void head_dloc_kernel(const CPtr3 logits, const CPtr3 sim, const CPtr3 loc_score,
                                                         const CPtr3 d_outbox, const CPtr3 d_loc, const float* __restrict__ mm,
                                                         const Geom3 g, float* __restrict__ dloc_map) {
  __shared__ double lds[(HF_T / 64 + 1) * 2];
  const int n = blockIdx.x;
  const float mn = mm[n * 4 + 0], mx = mm[n * 4 + 1];
  const int imn = __float_as_int(mm[n * 4 + 2]), imx = __float_as_int(mm[n * 4 + 3]);
  const float den = mx - mn + 1e-6f;
  double acc[2] = {0.0, 0.0};
  for (int p = threadIdx.x; p < g.P; p += HF_T) {
    const int s = scale_of(g, p), q = p - g.off[s], hw = g.hw[s];
    float dy = d_loc.p[s] ? d_loc.p[s][(size_t)n * hw + q] : 0.f;
    if (d_outbox.p[s]) {
      const float* l = logits.p[s] + ((size_t)n * hw + q) * g.ld[s];
      const float* d = d_outbox.p[s] + (size_t)n * 15 * hw + q;
      const float sm = sim.p[s][(size_t)n * hw + q];
      dy += (d[(size_t)4 * hw] * l[4] + d[(size_t)9 * hw] * l[9] + d[(size_t)14 * hw] * l[14]) * sm;
    }
    const float y = loc_score.p[s][(size_t)n * hw + q];
    dloc_map[(size_t)n * g.P + p] = dy / den;
    acc[0] += (double)dy; acc[1] += (double)dy * (double)y;
  }
  block_sum_d<2, HF_T / 64>(acc, lds);
  if (threadIdx.x == 0) {
    // y = (x - mn)/den:  d/dmn = sum dy*(y - 1)/den,  d/dmx = -sum dy*y/den
    dloc_map[(size_t)n * g.P + imn] += (float)((acc[1] - acc[0]) / (double)den);
    dloc_map[(size_t)n * g.P + imx] += (float)(-acc[1] / (double)den);
  }
}

// dX [(n,k)][Ppad] -> dobj_map[n][p] = sum_k dX*E8[p,k];  dE8x[p][k] = sum_n dX*obj_map[n,p]
__global__ __launch_bounds__(256) void head_fold_kernel(const float* __restrict__ dX, const float* __restrict__ E8,
                                                        const float* __restrict__ obj_map, int B, int P, int Ppad,
                                                        float* __restrict__ dobj_map, float* __restrict__ dE8x) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  float e[8], de[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { e[k] = E8[(size_t)p * 8 + k]; de[k] = 0.f; }
  for (int n = 0; n < B; ++n) {
    const float om = obj_map[(size_t)n * P + p];
    float d = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float x = dX[((size_t)n * 8 + k) * Ppad + p];
      d = fmaf(x, e[k], d);
      de[k] = fmaf(x, om, de[k]);
    }
    dobj_map[(size_t)n * P + p] = d;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) dE8x[(size_t)p * 8 + k] = de[k];
}

// d(logits) NHWC (pad channels zeroed) and d(sim) from d(outbox), d(only_obj) and the objectness chain
__global__ __launch_bounds__(HO_T) void head_dlogits_kernel(const CPtr3 logits, const CPtr3 sim, const CPtr3 loc_score,
                                                            const CPtr3 only_obj, const CPtr3 d_outbox, const CPtr3 d_only,
                                                            const float* __restrict__ obj_map, const float* __restrict__ objn,
                                                            const float* __restrict__ dobj_map, const Geom3 g,
                                                            const Ptr3 dlogits, const Ptr3 dsim) {
  __shared__ float red[HO_T / 64];
  const int n = blockIdx.x, tid = threadIdx.x;
  float dot = 0.f;
  if (dobj_map)
    for (int p = tid; p < g.P; p += HO_T) dot = fmaf(dobj_map[(size_t)n * g.P + p], obj_map[(size_t)n * g.P + p], dot);
  dot = block_sum_f(dot, red, HO_T / 64);
  const float inv = 1.f / fmaxf(objn[n], 1e-12f);
  for (int p = tid; p < g.P; p += HO_T) {
    const int s = scale_of(g, p), q = p - g.off[s], hw = g.hw[s];
    const size_t i = (size_t)n * hw + q;
    const float dobj = dobj_map ? (dobj_map[(size_t)n * g.P + p] - obj_map[(size_t)n * g.P + p] * dot) * inv : 0.f;
    const float sm = sim.p[s][i], lc = loc_score.p[s][i];
    float donly = dobj * sm + (d_only.p[s] ? d_only.p[s][i] : 0.f);
    float ds = dobj * only_obj.p[s][i];
    const float* l = logits.p[s] + i * g.ld[s];
    float* dl = dlogits.p[s] + i * g.ld[s];
    const float* d = d_outbox.p[s] ? d_outbox.p[s] + (size_t)n * 15 * hw + q : nullptr;
    const float third = donly / 3.f;
#pragma unroll
    for (int ch = 0; ch < 15; ++ch) {
      const float dv = d ? d[(size_t)ch * hw] : 0.f;
      if (ch % 5 == 4) { dl[ch] = dv * sm * lc + third; ds = fmaf(dv * l[ch], lc, ds); }
      else dl[ch] = dv;
    }
    for (int ch = 15; ch < g.ld[s]; ++ch) dl[ch] = 0.f;
    dsim.p[s][i] = ds;
  }
}

// dst[r][0:cols) = src[r][0:cols), dst[r][cols:ldd) = 0   (rows of any alignment: the (512, P) <-> (512, Ppad) copies)
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst, int ldd,
                                                       int rows, int cols, int cols_out) {
  const int r = blockIdx.y;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < cols_out; c += gridDim.x * 256)
    dst[(size_t)r * ldd + c] = c < cols ? src[(size_t)r * lds_ + c] : 0.f;
}

int fill_geom(Geom3& g, const int* hw, const int* ld, int pad32) {
  int off = 0;
  for (int s = 0; s < 3; ++s) {
    if (hw[s] <= 0 || ld[s] < 15) return -1;
    g.hw[s] = hw[s]; g.off[s] = off; g.ld[s] = ld[s]; off += hw[s];
  }
  g.P = off; g.Ppad = pad32 ? (off + 31) / 32 * 32 : off;
  return 0;
}

}  // namespace

extern "C" int dcn_pad_rows(const float* src, int lds, float* dst, int ldd, int rows, int cols, int cols_out, void* stream) {
  DCN_CHECK_ARG(src && dst && rows > 0 && cols > 0 && cols_out > 0 && lds >= cols && ldd >= cols_out, "pad_rows: bad argument");
  hipLaunchKernelGGL(pad_rows_kernel, dim3(cdiv(cols_out, 1024) > 0 ? cdiv(cols_out, 1024) : 1, rows), dim3(256), 0, (hipStream_t)stream,
                     src, lds, dst, ldd, rows, cols, cols_out);
  DCN_CHECK_LAUNCH("pad_rows");
  return DCN_OK;
}

extern "C" int dcn_locemb_fwd(const float* coord, const float* w, const float* b, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, int training, int64_t count,
                              int p, float* e8, float* xhat, float* stat, double* mom, void* stream) {
  DCN_CHECK_ARG(coord && w && b && gamma && beta && e8 && xhat && stat && mom, "locemb_fwd: null pointer");
  DCN_CHECK_ARG(p > 1, "locemb_fwd: %d positions", p);
  DCN_CHECK_ARG(training ? count > 1 : (running_mean && running_var), "locemb_fwd: eval mode needs the running statistics");
  hipLaunchKernelGGL(locemb_fwd_kernel, dim3(1), dim3(LE_T), 0, (hipStream_t)stream, coord, w, b, gamma, beta, running_mean, running_var,
                     momentum, eps, training, (long long)count, p, e8, xhat, stat, mom);
  DCN_CHECK_LAUNCH("locemb_fwd");
  return DCN_OK;
}

extern "C" int dcn_locemb_bwd(const float* coord, const float* w, const float* gamma, const float* beta, const float* xhat,
                              const float* stat, const float* de_a, const float* de_b, const double* dmom, int training, int p,
                              float* grads, void* stream) {
  DCN_CHECK_ARG(coord && w && gamma && beta && xhat && stat && grads, "locemb_bwd: null pointer");
  DCN_CHECK_ARG(p > 1, "locemb_bwd: %d positions", p);
  hipLaunchKernelGGL(locemb_bwd_kernel, dim3(1), dim3(LE_T), 0, (hipStream_t)stream, coord, w, gamma, beta, xhat, stat, de_a, de_b, dmom,
                     training, p, grads);
  DCN_CHECK_LAUNCH("locemb_bwd");
  return DCN_OK;
}

extern "C" int dcn_head_obj(const float* const* logits, const int* ld, const float* const* sim, const int* hw, const float* e8,
                            float* const* only_obj, float* obj_map, float* objn, float* x, int b, void* stream) {
  DCN_CHECK_ARG(logits && ld && sim && hw && e8 && only_obj && obj_map && objn && x && b > 0, "head_obj: null pointer");
  Geom3 g;
  DCN_CHECK_ARG(fill_geom(g, hw, ld, 1) == 0 && g.Ppad <= HO_T * HO_R, "head_obj: bad geometry (P=%d)", g.P);
  CPtr3 lg, sm; Ptr3 oo;
  for (int s = 0; s < 3; ++s) {
    DCN_CHECK_ARG(logits[s] && sim[s] && only_obj[s], "head_obj: null scale pointer");
    lg.p[s] = logits[s]; sm.p[s] = sim[s]; oo.p[s] = only_obj[s];
  }
  hipLaunchKernelGGL(head_obj_kernel, dim3(b), dim3(HO_T), 0, (hipStream_t)stream, lg, sm, e8, g, oo, obj_map, objn, x);
  DCN_CHECK_LAUNCH("head_obj");
  return DCN_OK;
}

extern "C" int dcn_locbn_fwd(const float* m, const double* mom, const float* b, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, float momentum, float eps, int training, int batch,
                             int64_t count, int c, float* mp, float* bp, float* saved, void* stream) {
  DCN_CHECK_ARG(m && mom && b && gamma && beta && mp && bp && saved && batch > 0, "locbn_fwd: null pointer");
  DCN_CHECK_ARG(c == LC, "locbn_fwd: channel count %d (built for %d)", c, LC);
  DCN_CHECK_ARG(training ? count > 1 : (running_mean && running_var), "locbn_fwd: eval mode needs the running statistics");
  hipLaunchKernelGGL(locbn_fwd_kernel, dim3(LC / 256), dim3(256), 0, (hipStream_t)stream, m, mom, b, gamma, beta, running_mean, running_var,
                     momentum, eps, training, batch, (long long)count, mp, bp, saved);
  DCN_CHECK_LAUNCH("locbn_fwd");
  return DCN_OK;
}

extern "C" int dcn_locbn_bwd(const float* m, const double* mom, const float* b, const float* gamma, const float* running_mean,
                             const float* saved, const float* dmp, int64_t dmp_bs, const float* dbp, int training, int batch,
                             int64_t count, int c, float* dm, float* out, double* dmom, void* stream) {
  DCN_CHECK_ARG(m && mom && b && gamma && saved && dmp && dbp && dm && out && batch > 0, "locbn_bwd: null pointer");
  DCN_CHECK_ARG(c == LC, "locbn_bwd: channel count %d (built for %d)", c, LC);
  DCN_CHECK_ARG(training ? dmom != nullptr : running_mean != nullptr, "locbn_bwd: dmom (train) / running_mean (eval) missing");
  hipLaunchKernelGGL(locbn_bwd_kernel, dim3(1), dim3(LC), 0, (hipStream_t)stream, m, mom, b, gamma, running_mean, saved, dmp,
                     (long long)(dmp_bs > 0 ? dmp_bs : 8 * LC), dbp, training, batch, (long long)count, dm, out, dmom);
  DCN_CHECK_LAUNCH("locbn_bwd");
  return DCN_OK;
}

extern "C" int dcn_head_final_fwd(const float* const* logits, const int* ld, const float* const* sim, const int* hw,
                                  const float* loc_map, float* const* outbox, float* const* loc_score, float* minmax, int b, void* stream) {
  DCN_CHECK_ARG(logits && ld && sim && hw && loc_map && outbox && loc_score && minmax && b > 0, "head_final_fwd: null pointer");
  Geom3 g;
  DCN_CHECK_ARG(fill_geom(g, hw, ld, 0) == 0, "head_final_fwd: bad geometry");
  CPtr3 lg, sm; Ptr3 ob, ls;
  for (int s = 0; s < 3; ++s) {
    DCN_CHECK_ARG(logits[s] && sim[s] && outbox[s] && loc_score[s], "head_final_fwd: null scale pointer");
    lg.p[s] = logits[s]; sm.p[s] = sim[s]; ob.p[s] = outbox[s]; ls.p[s] = loc_score[s];
  }
  hipLaunchKernelGGL(head_final_fwd_kernel, dim3(b), dim3(HF_T), 0, (hipStream_t)stream, lg, sm, loc_map, g, ob, ls, minmax);
  DCN_CHECK_LAUNCH("head_final_fwd");
  return DCN_OK;
}

extern "C" int dcn_head_dloc(const float* const* logits, const int* ld, const float* const* sim, const int* hw,
                             const float* const* loc_score, const float* const* d_outbox, const float* const* d_loc,
                             const float* minmax, float* dloc_map, int b, void* stream) {
  DCN_CHECK_ARG(logits && ld && sim && hw && loc_score && d_outbox && d_loc && minmax && dloc_map && b > 0, "head_dloc: null pointer");
  Geom3 g;
  DCN_CHECK_ARG(fill_geom(g, hw, ld, 0) == 0, "head_dloc: bad geometry");
  CPtr3 lg, sm, ls, dob, dl;
  for (int s = 0; s < 3; ++s) {
    DCN_CHECK_ARG(logits[s] && sim[s] && loc_score[s], "head_dloc: null scale pointer");
    lg.p[s] = logits[s]; sm.p[s] = sim[s]; ls.p[s] = loc_score[s]; dob.p[s] = d_outbox[s]; dl.p[s] = d_loc[s];
  }
  hipLaunchKernelGGL(head_dloc_kernel, dim3(b), dim3(HF_T), 0, (hipStream_t)stream, lg, sm, ls, dob, dl, minmax, g, dloc_map);
  DCN_CHECK_LAUNCH("head_dloc");
  return DCN_OK;
}

extern "C" int dcn_head_fold(const float* dx, const float* e8, const float* obj_map, int b, int p, int ppad, float* dobj_map,
                             float* de8, void* stream) {
  DCN_CHECK_ARG(dx && e8 && obj_map && dobj_map && de8 && b > 0 && p > 0 && ppad >= p, "head_fold: bad argument");
  hipLaunchKernelGGL(head_fold_kernel, dim3(cdiv(p, 256)), dim3(256), 0, (hipStream_t)stream, dx, e8, obj_map, b, p, ppad, dobj_map, de8);
  DCN_CHECK_LAUNCH("head_fold");
  return DCN_OK;
}

extern "C" int dcn_head_dlogits(const float* const* logits, const int* ld, const float* const* sim, const int* hw,
                                const float* const* loc_score, const float* const* only_obj, const float* const* d_outbox,
                                const float* const* d_only, const float* obj_map, const float* objn, const float* dobj_map,
                                float* const* dlogits, float* const* dsim, int b, void* stream) {
  DCN_CHECK_ARG(logits && ld && sim && hw && loc_score && only_obj && d_outbox && d_only && obj_map && objn && dlogits && dsim && b > 0,
                "head_dlogits: null pointer");
  Geom3 g;
  DCN_CHECK_ARG(fill_geom(g, hw, ld, 0) == 0, "head_dlogits: bad geometry");
  CPtr3 lg, sm, ls, oo, dob, don; Ptr3 dl, ds;
  for (int s = 0; s < 3; ++s) {
    DCN_CHECK_ARG(logits[s] && sim[s] && loc_score[s] && only_obj[s] && dlogits[s] && dsim[s], "head_dlogits: null scale pointer");
    lg.p[s] = logits[s]; sm.p[s] = sim[s]; ls.p[s] = loc_score[s]; oo.p[s] = only_obj[s]; dob.p[s] = d_outbox[s]; don.p[s] = d_only[s];
    dl.p[s] = dlogits[s]; ds.p[s] = dsim[s];
  }
  hipLaunchKernelGGL(head_dlogits_kernel, dim3(b), dim3(HO_T), 0, (hipStream_t)stream, lg, sm, ls, oo, dob, don, obj_map, objn, dobj_map,
                     g, dl, ds);
  DCN_CHECK_LAUNCH("head_dlogits");
  return DCN_OK;
}
