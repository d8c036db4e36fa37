// The caller-side pieces that drive the backward of the hot path, as kernels (the reference runs them as per-sample Python
// loops with .item() syncs):
//   target        build_target (train_DCNet.py:265-332) in compact form: per sample (best anchor, gi, gj | tx, ty, tw, th)
//   target_dense  the reference's dense (N,3,5,g,g) / (N,5,g,g) tensors from the compact form (API parity)
//   dense_loss    yolo_loss (:45-72), rank_loss (:173-203) and loc_loss (:205-220): all three read the maps at the
//                 positive cell and two log-sum-exps per sample; backward writes the dense d(outbox), d(loc) and the
//                 sparse d(sim), d(neg_sim) rows
//   contrastive   Interframe_contrastive_loss (:114-136) / Crossmodal_constrastive_loss (:140-166): rows of (q, k, M negatives),
//                 InfoNCE with T = 0.07; one wave per row, forward and backward
//   decode        evaluation decode (:764-810): global arg-max of the confidence over 3 scales x 3 anchors, box, xywh->xyxy
//   box_iou       utils/utils.py:76-104
// Sums over samples / rows are taken in a fixed order (deterministic), scalars stay on the device (no host sync).
#include "common.h"

namespace {

struct Grid3 { int g[3]; int off1[3]; int off3[3]; int P; };     // grid sides; first column of the scale in [P] / in [3P]
struct CP3 { const float* p[3]; };
struct P3 { float* p[3]; };

__host__ __device__ inline void grid3(Grid3& G, int size) {
  int o = 0;
  for (int s = 0; s < 3; ++s) { G.g[s] = size / (32 >> s); G.off1[s] = o; G.off3[s] = 3 * o; o += G.g[s] * G.g[s]; }
  G.P = o;
}

// anchors [3][3][2]: the reference's reversed table divided by (anchor_imsize / grid), rounded to float on the host
__global__ void target_kernel(const float* __restrict__ bbox, const float* __restrict__ anchors, int size, int N,
                              int* __restrict__ ti, float* __restrict__ tf) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  Grid3 G; grid3(G, size);
  const float lim = (float)(size - 1);
  float b[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) b[k] = fminf(fmaxf(bbox[n * 4 + k], 0.f), lim);      // :605
  const float fs = (float)size;
  const float cx = (b[0] + b[2]) / (2.f * fs), cy = (b[1] + b[3]) / (2.f * fs), w = (b[2] - b[0]) / fs, h = (b[3] - b[1]) / fs;   // :270-273
  float best = -1.f; int bn = 0;
  for (int s = 0; s < 3; ++s) {
    const float gw = w * (float)G.g[s], gh = h * (float)G.g[s];                    // :274
    for (int a = 0; a < 3; ++a) {
      const float aw = anchors[(s * 3 + a) * 2], ah = anchors[(s * 3 + a) * 2 + 1];
      const float inter = fmaxf(fminf(gw, aw), 0.f) * fmaxf(fminf(gh, ah), 0.f);
      const float iou = inter / (gw * gh + aw * ah - inter + 1e-16f);              // utils.bbox_iou
      if (iou > best) { best = iou; bn = s * 3 + a; }                              // first maximum (np.argmax, :305)
    }
  }
  const int s = bn / 3;
  const float g = (float)G.g[s];
  const float gx = cx * g, gy = cy * g, gw = w * g, gh = h * g;
  const int gi = (int)gx, gj = (int)gy;                                            // .long(): truncation (:312-313)
  ti[n * 4 + 0] = bn; ti[n * 4 + 1] = gi; ti[n * 4 + 2] = gj; ti[n * 4 + 3] = G.off1[s] + gj * G.g[s] + gi;
  tf[n * 4 + 0] = gx - (float)gi; tf[n * 4 + 1] = gy - (float)gj;                  // :314-315
  tf[n * 4 + 2] = logf(gw / anchors[bn * 2] + 1e-16f);                             // :319-320
  tf[n * 4 + 3] = logf(gh / anchors[bn * 2 + 1] + 1e-16f);
}

// dense tensors (pre-zeroed): bbox[s] (N,3,5,g,g), center[s] (N,5,g,g)  (:322-323)
__global__ void target_dense_kernel(const int* __restrict__ ti, const float* __restrict__ tf, int size, int N, const P3 box, const P3 ctr) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  Grid3 G; grid3(G, size);
  const int bn = ti[n * 4], gi = ti[n * 4 + 1], gj = ti[n * 4 + 2], s = bn / 3, a = bn % 3, g = G.g[s];
  for (int k = 0; k < 5; ++k) {
    const float v = k < 4 ? tf[n * 4 + k] : 1.f;
    box.p[s][((((size_t)n * 3 + a) * 5 + k) * g + gj) * g + gi] = v;
    ctr.p[s][(((size_t)n * 5 + k) * g + gj) * g + gi] = v;
  }
}

__device__ __forceinline__ float blk_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float blk_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// confidence logit j of sample n in the scale-concatenated [3P] order: scale, anchor, cell
__device__ __forceinline__ const float* conf_ptr(const CP3& ob, const Grid3& G, int n, int j) {
  const int s = j < G.off3[1] ? 0 : (j < G.off3[2] ? 1 : 2);
  const int gg = G.g[s] * G.g[s], r = j - G.off3[s], a = r / gg, cell = r - a * gg;
  return ob.p[s] + ((size_t)n * 15 + a * 5 + 4) * gg + cell;
}
__device__ __forceinline__ const float* map_ptr(const CP3& m, const Grid3& G, int n, int p) {
  const int s = p < G.off1[1] ? 0 : (p < G.off1[2] ? 1 : 2);
  return m.p[s] + (size_t)n * G.g[s] * G.g[s] + (p - G.off1[s]);
}

constexpr float RANK_MARGIN = 0.1f;

// vals [N][8]: squared coordinate errors (4), CE(conf), CE(loc), rank term, 0;  lse [N][2]
__global__ __launch_bounds__(256) void dense_loss_fwd_kernel(const CP3 outbox, const CP3 sim, const CP3 negsim, const CP3 loc,
                                                             const int* __restrict__ ti, const float* __restrict__ tf, int size, int N,
                                                             float* __restrict__ vals, float* __restrict__ lse) {
  __shared__ float red[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  Grid3 G; grid3(G, size);
  const int bn = ti[n * 4], gi = ti[n * 4 + 1], gj = ti[n * 4 + 2], cell = ti[n * 4 + 3];
  const int s = bn / 3, a = bn % 3, g = G.g[s], gg = g * g;
  // log-sum-exp of the 3P confidences and of the P location scores
  float mx = -INFINITY;
  for (int j = tid; j < 3 * G.P; j += 256) mx = fmaxf(mx, *conf_ptr(outbox, G, n, j));
  mx = blk_max(mx, red);
  float se = 0.f;
  for (int j = tid; j < 3 * G.P; j += 256) se += expf(*conf_ptr(outbox, G, n, j) - mx);
  const float lse_c = logf(blk_sum(se, red)) + mx;
  float mxl = -INFINITY;
  for (int p = tid; p < G.P; p += 256) mxl = fmaxf(mxl, *map_ptr(loc, G, n, p));
  mxl = blk_max(mxl, red);
  float sl = 0.f;
  for (int p = tid; p < G.P; p += 256) sl += expf(*map_ptr(loc, G, n, p) - mxl);
  const float lse_l = logf(blk_sum(sl, red)) + mxl;
  if (tid == 0) {
    const float* t = outbox.p[s] + ((size_t)n * 15 + a * 5) * gg + gj * g + gi;      // channel k at t[k*gg]
    float* v = vals + (size_t)n * 8;
    const float p0 = 1.f / (1.f + expf(-t[0])), p1 = 1.f / (1.f + expf(-t[gg]));
    const float d0 = p0 - tf[n * 4], d1 = p1 - tf[n * 4 + 1], d2 = t[2 * gg] - tf[n * 4 + 2], d3 = t[3 * gg] - tf[n * 4 + 3];
    v[0] = d0 * d0; v[1] = d1 * d1; v[2] = d2 * d2; v[3] = d3 * d3;
    v[4] = lse_c - t[4 * gg];                                                        // CE against the positive cell (:70)
    v[5] = lse_l - *map_ptr(loc, G, n, cell);
    const int cell_o = ti[(N - 1 - n) * 4 + 3];
    const float pp = *map_ptr(sim, G, n, cell), n1 = *map_ptr(negsim, G, n, cell), n2 = *map_ptr(sim, G, n, cell_o);
    v[6] = fmaxf(RANK_MARGIN + n1 - pp, 0.f) + fmaxf(RANK_MARGIN + n2 - pp, 0.f);   // :199
    v[7] = 0.f;
    lse[n * 2] = lse_c; lse[n * 2 + 1] = lse_l;
  }
}

// out[0] = yolo, out[1] = rank, out[2] = loc   (one workgroup; sums over n in index order)
__global__ __launch_bounds__(64) void dense_loss_reduce_kernel(const float* __restrict__ vals, int N, float* __restrict__ out) {
  const int k = threadIdx.x;
  if (k >= 7) return;
  __shared__ float tot[8];
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += vals[(size_t)n * 8 + k];
  tot[k] = s;
  __syncthreads();
  if (k == 0) {
    const float inv = 1.f / (float)N;
    out[0] = (((tot[0] * inv + tot[1] * inv) + tot[2] * inv) + tot[3] * inv) * 5.f + tot[4] * inv;     // :59-72
    out[1] = tot[6] / (float)(2 * N);                                                                    // :201
    out[2] = tot[5] * inv;
  }
}

// g[0..2] = upstream gradients of (yolo, rank, loc).  Every output element of sample n is written by block n.
__global__ __launch_bounds__(256) void dense_loss_bwd_kernel(const CP3 outbox, const CP3 sim, const CP3 negsim, const CP3 loc,
                                                             const int* __restrict__ ti, const float* __restrict__ tf,
                                                             const float* __restrict__ lse, const float* __restrict__ gup, int size, int N,
                                                             const P3 d_outbox, const P3 d_sim, const P3 d_negsim, const P3 d_loc) {
  const int n = blockIdx.x, tid = threadIdx.x;
  Grid3 G; grid3(G, size);
  const float gy = gup[0] / (float)N, gr = gup[1] / (float)(2 * N), gl = gup[2] / (float)N;
  const int bn = ti[n * 4], gi = ti[n * 4 + 1], gj = ti[n * 4 + 2], cell = ti[n * 4 + 3];
  const float lse_c = lse[n * 2], lse_l = lse[n * 2 + 1];
  for (int s = 0; s < 3; ++s) {
    const int gg = G.g[s] * G.g[s];
    const float* ob = outbox.p[s] + (size_t)n * 15 * gg;
    float* dob = d_outbox.p[s] + (size_t)n * 15 * gg;
    for (int i = tid; i < 15 * gg; i += 256) {
      const int ch = i / gg;
      dob[i] = (ch % 5 == 4) ? gy * expf(ob[i] - lse_c) : 0.f;
    }
    const float* lc = loc.p[s] + (size_t)n * gg;
    for (int i = tid; i < gg; i += 256) {
      d_loc.p[s][(size_t)n * gg + i] = gl * expf(lc[i] - lse_l);
      d_sim.p[s][(size_t)n * gg + i] = 0.f;
      d_negsim.p[s][(size_t)n * gg + i] = 0.f;
    }
  }
  __syncthreads();
  if (tid == 0) {
    const int s = bn / 3, a = bn % 3, g = G.g[s], gg = g * g;
    const size_t o = ((size_t)n * 15 + a * 5) * gg + gj * g + gi;
    const float* t = outbox.p[s] + o;
    float* d = d_outbox.p[s] + o;
    const float p0 = 1.f / (1.f + expf(-t[0])), p1 = 1.f / (1.f + expf(-t[gg]));
    d[0] = gy * 10.f * (p0 - tf[n * 4]) * p0 * (1.f - p0);               // w_coord 5 x d(mse) 2
    d[gg] = gy * 10.f * (p1 - tf[n * 4 + 1]) * p1 * (1.f - p1);
    d[2 * gg] = gy * 10.f * (t[2 * gg] - tf[n * 4 + 2]);
    d[3 * gg] = gy * 10.f * (t[3 * gg] - tf[n * 4 + 3]);
    d[4 * gg] -= gy;                                                      // softmax - onehot
    {
      const int sl = cell < G.off1[1] ? 0 : (cell < G.off1[2] ? 1 : 2);
      d_loc.p[sl][(size_t)n * G.g[sl] * G.g[sl] + cell - G.off1[sl]] -= gl;
    }
    const int cell_o = ti[(N - 1 - n) * 4 + 3];
    const float pp = *map_ptr(sim, G, n, cell), n1 = *map_ptr(negsim, G, n, cell), n2 = *map_ptr(sim, G, n, cell_o);
    const float a1 = RANK_MARGIN + n1 - pp > 0.f ? gr : 0.f, a2 = RANK_MARGIN + n2 - pp > 0.f ? gr : 0.f;
    const CP3 ds{{d_sim.p[0], d_sim.p[1], d_sim.p[2]}}, dn{{d_negsim.p[0], d_negsim.p[1], d_negsim.p[2]}};
    *const_cast<float*>(map_ptr(ds, G, n, cell)) -= a1 + a2;
    *const_cast<float*>(map_ptr(ds, G, n, cell_o)) += a2;
    *const_cast<float*>(map_ptr(dn, G, n, cell)) += a1;
  }
}

// ---- InfoNCE rows ------------------------------------------------------------------------------------------------
constexpr int CT_V4 = 4;          // E <= 1024
constexpr int CT_MAXM = 16;

__device__ __forceinline__ void ld_row(const float* __restrict__ p, int E, int lane, f32x4 (&v)[CT_V4]) {
#pragma unroll
  for (int k = 0; k < CT_V4; ++k) {
    const int c = (lane + 64 * k) * 4;
    v[k] = c < E ? *reinterpret_cast<const f32x4*>(p + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
__device__ __forceinline__ float dot_row(const f32x4 (&a)[CT_V4], const f32x4 (&b)[CT_V4]) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < CT_V4; ++k) s += a[k][0] * b[k][0] + a[k][1] * b[k][1] + a[k][2] * b[k][2] + a[k][3] * b[k][3];
  return wave_sum(s);
}

// loss_row[r] = logsumexp_i(s_i) - s_0,  s_0 = cos(q, k)/T,  s_m = cos(q, neg_m)/T        (one wave per row)
__global__ __launch_bounds__(256) void contrastive_fwd_kernel(const float* __restrict__ q, const float* __restrict__ pos,
                                                              const float* __restrict__ neg, int64_t R, int E, int M, float invT,
                                                              float* __restrict__ loss_row) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  f32x4 vq[CT_V4], vx[CT_V4];
  ld_row(q + r * E, E, lane, vq);
  const float iq = 1.f / fmaxf(sqrtf(dot_row(vq, vq)), 1e-12f);
  ld_row(pos + r * E, E, lane, vx);
  const float s0 = dot_row(vq, vx) * iq / fmaxf(sqrtf(dot_row(vx, vx)), 1e-12f) * invT;
  float sm[CT_MAXM], mx = s0;
#pragma unroll
  for (int m = 0; m < CT_MAXM; ++m) {
    sm[m] = -INFINITY;
    if (m < M) {
      ld_row(neg + (r * M + m) * E, E, lane, vx);
      sm[m] = dot_row(vq, vx) * iq / fmaxf(sqrtf(dot_row(vx, vx)), 1e-12f) * invT;
      mx = fmaxf(mx, sm[m]);
    }
  }
  float se = expf(s0 - mx);
#pragma unroll
  for (int m = 0; m < CT_MAXM; ++m) if (m < M) se += expf(sm[m] - mx);
  if (lane == 0) loss_row[r] = logf(se) + mx - s0;
}

// out[0] = scale * sum_r in[r]  (fixed tree: deterministic)
__global__ __launch_bounds__(1024) void sum_rows_kernel(const float* __restrict__ in, int64_t R, float scale, float* __restrict__ out) {
  __shared__ float red[1024];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < R; i += 1024) s += in[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// gradients of mean_r(loss_row) * gup[0]:  for x in {k, neg_m}: dx = g_x*(qh - xh*c_x)/(T*|x|),  dq = sum_x g_x*(xh - qh*c_x)/(T*|q|)
__global__ __launch_bounds__(256) void contrastive_bwd_kernel(const float* __restrict__ q, const float* __restrict__ pos,
                                                              const float* __restrict__ neg, int64_t R, int E, int M, float invT,
                                                              const float* __restrict__ gup, float scale,
                                                              float* __restrict__ dq, float* __restrict__ dpos, float* __restrict__ dneg) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float up = gup[0] * scale;
  f32x4 vq[CT_V4], vp[CT_V4], vx[CT_V4];
  ld_row(q + r * E, E, lane, vq);
  const float iq = 1.f / fmaxf(sqrtf(dot_row(vq, vq)), 1e-12f);
  ld_row(pos + r * E, E, lane, vp);
  const float ip = 1.f / fmaxf(sqrtf(dot_row(vp, vp)), 1e-12f);
  const float c0 = dot_row(vq, vp) * iq * ip;
  float cm[CT_MAXM], im[CT_MAXM], mx = c0 * invT;
#pragma unroll
  for (int m = 0; m < CT_MAXM; ++m) {
    cm[m] = 0.f; im[m] = 0.f;
    if (m < M) {
      ld_row(neg + (r * M + m) * E, E, lane, vx);
      im[m] = 1.f / fmaxf(sqrtf(dot_row(vx, vx)), 1e-12f);
      cm[m] = dot_row(vq, vx) * iq * im[m];
      mx = fmaxf(mx, cm[m] * invT);
    }
  }
  float se = expf(c0 * invT - mx);
#pragma unroll
  for (int m = 0; m < CT_MAXM; ++m) if (m < M) se += expf(cm[m] * invT - mx);
  const float g0 = (expf(c0 * invT - mx) / se - 1.f) * up * invT;
  f32x4 aq[CT_V4];                           // sum_x g_x * (xh - qh*c_x)
#pragma unroll
  for (int k = 0; k < CT_V4; ++k) {
    const int c = (lane + 64 * k) * 4;
    const f32x4 qh = vq[k] * iq, ph = vp[k] * ip;
    aq[k] = (ph - qh * c0) * g0;
    if (c < E) *reinterpret_cast<f32x4*>(dpos + r * E + c) = (qh - ph * c0) * (g0 * ip);
  }
#pragma unroll
  for (int m = 0; m < CT_MAXM; ++m)
    if (m < M) {
      const float gm = expf(cm[m] * invT - mx) / se * up * invT;
      ld_row(neg + (r * M + m) * E, E, lane, vx);
#pragma unroll
      for (int k = 0; k < CT_V4; ++k) {
        const int c = (lane + 64 * k) * 4;
        const f32x4 qh = vq[k] * iq, xh = vx[k] * im[m];
        aq[k] += (xh - qh * cm[m]) * gm;
        if (c < E) *reinterpret_cast<f32x4*>(dneg + (r * M + m) * E + c) = (qh - xh * cm[m]) * (gm * im[m]);
      }
    }
#pragma unroll
  for (int k = 0; k < CT_V4; ++k) {
    const int c = (lane + 64 * k) * 4;
    if (c < E) *reinterpret_cast<f32x4*>(dq + r * E + c) = aq[k] * iq;
  }
}

// ---- evaluation decode ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_kernel(const CP3 outbox, const float* __restrict__ anchors, int size, int N,
                                                     float* __restrict__ boxes, int* __restrict__ cellinfo) {
  __shared__ float sv[4]; __shared__ int si[4];
  const int n = blockIdx.x, tid = threadIdx.x;
  Grid3 G; grid3(G, size);
  float best = -INFINITY; int arg = 0x7fffffff;
  for (int j = tid; j < 3 * G.P; j += 256) {
    const float v = *conf_ptr(outbox, G, n, j);
    if (v > best) { best = v; arg = j; }                        // (ascending j per thread: first maximum)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(arg, o);
    if (ov > best || (ov == best && oi < arg)) { best = ov; arg = oi; }
  }
  if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = arg; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) if (sv[w] > best || (sv[w] == best && si[w] < arg)) { best = sv[w]; arg = si[w]; }
    const int s = arg < G.off3[1] ? 0 : (arg < G.off3[2] ? 1 : 2);                 // :779-787
    const int g = G.g[s], gg = g * g, r = arg - G.off3[s], a = r / gg, cell = r - a * gg, gj = cell / g, gi = cell - gj * g;
    const float* t = outbox.p[s] + ((size_t)n * 15 + a * 5) * gg + cell;
    const float stride = (float)(size / g);                                        // grid_size = 32 // 2^s (:789)
    const float x = (1.f / (1.f + expf(-t[0])) + (float)gi) * stride, y = (1.f / (1.f + expf(-t[gg])) + (float)gj) * stride;
    const float w = expf(t[2 * gg]) * anchors[(s * 3 + a) * 2] * stride, h = expf(t[3 * gg]) * anchors[(s * 3 + a) * 2 + 1] * stride;   // :805-809
    boxes[n * 4 + 0] = x - w / 2.f; boxes[n * 4 + 1] = y - h / 2.f; boxes[n * 4 + 2] = x + w / 2.f; boxes[n * 4 + 3] = y + h / 2.f;   // xywh2xyxy
    if (cellinfo) { cellinfo[n * 3] = s * 3 + a; cellinfo[n * 3 + 1] = gi; cellinfo[n * 3 + 2] = gj; }
  }
}

__global__ void box_iou_kernel(const float* __restrict__ b1, const float* __restrict__ b2, int N, float* __restrict__ iou) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float* a = b1 + n * 4; const float* b = b2 + n * 4;
  const float iw = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f), ih = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
  const float inter = iw * ih;
  iou[n] = inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter + 1e-16f);
}

int set3(CP3& c, const float* const* p) { for (int s = 0; s < 3; ++s) { if (!p || !p[s]) return -1; c.p[s] = p[s]; } return 0; }
int set3(P3& c, float* const* p) { for (int s = 0; s < 3; ++s) { if (!p || !p[s]) return -1; c.p[s] = p[s]; } return 0; }

}  // namespace

extern "C" int dcn_build_target(const float* bbox, const float* anchors, int size, int n, int* target_i, float* target_f, void* stream) {
  DCN_CHECK_ARG(bbox && anchors && target_i && target_f && n > 0 && size >= 32 && size % 32 == 0, "build_target: bad argument");
  hipLaunchKernelGGL(target_kernel, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, bbox, anchors, size, n, target_i, target_f);
  DCN_CHECK_LAUNCH("build_target");
  return DCN_OK;
}

extern "C" int dcn_target_dense(const int* target_i, const float* target_f, int size, int n, float* const* bbox_list,
                                float* const* center_list, void* stream) {
  P3 b, c;
  DCN_CHECK_ARG(target_i && target_f && n > 0 && set3(b, bbox_list) == 0 && set3(c, center_list) == 0, "target_dense: bad argument");
  hipLaunchKernelGGL(target_dense_kernel, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, target_i, target_f, size, n, b, c);
  DCN_CHECK_LAUNCH("target_dense");
  return DCN_OK;
}

extern "C" int dcn_dense_loss_fwd(const float* const* outbox, const float* const* sim, const float* const* negsim, const float* const* loc,
                                  const int* target_i, const float* target_f, int size, int n, float* vals, float* lse, float* out,
                                  void* stream) {
  CP3 ob, sm, ns, lc;
  DCN_CHECK_ARG(set3(ob, outbox) == 0 && set3(sm, sim) == 0 && set3(ns, negsim) == 0 && set3(lc, loc) == 0, "dense_loss_fwd: null map");
  DCN_CHECK_ARG(target_i && target_f && vals && lse && out && n > 0 && size >= 32 && size % 32 == 0, "dense_loss_fwd: bad argument");
  hipLaunchKernelGGL(dense_loss_fwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, ob, sm, ns, lc, target_i, target_f, size, n, vals, lse);
  DCN_CHECK_LAUNCH("dense_loss_fwd");
  hipLaunchKernelGGL(dense_loss_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, vals, n, out);
  DCN_CHECK_LAUNCH("dense_loss_reduce");
  return DCN_OK;
}

extern "C" int dcn_dense_loss_bwd(const float* const* outbox, const float* const* sim, const float* const* negsim, const float* const* loc,
                                  const int* target_i, const float* target_f, const float* lse, const float* grad_out, int size, int n,
                                  float* const* d_outbox, float* const* d_sim, float* const* d_negsim, float* const* d_loc, void* stream) {
  CP3 ob, sm, ns, lc; P3 dob, dsm, dns, dlc;
  DCN_CHECK_ARG(set3(ob, outbox) == 0 && set3(sm, sim) == 0 && set3(ns, negsim) == 0 && set3(lc, loc) == 0, "dense_loss_bwd: null map");
  DCN_CHECK_ARG(set3(dob, d_outbox) == 0 && set3(dsm, d_sim) == 0 && set3(dns, d_negsim) == 0 && set3(dlc, d_loc) == 0, "dense_loss_bwd: null output");
  DCN_CHECK_ARG(target_i && target_f && lse && grad_out && n > 0, "dense_loss_bwd: bad argument");
  hipLaunchKernelGGL(dense_loss_bwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, ob, sm, ns, lc, target_i, target_f, lse, grad_out,
                     size, n, dob, dsm, dns, dlc);
  DCN_CHECK_LAUNCH("dense_loss_bwd");
  return DCN_OK;
}

extern "C" int dcn_contrastive_fwd(const float* q, const float* pos, const float* neg, int64_t rows, int e, int m, float temperature,
                                   float* loss_rows, float* loss, void* stream) {
  DCN_CHECK_ARG(q && pos && neg && loss_rows && loss && rows > 0 && e > 0 && e % 4 == 0 && e <= 256 * CT_V4 && m > 0 && m <= CT_MAXM &&
                temperature > 0.f, "contrastive_fwd: bad argument (E=%d M=%d)", e, m);
  hipLaunchKernelGGL(contrastive_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, q, pos, neg, rows, e, m, 1.f / temperature, loss_rows);
  DCN_CHECK_LAUNCH("contrastive_fwd");
  hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, loss_rows, rows, 1.f / (float)rows, loss);
  DCN_CHECK_LAUNCH("contrastive sum");
  return DCN_OK;
}

extern "C" int dcn_contrastive_bwd(const float* q, const float* pos, const float* neg, int64_t rows, int e, int m, float temperature,
                                   const float* grad_out, float* dq, float* dpos, float* dneg, void* stream) {
  DCN_CHECK_ARG(q && pos && neg && grad_out && dq && dpos && dneg && rows > 0 && e > 0 && e % 4 == 0 && e <= 256 * CT_V4 && m > 0 &&
                m <= CT_MAXM && temperature > 0.f, "contrastive_bwd: bad argument (E=%d M=%d)", e, m);
  hipLaunchKernelGGL(contrastive_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, q, pos, neg, rows, e, m, 1.f / temperature,
                     grad_out, 1.f / (float)rows, dq, dpos, dneg);
  DCN_CHECK_LAUNCH("contrastive_bwd");
  return DCN_OK;
}

extern "C" int dcn_decode_boxes(const float* const* outbox, const float* anchors, int size, int n, float* boxes, int* cellinfo, void* stream) {
  CP3 ob;
  DCN_CHECK_ARG(set3(ob, outbox) == 0 && anchors && boxes && n > 0 && size >= 32 && size % 32 == 0, "decode_boxes: bad argument");
  hipLaunchKernelGGL(decode_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, ob, anchors, size, n, boxes, cellinfo);
  DCN_CHECK_LAUNCH("decode_boxes");
  return DCN_OK;
}

extern "C" int dcn_box_iou(const float* box1, const float* box2, int n, float* iou, void* stream) {
  DCN_CHECK_ARG(box1 && box2 && iou && n > 0, "box_iou: bad argument");
  hipLaunchKernelGGL(box_iou_kernel, dim3(cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, box1, box2, n, iou);
  DCN_CHECK_LAUNCH("box_iou");
  return DCN_OK;
}
