// The two correspondence-sampling heads of the training model (scale 0 only):
//
//  K9  inter-frame (model/DCNet_model.py:381-430): top-30 of the flattened HW0 x HW0 affinity of a frame pair
//      (torch.topk, sorted, :395), q = frame-1 feature at idx // HW0 (:407), k = frame-2 feature at idx % HW0 (:409),
//      10 negatives from frame 2 at host-drawn positions that skip the k position (:411-418).
//  K14 cross-modal (:625-637 + Crossmodal_corrspondence :41-112): vit = normalize(fvisu[0], over POSITIONS) (:629),
//      lag = normalize(context[:, :, 0::2], over WORDS) (:631-632), map = lag . vit (:634), Conv1d(L,L,3) along the
//      positions (:287-290,635; the Softmax over words that follows is monotone, F8), top-1 word per position (:48),
//      positives lag[:, word] (:66-70) and 5 negatives per (image, position) from image N-1 (:75-96).
//
// Forward kernels select and gather; backward kernels are deterministic gathers "by destination" (every destination row
// scans the short index lists / a host-built CSR), so there are no float atomics and a training step stays bitwise
// reproducible.  All of it is a few tens of MB per step: latency-bound, on its own stream under the head convolutions.
#include "common.h"

namespace {

constexpr int SK_MAXK = 64;       // top-k entries per pair (the reference uses 30)
constexpr int SK_T = 1024;

__device__ __forceinline__ unsigned ord_key(float v) {          // order-preserving float -> uint
  const unsigned b = __float_as_uint(v);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

// copy one E-float row with a wave (E % 4 == 0)
__device__ __forceinline__ void wave_copy_row(const float* __restrict__ src, float* __restrict__ dst, int E, int lane) {
  for (int c = lane * 4; c < E; c += 256) *reinterpret_cast<f32x4*>(dst + c) = *reinterpret_cast<const f32x4*>(src + c);
}

// One workgroup per frame pair: radix-select the top_k largest of cmap[pair][HW*HW] (ties: lowest flat index first),
// sort them (value descending, index ascending), then gather q / k / negatives.
__global__ __launch_bounds__(SK_T) void k9_fwd_kernel(const float* __restrict__ cmap, const float* __restrict__ fv, const int64_t* __restrict__ raw_neg,
                                                      int HW, int E, int top_k, int neg_n,
                                                      int64_t* __restrict__ index, int64_t* __restrict__ neg_idx,
                                                      float* __restrict__ frame, float* __restrict__ corr, float* __restrict__ negf) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_krem, s_cnt;
  __shared__ unsigned ck[SK_MAXK]; __shared__ int ci[SK_MAXK];
  __shared__ int s_idx[SK_MAXK];
  __shared__ unsigned scan[SK_T];
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = HW * HW;
  const float* x = cmap + (size_t)pair * n;
  // ---- radix select: find the key T of the top_k-th largest element and how many elements equal to T are taken ----
  unsigned prefix = 0, mask = 0, krem = (unsigned)top_k;
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += SK_T) {
      const unsigned k = ord_key(x[i]);
      if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned acc = 0; int d = 255;
      for (; d > 0; --d) { if (acc + hist[d] >= krem) break; acc += hist[d]; }
      s_prefix = prefix | ((unsigned)d << shift); s_krem = krem - acc;
    }
    __syncthreads();
    prefix = s_prefix; krem = s_krem; mask |= 0xFFu << shift;
    __syncthreads();
  }
  const unsigned T = prefix;               // krem = number of elements with key == T that belong to the top_k
  // ---- collect: every key > T, and the krem lowest-index elements with key == T ----
  if (tid == 0) s_cnt = 0;
  // (a) ties at the threshold: each thread owns a contiguous index segment, so an exclusive scan of the per-thread
  //     counts orders them by index
  const int seg = (n + SK_T - 1) / SK_T;
  const int lo = tid * seg, hi = min(n, lo + seg);
  unsigned eq = 0;
  for (int i = lo; i < hi; ++i) eq += ord_key(x[i]) == T;
  scan[tid] = eq;
  __syncthreads();
  for (int o = 1; o < SK_T; o <<= 1) {      // Hillis-Steele inclusive scan
    const unsigned v = tid >= o ? scan[tid - o] : 0u;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  unsigned ord = scan[tid] - eq;            // ordinal of this thread's first tie
  const int ngt = top_k - (int)krem;        // slots [0, ngt) hold the keys > T, [ngt, top_k) the ties
  for (int i = lo; i < hi && ord < krem; ++i)
    if (ord_key(x[i]) == T) { ck[ngt + ord] = T; ci[ngt + ord] = i; ++ord; }
  // (b) keys above the threshold: slot order is arbitrary, the sort below fixes the output order
  for (int i = tid; i < n; i += SK_T) {
    const unsigned k = ord_key(x[i]);
    if (k > T) { const unsigned s = atomicAdd(&s_cnt, 1u); if (s < SK_MAXK) { ck[s] = k; ci[s] = i; } }
  }
  __syncthreads();
  if (tid < top_k) {                        // rank by counting: value descending, index ascending
    const unsigned k = ck[tid]; const int i = ci[tid];
    int rank = 0;
    for (int j = 0; j < top_k; ++j) rank += (ck[j] > k) || (ck[j] == k && ci[j] < i);
    s_idx[rank] = i;
  }
  __syncthreads();
  if (tid < top_k) index[(size_t)pair * top_k + tid] = s_idx[tid];
  // ---- gather: rows j of frame / corr, rows (j, m) of negf ----
  const float* p1 = fv + (size_t)(2 * pair) * HW * E;
  const float* p2 = p1 + (size_t)HW * E;
  const int rows = top_k * (2 + neg_n);
  for (int r = wave; r < rows; r += SK_T / 64) {
    if (r < top_k) {
      wave_copy_row(p1 + (size_t)(s_idx[r] / HW) * E, frame + ((size_t)pair * top_k + r) * E, E, lane);
    } else if (r < 2 * top_k) {
      const int j = r - top_k;
      wave_copy_row(p2 + (size_t)(s_idx[j] % HW) * E, corr + ((size_t)pair * top_k + j) * E, E, lane);
    } else {
      const int e = r - 2 * top_k, j = e / neg_n;
      const size_t o = (size_t)pair * top_k * neg_n + e;
      const int64_t raw = raw_neg[o];
      const int64_t pos = raw + (raw >= (int64_t)(s_idx[j] % HW) ? 1 : 0);        // skip the removed element (:411-413)
      if (lane == 0) neg_idx[o] = pos;
      wave_copy_row(p2 + (size_t)pos * E, negf + o * E, E, lane);
    }
  }
}

// Gradient of the K9 gathers, by destination: block (pair, frame, group of GP positions), 2 channels per thread.  GP lanes first
// list, per position, the entries of the (direct | negative) lists that point at it — ascending, so the sums keep their order — and
// the channel threads then add those few rows (most positions have none and are a zero fill; scanning all 30 + 300 entries per
// position and channel cost 0.64 ms at 32 pairs of 52 x 52).
constexpr int K9B_GP = 16;
__global__ __launch_bounds__(256) void k9_bwd_kernel(const int64_t* __restrict__ index, const int64_t* __restrict__ neg_idx,
                                                     const float* __restrict__ d_frame, const float* __restrict__ d_corr,
                                                     const float* __restrict__ d_neg, int HW, int E, int top_k, int neg_n,
                                                     float* __restrict__ dfv) {
  extern __shared__ int lst[];              // [nl] list positions (direct, then negatives) | [GP] match counts | [GP][nl] matching entries
  const int pair = blockIdx.z, f = blockIdx.y, tid = threadIdx.x;
  const int nneg = top_k * neg_n, nl = top_k + (f == 1 ? nneg : 0), nl_max = top_k + nneg;
  int* cnt = lst + nl_max; int* match = cnt + K9B_GP;
  for (int j = tid; j < top_k; j += 256) {
    const int64_t i = index[(size_t)pair * top_k + j];
    lst[j] = f == 0 ? (int)(i / HW) : (int)(i % HW);
  }
  if (f == 1) for (int e = tid; e < nneg; e += 256) lst[top_k + e] = (int)neg_idx[(size_t)pair * nneg + e];
  __syncthreads();
  if (tid < K9B_GP) {
    const int pos = blockIdx.x * K9B_GP + tid;
    int m = 0;
    for (int j = 0; j < nl; ++j) if (lst[j] == pos) match[tid * nl_max + m++] = j;
    cnt[tid] = m;
  }
  __syncthreads();
  const float* dd = (f == 0 ? d_frame : d_corr) + (size_t)pair * top_k * E;
  const float* dn = d_neg + (size_t)pair * nneg * E;
  for (int g = 0; g < K9B_GP; ++g) {
    const int pos = blockIdx.x * K9B_GP + g;
    if (pos >= HW) break;
    const int m = cnt[g];
    for (int c = tid; c < E; c += 256) {
      float acc = 0.f;
      for (int i = 0; i < m; ++i) {
        const int j = match[g * nl_max + i];
        acc += j < top_k ? dd[(size_t)j * E + c] : dn[(size_t)(j - top_k) * E + c];
      }
      dfv[((size_t)(2 * pair + f) * HW + pos) * E + c] = acc;
    }
  }
}

// ---- K14 -----------------------------------------------------------------------------------------------------------
// vit[n][p][c] = v[n][p][c] / max(||v[n][:, c]||, 1e-12)   (normalised over the POSITIONS of each channel, :629)
__global__ __launch_bounds__(256) void colnorm_fwd_kernel(const float* __restrict__ v, int HW, int E, float* __restrict__ vit,
                                                          float* __restrict__ cnorm) {
  __shared__ float red[4][64];
  const int n = blockIdx.y, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const float* src = v + (size_t)n * HW * E;
  float ss = 0.f;
  if (c < E) {
    int p = ty;
    for (; p + 28 < HW; p += 32) {           // eight loads in flight, squared and added in position order
      float x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = src[(size_t)(p + 4 * k) * E + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) ss = fmaf(x[k], x[k], ss);
    }
    for (; p < HW; p += 4) { const float x = src[(size_t)p * E + c]; ss = fmaf(x, x, ss); }
  }
  red[ty][tx] = ss;
  __syncthreads();
  const float nrm = sqrtf(red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
  if (c >= E) return;
  if (ty == 0) cnorm[(size_t)n * E + c] = nrm;
  const float inv = 1.f / fmaxf(nrm, 1e-12f);
  for (int p = ty; p < HW; p += 4) vit[((size_t)n * HW + p) * E + c] = src[(size_t)p * E + c] * inv;
}

// dv = (dvit - vit * sum_p(dvit*vit)) / max(cnorm, eps), dvit = dq (+ extra for image n_extra)
__global__ __launch_bounds__(256) void colnorm_bwd_kernel(const float* __restrict__ vit, const float* __restrict__ cnorm,
                                                          const float* __restrict__ dq, const float* __restrict__ extra, int n_extra,
                                                          int HW, int E, float* __restrict__ dv) {
  __shared__ float red[4][64];
  const int n = blockIdx.y, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const bool ex = extra != nullptr && n == n_extra;
  float dot = 0.f;
  // eight positions of loads in flight per thread (in position order): one position per iteration ran at the latency of a load
  if (c < E) {
    int p = ty;
    for (; p + 28 < HW; p += 32) {
      float g[8], v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const size_t i = ((size_t)n * HW + p + 4 * k) * E + c;
        g[k] = dq[i] + (ex ? extra[(size_t)(p + 4 * k) * E + c] : 0.f); v[k] = vit[i];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) dot = fmaf(g[k], v[k], dot);
    }
    for (; p < HW; p += 4) {
      const size_t i = ((size_t)n * HW + p) * E + c;
      const float g = dq[i] + (ex ? extra[(size_t)p * E + c] : 0.f);
      dot = fmaf(g, vit[i], dot);
    }
  }
  red[ty][tx] = dot;
  __syncthreads();
  dot = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
  if (c >= E) return;
  const float inv = 1.f / fmaxf(cnorm[(size_t)n * E + c], 1e-12f);
  int p = ty;
  for (; p + 28 < HW; p += 32) {
    float g[8], v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t i = ((size_t)n * HW + p + 4 * k) * E + c;
      g[k] = dq[i] + (ex ? extra[(size_t)(p + 4 * k) * E + c] : 0.f); v[k] = vit[i];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) dv[((size_t)n * HW + p + 4 * k) * E + c] = (g[k] - v[k] * dot) * inv;
  }
  for (; p < HW; p += 4) {
    const size_t i = ((size_t)n * HW + p) * E + c;
    const float g = dq[i] + (ex ? extra[(size_t)p * E + c] : 0.f);
    dv[i] = (g - vit[i] * dot) * inv;
  }
}

// lag[n][l][c] = ctx[n][l][2c] / max(||ctx[n][:, 2c]||, 1e-12)   (F.interpolate(scale 0.5) = even channels; over WORDS)
__global__ __launch_bounds__(256) void lagnorm_fwd_kernel(const float* __restrict__ ctx, int L, int D, float* __restrict__ lag,
                                                          float* __restrict__ lnorm) {
  const int n = blockIdx.x, E = D / 2;
  for (int c = threadIdx.x; c < E; c += 256) {
    float ss = 0.f;
    for (int l = 0; l < L; ++l) { const float x = ctx[((size_t)n * L + l) * D + 2 * c]; ss = fmaf(x, x, ss); }
    const float nrm = sqrtf(ss);
    lnorm[(size_t)n * E + c] = nrm;
    const float inv = 1.f / fmaxf(nrm, 1e-12f);
    for (int l = 0; l < L; ++l) lag[((size_t)n * L + l) * E + c] = ctx[((size_t)n * L + l) * D + 2 * c] * inv;
  }
}

// cols[n][p] = argmax_lo ( bias[lo] + sum_li sum_t w[lo][li][t] * <lag[n][li], vit[n][p+t-1]> )   (first maximum)
constexpr int CM_MAXL = 32;
__global__ __launch_bounds__(1024) void crossmap_kernel(const float* __restrict__ lag, const float* __restrict__ vit,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        int L, int HW, int E, int64_t* __restrict__ cols, float* __restrict__ lvmap) {
  extern __shared__ float sm[];             // lv [L][HW + 2] | w_s [L*L*3] | b_s [L]   (lag[n], 40 KB, is read through L1/L2)
  float* lv = sm;
  float* w_s = lv + (size_t)L * (HW + 2);
  float* b_s = w_s + L * L * 3;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* lag_s = lag + (size_t)n * L * E;
  for (int i = tid; i < L * L * 3; i += 1024) w_s[i] = w[i];
  if (tid < L) { b_s[tid] = bias[tid]; lv[tid * (HW + 2)] = 0.f; lv[tid * (HW + 2) + HW + 1] = 0.f; }
  __syncthreads();
  for (int p = wave; p < HW; p += 16) {
    const float* row = vit + ((size_t)n * HW + p) * E;
    if (E <= 1024) {                          // the position's row stays in registers for all L words (same sums, same order)
      f32x4 a[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + 256 * k;
        a[k] = c < E ? *reinterpret_cast<const f32x4*>(row + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      for (int l = 0; l < L; ++l) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = lane * 4 + 256 * k;
          if (c < E) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(lag_s + (size_t)l * E + c);
            acc += a[k][0] * b[0] + a[k][1] * b[1] + a[k][2] * b[2] + a[k][3] * b[3];
          }
        }
        acc = wave_sum(acc);
        if (lane == 0) lv[l * (HW + 2) + p + 1] = acc;
      }
      continue;
    }
    for (int l = 0; l < L; ++l) {
      float acc = 0.f;
      for (int c = lane * 4; c < E; c += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(row + c), b = *reinterpret_cast<const f32x4*>(lag_s + (size_t)l * E + c);
        acc += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
      }
      acc = wave_sum(acc);
      if (lane == 0) lv[l * (HW + 2) + p + 1] = acc;
    }
  }
  __syncthreads();
  for (int p = tid; p < HW; p += 1024) {
    float best = -INFINITY; int arg = 0;
    for (int lo = 0; lo < L; ++lo) {
      float v = b_s[lo];
      for (int li = 0; li < L; ++li) {
        const float* r = lv + li * (HW + 2) + p;          // r[0..2] = positions p-1, p, p+1 (zero padded)
        const float* ww = w_s + (lo * L + li) * 3;
        v = fmaf(ww[0], r[0], v); v = fmaf(ww[1], r[1], v); v = fmaf(ww[2], r[2], v);
      }
      if (lvmap) lvmap[((size_t)n * L + lo) * HW + p] = v;
      if (v > best) { best = v; arg = lo; }
    }
    cols[(size_t)n * HW + p] = arg;
  }
}

// lag_pos[n][p] = lag[n][cols[n][p]];  neg_cross[n][p][m] = vit[N-1][neg[n][p][m]]      (one wave per (n, p))
__global__ __launch_bounds__(256) void k14_gather_kernel(const float* __restrict__ lag, const float* __restrict__ vit,
                                                         const int64_t* __restrict__ cols, const int64_t* __restrict__ neg,
                                                         int N, int L, int HW, int E, int neg_n,
                                                         float* __restrict__ lag_pos, float* __restrict__ neg_cross) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)N * HW) return;
  const int n = (int)(row / HW);
  wave_copy_row(lag + ((size_t)n * L + cols[row]) * E, lag_pos + (size_t)row * E, E, lane);
  const float* last = vit + (size_t)(N - 1) * HW * E;
  for (int m = 0; m < neg_n; ++m)
    wave_copy_row(last + (size_t)neg[row * neg_n + m] * E, neg_cross + ((size_t)row * neg_n + m) * E, E, lane);
}

// extra[p][c] = sum over the sources (ii, jj, m) whose negative is position p of image N-1, in ascending source order
__global__ __launch_bounds__(256) void k14_negscatter_kernel(const float* __restrict__ d_neg, const int* __restrict__ csr_off,
                                                             const int* __restrict__ csr_src, int E, float* __restrict__ extra) {
  const int p = blockIdx.x;
  const int lo = csr_off[p], hi = csr_off[p + 1];
  for (int c = threadIdx.x; c < E; c += 256) {
    float acc = 0.f;
    int i = lo;
    for (; i + 8 <= hi; i += 8) {            // eight rows in flight, added in source order
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = d_neg[(size_t)csr_src[i + k] * E + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k];
    }
    for (; i < hi; ++i) acc += d_neg[(size_t)csr_src[i] * E + c];
    extra[(size_t)p * E + c] = acc;
  }
}

// dctx[n][l][2c] = (dlag - lag*sum_l(dlag*lag)) / max(lnorm, eps),  dlag[n][l] = sum_{p: cols[n][p] == l} d_k[n][p];  odd channels 0
// Block = (image n, 128 channels).  The positions of every word l are listed first, ascending (20 lanes scan the HW column indices
// in LDS), so a channel's thread walks each position ONCE — the first form scanned all HW positions for each of the L words (54 k
// compares per thread, 64 blocks on 256 CUs: 0.79 ms at N = 64, HW = 2704); the sums run in the same order, bit for bit.
constexpr int DL_C = 128;
__global__ __launch_bounds__(DL_C) void k14_dlag_kernel(const float* __restrict__ lag, const float* __restrict__ lnorm,
                                                        const int64_t* __restrict__ cols, const float* __restrict__ d_k,
                                                        int L, int HW, int E, float* __restrict__ dctx) {
  extern __shared__ int dl_s[];             // [HW] column of every position, [HW] positions grouped by word, [CM_MAXL + 1] group starts
  int* col_s = dl_s; int* order = dl_s + HW; int* start = dl_s + 2 * HW;
  const int n = blockIdx.x, tid = threadIdx.x, D = 2 * E;
  for (int p = tid; p < HW; p += DL_C) col_s[p] = (int)cols[(size_t)n * HW + p];
  __syncthreads();
  if (tid < L) {                            // counts, then (after the prefix sum) the ascending position list of word tid
    int cnt = 0;
    for (int p = 0; p < HW; ++p) cnt += col_s[p] == tid;
    start[tid + 1] = cnt;
  }
  __syncthreads();
  if (tid == 0) { start[0] = 0; for (int l = 0; l < L; ++l) start[l + 1] += start[l]; }
  __syncthreads();
  if (tid < L) {
    int w = start[tid];
    for (int p = 0; p < HW; ++p) if (col_s[p] == tid) order[w++] = p;
  }
  __syncthreads();
  const int c = blockIdx.y * DL_C + tid;
  if (c >= E) return;
  float dl[CM_MAXL];
  float dot = 0.f;
  const float* dk = d_k + (size_t)n * HW * E + c;
#pragma unroll
  for (int l = 0; l < CM_MAXL; ++l) {
    dl[l] = 0.f;
    if (l < L) {
      float acc = 0.f;
      int i = start[l];
      const int hi = start[l + 1];
      for (; i + 8 <= hi; i += 8) {          // eight loads in flight, added in list order
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = dk[(size_t)order[i + k] * E];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
      }
      for (; i < hi; ++i) acc += dk[(size_t)order[i] * E];
      dl[l] = acc;
      dot = fmaf(acc, lag[((size_t)n * L + l) * E + c], dot);
    }
  }
  const float inv = 1.f / fmaxf(lnorm[(size_t)n * E + c], 1e-12f);
#pragma unroll
  for (int l = 0; l < CM_MAXL; ++l)
    if (l < L) {
      float* o = dctx + ((size_t)n * L + l) * D + 2 * c;
      o[0] = (dl[l] - lag[((size_t)n * L + l) * E + c] * dot) * inv;
      o[1] = 0.f;
    }
}

}  // namespace

extern "C" int dcn_k9_fwd(const float* cmap, const float* fv, const int64_t* raw_neg, int pairs, int hw, int e, int top_k, int neg_n,
                          int64_t* index, int64_t* neg_idx, float* frame, float* corr, float* negf, void* stream) {
  DCN_CHECK_ARG(cmap && fv && raw_neg && index && neg_idx && frame && corr && negf, "k9_fwd: null pointer");
  DCN_CHECK_ARG(pairs > 0 && hw > 1 && e % 4 == 0 && top_k > 0 && top_k <= SK_MAXK && neg_n >= 0, "k9_fwd: bad shape");
  DCN_CHECK_ARG((long long)hw * hw >= top_k && hw - 1 >= neg_n, "k9_fwd: %d positions cannot supply top-%d / %d negatives", hw, top_k, neg_n);
  hipLaunchKernelGGL(k9_fwd_kernel, dim3(pairs), dim3(SK_T), 0, (hipStream_t)stream, cmap, fv, raw_neg, hw, e, top_k, neg_n,
                     index, neg_idx, frame, corr, negf);
  DCN_CHECK_LAUNCH("k9_fwd");
  return DCN_OK;
}

extern "C" int dcn_k9_bwd(const int64_t* index, const int64_t* neg_idx, const float* d_frame, const float* d_corr, const float* d_neg,
                          int pairs, int hw, int e, int top_k, int neg_n, float* dfv, void* stream) {
  DCN_CHECK_ARG(index && neg_idx && d_frame && d_corr && d_neg && dfv && pairs > 0 && hw > 0 && e > 0, "k9_bwd: bad argument");
  const size_t lds = (size_t)((top_k + top_k * neg_n) * (K9B_GP + 1) + K9B_GP) * sizeof(int);
  DCN_CHECK_ARG(lds <= 64 * 1024, "k9_bwd: %zu bytes of LDS (top_k=%d, neg_n=%d)", lds, top_k, neg_n);
  hipLaunchKernelGGL(k9_bwd_kernel, dim3(cdiv(hw, K9B_GP), 2, pairs), dim3(256), lds, (hipStream_t)stream, index, neg_idx, d_frame, d_corr,
                     d_neg, hw, e, top_k, neg_n, dfv);
  DCN_CHECK_LAUNCH("k9_bwd");
  return DCN_OK;
}

extern "C" int dcn_colnorm_fwd(const float* v, int n, int hw, int e, float* vit, float* cnorm, void* stream) {
  DCN_CHECK_ARG(v && vit && cnorm && n > 0 && hw > 0 && e > 0, "colnorm_fwd: bad argument");
  hipLaunchKernelGGL(colnorm_fwd_kernel, dim3(cdiv(e, 64), n), dim3(256), 0, (hipStream_t)stream, v, hw, e, vit, cnorm);
  DCN_CHECK_LAUNCH("colnorm_fwd");
  return DCN_OK;
}

extern "C" int dcn_colnorm_bwd(const float* vit, const float* cnorm, const float* dq, const float* extra, int n_extra,
                               int n, int hw, int e, float* dv, void* stream) {
  DCN_CHECK_ARG(vit && cnorm && dq && dv && n > 0 && hw > 0 && e > 0, "colnorm_bwd: bad argument");
  hipLaunchKernelGGL(colnorm_bwd_kernel, dim3(cdiv(e, 64), n), dim3(256), 0, (hipStream_t)stream, vit, cnorm, dq, extra, n_extra, hw, e, dv);
  DCN_CHECK_LAUNCH("colnorm_bwd");
  return DCN_OK;
}

extern "C" int dcn_lagnorm_fwd(const float* context, int n, int l, int d, float* lag, float* lnorm, void* stream) {
  DCN_CHECK_ARG(context && lag && lnorm && n > 0 && l > 0 && d > 0 && d % 2 == 0, "lagnorm_fwd: bad argument");
  hipLaunchKernelGGL(lagnorm_fwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, context, l, d, lag, lnorm);
  DCN_CHECK_LAUNCH("lagnorm_fwd");
  return DCN_OK;
}

extern "C" int dcn_crossmap(const float* lag, const float* vit, const float* conv_w, const float* conv_b, int n, int l, int hw, int e,
                            int64_t* cols, float* lvmap, void* stream) {
  DCN_CHECK_ARG(lag && vit && conv_w && conv_b && cols && n > 0 && l > 0 && l <= CM_MAXL && hw > 0 && e % 4 == 0, "crossmap: bad argument");
  const size_t lds = ((size_t)l * (hw + 2) + (size_t)l * l * 3 + l) * sizeof(float);
  DCN_CHECK_ARG(lds <= 64 * 1024, "crossmap: %zu bytes of LDS needed (L=%d, HW=%d)", lds, l, hw);
  hipLaunchKernelGGL(crossmap_kernel, dim3(n), dim3(1024), lds, (hipStream_t)stream, lag, vit, conv_w, conv_b, l, hw, e, cols, lvmap);
  DCN_CHECK_LAUNCH("crossmap");
  return DCN_OK;
}

extern "C" int dcn_k14_gather(const float* lag, const float* vit, const int64_t* cols, const int64_t* neg, int n, int l, int hw, int e,
                              int neg_n, float* lag_pos, float* neg_cross, void* stream) {
  DCN_CHECK_ARG(lag && vit && cols && neg && lag_pos && neg_cross && n > 0 && l > 0 && hw > 0 && e % 4 == 0, "k14_gather: bad argument");
  hipLaunchKernelGGL(k14_gather_kernel, dim3(cdiv((int64_t)n * hw, 4)), dim3(256), 0, (hipStream_t)stream, lag, vit, cols, neg, n, l, hw, e,
                     neg_n, lag_pos, neg_cross);
  DCN_CHECK_LAUNCH("k14_gather");
  return DCN_OK;
}

extern "C" int dcn_k14_negscatter(const float* d_neg, const int* csr_off, const int* csr_src, int hw, int e, float* extra, void* stream) {
  DCN_CHECK_ARG(d_neg && csr_off && csr_src && extra && hw > 0 && e > 0, "k14_negscatter: bad argument");
  hipLaunchKernelGGL(k14_negscatter_kernel, dim3(hw), dim3(256), 0, (hipStream_t)stream, d_neg, csr_off, csr_src, e, extra);
  DCN_CHECK_LAUNCH("k14_negscatter");
  return DCN_OK;
}

extern "C" int dcn_k14_dlag(const float* lag, const float* lnorm, const int64_t* cols, const float* d_k, int n, int l, int hw, int e,
                            float* dcontext, void* stream) {
  DCN_CHECK_ARG(lag && lnorm && cols && d_k && dcontext && n > 0 && l > 0 && l <= CM_MAXL && hw > 0 && e > 0, "k14_dlag: bad argument");
  hipLaunchKernelGGL(k14_dlag_kernel, dim3(n, cdiv(e, DL_C)), dim3(DL_C), (size_t)(2 * hw + CM_MAXL + 1) * sizeof(int), (hipStream_t)stream,
                     lag, lnorm, cols, d_k, l, hw, e, dcontext);
  DCN_CHECK_LAUNCH("k14_dlag");
  return DCN_OK;
}

// ---- device-side negative sampling (round 6; SURVEY H3 option (ii), opt-in: grounding_model.sampler = "device") --------------------------
// The reference draws the negatives of both heads with Python's random.sample (model/DCNet_model.py:62-96, 394-420); the default path
// advances that very MT19937 stream on the host (sampling.cpp, bit-exact).  The K14 loop makes N*N*HW0 sample() calls of which N*HW0
// are kept: 11 M calls at 256 images, 0.19 s of one core — as long as the whole GPU step of configs[4].  This sampler draws ONLY what
// is kept, on the device, inside the captured step: same distribution (k distinct positions, uniform over the population with the
// excluded position removed), same exclusion rules, NOT the same numbers — a counter-based generator (Philox-4x32-10: Salmon et al.,
// SC'11) keyed by (seed, step, sample index), rejection exactly as CPython's randbelow (top bit_length(n) bits, redraw while >= n) and
// as random.sample's set branch (redraw duplicates).  state[0] = seed, state[1] = step counter (advanced by the last kernel of the
// sampler, so a replayed graph draws fresh negatives every step without the host).
namespace {

__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
  const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
  c[0] = h1 ^ c[1] ^ k0; c[1] = l1; c[2] = h0 ^ c[3] ^ k1; c[3] = l0;
}
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
}

// k distinct values of [0, pop), uniform (k <= 16); stream = (seed, step, sample id)
__device__ __forceinline__ void dsample_distinct(int pop, int k, unsigned long long seed, unsigned long long step, unsigned sample_id,
                                                 unsigned* sel) {
  const int bits = 32 - __clz((unsigned)pop);               // pop.bit_length()
  int have = 0;
  for (unsigned block = 0; have < k; ++block) {
    unsigned c[4] = {sample_id, block, (unsigned)step, (unsigned)(step >> 32)};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned r = c[w] >> (32 - bits);
      bool ok = r < (unsigned)pop && have < k;
      for (int e = 0; e < have; ++e) ok &= sel[e] != r;
      if (ok) sel[have++] = r;
    }
  }
}

// thread t < n9: the neg_n raw positions of K9 sample t (population hw - 1: the caller's kernel maps pos >= kp -> pos + 1);
// else K14 sample (ii, jj): neg_c positions of [0, hw), without jj when ii == n - 1 (the image the negatives are gathered from)
__global__ __launch_bounds__(256) void dsample_draw_kernel(const unsigned long long* __restrict__ state, int n9, int neg_n, int n, int hw,
                                                           int neg_c, int64_t* __restrict__ k9, int64_t* __restrict__ k14) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const unsigned long long seed = state[0], step = state[1];
  unsigned sel[16];
  if (t < n9) {
    dsample_distinct(hw - 1, neg_n, seed, step, (unsigned)t, sel);
    for (int e = 0; e < neg_n; ++e) k9[(size_t)t * neg_n + e] = sel[e];
    return;
  }
  const int u = t - n9;
  if (u >= n * hw) return;
  const int ii = u / hw, jj = u - ii * hw;
  const bool removed = ii == n - 1;
  dsample_distinct(removed ? hw - 1 : hw, neg_c, seed, step, 0x80000000u + (unsigned)u, sel);
  for (int e = 0; e < neg_c; ++e) k14[(size_t)u * neg_c + e] = (removed && sel[e] >= (unsigned)jj) ? sel[e] + 1 : sel[e];
}

// counting sort of the K14 table by position, as dcn_mt_sample_crossmodal_csr does on the host: block p counts the entries that drew p ...
__global__ __launch_bounds__(256) void dsample_count_kernel(const int64_t* __restrict__ k14, int total, int* __restrict__ cnt) {
  __shared__ int red[4];
  const int p = blockIdx.x;
  int c = 0;
  for (int i = threadIdx.x; i < total; i += 256) c += k14[i] == p;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) cnt[p] = red[0] + red[1] + red[2] + red[3];
}
// ... and lists them in ascending order behind the entries of the positions in front of it; block 0 advances the step counter
__global__ __launch_bounds__(256) void dsample_csr_kernel(const int64_t* __restrict__ k14, int total, int rows, const int* __restrict__ cnt,
                                                          int* __restrict__ csr_off, int* __restrict__ csr_src,
                                                          unsigned long long* __restrict__ state) {
  __shared__ int red[4]; __shared__ int s_base;
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int c = 0;
  for (int q = tid; q < p; q += 256) c += cnt[q];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) red[wave] = c;
  __syncthreads();
  if (tid == 0) {
    s_base = red[0] + red[1] + red[2] + red[3];
    csr_off[p] = s_base;
    if (p == rows - 1) csr_off[rows] = s_base + cnt[p];
    if (p == 0) state[1] += 1;
  }
  __syncthreads();
  int base = s_base;
  for (int i0 = 0; i0 < total; i0 += 256) {
    const int i = i0 + tid;
    const bool hit = i < total && k14[i] == p;
    const unsigned long long m = __ballot(hit);
    __syncthreads();                                   // (red of the previous chunk has been read)
    if (lane == 0) red[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += red[w];
    if (hit) csr_src[off + __popcll(m & ((1ull << lane) - 1ull))] = i;
    base += red[0] + red[1] + red[2] + red[3];
  }
}

}  // namespace

extern "C" int64_t dcn_device_sample_ws(int hw) { return hw; }      // ints: the per-position counts

// k9 [pairs][top_k][neg_n] raw positions, k14 [n][hw][neg_c], csr_off [hw + 1], csr_src [n*hw*neg_c]: the tensors dcn_mt_sample_* fill on
// the host, filled on the device from state = {seed, step}; ws: dcn_device_sample_ws(hw) ints.  Replaces the random.sample loops of
// model/DCNet_model.py:62-96 (keeping only the draws that are used) and :394-420 with the same distribution, not the same stream.
extern "C" int dcn_device_sample(uint64_t* state, int n, int top_k, int hw, int neg_n, int neg_c, int64_t* k9, int64_t* k14,
                                 int32_t* csr_off, int32_t* csr_src, int32_t* ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DCN_CHECK_ARG(state && k9 && k14 && csr_off && csr_src && ws, "device_sample: null pointer");
  DCN_CHECK_ARG(n >= 2 && n % 2 == 0 && top_k > 0 && hw > 1 && neg_n > 0 && neg_c > 0 && neg_n <= 16 && neg_c <= 16 && neg_n <= hw - 1 &&
                neg_c <= hw - 1, "device_sample: bad argument (n=%d hw=%d neg_n=%d neg_c=%d)", n, hw, neg_n, neg_c);
  const int64_t total = (int64_t)n * hw * neg_c;
  DCN_CHECK_ARG(total < (1LL << 31) && (int64_t)n * hw + (int64_t)(n / 2) * top_k < (1LL << 31), "device_sample: table too large");
  const int n9 = (n / 2) * top_k;
  hipLaunchKernelGGL(dsample_draw_kernel, dim3(cdiv((int64_t)n9 + (int64_t)n * hw, 256)), dim3(256), 0, stream,
                     (const unsigned long long*)state, n9, neg_n, n, hw, neg_c, k9, k14);
  hipLaunchKernelGGL(dsample_count_kernel, dim3(hw), dim3(256), 0, stream, k14, (int)total, ws);
  hipLaunchKernelGGL(dsample_csr_kernel, dim3(hw), dim3(256), 0, stream, k14, (int)total, hw, ws, csr_off, csr_src, (unsigned long long*)state);
  DCN_CHECK_LAUNCH("device_sample");
  return DCN_OK;
}
