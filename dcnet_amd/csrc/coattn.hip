// Inter-frame co-attention (model/DCNet_model.py:449-459, model/test_DCNet_model.py:259-274).
//
//   A[i,j]      = <f1_i, f2_j>                      (unit-norm features => A in [-1,1])
//   f1_attn[i]  = sum_j softmax_j(t*A[i,j]) f2_j
//   f2_attn[j]  = sum_i softmax_i(t*A[i,j]) f1_i
//
// MI355X design note.  fp32 MFMA peak is 157 TFLOP/s but HBM streams ~6 TB/s, i.e. the machine
// balance is only ~25 FLOP/B for this dtype.  A flash-style kernel would recompute A for each
// softmax direction (4 GEMM units forward, 14 in the backward).  Materialising E = exp(t*A - t)
// ONCE in HBM (0.94 GB for 32 pairs at 52x52 — 288 GB is there to be used) needs 3 GEMM units
// forward and 6 backward, at the price of ~6 streaming passes over E (<= 20 % of the GEMM time at
// HW = 2704, less at the coarser scales).  Because |A| <= 1 the softmax needs no running max:
// exp(t*A - t) is in [e^-2t, 1], so row sums and column sums of the same E give both directions.
// All GEMMs run on the shared engines: NT and NN forms on igemm.hip, TN on wgrad.hip.
#include "igemm.h"
#include "prof.h"

int tn_gemm_batched(const float* A, int lda, long long a_bs, const float* B, int ldb, long long b_bs,
                    float* C, int ldc, long long c_bs, const float* row_scale,
                    int M, int m_ld, int N, int K, int batch, int accumulate, hipStream_t stream,
                    const unsigned* amax_a = nullptr, const unsigned* amax_b = nullptr);

namespace {

constexpr int EXP_ROWS = 32;

// E = exp(t*A - t) in place (pad columns -> 0), row sums, per-row-block column partial sums.
// grid (ceil(hw/32), b); each thread owns FOUR consecutive columns (16-B accesses: with one float per lane this pass ran at
// 0.34 of the HBM rate) 4*tid, 4*tid + 1024, ... and walks the 32 rows, all 32 loads of a column group in flight together.
// SPLIT: E is written in the f16 two-piece split form the products on gemm3.hip read ([8 h | 8 l] per 8 columns, scale 2^13 for values
// <= 1): a thread's four columns are half of such a run — 8 bytes of the h piece, 8 of the l piece; the sums are those of the fp32 values.
typedef _Float16 f16x4c_t __attribute__((ext_vector_type(4)));
constexpr float E_SPLIT_SCALE = 8192.f;           // = pow2_scale(1.0): brings 1.0 below 2^14
template <bool SPLIT>
__global__ __launch_bounds__(256) void exp_sums_kernel(float* __restrict__ E, int hw, int ldE, float t,
                                                       float* __restrict__ rsum, float* __restrict__ colpart) {
  __shared__ float red[4][EXP_ROWS];
  const int b = blockIdx.y, rb = blockIdx.x;
  const int i0 = rb * EXP_ROWS;
  float* e = E + ((size_t)b * hw + i0) * ldE;
  float racc[EXP_ROWS];
#pragma unroll
  for (int r = 0; r < EXP_ROWS; ++r) racc[r] = 0.f;
  const int nrow = min(EXP_ROWS, hw - i0);
  for (int j = threadIdx.x * 4; j < ldE; j += 1024) {
    f32x4 v[EXP_ROWS];
#pragma unroll
    for (int r = 0; r < EXP_ROWS; ++r)
      v[r] = r < nrow ? *reinterpret_cast<const f32x4*>(e + (size_t)r * ldE + j) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < EXP_ROWS; ++r) {
      if (r < nrow) {
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (j + k < hw) ? expf(t * v[r][k] - t) : 0.f;
        if constexpr (SPLIT) {
          const f32x4 ts = o * E_SPLIT_SCALE;
          const f16x4c_t h = {(_Float16)ts[0], (_Float16)ts[1], (_Float16)ts[2], (_Float16)ts[3]};
          const f16x4c_t l = {(_Float16)(ts[0] - (float)h[0]), (_Float16)(ts[1] - (float)h[1]), (_Float16)(ts[2] - (float)h[2]),
                              (_Float16)(ts[3] - (float)h[3])};
          unsigned char* run = reinterpret_cast<unsigned char*>(e + (size_t)r * ldE + (j & ~7)) + (j & 4) * 2;
          *reinterpret_cast<f16x4c_t*>(run) = h;
          *reinterpret_cast<f16x4c_t*>(run + 16) = l;
        } else {
          *reinterpret_cast<f32x4*>(e + (size_t)r * ldE + j) = o;
        }
        cs += o; racc[r] += (o[0] + o[1]) + (o[2] + o[3]);
      }
    }
    *reinterpret_cast<f32x4*>(colpart + ((size_t)rb * gridDim.y + b) * ldE + j) = cs;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < EXP_ROWS; ++r) {
    const float s = wave_sum(racc[r]);
    if (lane == 0) red[wave][r] = s;
  }
  __syncthreads();
  if (threadIdx.x < EXP_ROWS && i0 + threadIdx.x < hw)
    rsum[(size_t)b * hw + i0 + threadIdx.x] = 1.f / (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// cinv[b][j] = 1 / sum_rb colpart[rb][b][j]
__global__ __launch_bounds__(256) void colsum_inv_kernel(const float* __restrict__ colpart, int nrb, int b, int hw, int ldE,
                                                         float* __restrict__ cinv) {
  const int j = blockIdx.x * 256 + threadIdx.x, bb = blockIdx.y;
  if (j >= hw) return;
  float s = 0.f;
  for (int rb = 0; rb < nrb; ++rb) s += colpart[((size_t)rb * b + bb) * ldE + j];
  cinv[(size_t)bb * hw + j] = 1.f / s;
}

// Abs-max words of the co-attention GEMM operands (DCN_AMAX_WORDS each, in the caller's workspace): slot 0 holds the
// constant 1 — unit-norm features, E = exp(t*A - t) <= 1 — the others start at 0 and are raised by the kernels that write
// or read the gradient operands.  With both words present a GEMM runs on the f16 two-piece split (igemm.hip).
enum { AM_ONE = 0, AM_DO1, AM_DO2, AM_DO1S, AM_DO2S, AM_DA, AM_DP1, AM_DP2, AM_SLOTS };
__global__ __launch_bounds__(256) void amax_init_kernel(unsigned* __restrict__ w) {
  for (int i = threadIdx.x; i < AM_SLOTS * DCN_AMAX_WORDS; i += 256) w[i] = i < DCN_AMAX_WORDS ? 0x3F800000u : 0u;
}

// one wave per row: delta[row] = <d[row], o[row]>, ds[row] = d[row] * inv[row]; abs-max of d and ds
__global__ __launch_bounds__(256) void rowdot_scale_kernel(const float* __restrict__ d, int ldd, int64_t bsd,
                                                           const float* __restrict__ o, int ldo, int64_t bso, int hw,
                                                           const float* __restrict__ inv, int64_t rows, int c,
                                                           float* __restrict__ delta, float* __restrict__ ds,
                                                           unsigned* __restrict__ amax_d, unsigned* __restrict__ amax_ds) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  float vmax = 0.f, s = 0.f;
  if (row < rows) {
    const int64_t bb = row / hw, i = row - bb * hw;
    s = inv[row];
    float acc = 0.f;
    for (int k = lane * 4; k < c; k += 256) {
      const f32x4 dv = *reinterpret_cast<const f32x4*>(d + bb * bsd + i * ldd + k);
      const f32x4 ov = *reinterpret_cast<const f32x4*>(o + bb * bso + i * ldo + k);
      acc += dv[0] * ov[0] + dv[1] * ov[1] + dv[2] * ov[2] + dv[3] * ov[3];
      *reinterpret_cast<f32x4*>(ds + row * c + k) = dv * s;
      vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(dv[0]), fabsf(dv[1]))), fmaxf(fabsf(dv[2]), fabsf(dv[3])));
    }
    acc = wave_sum(acc);
    if (lane == 0) delta[row] = acc;
  }
  // (s > 0: one max serves both words; the scaled tensor's maximum is bounded by max|d| * max s of the block's rows)
  const float vs = wave_max(vmax * s);
  amax_update_block(amax_d, vmax, red);
  __syncthreads();
  amax_update_block(amax_ds, vs, red);
}

// dA = t * E * ((dP1 - d1[i]) * rinv[i] + (dP2 - d2[j]) * cinv[j]), written over dP1; pad columns -> 0
// OSPLIT: dA is written in the split form as well, scaled by the word in amax_da, which then holds a BOUND (da_bound_kernel) instead
// of receiving the maximum.
template <bool ESPLIT, bool OSPLIT = false>
__global__ __launch_bounds__(256) void dA_kernel(const float* __restrict__ E, float* __restrict__ dP1, const float* __restrict__ dP2,
                                                 const float* __restrict__ rinv, const float* __restrict__ cinv,
                                                 const float* __restrict__ d1, const float* __restrict__ d2,
                                                 int hw, int ldE, float t, int64_t total4, unsigned* __restrict__ amax_da) {
  __shared__ float red[4];
  float vmax = 0.f;
  const int l4 = ldE >> 2;
  float s_out = 1.f;
  if constexpr (OSPLIT) {                         // (= igemm.hip pow2_scale of the bound)
    const int be = (int)((amax_read(amax_da) >> 23) & 0xFF);
    int ex = (be == 0 || be == 255) ? 0 : 14 - (be - 126);
    ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
    s_out = __uint_as_float((unsigned)(ex + 127) << 23);
  }
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (int64_t)gridDim.x * 256) {
    const int64_t row = idx / l4;                 // = b*hw + i
    const int j = (int)(idx - row * l4) * 4;
    const int64_t bb = row / hw;
    f32x4 e;
    if constexpr (ESPLIT) {                       // h + l of the split form: E to 22 bits
      const unsigned char* run = reinterpret_cast<const unsigned char*>(E + row * ldE + (j & ~7)) + (j & 4) * 2;
      const f16x4c_t h = *reinterpret_cast<const f16x4c_t*>(run), l = *reinterpret_cast<const f16x4c_t*>(run + 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) e[k] = ((float)h[k] + (float)l[k]) * (1.f / E_SPLIT_SCALE);
    } else {
      e = *reinterpret_cast<const f32x4*>(E + row * ldE + j);
    }
    const f32x4 p1 = *reinterpret_cast<const f32x4*>(dP1 + row * ldE + j);
    const f32x4 p2 = *reinterpret_cast<const f32x4*>(dP2 + row * ldE + j);
    const float ri = rinv[row], di = d1[row];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v = 0.f;
      if (j + k < hw) v = t * e[k] * ((p1[k] - di) * ri + (p2[k] - d2[bb * hw + j + k]) * cinv[bb * hw + j + k]);
      o[k] = v;
    }
    if constexpr (OSPLIT) {                       // (a lane pair shares a run of 8 columns: both have read their 16 bytes by now)
      const f32x4 ts = o * s_out;
      const f16x4c_t h = {(_Float16)ts[0], (_Float16)ts[1], (_Float16)ts[2], (_Float16)ts[3]};
      const f16x4c_t l = {(_Float16)(ts[0] - (float)h[0]), (_Float16)(ts[1] - (float)h[1]), (_Float16)(ts[2] - (float)h[2]),
                          (_Float16)(ts[3] - (float)h[3])};
      unsigned char* run = reinterpret_cast<unsigned char*>(dP1 + row * ldE + (j & ~7)) + (j & 4) * 2;
      *reinterpret_cast<f16x4c_t*>(run) = h;
      *reinterpret_cast<f16x4c_t*>(run + 16) = l;
    } else {
      *reinterpret_cast<f32x4*>(dP1 + row * ldE + j) = o;
      vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
    }
  }
  if constexpr (!OSPLIT) amax_update_block(amax_da, vmax, red);
}

// |dA_ij| = t E_ij |(dP1_ij - d1_i) r_i + (dP2_ij - d2_j) c_j| <= 2 t (max|dP1| + max|dP2|): E_ij r_i <= 1 and E_ij c_j <= 1 (an entry is
// one term of its row / column sum) and d1_i, d2_j are softmax-weighted means of dP1's row / dP2's column.  The f16 split needs a bound,
// not the maximum (a loose one costs the smallest entries their lowest bits: absolute error <= 2^-39 of the bound), so dA_kernel can
// write the split form directly instead of fp32 + a split pass over 0.94 GB.
__global__ __launch_bounds__(64) void da_bound_kernel(const unsigned* __restrict__ a1, const unsigned* __restrict__ a2, float t,
                                                      unsigned* __restrict__ out) {
  const float b = 2.f * t * (__uint_as_float(amax_read(a1)) + __uint_as_float(amax_read(a2)));
  out[threadIdx.x & (DCN_AMAX_WORDS - 1)] = __float_as_uint(b);
}

inline int ld_pad(int hw) { return (hw + 31) / 32 * 32; }

void gemm_params(IgemmParams& p, const float* A, int lda, long long a_bs, const float* B, int ldb, long long b_bs,
                 float* C, int ldc, long long c_bs, int M, int N, int K, int batch) {
  p = IgemmParams{};
  p.osy = p.osx = p.isy = p.isx = 1; p.dense_out = 1;
  p.in = A; p.ldi = lda; p.in_bs = a_bs; p.wt = B; p.ldw = ldb; p.wt_bs = b_bs; p.out = C; p.ldo = ldc; p.out_bs = c_bs;
  p.ldr = ldc;
  p.N = 1; p.Hi = 1; p.Wi = M; p.Ho = 1; p.Wo = M; p.Hs = 1; p.Ws = M; p.M = M;
  p.Ci = K; p.Co = N; p.ntaps = 1; p.batch = batch;
}

}  // namespace

// the products of a (b, hw, c) problem run on gemm3.hip (pre-split operands)?  A function of the shape, the precision mode and the
// "Gemm3" knob only: dcn_coattn_bwd must see what dcn_coattn_fwd saw (E is saved in the form the forward wrote).
static bool on_gemm3(int b, int hw, int c) { return c % 32 == 0 && gemm3_applicable(hw, c, hw, b) && gemm3_applicable(hw, hw, c, b); }

extern "C" int64_t dcn_coattn_fwd_ws(int b, int hw, int c) {
  (void)c;
  return (int64_t)cdiv(hw, EXP_ROWS) * b * ld_pad(hw) + AM_SLOTS * DCN_AMAX_WORDS;
}
extern "C" int64_t dcn_coattn_e_size(int b, int hw) { return (int64_t)b * hw * ld_pad(hw); }
// what the forward keeps for the backward in the caller's E buffer: E, and — on gemm3.hip — the split forms of f1 and f2 behind it (the
// backward multiplies by them four more times; it used to split them again: two passes over both tensors per scale)
extern "C" int64_t dcn_coattn_saved_size(int b, int hw, int c) {
  return dcn_coattn_e_size(b, hw) + (on_gemm3(b, hw, c) ? (int64_t)2 * b * hw * c : 0);
}

extern "C" int dcn_coattn_fwd(const float* f1, const float* f2, int ldf, int64_t bsf, float* f1_attn, float* f2_attn, int ldo,
                              int64_t bso, float* E, float* rinv, float* cinv, float* ws,
                              int b, int hw, int c, float temperature, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DCN_CHECK_ARG(f1 && f2 && f1_attn && E && rinv && cinv && ws, "coattn_fwd: null pointer");
  DCN_CHECK_ARG(b > 0 && hw > 0 && c > 0 && c % 32 == 0, "coattn_fwd: bad shape (c=%d must be a multiple of 32)", c);
  if (ldf <= 0) ldf = c;
  if (ldo <= 0) ldo = c;
  DCN_CHECK_ARG(ldf % 4 == 0 && ldo % 4 == 0, "coattn_fwd: ldf/ldo must be multiples of 4");
  if (bsf <= 0) bsf = (int64_t)hw * ldf;
  if (bso <= 0) bso = (int64_t)hw * ldo;
  DCN_CHECK_ARG(bsf % 4 == 0 && bso % 4 == 0, "coattn_fwd: batch strides must be multiples of 4");
  const int ldE = ld_pad(hw);
  unsigned* am = reinterpret_cast<unsigned*>(ws + (int64_t)cdiv(hw, EXP_ROWS) * b * ldE);
  hipLaunchKernelGGL(amax_init_kernel, dim3(1), dim3(256), 0, stream, am);
  DCN_CHECK_LAUNCH("coattn amax_init");
  const unsigned* one = am + AM_ONE * DCN_AMAX_WORDS;
  const int nrb = cdiv(hw, EXP_ROWS);
  const long long bsE = (long long)hw * ldE;
  if (on_gemm3(b, hw, c)) {
    // every product on gemm3.hip: f1, f2 split once (unit norm: the constant word), E written in split form by its own pass
    float* f1s = E + dcn_coattn_e_size(b, hw);               // (kept for the backward: dcn_coattn_saved_size)
    float* f2s = f1s + (int64_t)b * hw * c;
    const long long bss = (long long)hw * c;
    int rc = gemm3_presplit(f1, ldf, bsf, f1s, c, bss, b, hw, c, one, stream);
    if (rc) return rc;
    if ((rc = gemm3_presplit(f2, ldf, bsf, f2s, c, bss, b, hw, c, one, stream))) return rc;
    // 1. A = f1 . f2^T -> E                                              (NT)
    if ((rc = gemm3_launch(f1s, c, bss, 0, f2s, c, bss, 0, E, ldE, bsE, nullptr, 0, hw, hw, c, b, 0, one, one, stream))) return rc;
    // 2. E = exp(t*A - t) in split form, rinv, cinv
    const int pid = prof_begin(12, (double)b * hw * ldE * 8.0, stream);
    hipLaunchKernelGGL(exp_sums_kernel<true>, dim3(nrb, b), dim3(256), 0, stream, E, hw, ldE, temperature, rinv, ws);
    prof_end(pid, stream);
    DCN_CHECK_LAUNCH("exp_sums");
    hipLaunchKernelGGL(colsum_inv_kernel, dim3(cdiv(hw, 256), b), dim3(256), 0, stream, ws, nrb, b, hw, ldE, cinv);
    DCN_CHECK_LAUNCH("colsum_inv");
    // 3. f1_attn = diag(rinv) E f2                                       (NN, K = keys j)
    if ((rc = gemm3_launch(E, ldE, bsE, 0, f2s, c, bss, 1, f1_attn, ldo, bso, rinv, hw, hw, c, hw, b, 0, one, one, stream))) return rc;
    // 4. f2_attn = diag(cinv) E^T f1                                     (TN, K = queries i)
    if (f2_attn) rc = gemm3_launch(E, ldE, bsE, 1, f1s, c, bss, 1, f2_attn, ldo, bso, cinv, hw, hw, c, hw, b, 0, one, one, stream);
    return rc;
  }
  IgemmParams p;
  // 1. A = f1 . f2^T  -> E                                             (NT)
  gemm_params(p, f1, ldf, bsf, f2, ldf, bsf, E, ldE, (long long)hw * ldE, hw, hw, c, b);
  p.amax_a = one; p.amax_b = one;
  int rc = igemm_launch(p, stream);
  if (rc) return rc;
  // 2. E = exp(t*A - t), rinv = 1/rowsum, cinv = 1/colsum
  const int pid = prof_begin(12, (double)b * hw * ldE * 8.0, stream);
  hipLaunchKernelGGL(exp_sums_kernel<false>, dim3(nrb, b), dim3(256), 0, stream, E, hw, ldE, temperature, rinv, ws);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("exp_sums");
  hipLaunchKernelGGL(colsum_inv_kernel, dim3(cdiv(hw, 256), b), dim3(256), 0, stream, ws, nrb, b, hw, ldE, cinv);
  DCN_CHECK_LAUNCH("colsum_inv");
  // 3. f1_attn = diag(rinv) E f2                                       (NN, K = keys j)
  gemm_params(p, E, ldE, (long long)hw * ldE, f2, ldf, bsf, f1_attn, ldo, bso, hw, c, ldE, b);
  p.bmode = 1; p.kvalid = hw; p.row_scale = rinv; p.amax_a = one; p.amax_b = one;
  rc = igemm_launch(p, stream);
  if (rc) return rc;
  // 4. f2_attn = diag(cinv) E^T f1                                     (TN, K = queries i)
  if (f2_attn)
    rc = tn_gemm_batched(E, ldE, (long long)hw * ldE, f1, ldf, bsf, f2_attn, ldo, bso,
                         cinv, hw, ldE, c, hw, b, 0, stream, one, one);
  return rc;
}

extern "C" int64_t dcn_coattn_bwd_ws(int b, int hw, int c) {
  return (int64_t)2 * b * hw * ld_pad(hw) + (int64_t)2 * b * hw * c + (int64_t)2 * b * hw + AM_SLOTS * DCN_AMAX_WORDS +
         (on_gemm3(b, hw, c) ? (int64_t)2 * b * hw * c + 4 : 0);
}

extern "C" int dcn_coattn_bwd(const float* f1, const float* f2, int ldf, int64_t bsf,
                              const float* d_f1_attn, const float* d_f2_attn, int lddo, int64_t bsdo,
                              const float* f1_attn, const float* f2_attn, int ldo, int64_t bso,
                              const float* E, const float* rinv, const float* cinv,
                              float* d_f1, float* d_f2, int lddf, int64_t bsdf, int accumulate, float* ws,
                              int b, int hw, int c, float temperature, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DCN_CHECK_ARG(f1 && f2 && d_f1_attn && d_f2_attn && f1_attn && f2_attn && E && rinv && cinv && d_f1 && d_f2 && ws,
                "coattn_bwd: null pointer");
  DCN_CHECK_ARG(b > 0 && hw > 0 && c > 0 && c % 32 == 0, "coattn_bwd: bad shape");
  if (ldf <= 0) ldf = c;
  if (ldo <= 0) ldo = c;
  if (lddo <= 0) lddo = c;
  if (lddf <= 0) lddf = c;
  if (bsf <= 0) bsf = (int64_t)hw * ldf;
  if (bso <= 0) bso = (int64_t)hw * ldo;
  if (bsdo <= 0) bsdo = (int64_t)hw * lddo;
  if (bsdf <= 0) bsdf = (int64_t)hw * lddf;
  const int ldE = ld_pad(hw);
  const long long rows = (long long)b * hw;
  float* dP1 = ws;
  float* dP2 = dP1 + rows * ldE;
  float* dO1s = dP2 + rows * ldE;
  float* dO2s = dO1s + rows * c;
  float* del1 = dO2s + rows * c;
  float* del2 = del1 + rows;
  unsigned* am = reinterpret_cast<unsigned*>(del2 + rows);
  hipLaunchKernelGGL(amax_init_kernel, dim3(1), dim3(256), 0, stream, am);
  DCN_CHECK_LAUNCH("coattn amax_init");
  auto slot = [&](int i) { return am + i * DCN_AMAX_WORDS; };
  // 1. delta and pre-scaled upstream gradients
  hipLaunchKernelGGL(rowdot_scale_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, d_f1_attn, lddo, bsdo, f1_attn, ldo, bso, hw, rinv, rows, c, del1, dO1s,
                     slot(AM_DO1), slot(AM_DO1S));
  DCN_CHECK_LAUNCH("rowdot_scale");
  hipLaunchKernelGGL(rowdot_scale_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, stream, d_f2_attn, lddo, bsdo, f2_attn, ldo, bso, hw, cinv, rows, c, del2, dO2s,
                     slot(AM_DO2), slot(AM_DO2S));
  DCN_CHECK_LAUNCH("rowdot_scale");
  const int64_t total4 = rows * (ldE / 4);
  int64_t g = (total4 + 255) / 256; if (g > 8192) g = 8192;
  if (g > 1024) g = 1024;      // (one abs-max atomic per workgroup)
  const long long bsE = (long long)hw * ldE;
  if (on_gemm3(b, hw, c)) {
    // every product on gemm3.hip (E arrives in split form from dcn_coattn_fwd): the six narrow operands are split once each, dA in place
    float* dO1p = reinterpret_cast<float*>(((uintptr_t)(am + AM_SLOTS * DCN_AMAX_WORDS) + 15) & ~(uintptr_t)15);      // (b * hw may be odd)
    float* dO2p = dO1p + rows * c;
    const float* f1s = E + dcn_coattn_e_size(b, hw);         // the forward's split forms of f1, f2 (unit norm: the constant abs-max word)
    const float* f2s = f1s + rows * c;
    const long long bss = (long long)hw * c;
    int rc;
    if ((rc = gemm3_presplit(d_f1_attn, lddo, bsdo, dO1p, c, bss, b, hw, c, slot(AM_DO1), stream))) return rc;
    if ((rc = gemm3_presplit(d_f2_attn, lddo, bsdo, dO2p, c, bss, b, hw, c, slot(AM_DO2), stream))) return rc;
    if ((rc = gemm3_presplit(dO1s, c, bss, dO1s, c, bss, b, hw, c, slot(AM_DO1S), stream))) return rc;
    if ((rc = gemm3_presplit(dO2s, c, bss, dO2s, c, bss, b, hw, c, slot(AM_DO2S), stream))) return rc;
    // 2. dP1[i,j] = <dO1_i, f2_j>,  dP2[i,j] = <f1_i, dO2_j>                (NT x2)
    if ((rc = gemm3_launch(dO1p, c, bss, 0, f2s, c, bss, 0, dP1, ldE, bsE, nullptr, 0, hw, hw, c, b, 0, slot(AM_DO1), slot(AM_ONE), stream, slot(AM_DP1)))) return rc;
    if ((rc = gemm3_launch(f1s, c, bss, 0, dO2p, c, bss, 0, dP2, ldE, bsE, nullptr, 0, hw, hw, c, b, 0, slot(AM_ONE), slot(AM_DO2), stream, slot(AM_DP2)))) return rc;
    // 3. dA (over dP1; pads -> 0) in split form straight away, scaled by a bound from the two maxima the products above left
    hipLaunchKernelGGL(da_bound_kernel, dim3(1), dim3(64), 0, stream, slot(AM_DP1), slot(AM_DP2), temperature, slot(AM_DA));
    DCN_CHECK_LAUNCH("dA bound");
    const int pid_da = prof_begin(31, (double)total4 * 16.0 * 4.0, stream);
    hipLaunchKernelGGL((dA_kernel<true, true>), dim3((int)g), dim3(256), 0, stream, E, dP1, dP2, rinv, cinv, del1, del2, hw, ldE, temperature, total4, slot(AM_DA));
    prof_end(pid_da, stream);
    DCN_CHECK_LAUNCH("dA");
    // 4. d_f1 (+)= dA f2 + E (dO2 / colsum)                                 (NN x2)
    if ((rc = gemm3_launch(dP1, ldE, bsE, 0, f2s, c, bss, 1, d_f1, lddf, bsdf, nullptr, 0, hw, c, hw, b, accumulate, slot(AM_DA), slot(AM_ONE), stream))) return rc;
    if ((rc = gemm3_launch(E, ldE, bsE, 0, dO2s, c, bss, 1, d_f1, lddf, bsdf, nullptr, 0, hw, c, hw, b, 1, slot(AM_ONE), slot(AM_DO2S), stream))) return rc;
    // 5. d_f2 (+)= dA^T f1 + E^T (dO1 / rowsum)                             (TN x2)
    if ((rc = gemm3_launch(dP1, ldE, bsE, 1, f1s, c, bss, 1, d_f2, lddf, bsdf, nullptr, 0, hw, c, hw, b, accumulate, slot(AM_DA), slot(AM_ONE), stream))) return rc;
    return gemm3_launch(E, ldE, bsE, 1, dO1s, c, bss, 1, d_f2, lddf, bsdf, nullptr, 0, hw, c, hw, b, 1, slot(AM_ONE), slot(AM_DO1S), stream);
  }
  IgemmParams p;
  int rc;
  // 2. dP1[i,j] = <dO1_i, f2_j>,  dP2[i,j] = <f1_i, dO2_j>                  (NT x2)
  gemm_params(p, d_f1_attn, lddo, bsdo, f2, ldf, bsf, dP1, ldE, (long long)hw * ldE, hw, hw, c, b);
  p.amax_a = slot(AM_DO1); p.amax_b = slot(AM_ONE);
  if ((rc = igemm_launch(p, stream))) return rc;
  gemm_params(p, f1, ldf, bsf, d_f2_attn, lddo, bsdo, dP2, ldE, (long long)hw * ldE, hw, hw, c, b);
  p.amax_a = slot(AM_ONE); p.amax_b = slot(AM_DO2);
  if ((rc = igemm_launch(p, stream))) return rc;
  // 3. dA (over dP1)
  const int pid_da = prof_begin(31, (double)total4 * 16.0 * 4.0, stream);             // HBM-priced: E, dP1, dP2 read, dA written
  hipLaunchKernelGGL(dA_kernel<false>, dim3((int)g), dim3(256), 0, stream, E, dP1, dP2, rinv, cinv, del1, del2, hw, ldE, temperature, total4, slot(AM_DA));
  prof_end(pid_da, stream);
  DCN_CHECK_LAUNCH("dA");
  // 4. d_f1 (+)= dA f2 + E (dO2 / colsum)                                   (NN x2)
  gemm_params(p, dP1, ldE, (long long)hw * ldE, f2, ldf, bsf, d_f1, lddf, bsdf, hw, c, ldE, b);
  p.bmode = 1; p.kvalid = hw; p.accumulate = accumulate; p.amax_a = slot(AM_DA); p.amax_b = slot(AM_ONE);
  if ((rc = igemm_launch(p, stream))) return rc;
  gemm_params(p, E, ldE, (long long)hw * ldE, dO2s, c, (long long)hw * c, d_f1, lddf, bsdf, hw, c, ldE, b);
  p.bmode = 1; p.kvalid = hw; p.accumulate = 1; p.amax_a = slot(AM_ONE); p.amax_b = slot(AM_DO2S);
  if ((rc = igemm_launch(p, stream))) return rc;
  // 5. d_f2 (+)= dA^T f1 + E^T (dO1 / rowsum)                               (TN x2)
  rc = tn_gemm_batched(dP1, ldE, (long long)hw * ldE, f1, ldf, bsf, d_f2, lddf, bsdf,
                       nullptr, hw, ldE, c, hw, b, accumulate, stream, slot(AM_DA), slot(AM_ONE));
  if (rc) return rc;
  return tn_gemm_batched(E, ldE, (long long)hw * ldE, dO1s, c, (long long)hw * c, d_f2, lddf, bsdf,
                         nullptr, hw, ldE, c, hw, b, 1, stream, slot(AM_ONE), slot(AM_DO1S));
}
