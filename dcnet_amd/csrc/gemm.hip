// Plain fp32 GEMM entry points on the conv engines (igemm.hip: NT / NN forms, wgrad.hip: TN form) and the
// LSTM cell kernels.  They carry the language branch of DCNet (model/DCNet_model.py:124-188 RNNEncoder,
// :268-276 mapping_lang, :190-219 PhraseAttention): Linear layers and the BiLSTM's input / recurrent
// projections are GEMMs with M = batch rows, the gate non-linearities are an elementwise kernel.
#include "igemm.h"

int tn_gemm_batched(const float* A, int lda, long long a_bs, const float* B, int ldb, long long b_bs,
                    float* C, int ldc, long long c_bs, const float* row_scale,
                    int M, int m_ld, int N, int K, int batch, int accumulate, hipStream_t stream,
                    const unsigned* amax_a = nullptr, const unsigned* amax_b = nullptr);

namespace {

void gemm_params(IgemmParams& p, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K) {
  p = IgemmParams{};
  p.osy = p.osx = p.isy = p.isx = 1; p.dense_out = 1;
  p.in = A; p.ldi = lda; p.wt = B; p.ldw = ldb; p.out = C; p.ldo = ldc; p.ldr = ldc;
  p.N = 1; p.Hi = 1; p.Wi = M; p.Ho = 1; p.Wo = M; p.Hs = 1; p.Ws = M; p.M = M;
  p.Ci = K; p.Co = N; p.ntaps = 1; p.batch = 1;
}

// gates [n][4H] (i,f,g,o pre-activations) -> c, h.  One thread per (row, hidden unit).  Rows whose
// sequence has ended (t >= len) keep (h, c) and emit zeros (packed-sequence semantics).
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                            const float* __restrict__ h_prev, const int64_t* __restrict__ lens, int t,
                                                            float* __restrict__ act, float* __restrict__ c_out, float* __restrict__ h_out,
                                                            float* __restrict__ y, int ldy, int n, int H) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * H) return;
  const int r = idx / H, k = idx - r * H;
  const float* g = gates + (size_t)r * 4 * H;
  const float i = 1.f / (1.f + expf(-g[k])), f = 1.f / (1.f + expf(-g[H + k]));
  const float gg = tanhf(g[2 * H + k]), o = 1.f / (1.f + expf(-g[3 * H + k]));
  const float cp = c_prev[idx];
  const float c = f * cp + i * gg;
  const float tc = tanhf(c);
  const float h = o * tc;
  const bool live = lens == nullptr || t < lens[r];
  float* a = act + (size_t)r * 5 * H;           // saved for the backward: i, f, g, o, tanh(c)
  a[k] = i; a[H + k] = f; a[2 * H + k] = gg; a[3 * H + k] = o; a[4 * H + k] = tc;
  c_out[idx] = live ? c : cp;
  h_out[idx] = live ? h : h_prev[idx];
  y[(size_t)r * ldy + k] = live ? h : 0.f;
}

// dh (from the output at step t plus the recurrent path), dc_next -> dgates [n][4H], dc_prev, and the
// pass-through parts for finished rows.
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ dh_rec,
                                                            const float* __restrict__ dc_next, const float* __restrict__ act,
                                                            const float* __restrict__ c_prev, const int64_t* __restrict__ lens, int t,
                                                            float* __restrict__ dgates, int ldg, float* __restrict__ dc_prev,
                                                            float* __restrict__ dh_pass, int n, int H) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * H) return;
  const int r = idx / H, k = idx - r * H;
  const bool live = lens == nullptr || t < lens[r];
  const float* a = act + (size_t)r * 5 * H;
  float* dg = dgates + (size_t)r * ldg;
  const float dhr = dh_rec ? dh_rec[idx] : 0.f, dcn = dc_next ? dc_next[idx] : 0.f;
  if (!live) {             // state was copied through: gradients pass straight to the previous step
    dg[k] = 0.f; dg[H + k] = 0.f; dg[2 * H + k] = 0.f; dg[3 * H + k] = 0.f;
    dc_prev[idx] = dcn; dh_pass[idx] = dhr;
    return;
  }
  const float i = a[k], f = a[H + k], g = a[2 * H + k], o = a[3 * H + k], tc = a[4 * H + k];
  const float dh = dy[(size_t)r * lddy + k] + dhr;
  const float dc = dcn + dh * o * (1.f - tc * tc);
  dg[k] = dc * g * i * (1.f - i);
  dg[H + k] = dc * c_prev[idx] * f * (1.f - f);
  dg[2 * H + k] = dc * i * (1.f - g * g);
  dg[3 * H + k] = dh * tc * o * (1.f - o);
  dc_prev[idx] = dc * f;
  dh_pass[idx] = 0.f;
}

}  // namespace

// C[M][N] (+)= act(A[M][K] . B[N][K]^T + bias[N]) + residual      (rows of A and B are K-contiguous)
extern "C" int dcn_gemm_nt(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                           const float* bias, int act, const float* residual, int ldr, int accumulate, void* stream) {
  DCN_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && K % 32 == 0, "gemm_nt: bad argument (K=%d must be a multiple of 32)", K);
  IgemmParams p; gemm_params(p, A, lda, B, ldb, C, ldc, M, N, K);
  p.shift = bias; p.act = act; p.slope = 0.f; p.residual = residual; p.ldr = ldr > 0 ? ldr : ldc; p.accumulate = accumulate;
  return igemm_launch(p, (hipStream_t)stream);
}

// C[M][N] (+)= A[M][K] . B[K][N]      (B rows are N-contiguous; K may be any multiple of 32 with kvalid rows of B real)
extern "C" int dcn_gemm_nn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, int kvalid,
                           int accumulate, void* stream) {
  DCN_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && K % 32 == 0 && N % 4 == 0, "gemm_nn: bad argument (K=%d N=%d)", K, N);
  IgemmParams p; gemm_params(p, A, lda, B, ldb, C, ldc, M, N, K);
  p.bmode = 1; p.kvalid = kvalid > 0 ? kvalid : K; p.accumulate = accumulate;
  return igemm_launch(p, (hipStream_t)stream);
}

// C[M][N] (+)= A[K][M]^T . B[K][N]    (K is the strided dimension of both operands: weight-gradient form)
extern "C" int dcn_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
                           int accumulate, void* stream) {
  DCN_CHECK_ARG(M % 4 == 0, "gemm_tn: M=%d must be a multiple of 4", M);
  return tn_gemm_batched(A, lda, 0, B, ldb, 0, C, ldc, 0, nullptr, M, M, N, K, 1, accumulate, (hipStream_t)stream);
}

extern "C" int dcn_lstm_cell_fwd(const float* gates, const float* c_prev, const float* h_prev, const int64_t* lens, int t,
                                 float* act, float* c_out, float* h_out, float* y, int ldy, int n, int hidden, void* stream) {
  DCN_CHECK_ARG(gates && c_prev && h_prev && act && c_out && h_out && y && n > 0 && hidden > 0, "lstm_cell_fwd: bad argument");
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(cdiv((int64_t)n * hidden, 256)), dim3(256), 0, (hipStream_t)stream,
                     gates, c_prev, h_prev, lens, t, act, c_out, h_out, y, ldy, n, hidden);
  DCN_CHECK_LAUNCH("lstm_cell_fwd");
  return DCN_OK;
}

extern "C" int dcn_lstm_cell_bwd(const float* dy, int lddy, const float* dh_rec, const float* dc_next, const float* act,
                                 const float* c_prev, const int64_t* lens, int t, float* dgates, int ldg, float* dc_prev,
                                 float* dh_pass, int n, int hidden, void* stream) {
  DCN_CHECK_ARG(dy && act && c_prev && dgates && dc_prev && dh_pass && n > 0 && hidden > 0, "lstm_cell_bwd: bad argument");
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(cdiv((int64_t)n * hidden, 256)), dim3(256), 0, (hipStream_t)stream,
                     dy, lddy, dh_rec, dc_next, act, c_prev, lens, t, dgates, ldg > 0 ? ldg : 4 * hidden, dc_prev, dh_pass, n, hidden);
  DCN_CHECK_LAUNCH("lstm_cell_bwd");
  return DCN_OK;
}

// C[b][M][N] = A[b][M][K] . B[b][N][K]^T, `batch` independent problems (grid.y)
extern "C" int dcn_gemm_nt_batched(const float* A, int lda, int64_t a_bs, const float* B, int ldb, int64_t b_bs, float* C, int ldc,
                                   int64_t c_bs, int M, int N, int K, int batch, void* stream) {
  DCN_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && K % 32 == 0 && batch > 0, "gemm_nt_batched: bad argument (K=%d must be a multiple of 32)", K);
  DCN_CHECK_ARG(ldc >= N, "gemm_nt_batched: ldc=%d < N=%d", ldc, N);
  IgemmParams p; gemm_params(p, A, lda, B, ldb, C, ldc, M, N, K);
  p.batch = batch; p.in_bs = a_bs; p.wt_bs = b_bs; p.out_bs = c_bs;
  return igemm_launch(p, (hipStream_t)stream);
}
