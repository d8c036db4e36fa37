// Implicit-GEMM convolution engine (fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32).
//
//   out[pix(m)][co] (+)= epilogue( sum_{t < ntaps} sum_{c < Ci} in[src(m,t)][c] * wt[co][tap_w[t] + c] )
//
// m enumerates a (possibly strided) sub-grid of output pixels, src(m,t) is the gathered input
// pixel for tap t.  The same kernel serves: forward 1x1/3x3 stride 1/2, the data gradient
// (stride 1: flipped taps; stride 2: four parity classes with 1/2/2/4 taps each) and plain
// GEMMs (1 tap).  See igemm.hip for the tiling.
#pragma once
#include "common.h"

#define IGEMM_MAX_TAPS 16


struct IgemmParams {
  const float* in;        // gathered tensor, NHWC, pixel stride ldi
  const float* wt;        // [Co][ldw] K-contiguous rows
  const float* f8;        // optional device {sA, sB}: power-of-two operand scales of the fp8 path (dcn_f8_scale); null = off
  const unsigned* amax_a; // optional device word: float bits of max|in| (f16 two-piece split: the kernel derives its power-of-two
  const unsigned* amax_b; //   scales from these); both needed, else the launch stays on the bf16 three-piece split
  const void* wt16;       // non-null (bf16-operand mode): the filter bank converted to bf16, same [Co][ldw] layout (conv3.hip only)
  const float* b_scale;   // non-null: wt is pre-split (dcn_presplit_f16) with this power-of-two scale; igemm_will_presplit() says when
  unsigned* amax_out;     // optional device word: atomicMax of |stored values| (the abs-max of the tensor this launch produces)
  float* out;             // NHWC, pixel stride ldo
  const float* scale;     // per-Co, may be null
  const float* shift;     // per-Co, may be null
  const float* residual;  // same pixel indexing as out, stride ldr, may be null
  float* stats;           // [grid_m][2][Co] raw-result sum / sumsq partials, may be null
  const float* row_scale; // per output row m (index batch*M + m), may be null: out = acc*row_scale[m]
  int batch;              // grid.y; operands advance by the strides below (0 = shared)
  long long in_bs, wt_bs, out_bs;
  int bmode;              // 0: wt is [Co][ldw] (K contiguous, "NT"); 1: wt is [K][ldw] (N contiguous, "NN")
  int kvalid;             // bmode 1: rows k >= kvalid of wt read as zero (K tail)
  int N, Hi, Wi, Ci, ldi;
  int Ho, Wo;             // full output spatial dims (addressing)
  int Hs, Ws;             // sub-grid of output pixels covered by this launch
  int oy0, ox0, osy, osx; // output pixel = (oy0 + i*osy, ox0 + j*osx)
  int isy, isx;           // gathered pixel = (i*isy + dy[t], j*isx + dx[t])
  int Co, ldo, ldr, ldw;
  int M;                  // N*Hs*Ws
  int ntaps;              // taps; Ci must be a multiple of 32
  int cpt, kiters;        // filled by igemm_launch: K-steps per tap / in total for the chosen K-step size
  int act; float slope;
  int accumulate, dense_out, c4;
  int tap_dy[IGEMM_MAX_TAPS], tap_dx[IGEMM_MAX_TAPS], tap_w[IGEMM_MAX_TAPS];
  // ncls = 4: four sub-problems (the parity classes of a stride-2 data gradient) in one launch; class c has M / Hs / Ws / oy0 /
  // ox0 / ntaps below and its taps at tap_*[4c ...]; M, ntaps above then hold the largest class (tile and grid sizing)
  int ncls;
  int cls_M[4], cls_Hs[4], cls_Ws[4], cls_oy0[4], cls_ox0[4], cls_ntaps[4];
  // BatchNorm tap (stride-1 data gradients on conv1.hip / conv3.hip): the result is the complete gradient w.r.t. act(bn(bt_y)) (+ a
  // shortcut); `stats` then receives the partial sums of g = result * act'(.) and g * xhat per channel — what dcn_bn_act_bwd_reduce
  // would compute in a pass of its own — instead of sum / sum of squares.  bt_y dense [M][Co].
  const float* bt_y; const float* bt_mean; const float* bt_invstd; const float* bt_gamma; const float* bt_beta;
  int bt_act; float bt_slope;
  // fp8 storage (conv1.hip conv1b_kernel<..., F8>): one e8m0 byte per pixel of the gathered tensor / per filter of the bank
  const unsigned char* a_scale8; const unsigned char* b_scale8;
};
// [M][Co] tensors a launch's epilogue touches, in units of the output: the store, plus one read each for accumulate / shortcut / tapped
// BatchNorm input (the "algorithmic bytes" of a launch count every operand it must move once)
inline double epilogue_reads(const IgemmParams& p) { return 1.0 + (p.accumulate ? 1.0 : 0.0) + (p.residual ? 1.0 : 0.0) + (p.bt_y ? 1.0 : 0.0); }
// can this launch take a BatchNorm tap (i.e. does it run on conv1.hip / conv3.hip, whose statistics epilogues know the tap)?
bool igemm_tap_capable(const IgemmParams& p);

// rows of the stats partial buffer (= number of M-blocks) the launch will use
int igemm_grid_m(int M, int Co, int ntaps);
int igemm_launch(const IgemmParams& p, hipStream_t stream);
// true if a launch with these dimensions (rows = M x batch, filters Co, taps, channels Ci) would take a tile that reads a
// pre-split filter bank in the current precision mode: the caller then pre-splits (dcn_presplit_f16) and sets b_scale
bool igemm_will_presplit(long long rows, int Co, int ntaps, int Ci);

// conv3.hip: the 3x3 stride-1 strip kernel (f16 split, pre-split filter bank).  gran = output rows per statistics partial.
bool conv3_applicable(const IgemmParams& p, int precision, int gran);
int conv3_launch(const IgemmParams& p, int gran, hipStream_t stream);
void conv3_set_tuning(int key, int value);
// conv3x.hip: the same launches on v_mfma_f32_16x16x32_f16 (256 x 128 tile, two taps per MFMA)
bool conv3x_takes(const IgemmParams& p, int gran);
int conv3x_launch(const IgemmParams& p, int gran, hipStream_t stream);
void conv3x_set_tuning(int v);
// ... and its bf16-storage form (conv1b_launch hands it the 3x3 stride-1 launches): conv3b_bm = pixels per M-tile = rows per BatchNorm
// partial (0: the launch stays on the gathered tiles)
int conv3b_bm(int M, int Co, int Wi);
bool conv3b_takes(const IgemmParams& p);
int conv3b_launch(const IgemmParams& p, int out_f32, hipStream_t stream);
void conv3b_set_tuning(int v);

// conv1.hip: NT launches with both tiles by LDS-DMA (f16 split, pre-split filter bank): 1x1 layers, stride-2 layers, the parity
// classes of their data gradients, narrow 3x3 layers.  gran = output rows per BatchNorm partial row (128 | 256).
bool conv1_applicable(const IgemmParams& p, int precision, int gran);
int conv1_launch(const IgemmParams& p, int gran, hipStream_t stream);
void conv1_set_tuning(int key, int value);
// ... and its bf16-storage form (activations, gradients and filter banks bf16 in HBM): every forward / data-gradient launch with
// Ci % 32 == 0.  conv1b_grid_m = BatchNorm partial rows (M-tiles) of a launch.
int conv1b_grid_m(int M, int Co, int ntaps, int s1_w = 0);      // s1_w: map width when the launch is a 3x3 stride-1 convolution, else 0
int conv1b_launch(const IgemmParams& p, int out_f32, hipStream_t stream);
// ... and its fp8-storage form (e4m3 bytes + one e8m0 scale per pixel / per filter: a_scale8, b_scale8; Ci % 64 == 0, <= 12 taps)
int conv1q_grid_m(int M, int Co);
int conv1q_launch(const IgemmParams& p, int out_f32, hipStream_t stream);
// conv2b.hip: the same on 256 x 256 tiles (eight waves, gemm3.hip's two-group schedule) for launches with Co % 256 == 0 and enough tiles
bool conv2b_takes(int M, int Co, int ntaps);
int conv2b_launch(const IgemmParams& p, int out_f32, hipStream_t stream);
void conv2b_set_tuning(int v);


// stem.hip: the 4-channel 3x3 stride-1 stem directly on the vector ALU (forward).  scratch: >= 27*32 floats.
bool stem_applicable(const IgemmParams& p, const float* scratch);
int stem_launch(const IgemmParams& p, float* scratch, hipStream_t stream);
void stem_set_tuning(int v);

// nconv.hip: the data gradients of the 32 -> 64 and 64 -> 128 3x3 stride-2 layers with the filter bank in registers (persistent workgroups, f16 split)
bool dgrad2_applicable(int n, int h, int wd, int cin, int cout, int ksize, int stride, int accumulate);
// tap (optional): the BatchNorm + activation in front of this convolution — the kernel also writes the partial sums its backward starts with
struct DcnBnTap { const float* y; const float* mean; const float* invstd; const float* gamma; const float* beta; int act; float slope;
                  float* stats; int stats_rows; };
int nconv1_launch_b16(int mode, const void* x, int ldi, const void* w16, void* y, int ldo, float* stats, int stats_rows,
                      int n, int h, int wd, int stride, hipStream_t stream);
int dgrad2_launch_b16(const void* dy, int lddy, const void* wt16, void* dx, int n, int h, int wd, int cin, int accumulate, hipStream_t stream);
int dgrad2_grid(int n, int h, int wd, int cin);      // workgroups (= statistics rows of a tap) of a launch, -1: device query failed
int dgrad2_launch(const float* dy, int lddy, const float* wt, float* dx, int n, int h, int wd, int cin, int accumulate,
                  const uint32_t* amax_dy, const uint32_t* amax_w, const DcnBnTap* tap, hipStream_t stream);
void nconv_set_tuning(int v);
// gemm3.hip: batched GEMM on pre-split operands (the co-attention products)
void gemm3_set_tuning(int v);
void gemm3_set_h1(int v);
bool gemm3_applicable(int M, int N, int K, int batch);
int gemm3_presplit(const float* src, int ld, long long bs, float* dst, int ldd, long long bsd, int batch, int rows, int c,
                   const unsigned* amax, hipStream_t stream);
int gemm3_launch(const float* A, int lda, long long a_bs, int at, const float* B, int ldb, long long b_bs, int bt,
                 float* C, int ldc, long long c_bs, const float* row_scale, long long rs_bs,
                 int M, int N, int K, int batch, int accumulate, const unsigned* amax_a, const unsigned* amax_b, hipStream_t stream,
                 unsigned* amax_out = nullptr);
int igemm_precision();       // dcn_set_tuning("precision"): 4 = f16 two-piece split (the default)
// ... and the 3x3 layers between 32 and 64 channels: mode 0 = forward 32 -> 64 (stride 1 | 2, BatchNorm partial sums), mode 1 = data
// gradient of the stride-1 layer (64 -> 32)
// loader-side activation: the gathered tensor is the raw output of a conv + BatchNorm layer; x' = act(x * scale[c] + shift[c])
struct DcnPreAct { const float* scale; const float* shift; int act; float slope; };
bool nconv1_applicable(int mode, int n, int h, int wd, int cin, int cout, int ksize, int stride);
int nconv1_launch(int mode, const float* x, int ldi, const float* w, float* y, int ldo, float* stats, int stats_rows,
                  int n, int h, int wd, int stride, const uint32_t* amax_x, const uint32_t* amax_w, const DcnPreAct* pre, hipStream_t stream);
