// The stem: 3x3 stride-1 convolution of the 4-channel (RGB + pad) image to <= 32 filters, direct on the vector ALU.
//
// K = 27 is far too short for the matrix pipe: the implicit-GEMM c4 tile (igemm.hip) pads it to 64, stages a [256][16] tile per
// K-step for eight fp32 MFMAs and ran at 1.3 TB/s of output (1.09 ms for the 64 x 416 x 416 images of BASELINE.json configs[1];
// the 1.4 GB it writes take 0.3 ms at the copy rate of the HBM).  Here a thread owns one output pixel and all its filters:
//   * its nine taps are nine 16-B loads (out-of-image taps take the out-of-range buffer offset and read as zero: no branch),
//   * the filter bank is read [k = tap*3 + channel][32 filters]: a wave-uniform address, so the 32 weights of a k arrive through
//     the scalar cache in two s_load_dwordx16 and every v_fma takes its weight from a scalar register — no LDS, no broadcast,
//   * 27 x 32 fused multiply-adds per pixel (fp32, sequential over k: as exact as the fp32 MFMA it replaces),
//   * the block's [256 pixels][32 filters] go through LDS once: BatchNorm partial sums per 256 pixels (the layout of the igemm
//     epilogue: [row][2][Co]), then scale / shift / LeakyReLU and fully coalesced stores, abs-max of what is stored.
// Roofline: HBM (write of N*H*W*32 floats + read of N*H*W*4).
#include "igemm.h"
#include "prof.h"

namespace {

struct StemParams {
  const float* x;        // [M][4]
  const float* wk;       // [27][32]  (k = tap*3 + channel; filters >= Co are zero)
  float* y;              // [M][ldy]
  const float* scale; const float* shift;
  float* stats;          // [ceil(M/256)][2][Co] or null
  unsigned* amax_out;
  int N, H, W, M, Co, ldy, act; float slope;
};

// [Co][64] (k = tap*4 + channel, the c4 layout of ops.weight_to_ohwi) -> [27][32]
__global__ void stem_filters_kernel(const float* __restrict__ w, float* __restrict__ wk, int Co) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 27 * 32) return;
  const int k = i >> 5, co = i & 31, t = k / 3, c = k - 3 * t;
  wk[i] = co < Co ? w[co * 64 + t * 4 + c] : 0.f;
}

__global__ __launch_bounds__(256) void stem_kernel(const StemParams p) {
  constexpr int LDT = 36;                      // floats per pixel row in LDS (16-B aligned; b128 writes of 16 lanes hit 64 distinct banks)
  __shared__ __attribute__((aligned(16))) float tile[256 * LDT];
  __shared__ float part[8 * 2 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = blockIdx.x * 256 + tid;
  const bool valid = m < p.M;
  const int rem = m % (p.H * p.W), yy = rem / p.W, xx = rem - yy * p.W;
  const long long bytes = (long long)p.M * 16;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0u : (unsigned)bytes, 0x00020000);
  f32x4 xv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int dy = t / 3 - 1, dx = t % 3 - 1;
    const bool ok = valid && (unsigned)(yy + dy) < (unsigned)p.H && (unsigned)(xx + dx) < (unsigned)p.W;
    const unsigned off = ok ? (unsigned)(m + dy * p.W + dx) * 16u : 0x80000000u;
    xv[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
  }
  float acc[32];
#pragma unroll
  for (int co = 0; co < 32; ++co) acc[co] = 0.f;
  const float* __restrict__ wk = p.wk;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float xk = xv[t][c];
#pragma unroll
      for (int co = 0; co < 32; ++co) acc[co] = __builtin_fmaf(xk, wk[(t * 3 + c) * 32 + co], acc[co]);
    }

  // The accumulators go through LDS: a thread holds the 32 filters of ONE pixel (rows 128 B apart in y: a wave's 16-B stores
  // would touch 64 cache lines each, 8 partial writes per line — measured 0.63 ms, L2-request-bound); read back, 8 consecutive
  // lanes cover one pixel's 128 B and the block writes its 32 KB of y front to back.
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    *reinterpret_cast<f32x4*>(&tile[tid * LDT + 4 * q]) = v;
  }
  __syncthreads();
  if (p.stats) {                               // raw sums of this block's 256 pixels (pixels past the end contribute zero)
    {
      const int ch = tid & 31, pt = tid >> 5;  // 8 parts of 32 pixels
      float s = 0.f, ss = 0.f;
#pragma unroll 8
      for (int i = 0; i < 32; ++i) { const float v = tile[(pt * 32 + i) * LDT + ch]; s += v; ss = __builtin_fmaf(v, v, ss); }
      part[(pt * 2 + 0) * 32 + ch] = s; part[(pt * 2 + 1) * 32 + ch] = ss;
    }
    __syncthreads();
    if (tid < 64) {
      const int ch = tid & 31, which = tid >> 5;
      float t = 0.f;
#pragma unroll
      for (int pt = 0; pt < 8; ++pt) t += part[(pt * 2 + which) * 32 + ch];
      if (ch < p.Co) p.stats[((size_t)blockIdx.x * 2 + which) * p.Co + ch] = t;
    }
  }
  float vmax = 0.f;
  {
    const int q = tid & 7;                     // this thread's four filters, the same in every round
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    const bool live = 4 * q < p.Co;            // (Co % 4 == 0)
    if (live && p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + 4 * q);
    if (live && p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + 4 * q);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int px = it * 32 + (tid >> 3);
      const int mo = blockIdx.x * 256 + px;
      f32x4 v = *reinterpret_cast<const f32x4*>(&tile[px * LDT + 4 * q]);
      v = v * sc + sh;
      if (p.act == DCN_ACT_LEAKY) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * p.slope;
      }
      if (live && mo < p.M) {
        *reinterpret_cast<f32x4*>(p.y + (size_t)mo * p.ldy + 4 * q) = v;
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      }
    }
  }
  if (p.amax_out) {
    vmax = wave_max(vmax);
    if (lane == 0) amax_update(p.amax_out, vmax, blockIdx.x * 4 + wave);
  }
}

int g_stem_direct = 1;    // dcn_set_tuning("jstem", 0): the stem back on the implicit-GEMM c4 tile

}  // namespace

void stem_set_tuning(int v) { g_stem_direct = v; }

// can this forward launch run on the direct kernel?  (the caller's statistics buffer is sized for 256-row partials: Co <= 32)
bool stem_applicable(const IgemmParams& p, const float* scratch) {
  return g_stem_direct && scratch && p.c4 && p.Co <= 32 && p.Co % 4 == 0 && p.ldo % 4 == 0 && !p.residual && !p.accumulate && !p.row_scale &&
         p.batch <= 1 && p.isy == 1 && p.isx == 1 && p.ldi == 4 && p.Hs == p.Hi && p.Ws == p.Wi && p.M == p.N * p.Hi * p.Wi &&
         (long long)p.M * 16 < 0x7FFFFFF0LL && (p.act == DCN_ACT_NONE || p.act == DCN_ACT_LEAKY) &&
         (((uintptr_t)p.out | (uintptr_t)p.in | (uintptr_t)p.scale | (uintptr_t)p.shift | (uintptr_t)scratch) & 15) == 0;      // 16-B accesses
}

// scratch: >= 27*32 floats (the re-ordered filter bank of this launch)
int stem_launch(const IgemmParams& p, float* scratch, hipStream_t stream) {
  hipLaunchKernelGGL(stem_filters_kernel, dim3(4), dim3(256), 0, stream, p.wt, scratch, p.Co);
  DCN_CHECK_LAUNCH("stem_filters");
  StemParams s{};
  s.x = p.in; s.wk = scratch; s.y = p.out; s.scale = p.scale; s.shift = p.shift; s.stats = p.stats; s.amax_out = p.amax_out;
  s.N = p.N; s.H = p.Hi; s.W = p.Wi; s.M = p.M; s.Co = p.Co; s.ldy = p.ldo; s.act = p.act; s.slope = p.slope;
  const double alg_bytes = 4.0 * ((double)p.M * 4 + 27.0 * p.Co + (double)p.M * p.Co);
  const int pid = prof_begin(34, alg_bytes, stream);      // HBM-priced
  hipLaunchKernelGGL(stem_kernel, dim3(cdiv(p.M, 256)), dim3(256), 0, stream, s);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("stem");
  return DCN_OK;
}

// =====================================================================================================================
// Weight gradient of the stem with the BatchNorm + LeakyReLU backward applied on the fly.
//
//   dW[co][ky][kx][ci] = sum_p dY[p][co] * X[p + (ky-1, kx-1)][ci],   dY = gamma*invstd*(g - sum(g)/n - xhat*sum(g*xhat)/n)
//
// The stem has no data gradient, so its dY (1.4 GB at 64 x 416 x 416 x 32) exists only to feed this sum: dcn_bn_act_bwd_apply writes
// it and the c4 tile of wgrad.hip reads it back — 2.0 ms at the very end of the backward sweep, where nothing overlaps them.  Here
// one kernel reads y and dOut (the gradient w.r.t. the activation), forms dY per element exactly as bn_act_bwd_apply_kernel does,
// and contracts it with the im2col rows of X (9 taps x 3 channels = 27 of 32 columns) on v_mfma_f32_32x32x2_f32 — plain fp32, no
// operand scales: the whole contraction is 25 GFLOP.  K runs over PADDED positions (rows of W + 1 entries, the pad's dY is zero and
// the left / right taps of the border columns read zeros), 32 per step: 256 threads stage dY [32][32] and the im2col tile [32][27]
// in LDS (double buffered), wave w contracts positions 8w .. 8w+7 (four MFMAs), the four partial tiles meet in LDS at the end.
// One [32][64] slab per workgroup in the c4 layout of the stem's filter bank (k = tap*4 + channel), summed in a fixed order by
// reduce_slabs_kernel.  Roofline: HBM (y + dOut + X once: 3.0 GB).
int wgrad_reduce_slabs(const float* ws, float* dw, int64_t n4, int splits, hipStream_t stream);

namespace {

struct StemWParams {
  const float* x;                  // [M][4]
  const float* y;                  // raw convolution output [M][32]
  const float* dout; int lddo;     // gradient w.r.t. the activation (lddo in elements)
  int dout_b16;                    // bf16 storage: dout is a bf16 tensor (the data gradient of the layer behind the stem writes bf16)
  const float* mean; const float* invstd; const float* gamma; const float* beta; const float* sums;    // sums: [2][Co] = sum g, sum g*xhat
  float inv_count; int act; float slope;
  float* out;                      // slabs [grid][32][64]
  int N, H, W, Co;
  int Mp, kchunk;                  // padded positions N*H*(W+1); per workgroup (multiple of 32)
};

constexpr int SW_BP = 33;          // floats per im2col row in LDS (32-way bank conflict on the column writes otherwise)

__global__ __launch_bounds__(256) void stem_wgrad_bn_kernel(const StemWParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 32 * 32 + 2 * 32 * SW_BP];
  float (*sa)[32 * 32] = reinterpret_cast<float (*)[32 * 32]>(smem);                        // dY [buffer][position][filter]
  float (*sb)[32 * SW_BP] = reinterpret_cast<float (*)[32 * SW_BP]>(smem + 2 * 32 * 32);    // im2col [buffer][position][tap*3 + channel]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Wp = p.W + 1, NR = p.N * p.H;
  const int p_begin = blockIdx.x * p.kchunk, p_end = min(p.Mp, p_begin + p.kchunk);
  const int iters = (p_end - p_begin + 31) / 32;
  for (int i = tid; i < 2 * 32 * SW_BP; i += 256) (&sb[0][0])[i] = 0.f;      // columns 27..31 stay zero

  // ---- dY slot: position pa, filters ca..ca+3; im2col slot: position pb, taps tg (and 8 for tg == 0) -----------------------
  const int pa = tid >> 3, ca = (tid & 7) * 4;
  const int pb = tid & 31, tg = tid >> 5;
  const bool ca_on = ca < p.Co;
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = mu, ga = {1.f, 1.f, 1.f, 1.f}, be = mu, k1 = mu, k2 = mu;
  if (ca_on) {
    mu = *reinterpret_cast<const f32x4*>(p.mean + ca); is = *reinterpret_cast<const f32x4*>(p.invstd + ca);
    if (p.gamma) ga = *reinterpret_cast<const f32x4*>(p.gamma + ca);
    if (p.beta) be = *reinterpret_cast<const f32x4*>(p.beta + ca);
    k1 = *reinterpret_cast<const f32x4*>(p.sums + ca); k2 = *reinterpret_cast<const f32x4*>(p.sums + p.Co + ca);
  }
  const long long xbytes = (long long)NR * p.W * 16;
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, xbytes > 0x7FFFFFF0LL ? 0x7FFFFFF0u : (unsigned)xbytes, 0x00020000);

  int g_row = p_begin / Wp, g_col = p_begin - g_row * Wp, g_y = g_row % p.H, q_step = p_begin;      // scalar: first position of the step being loaded
  f32x4 yv, dv, xv0, xv1;
  bool a_ok = false;                                       // the dY slot of the loaded step is a real pixel (not a pad, not past the end)
  auto load_step = [&]() {
    {
      int col = g_col + pa, row = g_row;
      if (col >= Wp) { col -= Wp; ++row; }
      const bool ok = ca_on && col < p.W && q_step + pa < p_end;
      const size_t pix = (size_t)row * p.W + col;
      yv = ok ? *reinterpret_cast<const f32x4*>(p.y + pix * 32 + ca) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.dout_b16) {
        typedef __bf16 bf16x4s_t __attribute__((ext_vector_type(4)));
        const bf16x4s_t b = ok ? *reinterpret_cast<const bf16x4s_t*>(reinterpret_cast<const __bf16*>(p.dout) + pix * p.lddo + ca) : bf16x4s_t{0, 0, 0, 0};
        dv = f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
      } else
      dv = ok ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.dout + pix * p.lddo + ca)) : f32x4{0.f, 0.f, 0.f, 0.f};
      a_ok = ok;
    }
    {
      int col = g_col + pb, row = g_row, yy = g_y;
      if (col >= Wp) { col -= Wp; ++row; if (++yy == p.H) yy = 0; }
      auto tap = [&](int t) {
        const int iy = yy + t / 3 - 1, ix = col + t % 3 - 1;
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && row < NR;
        const unsigned off = ok ? (unsigned)(((row + t / 3 - 1) * p.W + ix) * 16) : 0x80000000u;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, off, 0, 0));
      };
      xv0 = tap(tg);
      if (tg == 0) xv1 = tap(8);
    }
    q_step += 32; g_col += 32;
    if (g_col >= Wp) { g_col -= Wp; ++g_row; if (++g_y == p.H) g_y = 0; }
  };
  auto store_step = [&](int buf) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (a_ok) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {                       // the arithmetic of bn_act_bwd_apply_kernel, term by term
        const float xh = (yv[k] - mu[k]) * is[k];
        float dd = dv[k];
        if (p.act == DCN_ACT_LEAKY && (ga[k] * xh + be[k]) <= 0.f) dd *= p.slope;
        o[k] = ga[k] * is[k] * (dd - k1[k] * p.inv_count - xh * k2[k] * p.inv_count);
      }
    }
    *reinterpret_cast<f32x4*>(&sa[buf][pa * 32 + ca]) = o;
    float* b = &sb[buf][pb * SW_BP + tg * 3];
    b[0] = xv0[0]; b[1] = xv0[1]; b[2] = xv0[2];
    if (tg == 0) { float* b8 = &sb[buf][pb * SW_BP + 24]; b8[0] = xv1[0]; b8[1] = xv1[1]; b8[2] = xv1[2]; }
  };

  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  __syncthreads();                                         // (the zero fill of sb)
  if (iters > 0) { load_step(); store_step(0); load_step(); }
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    const float* a = &sa[buf][(8 * wave + (lane >> 5)) * 32 + (lane & 31)];
    const float* b = &sb[buf][(8 * wave + (lane >> 5)) * SW_BP + (lane & 31)];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * i * 32], b[2 * i * SW_BP], acc, 0, 0, 0);
    store_step(buf ^ 1);                                   // step it + 1 (loaded during the previous step)
    load_step();                                           // step it + 2
    __syncthreads();
  }

  // ---- the four waves' partial tiles meet in LDS; D[m = filter][n = tap*3 + channel] -> slab[filter][tap*4 + channel] --------------
  float* red = smem;                                       // [wave][16][64]: 16 KB
  static_assert(sizeof(smem) >= 4 * 16 * 64 * sizeof(float), "exchange area");
#pragma unroll
  for (int q = 0; q < 16; ++q) red[(wave * 16 + q) * 64 + lane] = acc[q];
  __syncthreads();
  float* slab = p.out + (size_t)blockIdx.x * 32 * 64;
#pragma unroll
  for (int e8 = 0; e8 < 8; ++e8) {
    const int e = tid * 8 + e8, co = e >> 6, k = e & 63;
    float v = 0.f;
    if (k < 36 && (k & 3) < 3) {
      const int n = (k >> 2) * 3 + (k & 3);
      const int q = (co & 3) + 4 * (co >> 3), ln = ((co >> 2) & 1) * 32 + n;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += red[(w * 16 + q) * 64 + ln];
    }
    slab[e] = v;
  }
}

struct PlanSW { int grid, kchunk, Mp; };
PlanSW plan_sw(int n, int h, int wd) {
  PlanSW pl;
  pl.Mp = n * h * (wd + 1);
  int target = 1024;                     // slabs: 2048 of them cost 0.24 ms in reduce_slabs at the very end of the backward sweep
  const int max_splits = pl.Mp / 256 > 0 ? pl.Mp / 256 : 1;
  if (target > max_splits) target = max_splits;
  pl.kchunk = cdiv(cdiv(pl.Mp, target), 32) * 32;
  pl.grid = cdiv(pl.Mp, pl.kchunk);
  return pl;
}

}  // namespace

extern "C" int64_t dcn_stem_bwd_weight_bn_ws(int n, int h, int wd) { return (int64_t)plan_sw(n, h, wd).grid * 32 * 64; }

static int stem_bwd_impl(const float* x, const float* y, const float* dout, int dout_b16, int lddo,
                         const float* mean, const float* invstd, const float* gamma, const float* beta,
                         int act, float slope, const float* sums, int64_t count,
                         int n, int h, int wd, int cout, float* dw, float* ws, void* stream_);

extern "C" int dcn_stem_bwd_weight_bn(const float* x, const float* y, const float* dout, int lddo,
                                      const float* mean, const float* invstd, const float* gamma, const float* beta,
                                      int act, float slope, const float* sums, int64_t count,
                                      int n, int h, int wd, int cout, float* dw, float* ws, void* stream_) {
  return stem_bwd_impl(x, y, dout, 0, lddo, mean, invstd, gamma, beta, act, slope, sums, count, n, h, wd, cout, dw, ws, stream_);
}
// the same with dout as a bf16 tensor (bf16 storage: the stem's raw output y stays fp32, the gradient that reaches it is bf16)
extern "C" int dcn_stem_bwd_weight_bn_b16(const float* x, const float* y, const void* dout16, int lddo,
                                          const float* mean, const float* invstd, const float* gamma, const float* beta,
                                          int act, float slope, const float* sums, int64_t count,
                                          int n, int h, int wd, int cout, float* dw, float* ws, void* stream_) {
  return stem_bwd_impl(x, y, (const float*)dout16, 1, lddo, mean, invstd, gamma, beta, act, slope, sums, count, n, h, wd, cout, dw, ws, stream_);
}

static int stem_bwd_impl(const float* x, const float* y, const float* dout, int dout_b16, int lddo,
                         const float* mean, const float* invstd, const float* gamma, const float* beta,
                         int act, float slope, const float* sums, int64_t count,
                         int n, int h, int wd, int cout, float* dw, float* ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DCN_CHECK_ARG(x && y && dout && mean && invstd && sums && dw && ws, "stem_bwd_weight_bn: null pointer");
  DCN_CHECK_ARG(n > 0 && h >= 2 && wd >= 32 && cout > 0 && cout <= 32 && cout % 4 == 0 && count > 0,
                "stem_bwd_weight_bn: bad shape (n=%d h=%d wd=%d cout=%d: wd >= 32, cout <= 32 and a multiple of 4)", n, h, wd, cout);
  DCN_CHECK_ARG(act == DCN_ACT_NONE || act == DCN_ACT_LEAKY, "stem_bwd_weight_bn: act=%d", act);
  if (lddo <= 0) lddo = cout;
  DCN_CHECK_ARG(lddo % 4 == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)dout | (uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)sums |
                                    (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, "stem_bwd_weight_bn: 16-byte alignment");
  DCN_CHECK_ARG(cout == 32, "stem_bwd_weight_bn: y rows of %d floats (32 expected: the dense output of the stem)", cout);
  DCN_CHECK_ARG((long long)n * h * (wd + 1) < 0x7FFFFFF0LL && (long long)n * h * wd * 16 < 0x7FFFFFF0LL, "stem_bwd_weight_bn: tensor too large");
  const PlanSW pl = plan_sw(n, h, wd);
  StemWParams p{};
  p.x = x; p.y = y; p.dout = dout; p.dout_b16 = dout_b16; p.lddo = lddo; p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.beta = beta; p.sums = sums;
  p.inv_count = 1.f / (float)count; p.act = act; p.slope = slope; p.out = ws;
  p.N = n; p.H = h; p.W = wd; p.Co = cout; p.Mp = pl.Mp; p.kchunk = pl.kchunk;
  const int pid = prof_begin(39, 4.0 * ((double)n * h * wd * (32 + 32 + 4) + (double)pl.grid * 2048), stream);      // HBM-priced
  hipLaunchKernelGGL(stem_wgrad_bn_kernel, dim3(pl.grid), dim3(256), 0, stream, p);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("stem_wgrad_bn");
  return wgrad_reduce_slabs(ws, dw, 32 * 64 / 4, pl.grid, stream);
}
