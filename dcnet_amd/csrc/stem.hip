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
