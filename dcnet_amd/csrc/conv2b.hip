// bf16-storage convolution on 256 x 256 tiles: forward / data gradient of the layers with >= 256 filters (conv1.hip's conv1b serves the rest).
//
// Why a second bf16 kernel: conv1b's 128 x 128 tile stages 16 KB per K-step for 1.05 MFLOP — 64 FLOP per staged byte — and a CU takes in
// 24-70 GB/s whatever the instruction (MI355X_MICROARCH.md, gather table), so with ONE MFMA per product the kernel sits at 600-860 TFLOP/s
// and no tile aspect, ring depth or occupancy moves it.  What moves it is FLOP per staged byte: a 256 x 256
// tile stages 32 KB per 4.2 MFLOP = 128 FLOP/B.  This is gemm3.hip's schedule — eight waves, wave tile 128 x 64, a ring of four K-slices
// kept in flight by LDS-DMA across raw barriers, two wave groups running the loop one barrier apart so that one streams its MFMAs while
// the other reads its fragments and issues the next slice — with the operands of conv1b: a K-slice is 32 bf16 channels of one filter tap
// (64-byte rows, 16-byte chunk index XOR (row >> 2) & 3), the activation rows are GATHERED (per-row byte offset + bit mask of in-image taps,
// tap = a scalar pixel delta; out-of-image taps and rows past M are out-of-range DMA lanes = zeros), a lane's 16-byte fragment is an MFMA
// operand.  Epilogue = conv1b's (bf16 or fp32 store, BatchNorm partial sums of the values as stored / BatchNorm tap, scale-shift-
// activation, bf16 shortcut, accumulate); one statistics row per 256 output rows.
// Reference sites as conv1b: nn.Conv2d of model/darknet.py:172-191 and its autograd.  Roofline: MFMA, 2516.6 TFLOP/s.
#include "igemm.h"
#include "prof.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr unsigned C2_OOB = 0x80000000u;
constexpr int C2_BM = 256, C2_BN = 256, C2_STAGES = 4;
constexpr int C2_TILE = 256 * 64;                 // bytes of one operand tile of a K-slice: [256 rows][64 B]
constexpr int C2_STAGE = 2 * C2_TILE;
constexpr int C2_LDS = C2_STAGES * C2_STAGE;      // 128 KB

template <bool O32> struct Out2 { typedef __bf16 type; };
template <> struct Out2<true> { typedef float type; };

template <bool O32>
__global__ __launch_bounds__(512, 1) void conv2b_kernel(const IgemmParams p) {
  typedef typename Out2<O32>::type out_t;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem2b[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int gn = p.Co / C2_BN;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int bm = __builtin_amdgcn_readfirstlane(lin / gn), bn = lin - bm * gn;
  const int M = p.M, m0 = bm * C2_BM, n0 = bn * C2_BN;
  const int cpt = p.Ci >> 5;                      // 32-channel slices per tap
  const int kslices = p.ntaps * cpt;
  const int hsws = p.Hs * p.Ws;
  const bool plain = p.ntaps == 1 && p.dense_out && p.isy == 1 && p.isx == 1 && p.tap_dy[0] == 0 && p.tap_dx[0] == 0 &&
                     p.Ws == p.Wi && p.Hs == p.Hi;
  const __bf16* in16 = reinterpret_cast<const __bf16*>(p.in);
  const __bf16* wt16 = reinterpret_cast<const __bf16*>(p.wt);

  // ---- descriptors: activations from the tile's first image on, the filter bank whole -------------------------------------------------
  const int img0 = m0 / hsws;
  const long long img = (long long)p.Hi * p.Wi * p.ldi;               // elements per image of the gathered tensor
  const __bf16* a_base = in16 + (long long)img0 * img;
  const long long a_bytes = ((long long)(p.N - img0) * img - p.ldi + p.Ci) * 2;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, a_bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0 : (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)wt16, 0, (int)((long long)p.Co * p.ldw * 2), 0x00020000);

  // ---- per-lane source offsets of this wave's four 1-KiB pieces per K-slice (pieces 2 wave, 2 wave + 1 of each tile) ------------------
  // piece j = rows 16 j .. 16 j + 15; LDS position (row, c') = (16 j + lane / 4, lane % 4) holds chunk c = c' ^ ((row >> 2) & 3)
  unsigned voff[4], msk[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int row = 16 * (2 * wave + e) + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
    const int m = m0 + row;
    voff[e] = C2_OOB; msk[e] = 0;
    if (m < M) {
      if (plain) { voff[e] = (unsigned)((m - img0 * hsws) * p.ldi * 2 + c * 16); msk[e] = 1u; }
      else {
        const int n = m / hsws, rem = m - n * hsws;
        const int i = rem / p.Ws, jx = rem - i * p.Ws;
        const int iy0 = i * p.isy, ix0 = jx * p.isx;
        voff[e] = (unsigned)((((n - img0) * p.Hi + iy0) * p.Wi + ix0) * p.ldi * 2 + c * 16);
        unsigned mk = 0;
        for (int t = 0; t < p.ntaps; ++t)
          if ((unsigned)(iy0 + p.tap_dy[t]) < (unsigned)p.Hi && (unsigned)(ix0 + p.tap_dx[t]) < (unsigned)p.Wi) mk |= 1u << t;
        msk[e] = mk;
      }
    }
    voff[2 + e] = (unsigned)((n0 + row) * p.ldw * 2 + c * 16);        // (Co is a multiple of 256: every filter row exists)
  }
  const int my_dst = 2 * wave * 1024;

  int k_done = 0, k_tap = 0, k_c = 0;
  // this wave's pieces of its next K-slice: two of the activation tile, two of the filter tile (past the end: no-ops that count in vmcnt)
  auto issue = [&]() {
    const bool live = k_done < kslices;
    const unsigned bit = 1u << k_tap;
    int delta = 0, sa_ = 0, sb_ = 0;
    if (live) {
      delta = (p.tap_dy[k_tap] * p.Wi + p.tap_dx[k_tap]) * p.ldi * 2;
      sa_ = k_c * 64;
      sb_ = (p.tap_w[k_tap] + k_c * 32) * 2;
    }
    unsigned char* st = smem2b + (k_done & (C2_STAGES - 1)) * C2_STAGE + my_dst;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_void*)(st + e * 1024), 16,
                                               (int)((live && (msk[e] & bit)) ? voff[e] + (unsigned)delta : C2_OOB), sa_, 0, 0);
#pragma unroll
    for (int e = 0; e < 2; ++e)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (lds_void*)(st + C2_TILE + e * 1024), 16, (int)(live ? voff[2 + e] : C2_OOB), sb_, 0, 0);
    ++k_done; ++k_c;
    if (k_c == cpt) { k_c = 0; ++k_tap; }
  };

  // ---- fragment addresses (bytes inside a stage): MFMA k-block kb of a slice, lane half g: chunk 2 kb + g of the row --------------------
  const int g = lane >> 5;
  int a_rd[4][2], b_rd[2][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int ar = wm * 128 + mi * 32 + (lane & 31);
      a_rd[mi][kb] = ar * 64 + (((2 * kb + g) ^ ((ar >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int br = wn * 64 + ni * 32 + (lane & 31);
      b_rd[ni][kb] = C2_TILE + br * 64 + (((2 * kb + g) ^ ((br >> 2) & 3)) << 4);
    }
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  for (int s = 0; s < C2_STAGES - 1; ++s) issue();

  // gemm3.hip's loop: two groups of four waves (wm = 0 | 1: one wave of each per SIMD) ONE BARRIER APART —
  //   group 0:  b1 [read s, issue] b2 [MFMA s, wait s+1] b1 ...
  //   group 1:     b0              b1 [read s, issue, wait s+1]  b2 [MFMA s]  b1 ...
  // a slice is read only behind every wave's counted vmcnt for it AND a barrier; a stage is refilled only behind a barrier its last reader
  // reached with its reads retired (lgkmcnt(0) in front of b2).
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"((C2_STAGES - 2) * 4) : "memory");       // slice 0 of this wave
  if (wm == 1) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < kslices; ++it) {
    __builtin_amdgcn_s_barrier();                                                     // b1
    asm volatile("" ::: "memory");
    const unsigned char* st = smem2b + (it & (C2_STAGES - 1)) * C2_STAGE;
    bf16x8_t af[4][2], bf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) bf[ni][kb] = *reinterpret_cast<const bf16x8_t*>(st + b_rd[ni][kb]);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) af[mi][kb] = *reinterpret_cast<const bf16x8_t*>(st + a_rd[mi][kb]);
    }
    issue();                                                                          // slice it + 3, into the stage of slice it - 1; behind the reads
    if (wm == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((C2_STAGES - 2) * 4) : "memory");      // slice it + 1 of this wave
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                                     // b2
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][kb], bf[ni][kb], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (wm == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((C2_STAGES - 2) * 4) : "memory");      // slice it + 1 of this wave
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();
  // the no-op pieces issued past the end may still be pending LDS writes: drain before LDS is reused by the statistics
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue (conv1b's, on this kernel's accumulator map: rows m0 + wm 128 + mi 32 + 4 g + (r & 3) + 8 (r >> 2); filters n0 + wn 64 + ni 32 + lane % 32)
  out_t* __restrict__ gout = reinterpret_cast<out_t*>(p.out);
  auto out_pix = [&](int m) -> size_t {
    if (p.dense_out) return (size_t)m;
    const int n = m / hsws, rem = m - n * hsws;
    const int i = rem / p.Ws, jx = rem - i * p.Ws;
    return ((size_t)n * p.Ho + p.oy0 + i * p.osy) * p.Wo + p.ox0 + jx * p.osx;
  };
  auto row_of = [&](int mi, int r) { return m0 + wm * 128 + mi * 32 + 4 * g + (r & 3) + 8 * (r >> 2); };
  if (p.accumulate) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = row_of(mi, r);
        if (m >= M) continue;
        const size_t pix = out_pix(m);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni][r] += (float)gout[pix * p.ldo + n0 + wn * 64 + ni * 32 + (lane & 31)];
      }
  }
  if (p.stats) {                                         // one partial row per 256 output rows (rows >= M gathered zeros)
    float* red = reinterpret_cast<float*>(smem2b);       // [2][2 row groups wm][256 filters]
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      float s = 0.f, ss = 0.f;
      const int co = n0 + wn * 64 + ni * 32 + (lane & 31);
      if (p.bt_y) {                                      // BatchNorm tap (igemm.h): the terms of bn_act_bwd's reduce pass, on the gradient as stored
        const float mu = p.bt_mean[co], is = p.bt_invstd[co], ga = p.bt_gamma ? p.bt_gamma[co] : 1.f, be = p.bt_beta ? p.bt_beta[co] : 0.f;
        const __bf16* yb = reinterpret_cast<const __bf16*>(p.bt_y) + co;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          float yv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) { const int m = row_of(mi, r); yv[r] = (float)yb[(size_t)(m < M ? m : M - 1) * p.Co]; }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = row_of(mi, r);
            const float xh = (yv[r] - mu) * is;
            float gg = O32 ? acc[mi][ni][r] : (float)(__bf16)acc[mi][ni][r];
            if (p.bt_act == DCN_ACT_LEAKY) gg = (ga * xh + be <= 0.f) ? gg * p.bt_slope : gg;
            gg = m < M ? gg : 0.f;
            s += gg; ss += gg * xh;
          }
        }
      } else {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = O32 ? acc[mi][ni][r] : (float)(__bf16)acc[mi][ni][r];
            s += v; ss = __builtin_fmaf(v, v, ss);
          }
      }
      s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
      if (lane < 32) {
        red[(0 * 2 + wm) * C2_BN + wn * 64 + ni * 32 + lane] = s;
        red[(1 * 2 + wm) * C2_BN + wn * 64 + ni * 32 + lane] = ss;
      }
    }
    __syncthreads();
    {
      const int which = tid >> 8, col = tid & 255;       // 512 threads: one (sum | sum of squares, filter) each
      p.stats[((size_t)bm * 2 + which) * p.Co + n0 + col] = red[(which * 2 + 0) * C2_BN + col] + red[(which * 2 + 1) * C2_BN + col];
    }
  }
  float sc[2], sh[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int co = n0 + wn * 64 + ni * 32 + (lane & 31);
    sc[ni] = p.scale ? p.scale[co] : 1.f;
    sh[ni] = p.shift ? p.shift[co] : 0.f;
  }
  const __bf16* res16 = reinterpret_cast<const __bf16*>(p.residual);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = row_of(mi, r);
      if (m >= M) continue;
      const size_t pix = out_pix(m);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int co = n0 + wn * 64 + ni * 32 + (lane & 31);
        float v = acc[mi][ni][r] * sc[ni] + sh[ni];
        if (p.act == DCN_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
        if (res16) v += (float)res16[pix * p.ldr + co];
        gout[pix * p.ldo + co] = (out_t)v;
      }
    }
}

int g_conv2b = 256;       // dcn_set_tuning("2btile", min tiles; 0 = off; negative: also the 1-tap launches): multi-tap launches with that many 256 x 256 tiles run here

template <bool O32>
int launch2b(const IgemmParams& p, hipStream_t stream) {
  static DcnPerDeviceFlag attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2b_kernel<O32>), hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS);
  const int grid = cdiv(p.M, C2_BM) * (p.Co / C2_BN);
  const double k_alg = (double)p.ntaps * p.Ci;
  const double alg_bytes = 2.0 * ((double)p.N * p.Hi * p.Wi * p.Ci + (double)p.Co * k_alg) +
                           ((O32 ? 4.0 : 2.0) * (1.0 + (p.accumulate ? 1.0 : 0.0) + (p.residual ? 1.0 : 0.0)) + (p.bt_y ? 2.0 : 0.0)) * (double)p.M * p.Co;
  const int pid = prof_begin(47, 2.0 * (double)p.M * p.Co * k_alg, stream, alg_bytes);
  hipLaunchKernelGGL((conv2b_kernel<O32>), dim3(grid), dim3(512), C2_LDS, stream, p);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("conv2b");
  return DCN_OK;
}

}  // namespace

void conv2b_set_tuning(int v) { g_conv2b = v; }

// does a bf16-storage launch of M rows x Co filters run on the 256 x 256 tile?  (a function of the shape and the knob only: the caller
// sizes its BatchNorm partial rows with it)
// Multi-tap launches only (tools/bench_b16.py --set 2btile=0 --ab 2btile=256, N = 64): 512->512 3x3 @52 0.94 -> 0.78 ms (1046 TFLOP/s), 128->256
// 3x3 @52 0.168 -> 0.154, 128->256 stride 2 @104 0.155 -> 0.147; the 1x1 layers LOSE on it — short K loops, one workgroup per CU: 512->512 1x1
// @52 0.183 -> 0.202, the data gradient of 256->128 @52 0.044 -> 0.067 — and stay on conv1b (g_conv2b < 0: |g_conv2b| tiles, every tap count).
bool conv2b_takes(int M, int Co, int ntaps) {
  const int need = g_conv2b < 0 ? -g_conv2b : g_conv2b;
  return g_conv2b != 0 && (ntaps >= 4 || g_conv2b < 0) && Co % C2_BN == 0 && (long long)cdiv(M, C2_BM) * (Co / C2_BN) >= need;
}

int conv2b_launch(const IgemmParams& p, int out_f32, hipStream_t stream) {
  return out_f32 ? launch2b<true>(p, stream) : launch2b<false>(p, stream);
}
