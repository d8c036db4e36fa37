// Implicit-GEMM convolution on the CDNA4 fp32 matrix pipe.
//
// Roofline: MFMA.  v_mfma_f32_32x32x2_f32 = 4096 FLOP / 64 cycles / SIMD -> 157.3 TFLOP/s chip peak
// (MI355X_MICROARCH.md: fp32 matrix rate == fp32 vector rate; there is no xf32/TF32 on gfx950).
//
// Tiling (one workgroup = 4 waves of 64):
//   (BMODE 1, "NN": the B operand is N-contiguous in HBM — e.g. P.V of the co-attention — so its LDS
//    tile is a straight [k][n] copy and fragments are four ds_read_b32 per lane instead of one b128.)
//   block tile BM x BN of the GEMM [M = pixels] x [N = Cout], K-step 32 channels of one tap;
//   A (activations, NHWC => K-contiguous rows) and B (weights OHWI => K-contiguous rows) are
//   staged global -> registers -> LDS (T14 split: issue the loads of step k+1 before the MFMAs
//   of step k, write them to the other LDS buffer after), one barrier per K-step;
//   LDS rows are padded 32 -> 36 floats, which makes the ds_read_b128 fragment reads
//   conflict-free (16-B slot index 9*row mod 16 is a bijection over a 16-lane group);
//   each lane reads 4 consecutive k per ds_read_b128 and feeds them to 4 MFMAs: lane half h
//   supplies k = 8*kk + 4*h + e to the e-th MFMA in BOTH operands, so the k-permutation cancels;
//   each wave owns a (BM/WM) x (BN/WN) sub-tile = MI x NI accumulators of 32x32 (16 VGPRs each).
//   C/D layout: col = lane & 31 -> output channel, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> pixel,
//   so every store instruction writes two full 128-B channel segments.
// Epilogue fuses: per-channel sum/sumsq partials of the raw result (train-mode BatchNorm
// statistics, deterministic: no atomics), scale/shift (folded eval BatchNorm or bias),
// LeakyReLU/ReLU, the shortcut add, and accumulate-into-destination.
#include "igemm.h"
#include "prof.h"

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

template <int BM, int BN, int WM, int WN, int BMODE>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
  constexpr int A_LD = BM / 32, B_LD = BN / 32;   // float4 global loads per thread per K-step
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [2][BM][LDS_LD]
  float* Bs = smem + 2 * BM * LDS_LD;    // BMODE 0: [2][BN][LDS_LD]   BMODE 1: [2][BK][BN]
  constexpr int B_TILE = BMODE == 0 ? BN * LDS_LD : BK * BN;
  const float* __restrict__ gin = p.in + (long long)blockIdx.y * p.in_bs;
  const float* __restrict__ gwt = p.wt + (long long)blockIdx.y * p.wt_bs;
  float* __restrict__ gout = p.out + (long long)blockIdx.y * p.out_bs;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int gn = (p.Co + BN - 1) / BN;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int bm = lin / gn, bn = lin - bm * gn;

  // ---- per-thread gather coordinates for the A rows this thread stages -----------------
  const int chunk = tid & 7;             // which float4 of the 32-wide K-step
  const int row0 = tid >> 3;             // rows row0 + 32*j
  int a_nb[A_LD], a_iy[A_LD], a_ix[A_LD];
  {
    const int hsws = p.Hs * p.Ws;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int m = bm * BM + row0 + 32 * j;
      if (m < p.M) {
        const int n = m / hsws, rem = m - n * hsws;
        const int i = rem / p.Ws, jx = rem - i * p.Ws;
        a_nb[j] = n * p.Hi; a_iy[j] = i * p.isy; a_ix[j] = jx * p.isx;
      } else {
        a_nb[j] = -1; a_iy[j] = 0; a_ix[j] = 0;
      }
    }
  }
  f32x4 a_reg[A_LD], b_reg[B_LD];

  auto load_tiles = [&](int it) {
    int dy, dx, koff_a, koff_b;
    bool tap_ok = true;
    if (p.c4) {                          // first layer: Ci == 4, eight 3x3 taps per K-step
      const int t = it * 8 + chunk;
      tap_ok = t < p.ntaps;
      const int r = t / 3;
      dy = r - 1; dx = t - 3 * r - 1;
      koff_a = 0; koff_b = it * 32 + chunk * 4;
    } else {
      const int t = it / p.cpt;
      const int c0 = (it - t * p.cpt) * 32;
      dy = p.tap_dy[t]; dx = p.tap_dx[t];
      koff_a = c0 + chunk * 4; koff_b = p.tap_w[t] + c0 + chunk * 4;
    }
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int iy = a_iy[j] + dy, ix = a_ix[j] + dx;
      const bool ok = tap_ok && a_nb[j] >= 0 && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(gin + ((size_t)(a_nb[j] + iy) * p.Wi + ix) * p.ldi + koff_a);
      a_reg[j] = v;
    }
    if (BMODE == 0) {
#pragma unroll
      for (int j = 0; j < B_LD; ++j) {
        const int co = bn * BN + row0 + 32 * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (co < p.Co) v = *reinterpret_cast<const f32x4*>(gwt + (size_t)co * p.ldw + koff_b);
        b_reg[j] = v;
      }
    } else {                             // wt rows are K, columns are output channels
      const int kbase = koff_b - chunk * 4;
#pragma unroll
      for (int j = 0; j < B_LD; ++j) {
        const int idx = tid + 256 * j;
        const int k = idx / (BN / 4), n4 = (idx - k * (BN / 4)) * 4;
        const int col = bn * BN + n4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kbase + k < p.kvalid && col < p.Co) v = *reinterpret_cast<const f32x4*>(gwt + (size_t)(kbase + k) * p.ldw + col);
        b_reg[j] = v;
      }
    }
  };
  auto store_tiles = [&](int buf) {
    float* a = As + buf * BM * LDS_LD;
    float* b = Bs + buf * B_TILE;
#pragma unroll
    for (int j = 0; j < A_LD; ++j)
      *reinterpret_cast<f32x4*>(a + (row0 + 32 * j) * LDS_LD + chunk * 4) = a_reg[j];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      if (BMODE == 0) *reinterpret_cast<f32x4*>(b + (row0 + 32 * j) * LDS_LD + chunk * 4) = b_reg[j];
      else *reinterpret_cast<f32x4*>(b + (tid + 256 * j) * 4) = b_reg[j];
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int a_frag = (wm * (BM / WM) + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
  const int b_frag = BMODE == 0 ? (wn * (BN / WN) + (lane & 31)) * LDS_LD + (lane >> 5) * 4
                                : (lane >> 5) * 4 * BN + wn * (BN / WN) + (lane & 31);

  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int it = 0; it < p.kiters; ++it) {
    const int cur = it & 1;
    if (it + 1 < p.kiters) load_tiles(it + 1);
    const float* a = As + cur * BM * LDS_LD + a_frag;
    const float* b = Bs + cur * B_TILE + b_frag;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *reinterpret_cast<const f32x4*>(a + mi * 32 * LDS_LD + kk * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (BMODE == 0) bf[ni] = *reinterpret_cast<const f32x4*>(b + ni * 32 * LDS_LD + kk * 8);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) bf[ni][e] = b[(kk * 8 + e) * BN + ni * 32];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][e], bf[ni][e], acc[mi][ni], 0, 0, 0);
    }
    if (it + 1 < p.kiters) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue ----------------------------------------------------------------------------
  // (a) BatchNorm batch statistics of the raw result.  Rows beyond M gathered zeros, so they add 0.
  if (p.stats) {
    float* red = smem;                    // [2][WM][BN]  (LDS is free after the last barrier)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc[mi][ni][r]; s += v; ss += v * v; }
      s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
      if (lane < 32) {
        const int col = wn * (BN / WN) + ni * 32 + lane;
        red[(0 * WM + wm) * BN + col] = s;
        red[(1 * WM + wm) * BN + col] = ss;
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, col = tid - which * BN;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) t += red[(which * WM + w) * BN + col];
      const int co = bn * BN + col;
      if (co < p.Co) p.stats[((size_t)bm * 2 + which) * p.Co + co] = t;   // (stats are not batched)
    }
  }
  // (b) scale/shift, activation, residual, store
  float sc[NI], sh[NI];
  int co_[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int co = bn * BN + wn * (BN / WN) + ni * 32 + (lane & 31);
    co_[ni] = co;
    sc[ni] = (p.scale && co < p.Co) ? p.scale[co] : 1.f;
    sh[ni] = (p.shift && co < p.Co) ? p.shift[co] : 0.f;
  }
  const int hsws = p.Hs * p.Ws;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * (BM / WM) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int m = bm * BM + row;
      if (m >= p.M) continue;
      size_t pix;
      if (p.dense_out) {
        pix = (size_t)m;
      } else {
        const int n = m / hsws, rem = m - n * hsws;
        const int i = rem / p.Ws, jx = rem - i * p.Ws;
        pix = ((size_t)n * p.Ho + p.oy0 + i * p.osy) * p.Wo + p.ox0 + jx * p.osx;
      }
      const float rs = p.row_scale ? p.row_scale[(size_t)blockIdx.y * p.M + m] : 1.f;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (co_[ni] >= p.Co) continue;
        float v = acc[mi][ni][r] * rs * sc[ni] + sh[ni];
        if (p.act == DCN_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
        if (p.residual) v += p.residual[pix * p.ldr + co_[ni]];
        float* o = gout + pix * p.ldo + co_[ni];
        if (p.accumulate) v += *o;
        *o = v;
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int BMODE>
int launch_variant(const IgemmParams& p, hipStream_t stream) {
  const int gm = cdiv(p.M, BM), gn = cdiv(p.Co, BN);
  const size_t lds = (size_t)2 * (BM * LDS_LD + (BMODE == 0 ? BN * LDS_LD : BK * BN)) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, WM, WN, BMODE>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const int nb = p.batch > 0 ? p.batch : 1;
  const int tag = BMODE == 1 ? (BN == 128 ? 3 : 4) : (BN == 128 ? 0 : (BN == 64 ? 1 : 2));
  const double k_alg = p.c4 ? 27.0 : (double)p.ntaps * (p.bmode == 1 && p.kvalid > 0 ? p.kvalid : p.Ci);
  const int pid = prof_begin(tag, 2.0 * nb * (double)p.M * p.Co * k_alg, stream);
  hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, BMODE>), dim3(gm * gn, nb), dim3(256), lds, stream, p);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("igemm");
  return DCN_OK;
}

// tile choice: narrow-N tiles for the 32/64-channel layers so no MFMA column is wasted
inline int tile_bm(int Co) { return Co <= 32 ? 256 : 128; }

}  // namespace

int igemm_grid_m(int M, int Co) { return cdiv(M, tile_bm(Co)); }

int igemm_launch(const IgemmParams& p, hipStream_t stream) {
  DCN_CHECK_ARG(p.in && p.wt && p.out, "igemm: null pointer");
  DCN_CHECK_ARG(p.M > 0 && p.Co > 0 && p.kiters > 0, "igemm: empty problem (M=%d Co=%d kiters=%d)", p.M, p.Co, p.kiters);
  DCN_CHECK_ARG(p.c4 ? (p.Ci == 4) : (p.Ci % 32 == 0), "igemm: Ci=%d must be a multiple of 32 (or 4 in c4 mode)", p.Ci);
  DCN_CHECK_ARG(p.ntaps >= 1 && p.ntaps <= IGEMM_MAX_TAPS, "igemm: ntaps=%d", p.ntaps);
  DCN_CHECK_ARG(p.ldi % 4 == 0 && p.ldw % 4 == 0, "igemm: ldi=%d ldw=%d must be multiples of 4 floats", p.ldi, p.ldw);
  DCN_CHECK_ARG(((uintptr_t)p.in & 15) == 0 && ((uintptr_t)p.wt & 15) == 0, "igemm: in/wt must be 16-byte aligned");
  DCN_CHECK_ARG(p.stats == nullptr || p.batch <= 1, "igemm: stats are not supported on batched launches");
  if (p.bmode == 1) {
    DCN_CHECK_ARG(p.ntaps == 1 && !p.c4 && p.Co % 4 == 0, "igemm: NN mode needs one tap and Co %% 4 == 0 (Co=%d)", p.Co);
    if (p.Co <= 64) return launch_variant<128, 64, 2, 2, 1>(p, stream);
    return launch_variant<128, 128, 2, 2, 1>(p, stream);
  }
  if (p.Co <= 32) return launch_variant<256, 32, 4, 1, 0>(p, stream);
  if (p.Co <= 64) return launch_variant<128, 64, 2, 2, 0>(p, stream);
  return launch_variant<128, 128, 2, 2, 0>(p, stream);
}
