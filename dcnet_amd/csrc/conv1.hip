// Implicit-GEMM convolution with both operand tiles brought in by LDS-DMA: every 1x1 layer (forward and data gradient), and the
// multi-tap launches that have no strip form — stride-2 3x3 layers, the parity classes of their data gradients, 3x3 layers with
// 32 / 64 filters on the 104..416-wide maps.
//
// Why a kernel of its own: the 1x1 layers are the time-dominant launches of the step (round-2 verdict: 26 % of it at 0.12-0.16 of
// the MFMA ceiling) and they are NOT matrix-pipe work — 256->128 channels on the 52x52 maps is 43 FLOP/B against a machine balance
// of 105, i.e. bound by HBM (265 MB: 0.053 ms at 5 TB/s) — yet the implicit-GEMM tile ran them at 2.6 TB/s.  Its loader stages
// through registers, two K-steps deep: 16 KB in flight per workgroup, 48 KB per CU, where 8 TB/s x ~2.5 us of loaded latency
// asks for ~80 KB per CU; and every K-step pays the f16 split + LDS stores of both tiles in front of 12 MFMAs.
//
// Here (cdna_hip_programming.md section 5, "Async global->LDS copy" / "Pipelining across barriers"):
//   * A (fp32 activations, K-contiguous rows) and B (the pre-split filter bank of dcn_prepare_filters: [8 h | 8 l] f16 per 8 k)
//     go global -> LDS with `buffer_load_dwordx4 ... lds` (no VGPRs, no ds_write): rings of K-steps (8 KB per tile and K-step)
//     kept in flight across the barriers (counted `s_waitcnt vmcnt`, raw `s_barrier`), two workgroups per CU: 80 KB of
//     activations in flight per CU.  The LDS image of an LDS-DMA is lane-linear, so both swizzles live in the per-lane SOURCE address.
//   * the f16 split of A happens on the fragment, in registers: a wave owns 32 rows x 128 filters (no other wave reads its rows),
//     so every activation is split exactly once per N-tile: 24 vector-ALU operations per 12 MFMAs.
//   * epilogue = igemm.hip's (BatchNorm partials, scale/shift, activation, shortcut, accumulate, abs-max).
// Roofline: whichever binds per layer — HBM for K <= 512 on the long maps, MFMA (838.9 TFLOP/s, three f16 MFMAs per product) for
// the 1024-channel layers of the 13x13 maps; bench.py prices every launch against max(FLOP / 838.9 T, bytes / 8 TB/s).
#include "igemm.h"
#include "prof.h"

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

// A tile of a K-step: [BM rows][16 k fp32] — 64-B rows, 16-B chunks XOR-swizzled by (row >> 2) & 3.
// B tile: 2 planes (h, l) of [BN filters][16 k f16] — 32-B rows, halves swapped in rows 8-15 (mod 16).
constexpr unsigned C1_OOB = 0x80000000u;         // beyond every descriptor: an LDS-DMA lane with this offset writes ZEROS (tools/ldsdma_oob_probe.hip)

__device__ __forceinline__ float c1_pow2_scale(unsigned amax_bits) {      // = igemm.hip pow2_scale
  const int be = (int)((amax_bits >> 23) & 0xFF);
  if (be == 0 || be == 255) return 1.f;
  int e = 14 - (be - 126);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return __uint_as_float((unsigned)(e + 127) << 23);
}

// SA / SB: ring depths (K-steps) of the activation / filter tiles.  The loads are WAVE-SPECIALISED — waves 0-1 bring in the
// 1-KiB pieces of an A tile, waves 2-3 those of a B tile — because `s_waitcnt vmcnt` retires a wave's loads in issue order: a
// wave that loaded both kinds could keep only as many A tiles in flight as B tiles.  A comes from HBM, B from L2 (the bank of this
// filter tile is shared by every M-tile).  MI x NI: 32x32 accumulator blocks per wave — tile BM = 128 MI rows (four waves stacked
// along M), BN = 32 NI filters: (1,4) wide layers, (1,2) / (2,2) 64 filters, (2,1) 32 filters.
// K runs over (tap, 16-channel step); a tap that leaves the image for a row is that lane's out-of-range offset (zeros), exactly
// like the buffer loads of igemm.hip: per row a byte offset and a bit mask of in-image taps, computed once.
// Waves per SIMD the 128 x 128 build must fit: its rings are sized for FOUR workgroups per CU (40 KB of LDS each), which also takes <= 128
// registers — the build had crept to 129 (three workgroups per CU) without anything failing.  -DC1_OCC=2 restores the old bound (A/B).
#ifndef C1_OCC
#define C1_OCC 4
#endif
template <int SA, int SB, int NI, int MI>
__global__ __launch_bounds__(256, (NI == 4 && MI == 1) ? C1_OCC : 2) void conv1_kernel(const IgemmParams p) {
  constexpr int BM = 128 * MI, BN = 32 * NI;
  constexpr int ASTAGE = BM * 64, BPLANE = BN * 32, BSTAGE = 2 * BPLANE;
  constexpr int AP = 4 * MI;                      // A pieces per loader wave and K-step (16 rows each)
  constexpr int NLD = AP > NI ? AP : NI;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem1[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gn = p.Co / BN;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int bm = lin / gn, bn = lin - bm * gn;
  const int M = p.M, m0 = bm * BM;
  const int cpt = p.Ci >> 4;                      // 16-channel steps per tap
  const int kiters = p.ntaps * cpt;
  const int hsws = p.Hs * p.Ws;
  const bool plain = p.ntaps == 1 && p.dense_out && p.isy == 1 && p.isx == 1 && p.tap_dy[0] == 0 && p.tap_dx[0] == 0 &&
                     p.Ws == p.Wi && p.Hs == p.Hi;            // GEMM rows: row m IS pixel m (no divisions)

  const float sa = c1_pow2_scale(amax_read(p.amax_a)), sb = p.b_scale[0];

  // ---- descriptors ------------------------------------------------------------------------------------------------
  const int n0 = m0 / hsws;                                           // image of the tile's first row
  const long long img = (long long)p.Hi * p.Wi * p.ldi;               // floats per image of the gathered tensor
  const float* a_base = p.in + (long long)n0 * img;
  const long long a_bytes = ((long long)(p.N - n0) * img - p.ldi + p.Ci) * 4;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, a_bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0 : (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, (int)((long long)p.Co * p.ldw * 4), 0x00020000);

  // ---- per-lane source offsets of this wave's 1-KiB LDS-DMA pieces per K-step (waves 0-1: A, waves 2-3: B) --------------------
  // A piece j (16 rows): LDS position (row, c') = (16 j + lane / 4, lane % 4) holds the row's chunk c = c' ^ ((row >> 2) & 3).
  // B piece (plane = wave & 1, e): 32 filters: LDS position (row, halfpos) = (32 e + lane / 2, lane & 1) holds k-half
  // kh = halfpos ^ ((row >> 3) & 1) of that plane: bank chunk 2 kh + plane of the K-step's 64 bytes.
  const bool loads_a = wave < 2;
  unsigned voff[NLD], msk[NLD];
#pragma unroll
  for (int e = 0; e < NLD; ++e) { voff[e] = C1_OOB; msk[e] = 0; }
  if (loads_a) {
#pragma unroll
    for (int e = 0; e < AP; ++e) {
      const int j = AP * (wave & 1) + e;
      const int row = 16 * j + (lane >> 2), cp = lane & 3, c = cp ^ ((row >> 2) & 3);
      const int m = m0 + row;
      if (m < M) {
        if (plain) { voff[e] = (unsigned)((m - n0 * hsws) * p.ldi * 4 + c * 16); msk[e] = 1u; }
        else {
          const int n = m / hsws, rem = m - n * hsws;
          const int i = rem / p.Ws, jx = rem - i * p.Ws;
          const int iy0 = i * p.isy, ix0 = jx * p.isx;
          voff[e] = (unsigned)((((n - n0) * p.Hi + iy0) * p.Wi + ix0) * p.ldi * 4 + c * 16);
          unsigned mk = 0;
          for (int t = 0; t < p.ntaps; ++t)
            if ((unsigned)(iy0 + p.tap_dy[t]) < (unsigned)p.Hi && (unsigned)(ix0 + p.tap_dx[t]) < (unsigned)p.Wi) mk |= 1u << t;
          msk[e] = mk;
        }
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < NI; ++e) {
      const int plane = wave & 1, rr = 32 * e + (lane >> 1), kh = (lane & 1) ^ ((rr >> 3) & 1);
      voff[e] = (unsigned)((bn * BN + rr) * p.ldw * 4 + (2 * kh + plane) * 16);
      msk[e] = 0xFFFFu;
    }
  }
  const __amdgpu_buffer_rsrc_t my_rs = loads_a ? a_rs : b_rs;
  const int my_dst = loads_a ? AP * (wave & 1) * 1024 : SA * ASTAGE + (wave & 1) * BPLANE;   // first piece of this wave inside a ring slot
  const int my_n = loads_a ? AP : NI;

  // wave-uniform K iterator of this wave's loads: (tap, channel step) of the next K-step to issue
  int k_tap = 0, k_c = 0, k_done = 0;
  auto issue = [&]() {                     // this wave's pieces of its next K-step (past the end: no-ops that still count in vmcnt)
    const bool live = k_done < kiters;
    const unsigned bit = 1u << k_tap;
    int delta = 0; unsigned soff = 0;
    if (live) {
      if (loads_a) { delta = (p.tap_dy[k_tap] * p.Wi + p.tap_dx[k_tap]) * p.ldi * 4; soff = (unsigned)k_c * 64u; }
      else soff = (unsigned)(p.tap_w[k_tap] + k_c * 16) * 4u;
    }
    unsigned char* st = smem1 + (loads_a ? (k_done % SA) * ASTAGE : (k_done % SB) * BSTAGE) + my_dst;
#pragma unroll
    for (int e = 0; e < NLD; ++e)
      if (e < my_n)
        // (explicit int casts: with unsigned arguments hipcc 7.2 silently drops the instantiation of the whole kernel template)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(my_rs, (lds_void*)(st + e * 1024), 16,
                                                 (int)((live && (msk[e] & bit)) ? voff[e] + (unsigned)delta : C1_OOB), (int)soff, 0, 0);
    ++k_done; ++k_c;
    if (k_c == cpt) { k_c = 0; ++k_tap; }
  };

  // ---- fragment addresses ----------------------------------------------------------------------------------------------
  const int kh = lane >> 5;
  int a_rd0[MI], a_rd1[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int ar = (wave * MI + mi) * 32 + (lane & 31);
    a_rd0[mi] = ar * 64 + (((2 * kh) ^ ((ar >> 2) & 3)) << 4);
    a_rd1[mi] = ar * 64 + (((2 * kh + 1) ^ ((ar >> 2) & 3)) << 4);
  }
  int b_rd[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int row = ni * 32 + (lane & 31);
    b_rd[ni] = SA * ASTAGE + row * 32 + (((kh ^ (row >> 3)) & 1) << 4);
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // ---- prologue: SA - 1 activation tiles / SB - 1 filter tiles in flight --------------------------------------------------
  for (int s = 0; s < (loads_a ? SA : SB) - 1; ++s) issue();

  for (int it = 0; it < kiters; ++it) {
    // this wave's pieces of K-step `it` have landed (all but the younger K-steps' loads), then everybody's
    if (loads_a) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((SA - 2) * AP) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((SB - 2) * NI) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // the slots read during the previous iteration are free (every wave has passed the barrier): refill them
    issue();
    const unsigned char* st = smem1 + (it % SA) * ASTAGE;
    const unsigned char* sb_ = smem1 + (it % SB) * BSTAGE;
    f16x8_t bh[NI], bl[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      bh[ni] = *reinterpret_cast<const f16x8_t*>(sb_ + b_rd[ni]);
      bl[ni] = *reinterpret_cast<const f16x8_t*>(sb_ + BPLANE + b_rd[ni]);
    }
    f16x8_t ah[MI], al[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(st + a_rd0[mi]), x1 = *reinterpret_cast<const f32x4*>(st + a_rd1[mi]);
      // x * s = h + l, two f16 (round to nearest): 11 + 11 significant bits
      const f32x4 t0 = x0 * sa, t1 = x1 * sa;
#pragma unroll
      for (int e = 0; e < 4; ++e) { ah[mi][e] = (_Float16)t0[e]; ah[mi][4 + e] = (_Float16)t1[e]; }
#pragma unroll
      for (int e = 0; e < 4; ++e) { al[mi][e] = (_Float16)(t0[e] - (float)ah[mi][e]); al[mi][4 + e] = (_Float16)(t1[e] - (float)ah[mi][4 + e]); }
    }
    // smallest terms first: (l,h) (h,l) (h,h)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
  }
  // the no-op pieces issued past the end may still be pending LDS writes: drain before LDS is reused / the workgroup ends
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float dq = 1.f / (sa * sb);           // powers of two: exact
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] *= dq;

  // ---- epilogue (igemm.hip's) -------------------------------------------------------------------------------------------
  float* __restrict__ gout = p.out;
  // output pixel of accumulator register r of block mi: row m = rbase(mi) + (r & 3) + 8 (r >> 2)
  auto out_pix = [&](int m) -> size_t {
    if (p.dense_out) return (size_t)m;
    const int n = m / hsws, rem = m - n * hsws;
    const int i = rem / p.Ws, jx = rem - i * p.Ws;
    return ((size_t)n * p.Ho + p.oy0 + i * p.osy) * p.Wo + p.ox0 + jx * p.osx;
  };
  if (p.accumulate) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wave * MI + mi) * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
        if (m >= M) continue;
        const size_t pix = out_pix(m);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni][r] += gout[pix * p.ldo + bn * BN + ni * 32 + (lane & 31)];
      }
  }
  if (p.stats) {                                         // one partial row per BM output rows (rows >= M gathered zeros)
    float* red = reinterpret_cast<float*>(smem1);      // [2][4 waves][BN]
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float s = 0.f, ss = 0.f;
      if (p.bt_y) {                                        // BatchNorm tap (igemm.h): channel_partials_kernel<1>'s terms (bn.hip)
        const int co = bn * BN + ni * 32 + (lane & 31);
        const float mu = p.bt_mean[co], is = p.bt_invstd[co], ga = p.bt_gamma ? p.bt_gamma[co] : 1.f, be = p.bt_beta ? p.bt_beta[co] : 0.f;
        // all loads of the block first (rows past the end clamped, their terms dropped below): one wait instead of one per element
        const float* yb = p.bt_y + co;
        float yv[MI][16];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wave * MI + mi) * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
            yv[mi][r] = yb[(size_t)(m < M ? m : M - 1) * p.Co];
          }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wave * MI + mi) * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
            const float xh = (yv[mi][r] - mu) * is;
            float g = acc[mi][ni][r];
            if (p.bt_act == DCN_ACT_LEAKY) g = (ga * xh + be <= 0.f) ? g * p.bt_slope : g;
            g = m < M ? g : 0.f;
            s += g; ss += g * xh;
          }
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float v = acc[mi][ni][r]; s += v; ss = __builtin_fmaf(v, v, ss); }      // (fma, spelled out: the sums are bitwise those of the build without the tap)
      }
      s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
      if (lane < 32) {
        red[(0 * 4 + wave) * BN + ni * 32 + lane] = s;
        red[(1 * 4 + wave) * BN + ni * 32 + lane] = ss;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 2 * BN; idx += 256) {
      const int which = idx / BN, col = idx - which * BN;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t += red[(which * 4 + w) * BN + col];
      p.stats[((size_t)bm * 2 + which) * p.Co + bn * BN + col] = t;
    }
  }
  float sc[NI], sh[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int co = bn * BN + ni * 32 + (lane & 31);
    sc[ni] = p.scale ? p.scale[co] : 1.f;
    sh[ni] = p.shift ? p.shift[co] : 0.f;
  }
  float vmax = 0.f;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (wave * MI + mi) * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
      if (m >= M) continue;
      const size_t pix = out_pix(m);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int co = bn * BN + ni * 32 + (lane & 31);
        float v = acc[mi][ni][r] * sc[ni] + sh[ni];
        if (p.act == DCN_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
        if (p.residual) v += p.residual[pix * p.ldr + co];
        gout[pix * p.ldo + co] = v;
        vmax = fmaxf(vmax, fabsf(v));
      }
    }
  if (p.amax_out) {
    vmax = wave_max(vmax);
    if (lane == 0) amax_update(p.amax_out, vmax, blockIdx.x * 4 + wave);
  }
}

// ---- bf16 storage (BASELINE.json configs[2]) ------------------------------------------------------------------------------------
// The same ring kernel on tensors that ARE bf16 in HBM: activations / raw conv outputs / gradients [pixels][C] bf16, the filter
// bank [Co][taps*Ci] bf16 (dcn_prepare_filters' b16 / tb16 forms).  A K-step is 32 channels = the same 64 bytes per row, so the
// LDS-DMA geometry of the activation tile is unchanged and the filter tile takes it over (64-byte rows, chunk XOR swizzle: one
// plane instead of the h | l pair).  A lane's 16-byte fragment is 8 bf16 = one MFMA operand: no split, no vector ALU in the
// loop, one v_mfma_f32_32x32x16_bf16 per product instead of three (ceiling 2516.6 TFLOP/s), half the bytes per element.
// fp32 accumulate; the raw result is rounded to bf16 BEFORE the BatchNorm partial sums, so the statistics are those of the stored
// tensor.  Serves every forward / data-gradient launch with Ci % 32 == 0 (1x1, 3x3 of either stride, parity classes).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <bool O32> struct OutT { typedef __bf16 type; };
template <> struct OutT<true> { typedef float type; };

// F8 (fp8 STORAGE, BASELINE.json configs[4]; round 5): the gathered tensor and the bank are OCP e4m3 bytes with ONE e8m0 scale per row —
// per pixel of the gathered tensor (p.a_scale8[pixel]) and per filter (p.b_scale8[filter]) — and the products run on the block-scaled
// v_mfma_scale_f32_32x32x64_f8f6f4 (2 x the bf16 rate; every 32-k block of a row is handed the row's scale).  A K-step is 64 channels = the
// same 64 bytes per row, so the rings, the swizzle and the epilogue are conv1b's; a lane's fragment is the 32 bytes of its k-half.  The
// scale bytes a lane needs — its rows' pixel under each tap, its four filters — are read once, before the first LDS-DMA is issued.
// tools/fp8_probe.hip measured this loop at 1.5-2.1 x the bf16 loop on the step's layer shapes (half the staged bytes per FLOP).
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
template <int I> struct IC1 { static constexpr int value = I; };
template <int N, int I = 0, typename F> __device__ __forceinline__ void static_for1(F&& f) {
  if constexpr (I < N) { f(IC1<I>{}); static_for1<N, I + 1>(f); }
}

template <int SA, int SB, int NI, int MI, bool O32, bool F8 = false>
__global__ __launch_bounds__(256, 2) void conv1b_kernel(const IgemmParams p) {
  typedef typename OutT<O32>::type out_t;
  constexpr int ESZ = F8 ? 1 : 2, KSTEP = 64 / ESZ;        // bytes per element, channels per K-step
  constexpr int BM = 128 * MI, BN = 32 * NI;
  constexpr int ASTAGE = BM * 64, BSTAGE = BN * 64;
  constexpr int AP = 4 * MI;                      // A pieces (16 rows x 64 B) per loader wave and K-step
  constexpr int NLD = AP > NI ? AP : NI;          // B: 2 NI pieces per K-step over two waves
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem1[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gn = p.Co / BN;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int bm = lin / gn, bn = lin - bm * gn;
  const int M = p.M, m0 = bm * BM;
  const int cpt = p.Ci / KSTEP;                   // 32-channel (fp8: 64-channel) steps per tap
  const int kiters = p.ntaps * cpt;
  const int hsws = p.Hs * p.Ws;
  const bool plain = p.ntaps == 1 && p.dense_out && p.isy == 1 && p.isx == 1 && p.tap_dy[0] == 0 && p.tap_dx[0] == 0 &&
                     p.Ws == p.Wi && p.Hs == p.Hi;

  const int n0 = m0 / hsws;
  const long long img = (long long)p.Hi * p.Wi * p.ldi;               // elements per image of the gathered tensor
  const char* a_base = reinterpret_cast<const char*>(p.in) + (long long)n0 * img * ESZ;
  const long long a_bytes = ((long long)(p.N - n0) * img - p.ldi + p.Ci) * ESZ;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, a_bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0 : (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, (int)((long long)p.Co * p.ldw * ESZ), 0x00020000);

  // per-lane source offsets of this wave's 1-KiB pieces: LDS position (row, c') = (16 j + lane / 4, lane % 4) of a tile with 64-byte
  // rows holds the row's 16-byte chunk c = c' ^ ((row >> 2) & 3) — activations (waves 0-1) and filters (waves 2-3) alike
  const bool loads_a = wave < 2;
  unsigned voff[NLD], msk[NLD];
#pragma unroll
  for (int e = 0; e < NLD; ++e) { voff[e] = C1_OOB; msk[e] = 0; }
  if (loads_a) {
#pragma unroll
    for (int e = 0; e < AP; ++e) {
      const int j = AP * (wave & 1) + e;
      const int row = 16 * j + (lane >> 2), cp = lane & 3, c = cp ^ ((row >> 2) & 3);
      const int m = m0 + row;
      if (m < M) {
        if (plain) { voff[e] = (unsigned)((m - n0 * hsws) * p.ldi * ESZ + c * 16); msk[e] = 1u; }
        else {
          const int n = m / hsws, rem = m - n * hsws;
          const int i = rem / p.Ws, jx = rem - i * p.Ws;
          const int iy0 = i * p.isy, ix0 = jx * p.isx;
          voff[e] = (unsigned)((((n - n0) * p.Hi + iy0) * p.Wi + ix0) * p.ldi * ESZ + c * 16);
          unsigned mk = 0;
          for (int t = 0; t < p.ntaps; ++t)
            if ((unsigned)(iy0 + p.tap_dy[t]) < (unsigned)p.Hi && (unsigned)(ix0 + p.tap_dx[t]) < (unsigned)p.Wi) mk |= 1u << t;
          msk[e] = mk;
        }
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < NI; ++e) {
      const int j = NI * (wave & 1) + e;
      const int row = 16 * j + (lane >> 2), cp = lane & 3, c = cp ^ ((row >> 2) & 3);
      voff[e] = (unsigned)((bn * BN + row) * p.ldw * ESZ + c * 16);
      msk[e] = 0xFFFFu;
    }
  }
  const __amdgpu_buffer_rsrc_t my_rs = loads_a ? a_rs : b_rs;
  const int my_dst = loads_a ? AP * (wave & 1) * 1024 : SA * ASTAGE + NI * (wave & 1) * 1024;
  const int my_n = loads_a ? AP : NI;

  int k_tap = 0, k_c = 0, k_done = 0;
  auto issue = [&]() {
    const bool live = k_done < kiters;
    const unsigned bit = 1u << k_tap;
    int delta = 0; unsigned soff = 0;
    if (live) {
      if (loads_a) { delta = (p.tap_dy[k_tap] * p.Wi + p.tap_dx[k_tap]) * p.ldi * ESZ; soff = (unsigned)k_c * 64u; }
      else soff = (unsigned)(p.tap_w[k_tap] + k_c * KSTEP) * (unsigned)ESZ;
    }
    unsigned char* st = smem1 + (loads_a ? (k_done % SA) * ASTAGE : (k_done % SB) * BSTAGE) + my_dst;
#pragma unroll
    for (int e = 0; e < NLD; ++e)
      if (e < my_n)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(my_rs, (lds_void*)(st + e * 1024), 16,
                                                 (int)((live && (msk[e] & bit)) ? voff[e] + (unsigned)delta : C1_OOB), (int)soff, 0, 0);
    ++k_done; ++k_c;
    if (k_c == cpt) { k_c = 0; ++k_tap; }
  };

  // fragments: MFMA k-block kb (16 channels) of a K-step, lane half kh: chunk 2 kb + kh of the row
  const int kh = lane >> 5;
  int a_rd[MI][2], b_rd[NI][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int ar = (wave * MI + mi) * 32 + (lane & 31);
      a_rd[mi][kb] = ar * 64 + ((((F8 ? 2 * kh + kb : 2 * kb + kh)) ^ ((ar >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int row = ni * 32 + (lane & 31);
      b_rd[ni][kb] = SA * ASTAGE + row * 64 + ((((F8 ? 2 * kh + kb : 2 * kb + kh)) ^ ((row >> 2) & 3)) << 4);
    }
  }
  // fp8: the e8m0 scale bytes this lane's MFMAs take — per accumulator block row its pixel's byte under every tap (packed four to a
  // register; a tap outside the image gathers zeros: any scale), per filter block its filter's byte.  Plain loads, before any LDS-DMA.
  unsigned sa_pk[MI][3], sb_pk[2];
  if constexpr (F8) {
    const unsigned char* as8 = reinterpret_cast<const unsigned char*>(p.a_scale8);
    const unsigned char* bs8 = reinterpret_cast<const unsigned char*>(p.b_scale8);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + (wave * MI + mi) * 32 + (lane & 31);
      long long pix = -1; int iy0 = 0, ix0 = 0;
      if (m < M) {
        if (plain) pix = m;
        else {
          const int n = m / hsws, rem = m - n * hsws;
          const int i = rem / p.Ws, jx = rem - i * p.Ws;
          iy0 = i * p.isy; ix0 = jx * p.isx;
          pix = ((long long)n * p.Hi + iy0) * p.Wi + ix0;
        }
      }
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        unsigned pk = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int t = 4 * w + b;
          unsigned v = 127u;
          if (t < p.ntaps && pix >= 0 && (plain || ((unsigned)(iy0 + p.tap_dy[t]) < (unsigned)p.Hi && (unsigned)(ix0 + p.tap_dx[t]) < (unsigned)p.Wi)))
            v = as8[pix + p.tap_dy[t] * p.Wi + p.tap_dx[t]];
          pk |= v << (8 * b);
        }
        sa_pk[mi][w] = pk;
      }
    }
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      unsigned pk = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) { const int ni = 4 * w + b; if (ni < NI) pk |= (unsigned)bs8[bn * BN + ni * 32 + (lane & 31)] << (8 * b); }
      sb_pk[w] = pk;
    }
    // (the values leave this block as fresh definitions: the compiler's wait for the loads stays here, in front of the loop, instead of
    //  becoming a vmcnt(0) at their first use while LDS-DMA pieces are in flight)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int w = 0; w < 3; ++w) asm volatile("" : "+v"(sa_pk[mi][w]));
    asm volatile("" : "+v"(sb_pk[0]), "+v"(sb_pk[1]));
  }
  int c_tap = 0, c_kc = 0;                        // (tap, channel step) of the K-step being MULTIPLIED (the loads run ahead)

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  for (int s = 0; s < (loads_a ? SA : SB) - 1; ++s) issue();

  for (int it = 0; it < kiters; ++it) {
    if (loads_a) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((SA - 2) * AP) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((SB - 2) * NI) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue();
    const unsigned char* st = smem1 + (it % SA) * ASTAGE;
    const unsigned char* sb_ = smem1 + (it % SB) * BSTAGE;
    bf16x8_t af[MI][2], bf[NI][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[ni][kb] = *reinterpret_cast<const bf16x8_t*>(sb_ + b_rd[ni][kb]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi][kb] = *reinterpret_cast<const bf16x8_t*>(st + a_rd[mi][kb]);
    }
    if constexpr (F8) {
      // this K-step's tap is wave-uniform: the pixel's scale byte under it, from the packed registers
      unsigned sa_now[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const unsigned w = c_tap < 4 ? sa_pk[mi][0] : (c_tap < 8 ? sa_pk[mi][1] : sa_pk[mi][2]);
        sa_now[mi] = (w >> (8 * (c_tap & 3))) & 0xFFu;
      }
      if (++c_kc == cpt) { c_kc = 0; ++c_tap; }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const v4i_t a0 = __builtin_bit_cast(v4i_t, af[mi][0]), a1 = __builtin_bit_cast(v4i_t, af[mi][1]);
        const v8i_t a8 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        static_for1<NI>([&](auto nic) {
          constexpr int ni = decltype(nic)::value;
          const v4i_t b0 = __builtin_bit_cast(v4i_t, bf[ni][0]), b1 = __builtin_bit_cast(v4i_t, bf[ni][1]);
          const v8i_t b8 = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
          acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[mi][ni], 0, 0, 0, (int)sa_now[mi], ni & 3, (int)sb_pk[ni >> 2]);
        });
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][kb], bf[ni][kb], acc[mi][ni], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue ---------------------------------------------------------------------------------------------------------
  out_t* __restrict__ gout = reinterpret_cast<out_t*>(p.out);
  auto out_pix = [&](int m) -> size_t {
    if (p.dense_out) return (size_t)m;
    const int n = m / hsws, rem = m - n * hsws;
    const int i = rem / p.Ws, jx = rem - i * p.Ws;
    return ((size_t)n * p.Ho + p.oy0 + i * p.osy) * p.Wo + p.ox0 + jx * p.osx;
  };
  if (p.accumulate) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wave * MI + mi) * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
        if (m >= M) continue;
        const size_t pix = out_pix(m);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni][r] += (float)gout[pix * p.ldo + bn * BN + ni * 32 + (lane & 31)];
      }
  }
  if (p.stats) {
    float* red = reinterpret_cast<float*>(smem1);      // [2][4 waves][BN]
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float s = 0.f, ss = 0.f;
      if (p.bt_y) {                                      // BatchNorm tap (igemm.h): the terms of bn_act_bwd's reduce pass
        const int co = bn * BN + ni * 32 + (lane & 31);
        const float mu = p.bt_mean[co], is = p.bt_invstd[co], ga = p.bt_gamma ? p.bt_gamma[co] : 1.f, be = p.bt_beta ? p.bt_beta[co] : 0.f;
        const __bf16* yb = reinterpret_cast<const __bf16*>(p.bt_y) + co;
        float yv[MI][16];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wave * MI + mi) * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
            yv[mi][r] = (float)yb[(size_t)(m < M ? m : M - 1) * p.Co];
          }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wave * MI + mi) * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
            const float xh = (yv[mi][r] - mu) * is;
            float g = O32 ? acc[mi][ni][r] : (float)(__bf16)acc[mi][ni][r];      // (the gradient as it is stored)
            if (p.bt_act == DCN_ACT_LEAKY) g = (ga * xh + be <= 0.f) ? g * p.bt_slope : g;
            g = m < M ? g : 0.f;
            s += g; ss += g * xh;
          }
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {             // the statistics of the tensor as it is stored (rounded to bf16)
            const float v = O32 ? acc[mi][ni][r] : (float)(__bf16)acc[mi][ni][r];
            s += v; ss = __builtin_fmaf(v, v, ss);
          }
      }
      s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
      if (lane < 32) {
        red[(0 * 4 + wave) * BN + ni * 32 + lane] = s;
        red[(1 * 4 + wave) * BN + ni * 32 + lane] = ss;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 2 * BN; idx += 256) {
      const int which = idx / BN, col = idx - which * BN;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t += red[(which * 4 + w) * BN + col];
      p.stats[((size_t)bm * 2 + which) * p.Co + bn * BN + col] = t;
    }
  }
  float sc[NI], sh[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int co = bn * BN + ni * 32 + (lane & 31);
    sc[ni] = p.scale ? p.scale[co] : 1.f;
    sh[ni] = p.shift ? p.shift[co] : 0.f;
  }
  const __bf16* res16 = reinterpret_cast<const __bf16*>(p.residual);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (wave * MI + mi) * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
      if (m >= M) continue;
      const size_t pix = out_pix(m);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int co = bn * BN + ni * 32 + (lane & 31);
        float v = acc[mi][ni][r] * sc[ni] + sh[ni];
        if (p.act == DCN_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
        if (res16) v += (float)res16[pix * p.ldr + co];
        gout[pix * p.ldo + co] = (out_t)v;
      }
    }
}

int g_conv1 = 1;          // dcn_set_tuning("1x1dma", 0): back on the implicit-GEMM tiles of igemm.hip; 2: 1x1 launches only
int g_conv1_stages = 32;  // dcn_set_tuning("1stages", 10 * SA + SB): ring depths.  Default 3 activation + 2 filter K-steps = 40 KB at the 128 x 128 tile:
                          // FOUR workgroups per CU (measured per layer, tools/bench_convs.py --set 1stages=..: 32 < 33 < 42 < 44 ~ 63 << 84: occupancy beats ring depth)

template <int SA, int SB, int NI, int MI>
int launch1(const IgemmParams& p, hipStream_t stream) {
  constexpr int BM = 128 * MI, BN = 32 * NI;
  static DcnPerDeviceFlag attr_once;
  const size_t lds = (size_t)SA * BM * 64 + (size_t)SB * 2 * BN * 32;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1_kernel<SA, SB, NI, MI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int gm = cdiv(p.M, BM), gn = p.Co / BN;
  const double k_alg = (double)p.ntaps * p.Ci;
  // algorithmic bytes: the gathered tensor once (its N*Hi*Wi pixels of Ci), the filter bank, the output
  // (every operand the launch must read once: gathered tensor, bank, and — where the epilogue asks for them — the tensor accumulated
  //  onto, the shortcut, the tapped BatchNorm input)
  const double alg_bytes = 4.0 * ((double)p.N * p.Hi * p.Wi * p.Ci + (double)p.Co * k_alg + (double)p.M * p.Co * epilogue_reads(p));
  const int pid = prof_begin(35, 2.0 * (double)p.M * p.Co * k_alg, stream, alg_bytes);
  hipLaunchKernelGGL((conv1_kernel<SA, SB, NI, MI>), dim3(gm * gn), dim3(256), lds, stream, p);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("conv1");
  return DCN_OK;
}

template <int NI, int MI>
int launch1_ring(const IgemmParams& p, hipStream_t stream) {
  if constexpr (NI == 4 && MI == 1) {
    switch (g_conv1_stages) {
      case 33: return launch1<3, 3, NI, MI>(p, stream);
      case 42: return launch1<4, 2, NI, MI>(p, stream);
      case 44: return launch1<4, 4, NI, MI>(p, stream);
      case 63: return launch1<6, 3, NI, MI>(p, stream);
      default: break;
    }
  }
  return launch1<3, 2, NI, MI>(p, stream);
}

// (MI, NI) for a launch, 0 = not on this kernel.  gran = rows per BatchNorm partial the caller sized its buffer for (128 | 256).
int g_conv1_fill = 0;     // dcn_set_tuning("1fill", n): grids of fewer 128 x 128 workgroups than n take 128 x 64 tiles (twice the workgroups).
                          // Measured on the 13x13 maps (340 workgroups for 1024 slots), tools/bench_convs.py --ab 1fill=0 --ab-default 512:
                          // 1024->512 0.058 -> 0.064 ms, 512->512 data gradient 0.039 -> 0.048: half the MFMAs per barrier costs more than
                          // the second round of workgroups brings; off.

int g_conv1_wide = 1024;  // dcn_set_tuning("1wide", min workgroups; 0 = off): 128 x 256 tiles (a wave owns 32 rows x 256 filters: half the split work and
                          // 0.75 instead of 0.83 fragment reads per MFMA) where the filter count allows and that many workgroups remain.
                          // tools/bench_convs.py --set 1wide=0|1, forward / data gradient: 1024->512 @52 0.806 -> 0.707 / 0.757 -> 0.691 ms,
                          // 512->512 @52 0.419 -> 0.372 / 0.433 -> 0.397, 256->512 @52 0.267 -> 0.235 / 0.225 -> 0.205; the short grids lose
                          // (512->256 @26, 338 workgroups: 0.062 -> 0.065; 1024->512 @13, 170: 0.070 -> 0.086): from 1024 workgroups on

int g_conv1_tall = 0;      // dcn_set_tuning("1tall", min workgroups; 0 = off): 256 x 128 tiles on FOUR waves (a wave owns 64 rows x 128 filters: 6 KB of fragment
                          // reads per 12 MFMAs instead of 10, two workgroups per CU) for launches without BatchNorm partials
int conv1_shape(const IgemmParams& p, int gran) {
  int ni = p.Co % 128 == 0 ? 4 : (p.Co % 64 == 0 ? 2 : (p.Co % 32 == 0 ? 1 : 0));
  if (!ni) return 0;
  if (g_conv1_tall && ni == 4 && !p.stats && (long long)cdiv(p.M, 256) * (p.Co / 128) >= g_conv1_tall) return 24;
  if (g_conv1_wide && p.Co % 256 == 0 && (!p.stats || gran == 128) && (long long)cdiv(p.M, 128) * (p.Co / 256) >= g_conv1_wide) return 18;
  // (experiment knob, off by default — see g_conv1_fill)
  if (ni == 4 && (!p.stats || gran == 128) && (long long)cdiv(p.M, 128) * (p.Co / 128) < g_conv1_fill) ni = 2;
  int mi;
  if (p.stats) { mi = gran == 128 ? 1 : (gran == 256 ? 2 : 0); }
  else mi = ni == 4 ? 1 : 2;
  if (!mi || mi * ni > 4 || (mi == 1 && ni == 1)) return 0;
  return mi * 10 + ni;
}

// bf16 storage: (MI, NI) of a launch; 0 = no tile (Co not a multiple of 32).  One BatchNorm partial row per 128 MI output rows.
int g_conv1b_wide = 1024; // dcn_set_tuning("bwide", min workgroups; 0 = off): 128 x 256 tiles from that many workgroups on
int g_conv1b_tall = 0;    // dcn_set_tuning("btall", min workgroups; 0 = off): 256 x 128 tiles (a wave owns 64 rows x 128 filters)
int conv1b_shape(int M, int Co) {
  if (Co % 32 != 0) return 0;
  if (g_conv1b_wide && Co % 256 == 0 && (long long)cdiv(M, 128) * (Co / 256) >= g_conv1b_wide) return 18;
  if (g_conv1b_tall && Co % 128 == 0 && (long long)cdiv(M, 256) * (Co / 128) >= g_conv1b_tall) return 24;
  if (Co % 128 == 0) return 14;
  if (Co % 64 == 0) return 12;
  return 21;
}

template <int NI, int MI, bool O32, bool F8 = false>
int launch1b(const IgemmParams& p, hipStream_t stream) {
  constexpr int SA = 3, SB = 2;
  constexpr int BM = 128 * MI, BN = 32 * NI;
  static DcnPerDeviceFlag attr_once;
  const size_t lds = (size_t)SA * BM * 64 + (size_t)SB * BN * 64;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1b_kernel<SA, SB, NI, MI, O32, F8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int gm = cdiv(p.M, BM), gn = p.Co / BN;
  const double k_alg = (double)p.ntaps * p.Ci;
  const double alg_bytes = (F8 ? 1.0 : 2.0) * ((double)p.N * p.Hi * p.Wi * p.Ci + (double)p.Co * k_alg) +
                           ((O32 ? 4.0 : 2.0) * (1.0 + (p.accumulate ? 1.0 : 0.0) + (p.residual ? 1.0 : 0.0)) + (p.bt_y ? 2.0 : 0.0)) * (double)p.M * p.Co;
  const int pid = prof_begin(F8 ? 48 : 41, 2.0 * (double)p.M * p.Co * k_alg, stream, alg_bytes);
  hipLaunchKernelGGL((conv1b_kernel<SA, SB, NI, MI, O32, F8>), dim3(gm * gn), dim3(256), lds, stream, p);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("conv1b");
  return DCN_OK;
}

}  // namespace

void conv1_set_tuning(int key, int value) {
  if (key == 6) { g_conv1_tall = value; return; }
  if (key == 0) g_conv1 = value; else if (key == 1) g_conv1_stages = value; else if (key == 3) g_conv1_wide = value;
  else if (key == 4) g_conv1b_wide = value; else if (key == 5) g_conv1b_tall = value; else g_conv1_fill = value;
}

// shape part of the decision.  The kernel only takes launches that igemm.hip would run on its f16-split tiles WITH the pre-split
// bank (b_scale set by conv.hip under igemm_will_presplit): the arithmetic — which products, in which order — is then the same and
// the results are bitwise those of the tile it replaces.  (Taking the 32 / 64-filter 1x1 layers and parity classes from the
// fp32-pipe narrow tiles as well measured -0.2 ms per step; it changes their rounding, and with it which LeakyReLU kinks the
// ill-conditioned N = 4 gradient test sits on.)  Multi-tap launches: stride-2 layers and the parity classes of their data
// gradients; the 3x3 stride-1 layers re-gather every pixel nine times here and belong to the strip kernel / the 256 x 64 tile
// (measured: 32 -> 64 @208 forward 0.745 vs 0.731 ms, data gradient 0.778 vs 0.722).
bool conv1_will_take(long long rows, int Co, int ntaps, int Ci) {
  if (!g_conv1 || rows < 1024 || Co % 64 != 0 || Ci % 16 != 0 || Ci < 32 || ntaps < 1 || ntaps > 16) return false;
  if (g_conv1 == 2 && ntaps != 1) return false;
  return true;
}

// NT launches with a pre-split B bank and an abs-max word for A: 1x1 convolutions and their data gradients (plain GEMM rows), and
// gathered multi-tap launches (stride-2 layers, parity classes of their data gradients, narrow 3x3 layers)
bool conv1_applicable(const IgemmParams& p, int precision, int gran) {
  if (!g_conv1 || precision != 4 || !p.b_scale || !p.amax_a || p.wt16 || p.f8 || p.c4 || p.bmode != 0 || p.batch > 1 || p.row_scale || p.ncls)
    return false;
  if (!conv1_will_take(p.M, p.Co, p.ntaps, p.Ci)) return false;
  if (p.ntaps > 1 && p.isy == 1 && p.osy == 1) return false;            // stride-1 multi-tap: strip kernel / 256 x 64 tile
  if ((long long)p.M != (long long)p.N * p.Hs * p.Ws) return false;
  if (p.Hs * p.Ws < 1 || (256 / (p.Hs * p.Ws) + 2) * (long long)p.Hi * p.Wi * p.ldi * 4 >= 0x7FFFFFF0LL || (long long)p.Co * p.ldw * 4 >= 0x7FFFFFF0LL) return false;
  return conv1_shape(p, gran) != 0;
}

int conv1_launch(const IgemmParams& p, int gran, hipStream_t stream) {
  switch (conv1_shape(p, gran)) {
    case 18: return launch1_ring<8, 1>(p, stream);
    case 14: return launch1_ring<4, 1>(p, stream);
    case 24: return launch1_ring<4, 2>(p, stream);
    case 12: return launch1_ring<2, 1>(p, stream);
    case 22: return launch1_ring<2, 2>(p, stream);
    case 21: return launch1_ring<1, 2>(p, stream);
    default: dcn_set_error("conv1: no tile for this launch"); return DCN_ERR_ARG;
  }
}

// ---- bf16 storage ------------------------------------------------------------------------------------------------------
int conv1b_grid_m(int M, int Co, int ntaps, int s1_w) {
  if (ntaps == 9 && s1_w > 0) { const int bm3 = conv3b_bm(M, Co, s1_w); if (bm3) return cdiv(M, bm3); }      // conv3.hip's strip kernel
  if (conv2b_takes(M, Co, ntaps)) return cdiv(M, 256);              // conv2b.hip's 256 x 256 tile
  const int sh = conv1b_shape(M, Co);
  return sh ? cdiv(M, 128 * (sh / 10)) : 0;
}

// p.in / p.residual / p.bt_y / p.wt point at bf16 data, p.out at bf16 (out_f32 = 0) or fp32 data; no abs-max words, no f8.
int conv1b_launch(const IgemmParams& p, int out_f32, hipStream_t stream) {
  if (p.Ci % 32 != 0 || p.c4 || p.bmode != 0 || p.batch > 1 || p.row_scale || p.ncls || p.ntaps < 1 || p.ntaps > 16 ||
      (long long)p.M != (long long)p.N * p.Hs * p.Ws) {
    dcn_set_error("conv1b: launch form not supported on bf16 storage (Ci=%d taps=%d)", p.Ci, p.ntaps); return DCN_ERR_ARG;
  }
  if (p.Hs * p.Ws < 1 || (256 / (p.Hs * p.Ws) + 2) * (long long)p.Hi * p.Wi * p.ldi * 2 >= 0x7FFFFFF0LL || (long long)p.Co * p.ldw * 2 >= 0x7FFFFFF0LL) {
    dcn_set_error("conv1b: tensor beyond the 2 GiB buffer window"); return DCN_ERR_ARG;
  }
  if (conv3b_takes(p)) return conv3b_launch(p, out_f32, stream);                          // 3x3 stride 1: the strip kernel (every row staged once, not nine times)
  // conv1b_grid_m() sized the caller's BatchNorm partial rows from conv3b_bm() alone: a stride-1 3x3 launch that conv3b_bm accepts and
  // conv3b_takes then rejects (a tensor beyond the 2 GiB window) must not fall through to a 128-row tile writing cdiv(M, 128) rows into a
  // buffer sized for cdiv(M, 256) (round-5 advice; reachable only with dcn_set_tuning("3h16", 1))
  if (p.stats && p.ntaps == 9 && p.isy == 1 && p.isx == 1 && p.osy == 1 && p.osx == 1 && p.dense_out && conv3b_bm(p.M, p.Co, p.Wi) != 0) {
    dcn_set_error("conv1b: the strip kernel sized this launch's BatchNorm partials but cannot take it (tensor beyond its 2 GiB window)");
    return DCN_ERR_ARG;
  }
  if (conv2b_takes(p.M, p.Co, p.ntaps)) return conv2b_launch(p, out_f32, stream);      // >= 256 filters on a long grid: 256 x 256 tiles, eight waves
  switch (conv1b_shape(p.M, p.Co)) {
    case 18: return out_f32 ? launch1b<8, 1, true>(p, stream) : launch1b<8, 1, false>(p, stream);
    case 14: return out_f32 ? launch1b<4, 1, true>(p, stream) : launch1b<4, 1, false>(p, stream);
    case 24: return out_f32 ? launch1b<4, 2, true>(p, stream) : launch1b<4, 2, false>(p, stream);
    case 12: return out_f32 ? launch1b<2, 1, true>(p, stream) : launch1b<2, 1, false>(p, stream);
    case 21: return out_f32 ? launch1b<1, 2, true>(p, stream) : launch1b<1, 2, false>(p, stream);
    default: dcn_set_error("conv1b: Co=%d is not a multiple of 32", p.Co); return DCN_ERR_ARG;
  }
}

// ---- fp8 storage ---------------------------------------------------------------------------------------------------------
// p.in / p.wt point at e4m3 bytes (strides in elements = bytes), p.a_scale8 / p.b_scale8 at the e8m0 row scales, p.residual / p.bt_y at
// bf16 data, p.out at bf16 (out_f32 = 0) or fp32 data.  One BatchNorm partial row per 128 MI output rows (conv1q_grid_m).
int conv1q_grid_m(int M, int Co) {
  const int sh = conv1b_shape(M, Co);
  return sh ? cdiv(M, 128 * (sh / 10)) : 0;
}

int conv1q_launch(const IgemmParams& p, int out_f32, hipStream_t stream) {
  if (p.Ci % 64 != 0 || !p.a_scale8 || !p.b_scale8 || p.c4 || p.bmode != 0 || p.batch > 1 || p.row_scale || p.ncls || p.ntaps < 1 || p.ntaps > 12 ||
      (long long)p.M != (long long)p.N * p.Hs * p.Ws) {
    dcn_set_error("conv1q: launch form not supported on fp8 storage (Ci=%d taps=%d)", p.Ci, p.ntaps); return DCN_ERR_ARG;
  }
  if (p.Hs * p.Ws < 1 || (256 / (p.Hs * p.Ws) + 2) * (long long)p.Hi * p.Wi * p.ldi >= 0x7FFFFFF0LL || (long long)p.Co * p.ldw >= 0x7FFFFFF0LL) {
    dcn_set_error("conv1q: tensor beyond the 2 GiB buffer window"); return DCN_ERR_ARG;
  }
  switch (conv1b_shape(p.M, p.Co)) {
    case 18: return out_f32 ? launch1b<8, 1, true, true>(p, stream) : launch1b<8, 1, false, true>(p, stream);
    case 14: return out_f32 ? launch1b<4, 1, true, true>(p, stream) : launch1b<4, 1, false, true>(p, stream);
    case 24: return out_f32 ? launch1b<4, 2, true, true>(p, stream) : launch1b<4, 2, false, true>(p, stream);
    case 12: return out_f32 ? launch1b<2, 1, true, true>(p, stream) : launch1b<2, 1, false, true>(p, stream);
    case 21: return out_f32 ? launch1b<1, 2, true, true>(p, stream) : launch1b<1, 2, false, true>(p, stream);
    default: dcn_set_error("conv1q: Co=%d is not a multiple of 32", p.Co); return DCN_ERR_ARG;
  }
}
