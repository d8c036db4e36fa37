// 1x1 convolution / plain NT GEMM (forward and data gradient of every 1x1 layer) with both operand tiles brought in by LDS-DMA.
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

constexpr int C1_BM = 128, C1_BN = 128;
constexpr int C1_ASTAGE = C1_BM * 64;            // [128 rows][16 k fp32]: 64-B rows, 16-B chunks XOR-swizzled by (row >> 2) & 3
constexpr int C1_BPLANE = C1_BN * 32;            // [128 filters][16 k f16]: 32-B rows, halves swapped in rows 8-15 (mod 16)
constexpr unsigned C1_OOB = 0x80000000u;

__device__ __forceinline__ float c1_pow2_scale(unsigned amax_bits) {      // = igemm.hip pow2_scale
  const int be = (int)((amax_bits >> 23) & 0xFF);
  if (be == 0 || be == 255) return 1.f;
  int e = 14 - (be - 126);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return __uint_as_float((unsigned)(e + 127) << 23);
}

// SA / SB: ring depths (K-steps) of the activation / filter tiles.  The loads are WAVE-SPECIALISED — waves 0-1 bring in the eight
// 1-KiB pieces of an A tile, waves 2-3 those of a B tile — because `s_waitcnt vmcnt` retires a wave's loads in issue order: a
// wave that loaded both kinds could keep only as many A tiles in flight as B tiles.  A comes from HBM (deep ring: SA - 1 tiles
// of 8 KB in flight per workgroup), B from L2 (the bank of this filter tile is shared by every M-tile: a shallow ring).
template <int SA, int SB>
__global__ __launch_bounds__(256, 2) void conv1_kernel(const IgemmParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem1[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gn = p.Co / C1_BN;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int bm = lin / gn, bn = lin - bm * gn;
  const int M = p.M, m0 = bm * C1_BM;
  const int kiters = p.Ci >> 4;

  const float sa = c1_pow2_scale(amax_read(p.amax_a)), sb = p.b_scale[0];

  // ---- descriptors ------------------------------------------------------------------------------------------------
  const float* a_base = p.in + (long long)m0 * p.ldi;
  const long long a_bytes = ((long long)(M - m0 - 1) * p.ldi + p.Ci) * 4;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, a_bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0 : (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, (int)((long long)p.Co * p.ldw * 4), 0x00020000);

  // ---- per-lane source offsets of this wave's four 1-KiB LDS-DMA pieces per K-step (waves 0-1: A, waves 2-3: B) --------------
  // A piece j (16 rows): LDS position (row, c') = (16 j + lane / 4, lane % 4) holds the row's chunk c = c' ^ ((row >> 2) & 3).
  // Rows past M are clamped to the last row (finite values; the epilogue masks them).
  // B piece j: plane = j >> 2, 32 filters: LDS position (row, halfpos) = (32 (j & 3) + lane / 2, lane & 1) holds k-half
  // kh = halfpos ^ ((row >> 3) & 1) of plane `plane`: bank chunk 2 kh + plane of the K-step's 64 bytes.
  const bool loads_a = wave < 2;
  unsigned voff[4];
  int dst[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = 4 * (wave & 1) + e;
    if (loads_a) {
      const int row = 16 * j + (lane >> 2), cp = lane & 3, c = cp ^ ((row >> 2) & 3);
      int m = m0 + row; m = m < M ? m : M - 1;
      voff[e] = (unsigned)((m - m0) * p.ldi * 4 + c * 16);
      dst[e] = j * 1024;
    } else {
      const int plane = j >> 2, rr = 32 * (j & 3) + (lane >> 1), kh = (lane & 1) ^ ((rr >> 3) & 1);
      voff[e] = (unsigned)((bn * C1_BN + rr) * p.ldw * 4 + (2 * kh + plane) * 16);
      dst[e] = SA * C1_ASTAGE + plane * C1_BPLANE + (j & 3) * 1024;
    }
  }
  const unsigned b_k0 = (unsigned)p.tap_w[0] * 4;
  const __amdgpu_buffer_rsrc_t my_rs = loads_a ? a_rs : b_rs;

  auto issue = [&](int it) {               // this wave's four pieces of K-step `it` (past the end: no-ops that still count in vmcnt)
    const int kt = loads_a ? it + SA - 1 : it + SB - 1;          // the K-step whose ring slot the previous iteration freed
    const bool live = kt < kiters;
    const unsigned soff = live ? (unsigned)kt * 64u + (loads_a ? 0u : b_k0) : 0u;
    unsigned char* st = smem1 + (loads_a ? (kt % SA) * C1_ASTAGE : (kt % SB) * 2 * C1_BPLANE);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(my_rs, (lds_void*)(st + dst[e]), 16, live ? voff[e] : C1_OOB, soff, 0, 0);
  };

  // ---- fragment addresses ----------------------------------------------------------------------------------------------
  const int ar = wave * 32 + (lane & 31), kh = lane >> 5;
  const int a_rd0 = ar * 64 + (((2 * kh) ^ ((ar >> 2) & 3)) << 4);
  const int a_rd1 = ar * 64 + (((2 * kh + 1) ^ ((ar >> 2) & 3)) << 4);
  int b_rd[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int row = ni * 32 + (lane & 31);
    b_rd[ni] = SA * C1_ASTAGE + row * 32 + (((kh ^ (row >> 3)) & 1) << 4);
  }

  f32x16 acc[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

  // ---- prologue: SA - 1 activation tiles / SB - 1 filter tiles in flight --------------------------------------------------
  for (int s = -(loads_a ? SA : SB) + 1; s < 0; ++s) issue(s);

  for (int it = 0; it < kiters; ++it) {
    // this wave's pieces of K-step `it` have landed (all but the younger K-steps' 4 loads each), then everybody's
    if (loads_a) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((SA - 2) * 4) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((SB - 2) * 4) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // the slots read during the previous iteration are free (every wave has passed the barrier): refill them
    issue(it);
    const unsigned char* st = smem1 + (it % SA) * C1_ASTAGE;
    const unsigned char* sb_ = smem1 + (it % SB) * 2 * C1_BPLANE;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(st + a_rd0), x1 = *reinterpret_cast<const f32x4*>(st + a_rd1);
    f16x8_t bh[4], bl[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      bh[ni] = *reinterpret_cast<const f16x8_t*>(sb_ + b_rd[ni]);
      bl[ni] = *reinterpret_cast<const f16x8_t*>(sb_ + C1_BPLANE + b_rd[ni]);
    }
    // x * s = h + l, two f16 (round to nearest): 11 + 11 significant bits
    const f32x4 t0 = x0 * sa, t1 = x1 * sa;
    f16x8_t ah, al;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ah[e] = (_Float16)t0[e]; ah[4 + e] = (_Float16)t1[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      al[e] = (_Float16)(t0[e] - (float)ah[e]); al[4 + e] = (_Float16)(t1[e] - (float)ah[4 + e]);
    }
    // smallest terms first: (l,h) (h,l) (h,h)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ni], acc[ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ni], acc[ni], 0, 0, 0);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ni], acc[ni], 0, 0, 0);
  }
  // the no-op pieces issued past the end may still be pending LDS writes: drain before LDS is reused / the workgroup ends
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float dq = 1.f / (sa * sb);           // powers of two: exact
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) acc[ni] *= dq;

  // ---- epilogue (igemm.hip's; output pixel = row m; rows >= M are duplicates of row M-1 and masked) ------------------------
  float* __restrict__ gout = p.out;
  const int rbase = m0 + wave * 32 + 4 * kh;         // row of register r: rbase + (r & 3) + 8 * (r >> 2)
  if (p.accumulate) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = rbase + (r & 3) + 8 * (r >> 2);
      if (m >= M) continue;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[ni][r] += gout[(size_t)m * p.ldo + bn * C1_BN + ni * 32 + (lane & 31)];
    }
  }
  if (p.stats) {
    float* red = reinterpret_cast<float*>(smem1);      // [2][4 waves][128]
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = rbase + (r & 3) + 8 * (r >> 2);
        const float v = m < M ? acc[ni][r] : 0.f;
        s += v; ss += v * v;
      }
      s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
      if (lane < 32) {
        red[(0 * 4 + wave) * C1_BN + ni * 32 + lane] = s;
        red[(1 * 4 + wave) * C1_BN + ni * 32 + lane] = ss;
      }
    }
    __syncthreads();
    {
      const int which = tid >> 7, col = tid & 127;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t += red[(which * 4 + w) * C1_BN + col];
      p.stats[((size_t)bm * 2 + which) * p.Co + bn * C1_BN + col] = t;
    }
  }
  float sc[4], sh[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int co = bn * C1_BN + ni * 32 + (lane & 31);
    sc[ni] = p.scale ? p.scale[co] : 1.f;
    sh[ni] = p.shift ? p.shift[co] : 0.f;
  }
  float vmax = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = rbase + (r & 3) + 8 * (r >> 2);
    if (m >= M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int co = bn * C1_BN + ni * 32 + (lane & 31);
      float v = acc[ni][r] * sc[ni] + sh[ni];
      if (p.act == DCN_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
      if (p.residual) v += p.residual[(size_t)m * p.ldr + co];
      gout[(size_t)m * p.ldo + co] = v;
      vmax = fmaxf(vmax, fabsf(v));
    }
  }
  if (p.amax_out) {
    vmax = wave_max(vmax);
    if (lane == 0) amax_update(p.amax_out, vmax, blockIdx.x * 4 + wave);
  }
}

int g_conv1 = 1;          // dcn_set_tuning("1x1dma", 0): 1x1 layers back on the implicit-GEMM tile
int g_conv1_stages = 32;  // dcn_set_tuning("1stages", 10 * SA + SB): ring depths.  Default 3 activation + 2 filter K-steps = 40 KB: FOUR workgroups per CU (measured per layer, tools/bench_convs.py --set 1stages=..: 32 < 33 < 42 < 44 ~ 63 << 84: occupancy beats ring depth)

template <int SA, int SB>
int launch1(const IgemmParams& p, hipStream_t stream) {
  static bool attr_done = false;
  const size_t lds = (size_t)SA * C1_ASTAGE + (size_t)SB * 2 * C1_BPLANE;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1_kernel<SA, SB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const int gm = cdiv(p.M, C1_BM), gn = p.Co / C1_BN;
  const double alg_bytes = 4.0 * ((double)p.M * p.Ci + (double)p.Co * p.Ci + (double)p.M * p.Co);
  const int pid = prof_begin(35, 2.0 * (double)p.M * p.Co * p.Ci, stream, alg_bytes);
  hipLaunchKernelGGL((conv1_kernel<SA, SB>), dim3(gm * gn), dim3(256), lds, stream, p);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("conv1");
  return DCN_OK;
}

}  // namespace

void conv1_set_tuning(int key, int value) { if (key == 0) g_conv1 = value; else g_conv1_stages = value; }

// plain NT GEMM rows with a pre-split B bank and an abs-max word for A: the 1x1 convolutions and their data gradients
bool conv1_applicable(const IgemmParams& p, int precision) {
  if (!g_conv1 || precision != 4 || !p.b_scale || !p.amax_a || p.wt16 || p.f8 || p.c4 || p.bmode != 0 || p.batch > 1 || p.row_scale || p.ncls)
    return false;
  if (p.ntaps != 1 || p.tap_dy[0] != 0 || p.tap_dx[0] != 0 || p.isy != 1 || p.isx != 1 || !p.dense_out) return false;
  if (p.Hs != p.Hi || p.Ws != p.Wi || (long long)p.M != (long long)p.N * p.Hi * p.Wi) return false;
  if (p.Co % C1_BN != 0 || p.Ci % 16 != 0 || p.Ci < 32 || p.M < 1024) return false;
  if ((long long)C1_BM * p.ldi * 4 >= 0x7FFFFFF0LL || (long long)p.Co * p.ldw * 4 >= 0x7FFFFFF0LL) return false;
  return true;
}

int conv1_launch(const IgemmParams& p, hipStream_t stream) {
  switch (g_conv1_stages) {
    case 33: return launch1<3, 3>(p, stream);
    case 32: return launch1<3, 2>(p, stream);
    case 42: return launch1<4, 2>(p, stream);
    case 44: return launch1<4, 4>(p, stream);
    case 43: return launch1<4, 3>(p, stream);
    case 53: return launch1<5, 3>(p, stream);
    case 84: return launch1<8, 4>(p, stream);
    case 64: return launch1<6, 4>(p, stream);
    case 63: return launch1<6, 3>(p, stream);
    default: return launch1<3, 2>(p, stream);
  }
}
