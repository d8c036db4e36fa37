// 3x3 stride-1 convolution (forward and data gradient) with the activation strip resident in LDS.
//
// The implicit-GEMM tile of igemm.hip stages a fresh [BM pixels][16 channels] activation tile per tap: every activation is
// loaded from L2, split into f16 pieces and written to LDS nine times, and a workgroup meets at a barrier every 12 MFMAs.
// Measured on the 256->512 layer of the 52x52 map (tools/bench_convs.py --ab abl=N, timing ablations): without the loop's
// global loads -31 %, without the split + LDS stores -24 %, with one of the three MFMA terms -46 % — a pipeline in which the
// operand staging weighs as much as the matrix pipe.  Here the staging is done ONCE per 16 channels:
//
//   * an M-tile is BM consecutive pixels of the flattened (image, row, column) space; the 9 taps of those pixels read the
//     pixels [m0 - W - 1, m0 + BM + W + 1): one contiguous strip of S = BM + 2W + 2 pixels, (1.1 - 1.8) x BM on the 13..104
//     wide maps instead of 9 x BM.  The strip's 16 channels are split into two f16 planes (x*s = h + l, igemm.hip) and kept
//     in LDS for all nine taps; the MFMA A-fragment of tap (dy, dx) is the same ds_read_b128 at strip position
//     row + (dy+1)*W + (dx+1).  Taps that fall outside the image (left/right/top/bottom edge; the strip holds the wrapped
//     neighbour there) redirect the lane's read to a row of zeros: one v_cndmask per fragment, no branch.
//   * the filter bank arrives pre-split (dcn_presplit_f16): its [BN filters][16 k] tile per tap is a 16-B copy.  Three taps
//     (one filter row) are staged per barrier: 36 MFMAs per wave between barriers instead of 12.
//   * 32-byte LDS rows whose two 16-B halves swap places in positions 8-15 (mod 16): conflict-free ds_read_b128 for ANY
//     strip shift (a 16-lane group reads positions c + {0-3, 12-15, 20-27}: equal 2*(pos mod 8) only 8 or 24 positions
//     apart, where the swap bit differs) and for the ds_write_b64 of the split pieces.
//
// Roofline: MFMA, 838.9 TFLOP/s (three v_mfma_f32_32x32x16_f16 per product).  Vector ALU per MFMA: ~1 (was 4.5-8.5).
// Epilogue = igemm.hip's (BatchNorm statistics partials, scale/shift, LeakyReLU, shortcut add, accumulate, abs-max).
#include "igemm.h"
#include "prof.h"
#include <type_traits>

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned OOB3 = 0x80000000u;

__device__ __forceinline__ f32x4 ld16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc3(const float* base, long long bytes) {
  const unsigned n = bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0u : (bytes < 0 ? 0u : (unsigned)bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, n, 0x00020000);
}
__device__ __forceinline__ float pow2_scale3(unsigned amax_bits) {      // = igemm.hip pow2_scale
  const int be = (int)((amax_bits >> 23) & 0xFF);
  if (be == 0 || be == 255) return 1.f;
  int e = 14 - (be - 126);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return __uint_as_float((unsigned)(e + 127) << 23);
}
// byte offset of the 16-B half `half` of strip position / filter row `pos` inside a plane of 32-B rows
__device__ __forceinline__ int row32(int pos, int half) { return pos * 32 + (((half ^ (pos >> 3)) & 1) << 4); }

// WM x WN waves, each a 64 x 64 sub-tile (2 x 2 accumulators of 32x32): BM = 64*WM pixels, BN = 64*WN filters.
// A_LD: 16-B strip loads per thread and 16 channels (>= ceil(4*S / threads); excess loads are predicated off).
// NP = 2: fp32-accurate f16 two-piece split (pre-split filter bank, three MFMAs per product).  NP = 1: bf16 operands, one plane,
// one MFMA per product — the bf16-operand mode (BASELINE.json configs[2]); the filter bank arrives converted to bf16
// (IgemmParams::wt16, dcn_prepare_filters), activations are rounded when the strip is staged.
// ABL (builds with -DC3_ABL=1 only, dcn_set_tuning("3abl", bits); results are WRONG, timing experiments): 1 = no filter loads in the loop,
// 2 = no strip loads in the loop, 4 = no filter / strip LDS stores in the loop, 8 = no epilogue, 16 = one of the three MFMA terms
#ifndef C3_ABL
#define C3_ABL 0
#endif
// LS ("load spread"): the global loads of an iteration are issued one by one BEHIND its MFMA groups instead of all at its start — every wave
// leaves the barrier at the same time, and a burst of 7-10 loads per thread from all of them fills the memory path's queue: a load then sits in
// the issue stage with no MFMA behind it (gemm3.hip measured the same for its LDS-DMA: 0.92 -> 0.70 ms with the DMA behind the fragment reads)
// IN16 (NP = 1, bf16 STORAGE — BASELINE.json configs[2]): the gathered tensor, the shortcut, the accumulated-onto tensor and the tapped
// BatchNorm input ARE bf16 tensors (strides in elements), the bank is the bf16 bank; a 16-byte strip piece is 8 channels and goes to its
// half of the position's 32-byte row as loaded (no conversion, two pieces per position and channel step instead of four); the result is
// stored as bf16 (O32: fp32), and the BatchNorm partial sums are those of the values as stored (conv1.hip conv1b_kernel's epilogue).
template <int WM, int WN, int A_LD, int NP = 2, int ABL = 0, int LS = 0, bool IN16 = false, bool O32 = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv3_kernel(const IgemmParams p, const int S, const int gran) {
  static_assert(!IN16 || NP == 1, "bf16 storage: one plane");
  constexpr int NT = 64 * WM * WN, BM = 64 * WM, BN = 64 * WN;
  constexpr int CHB = 2 * NP;                 // 16-B chunks per filter row and tap (16 k: 64 B pre-split, 32 B bf16)
  constexpr int B_LD = (BN * CHB + NT - 1) / NT;          // 16-B filter loads per thread and tap (the last may be predicated off)
  static_assert(A_LD % 2 == 0, "strip pieces are split in two halves (taps 0 and 1)");
  constexpr int PB = BN * 32;                 // bytes per filter plane of one tap
  // plane 1 sits 64 B past a multiple of 128: a ds_write_b128 wave-instruction stores the (plane 0, plane 1) chunks of the same
  // row side by side in lane order, and with planes a multiple of 128 B apart they fell on the same banks (PMC:
  // SQ_LDS_BANK_CONFLICT 23 % of SQ_LDS_IDX_ACTIVE)
  constexpr int PB1 = PB + 64, SLOT = NP * PB + 128;     // offset of plane 1 inside a tap slot / bytes per tap slot (last 16 B: dump)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  const int PA = (S + 2) * 32;                // bytes per strip plane (+ the zero row at position S, + a dump row for predicated-off stores)
  unsigned char* const Abase = smem3;                     // [2 buffers][NP planes][PA]
  unsigned char* const Bbase = smem3 + 2 * NP * PA;       // [2 buffers][3 taps][SLOT]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, half = lane >> 5;
  const int gn = (p.Co + BN - 1) / BN;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int bm = lin / gn, bn = lin - bm * gn;
  const int W = p.Wi, H = p.Hi, M = p.M;
  const int m0 = bm * BM;
  float sa = 1.f, sb = 1.f;
  if constexpr (NP == 2) { sa = pow2_scale3(amax_read(p.amax_a)); sb = p.b_scale[0]; }

  // ---- descriptors ---------------------------------------------------------------------------------
  const int lin0 = m0 - W - 1;                                     // pixel of strip position 0
  const int base_px = lin0 > 0 ? lin0 : 0;
  constexpr int AESZ = IN16 ? 2 : 4;           // bytes per element of the gathered tensor
  constexpr int PPP = IN16 ? 2 : 4;            // 16-byte pieces per strip position and 16-channel step
  const float* a_base = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.in) + (long long)base_px * p.ldi * AESZ);
  const __amdgpu_buffer_rsrc_t a_rs = rsrc3(a_base, ((long long)(M - base_px - 1) * p.ldi + p.Ci) * AESZ);
  const __amdgpu_buffer_rsrc_t b_rs = NP == 2 ? rsrc3(p.wt, (long long)p.Co * p.ldw * 4)
                                              : rsrc3(reinterpret_cast<const float*>(p.wt16), (long long)p.Co * p.ldw * 2);
  constexpr int ESZ = NP == 2 ? 4 : 2;         // bytes per filter element in the bank

  // ---- per-thread staging state --------------------------------------------------------------------
  unsigned a_off[A_LD]; int a_st[A_LD];      // global byte offset (channel step 0) / LDS byte offset inside plane 0
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    const int idx = tid + j * NT, pos = idx / PPP, q = idx % PPP;
    const int px = lin0 + pos;
    a_off[j] = (pos < S && px >= 0 && px < M) ? (unsigned)((px - base_px) * p.ldi * AESZ + q * 16) : OOB3;
    if constexpr (IN16) a_st[j] = pos < S ? row32(pos, q) : (S + 1) * 32 + q * 16;
    else a_st[j] = pos < S ? row32(pos, q >> 1) + (q & 1) * 8 : (S + 1) * 32 + q * 8;      // (no branch in the loop: beyond the strip -> dump row)
  }
  unsigned b_off[B_LD]; int b_st[B_LD];
#pragma unroll
  for (int l = 0; l < B_LD; ++l) {
    const int idx = tid + l * NT, row = idx / CHB, chunk = idx % CHB;
    const int co = bn * BN + row;
    const bool on = idx < BN * CHB;
    b_off[l] = (on && co < p.Co) ? (unsigned)(co * p.ldw * ESZ + chunk * 16) : OOB3;
    // pre-split: plane = chunk & 1, k-half = chunk >> 1; bf16: one plane, k-half = chunk; threads without a chunk: dump slot
    b_st[l] = !on ? SLOT - 16 : (NP == 2 ? (chunk & 1) * PB1 + row32(row, chunk >> 1) : row32(row, chunk));
  }
  // rows of this lane (one per 32x32 block along M): in-image tap masks
  unsigned msk[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int m = m0 + wm * 64 + mi * 32 + (lane & 31);
    unsigned v = 0;
    if (m < M) {
      const int rem = m % (H * W), y = rem / W, x = rem - y * W;
      for (int t = 0; t < 9; ++t)
        if ((unsigned)(y + p.tap_dy[t]) < (unsigned)H && (unsigned)(x + p.tap_dx[t]) < (unsigned)W) v |= 1u << t;
    }
    msk[mi] = v;
  }
  const int a_row0 = wm * 64 + (lane & 31);
  int b_fr[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) b_fr[ni] = row32(wn * 64 + ni * 32 + (lane & 31), half);

  const int nch = p.Ci >> 4;                  // 16-channel steps (even: Ci % 32 == 0)
  const int iters = 3 * nch;                  // one iteration = one filter row (3 taps) of one channel step

  f32x4 a_reg[A_LD], b_r0[3 * B_LD], b_r1[3 * B_LD];
  auto load_a = [&](int cc) {
#pragma unroll
    for (int j = 0; j < A_LD; ++j) a_reg[j] = ld16(a_rs, a_off[j], (unsigned)cc * (16u * AESZ));
  };
  auto load_b = [&](f32x4* br, int it) {      // the three taps of iteration `it`
    const int cc = it / 3, g = it - 3 * cc;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const unsigned soff = (unsigned)(p.tap_w[3 * g + j] + cc * 16) * (unsigned)ESZ;
#pragma unroll
      for (int l = 0; l < B_LD; ++l) br[j * B_LD + l] = ld16(b_rs, b_off[l], soff);
    }
  };
  auto store_a_piece = [&](unsigned char* abuf, int j) {            // x*s = h + l, two f16 planes (NP = 1: one bf16 plane)
    if constexpr (IN16) { *reinterpret_cast<f32x4*>(abuf + a_st[j]) = a_reg[j]; return; }      // 8 bf16 channels as loaded
    if constexpr (NP == 1) {
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      const bf16x4_t b = {(__bf16)a_reg[j][0], (__bf16)a_reg[j][1], (__bf16)a_reg[j][2], (__bf16)a_reg[j][3]};
      *reinterpret_cast<uint2*>(abuf + a_st[j]) = __builtin_bit_cast(uint2, b);
      return;
    }
    const f32x4 t = a_reg[j] * sa;
    const f16x4_t h = {(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
    const f16x4_t l = {(_Float16)(t[0] - (float)h[0]), (_Float16)(t[1] - (float)h[1]), (_Float16)(t[2] - (float)h[2]),
                       (_Float16)(t[3] - (float)h[3])};
    *reinterpret_cast<uint2*>(abuf + a_st[j]) = __builtin_bit_cast(uint2, h);
    *reinterpret_cast<uint2*>(abuf + PA + a_st[j]) = __builtin_bit_cast(uint2, l);
  };
  auto store_b_tap = [&](unsigned char* bbuf, const f32x4* br, int j) {
#pragma unroll
    for (int l = 0; l < B_LD; ++l) *reinterpret_cast<f32x4*>(bbuf + j * SLOT + b_st[l]) = br[j * B_LD + l];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // ---- prologue --------------------------------------------------------------------------------------
  if (tid < 4 * NP) {                         // the zero rows: [buffer][plane] x two 16-B halves
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(Abase + (tid >> 1) * PA + S * 32 + (tid & 1) * 16) = z;
  }
  load_a(0);
  load_b(b_r1, 0);
#pragma unroll
  for (int j = 0; j < A_LD; ++j) store_a_piece(Abase, j);
#pragma unroll
  for (int j = 0; j < 3; ++j) store_b_tap(Bbase, b_r1, j);
  load_b(b_r1, 1);
  __syncthreads();

  // ---- main loop ---------------------------------------------------------------------------------------
  // iteration it = 3*cc + G (channel step cc, filter row G): MFMAs on strip buffer cc & 1 and filter buffer it & 1; the filter
  // tile of it + 1 (in `bold`, loaded one iteration ago) goes to the other filter buffer in the MFMAs' shadow; the loads of
  // it + 2 are issued first, into `bnew`.  The strip of cc + 1 is loaded at G == 0 and split into the other strip buffer at
  // G == 2.  The loop is unrolled over two channel steps x three filter rows, so G, both buffer parities and the register
  // stage are compile-time constants: tap shifts and filter offsets stay in scalar registers, every LDS address is
  // base + constant.
  f16x8_t af[2][2][2], bf[2][2][2];          // [pipeline stage][block][plane]
  // fragments of tap J of the iteration with constants (G, CP, IP) into stage ST
  auto read_frags = [&](const int G, const int CP, const int IP, const int J, const int ST) {
    const unsigned char* ab = Abase + CP * NP * PA;
    const unsigned char* bb = Bbase + IP * 3 * SLOT;
    const int t = 3 * G + J;
    const int sh = (p.tap_dy[t] + 1) * W + p.tap_dx[t] + 1;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int pos = a_row0 + mi * 32 + sh;
      const int off = ((msk[mi] >> t) & 1) ? row32(pos, half) : S * 32;
      af[ST][mi][0] = *reinterpret_cast<const f16x8_t*>(ab + off);
      if constexpr (NP == 2) af[ST][mi][1] = *reinterpret_cast<const f16x8_t*>(ab + PA + off);
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      bf[ST][ni][0] = *reinterpret_cast<const f16x8_t*>(bb + J * SLOT + b_fr[ni]);
      if constexpr (NP == 2) bf[ST][ni][1] = *reinterpret_cast<const f16x8_t*>(bb + J * SLOT + PB1 + b_fr[ni]);
    }
  };
  // one of the three cross terms, smallest first: (l,h) (h,l) (h,h)
  auto mfma_term = [&](const int ST, const int term) {
    if constexpr (NP == 1) {                  // bf16 operands: the one product, in the slot of the (h,h) term
      typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
      if (term != 2) return;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[ST][mi][0]),
                                                                __builtin_bit_cast(bf16x8_t, bf[ST][ni][0]), acc[mi][ni], 0, 0, 0);
      return;
    }
    const int qa = term == 0 ? 1 : 0, qb = term == 1 ? 1 : 0;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ST][mi][qa], bf[ST][ni][qb], acc[mi][ni], 0, 0, 0);
  };
  // Issue order of one iteration (fixed with scheduling barriers; the fragments of its tap 0 were read during the previous
  // iteration's tap 2):  global loads of it + 2 | tap 0: reads of tap 1, 4 MFMAs, filter stores, 4 MFMAs, strip split, 4 MFMAs |
  // tap 1: the same with the reads of tap 2 | barrier (every store of this iteration is done) | reads of the NEXT iteration's
  // tap 0 | tap 2: 12 MFMAs.  An LDS read is always issued at least 8 MFMAs (256 cycles of the matrix pipe) before its use.
  auto body = [&](auto G_, auto CP_, const int cc) {
    constexpr int G = decltype(G_)::value, CP = decltype(CP_)::value;
    constexpr int IP = (CP + G) & 1;          // it & 1
    constexpr int G2 = (G + 2) % 3, C2 = (G + 2) / 3;      // filter row / channel-step advance of iteration it + 2
    constexpr int GN = (G + 1) % 3, CPN = G == 2 ? (CP ^ 1) : CP;      // the next iteration's constants
    const int it = 3 * cc + G;
    f32x4* bnew = IP ? b_r1 : b_r0;
    const f32x4* bold = IP ? b_r0 : b_r1;
    // Every load and store below is unconditional: a load inside a (wave-uniform) branch makes the s_waitcnt in front of the
    // older stage's stores conservative (it then also waits for the loads just issued).  Past the end of the K loop the loads
    // take the out-of-range offset (they return zero without touching memory) and the stores fill a buffer nobody reads.
    const bool in_b = (ABL & 1) ? false : it + 2 < iters;
    const bool in_a = (ABL & 2) ? false : cc + 1 < nch;
    // load q of this iteration: the 3 B_LD filter pieces of iteration it + 2 first (stored one iteration from now), then — G == 0 — the
    // A_LD strip pieces of channel step cc + 1 (stored at G == 2)
    constexpr int NLOAD = 3 * B_LD + (G == 0 ? A_LD : 0);
    auto load_q = [&](const int q) {
      if (q < 3 * B_LD) {
        const int j = q / B_LD, l = q - j * B_LD;
        const unsigned soff = (unsigned)(p.tap_w[3 * G2 + j] + (cc + C2) * 16) * (unsigned)ESZ;
        bnew[q] = ld16(b_rs, in_b ? b_off[l] : OOB3, in_b ? soff : 0u);
      } else if (q < NLOAD) {
        a_reg[q - 3 * B_LD] = ld16(a_rs, in_a ? a_off[q - 3 * B_LD] : OOB3, in_a ? (unsigned)(cc + 1) * (16u * AESZ) : 0u);
      }
    };
    // LS: slot s of the iteration's nine MFMA groups takes loads [lo(s), lo(s + 1)): the filter pieces behind the first groups (they are
    // stored early in the next iteration), the strip pieces behind the later ones
    constexpr int BSL = 3 * B_LD <= 4 ? 3 * B_LD : 4;             // slots the filter loads spread over
    auto slot_loads = [&](const int sl) {
      if constexpr (LS == 0) return;
#pragma unroll
      for (int q = 0; q < NLOAD; ++q) {
        const int home = q < 3 * B_LD ? q * BSL / (3 * B_LD) : BSL + (q - 3 * B_LD) * (9 - BSL) / (A_LD > 0 ? A_LD : 1);
        if (home == sl) load_q(q);
      }
    };
    if constexpr (LS == 0) {
#pragma unroll
      for (int q = 0; q < NLOAD; ++q) load_q(q);
    }
    unsigned char* an = Abase + (CP ^ 1) * NP * PA;
    unsigned char* bnx = Bbase + (IP ^ 1) * 3 * SLOT;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int st = (IP + j) & 1;
      read_frags(G, CP, IP, j + 1, st ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(ABL & 16)) mfma_term(st, 0);
      slot_loads(3 * j);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(ABL & 4)) { if (j == 0) { store_b_tap(bnx, bold, 0); store_b_tap(bnx, bold, 1); } else store_b_tap(bnx, bold, 2); }
      if constexpr (!(ABL & 16)) mfma_term(st, 1);
      slot_loads(3 * j + 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (G == 2 && !(ABL & 4)) {
#pragma unroll
        for (int q = j * (A_LD / 2); q < (j + 1) * (A_LD / 2); ++q) store_a_piece(an, q);
      }
      mfma_term(st, 2);
      slot_loads(3 * j + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    read_frags(GN, CPN, IP ^ 1, 0, IP ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 16)) {
      mfma_term(IP, 0); slot_loads(6); __builtin_amdgcn_sched_barrier(0);
      mfma_term(IP, 1); slot_loads(7); __builtin_amdgcn_sched_barrier(0);
    }
    mfma_term(IP, 2); slot_loads(8);
    __builtin_amdgcn_sched_barrier(0);
  };
  read_frags(0, 0, 0, 0, 0);
  {
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    for (int cc = 0; cc < nch; cc += 2) {
      body(I0{}, I0{}, cc); body(I1{}, I0{}, cc); body(I2{}, I0{}, cc);
      body(I0{}, I1{}, cc + 1); body(I1{}, I1{}, cc + 1); body(I2{}, I1{}, cc + 1);
    }
  }
  __syncthreads();                            // (the statistics reduction below reuses LDS)
  if constexpr ((ABL & 8) != 0) {             // (timing only: one store per lane keeps the accumulators alive)
    float t_ = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) t_ += acc[mi][ni][r];
    if (t_ == 12345.678f) p.out[tid] = t_;
    return;
  }

  if constexpr (IN16) {
    // ---- epilogue on bf16 tensors (conv1.hip conv1b_kernel's): accumulate, partial sums of the values AS STORED, scale/shift, activation,
    //      shortcut, bf16 | fp32 store; one partial row per M-tile (gran = BM) ----------------------------------------------------------
    typedef typename std::conditional<O32, float, __bf16>::type out_t;
    out_t* __restrict__ gout16 = reinterpret_cast<out_t*>(p.out);
    auto row_of = [&](int mi, int r) { return m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half; };
    if (p.accumulate) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row_of(mi, r);
          if (m >= M) continue;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const int co = bn * BN + wn * 64 + ni * 32 + (lane & 31);
            if (co < p.Co) acc[mi][ni][r] += (float)gout16[(size_t)m * p.ldo + co];
          }
        }
    }
    if (p.stats) {
      float* red = reinterpret_cast<float*>(smem3);       // [2][WM][BN]
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        float s_ = 0.f, ss_ = 0.f;
        const int co = bn * BN + wn * 64 + ni * 32 + (lane & 31);
        if (p.bt_y && co < p.Co) {                         // BatchNorm tap: the terms of bn_act_bwd's reduce pass, y bf16
          const float mu = p.bt_mean[co], is = p.bt_invstd[co], ga = p.bt_gamma ? p.bt_gamma[co] : 1.f, be = p.bt_beta ? p.bt_beta[co] : 0.f;
          const __bf16* yb = reinterpret_cast<const __bf16*>(p.bt_y) + co;
          float yv[2][16];
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const int m = row_of(mi, r); yv[mi][r] = (float)yb[(size_t)(m < M ? m : M - 1) * p.Co]; }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int m = row_of(mi, r);
              const float xh = (yv[mi][r] - mu) * is;
              float g = O32 ? acc[mi][ni][r] : (float)(__bf16)acc[mi][ni][r];      // (the gradient as it is stored)
              if (p.bt_act == DCN_ACT_LEAKY) g = (ga * xh + be <= 0.f) ? g * p.bt_slope : g;
              g = m < M ? g : 0.f;
              s_ += g; ss_ += g * xh;
            }
        } else if (!p.bt_y) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = O32 ? acc[mi][ni][r] : (float)(__bf16)acc[mi][ni][r];
              s_ += v; ss_ = __builtin_fmaf(v, v, ss_);
            }
        }
        s_ += __shfl_xor(s_, 32); ss_ += __shfl_xor(ss_, 32);
        if (lane < 32) {
          const int col = wn * 64 + ni * 32 + lane;
          red[(0 * WM + wm) * BN + col] = s_;
          red[(1 * WM + wm) * BN + col] = ss_;
        }
      }
      __syncthreads();
      for (int idx = tid; idx < 2 * BN; idx += NT) {
        const int which = idx / BN, col = idx - which * BN;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) t += red[(which * WM + w) * BN + col];
        const int co = bn * BN + col;
        if (co < p.Co) p.stats[((size_t)bm * 2 + which) * p.Co + co] = t;
      }
    }
    const __bf16* res16 = reinterpret_cast<const __bf16*>(p.residual);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int co = bn * BN + wn * 64 + ni * 32 + (lane & 31);
      if (co >= p.Co) continue;
      const float sc = p.scale ? p.scale[co] : 1.f, sh = p.shift ? p.shift[co] : 0.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = row_of(mi, r);
          if (m >= M) continue;
          float v = acc[mi][ni][r] * sc + sh;
          if (p.act == DCN_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
          if (res16) v += (float)res16[(size_t)m * p.ldr + co];
          gout16[(size_t)m * p.ldo + co] = (out_t)v;
        }
    }
    return;
  }

  const float dq = 1.f / (sa * sb);           // powers of two: exact
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] *= dq;

  // ---- epilogue (igemm.hip's; output pixel index = flattened row m) ----------------------------------------
  float* __restrict__ gout = p.out;
  if (p.accumulate) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int co = bn * BN + wn * 64 + ni * 32 + (lane & 31);
          if (co < p.Co) acc[mi][ni][r] += gout[(size_t)m * p.ldo + co];
        }
      }
  }
  if (p.stats) {
    // one partial row per `gran` output pixels (the row count the caller sized the buffer for): groups of WM*64/gran slabs
    float* red = reinterpret_cast<float*>(smem3);       // [2][WM][BN]  (LDS is free after the last barrier)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      float s = 0.f, ss = 0.f;
      const int co = bn * BN + wn * 64 + ni * 32 + (lane & 31);
      if (p.bt_y && co < p.Co) {                           // BatchNorm tap (igemm.h): channel_partials_kernel<1>'s terms (bn.hip)
        const float mu = p.bt_mean[co], is = p.bt_invstd[co], ga = p.bt_gamma ? p.bt_gamma[co] : 1.f, be = p.bt_beta ? p.bt_beta[co] : 0.f;
        // all loads of the block first (rows past the end clamped, their terms dropped below): one wait instead of one per element
        const float* yb = p.bt_y + co;
        float yv[2][16];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            yv[mi][r] = yb[(size_t)(m < M ? m : M - 1) * p.Co];
          }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float xh = (yv[mi][r] - mu) * is;
            float g = acc[mi][ni][r];
            if (p.bt_act == DCN_ACT_LEAKY) g = (ga * xh + be <= 0.f) ? g * p.bt_slope : g;
            g = m < M ? g : 0.f;
            s += g; ss += g * xh;
          }
      } else if (!p.bt_y) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const float v = acc[mi][ni][r]; s += v; ss = __builtin_fmaf(v, v, ss); }      // (fma, spelled out: the sums are bitwise those of the build without the tap)
      }
      s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
      if (lane < 32) {
        const int col = wn * 64 + ni * 32 + lane;
        red[(0 * WM + wm) * BN + col] = s;
        red[(1 * WM + wm) * BN + col] = ss;
      }
    }
    __syncthreads();
    const int groups = BM / gran, spg = WM / groups;     // launch side guarantees gran in {BM, BM/2} and gran >= 64... slabs per group
    const int rows_total = (M + gran - 1) / gran;
    for (int idx = tid; idx < 2 * BN * groups; idx += NT) {
      const int gq = idx / (2 * BN), rest = idx - gq * 2 * BN;
      const int which = rest / BN, col = rest - which * BN;
      float t = 0.f;
      for (int w = 0; w < spg; ++w) t += red[(which * WM + gq * spg + w) * BN + col];
      const int co = bn * BN + col, srow = bm * groups + gq;
      if (co < p.Co && srow < rows_total) p.stats[((size_t)srow * 2 + which) * p.Co + co] = t;
    }
  }
  float sc[2], sh2[2]; int co_[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int co = bn * BN + wn * 64 + ni * 32 + (lane & 31);
    co_[ni] = co;
    sc[ni] = (p.scale && co < p.Co) ? p.scale[co] : 1.f;
    sh2[ni] = (p.shift && co < p.Co) ? p.shift[co] : 0.f;
  }
  float vmax = 0.f;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (m >= M) continue;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (co_[ni] >= p.Co) continue;
        float v = acc[mi][ni][r] * sc[ni] + sh2[ni];
        if (p.act == DCN_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
        if (p.residual) v += p.residual[(size_t)m * p.ldr + co_[ni]];
        gout[(size_t)m * p.ldo + co_[ni]] = v;
        vmax = fmaxf(vmax, fabsf(v));
      }
    }
  if (p.amax_out) {
    vmax = wave_max(vmax);
    if (lane == 0) amax_update(p.amax_out, vmax, blockIdx.x * (WM * WN) + wave);
  }
}

int g_conv3 = 1;          // dcn_set_tuning("3x3strip", 0): 3x3 stride-1 layers back on the implicit-GEMM tile
int g_conv3_bm = 0;       // dcn_set_tuning("3bm", 128|256): force the strip kernel's M tile (0 = automatic)
int g_conv3_abl = 0;      // dcn_set_tuning("3abl", bits): timing ablations (C3_ABL builds only)
int g_conv3_m16 = 1;      // dcn_set_tuning("3m16", 0|1|2): conv3x.hip (16x16x32 MFMAs) for the launches that take the 256-row tile (1) / for every strip launch it fits (2)
int g_conv3_ls = 1;       // dcn_set_tuning("3ls", 0): every global load of an iteration at its start again (A/B switch; LS = 1 measured 1-6 % faster per layer)

template <int WM, int WN, int A_LD, int NP = 2, int ABL = 0, int LS = 0, bool IN16 = false, bool O32 = false>
int launch3(const IgemmParams& p, int gran, hipStream_t stream) {
  constexpr int NT = 64 * WM * WN, BM = 64 * WM, BN = 64 * WN;
  const int S = BM + 2 * p.Wi + 2;
  const size_t lds = (size_t)2 * NP * (S + 2) * 32 + (size_t)2 * 3 * (NP * BN * 32 + 128);
  static DcnPerDeviceFlag attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_kernel<WM, WN, A_LD, NP, ABL, LS, IN16, O32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int gm = cdiv(p.M, BM), gn = cdiv(p.Co, BN);
  const double k_alg = 9.0 * p.Ci;
  const double alg_bytes = IN16 ? 2.0 * ((double)p.N * p.Hi * p.Wi * p.Ci + (double)p.Co * k_alg) +
                                      ((O32 ? 4.0 : 2.0) * (1.0 + (p.accumulate ? 1.0 : 0.0)) + (p.residual ? 2.0 : 0.0) + (p.bt_y ? 2.0 : 0.0)) * (double)p.M * p.Co
                                : 4.0 * ((double)p.N * p.Hi * p.Wi * p.Ci + (double)p.Co * k_alg + (double)p.M * p.Co * epilogue_reads(p));
  const int pid = prof_begin(IN16 ? 50 : (NP == 1 ? 33 : (WM == 4 ? 28 : 29)), 2.0 * (double)p.M * p.Co * k_alg, stream, alg_bytes);
  hipLaunchKernelGGL((conv3_kernel<WM, WN, A_LD, NP, ABL, LS, IN16, O32>), dim3(gm * gn), dim3(NT), lds, stream, p, S, gran);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("conv3");
  return DCN_OK;
}

constexpr int A_LD_MAX = 4;   // strip loads per thread: S <= threads pixels.  (A 12-load build for the 104/208-wide maps with 64 filters
                              //  spilled registers and lost to the implicit-GEMM tile: 0.68 vs 0.55 ms, 1.24 vs 0.72 ms.)
template <int WM, int WN>
int launch3_ld(const IgemmParams& p, int gran, hipStream_t stream) {
  if (p.wt16) return launch3<WM, WN, A_LD_MAX, 1>(p, gran, stream);       // bf16 operands
#if C3_ABL
  switch (g_conv3_abl) {
    case 1: return launch3<WM, WN, A_LD_MAX, 2, 1>(p, gran, stream);
    case 3: return launch3<WM, WN, A_LD_MAX, 2, 3>(p, gran, stream);
    case 7: return launch3<WM, WN, A_LD_MAX, 2, 7>(p, gran, stream);
    case 8: return launch3<WM, WN, A_LD_MAX, 2, 8>(p, gran, stream);
    case 15: return launch3<WM, WN, A_LD_MAX, 2, 15>(p, gran, stream);
    case 16: return launch3<WM, WN, A_LD_MAX, 2, 16>(p, gran, stream);
    default: break;
  }
#endif
  if (g_conv3_ls) return launch3<WM, WN, A_LD_MAX, 2, 0, 1>(p, gran, stream);
  return launch3<WM, WN, A_LD_MAX>(p, gran, stream);
}

// tile choice: 256 pixels x 128 filters (8 waves, one workgroup per CU) or 128 x 128 (4 waves, two per CU), see below
int conv3_tile(const IgemmParams& p, int gran, int* a_need) {
  int wmm = 4;
  const int wn = 2;
  if (wn == 2 && gran == 128) {
    const int gn = cdiv(p.Co, 128);
    const long long wgs256 = (long long)cdiv(p.M, 256) * gn;
    // Measured per layer at N = 64 (tools/bench_convs.py --ab 3bm=128): the 128-row tile (two workgroups per CU, epilogues
    // overlap the other's K loop) wins 4-9 % with 2 or >= 8 filter tiles, the 256-row tile (half the filter traffic, shorter
    // strip per output) wins 7-11 % with 1 or 4; both fill the same number of rounds.
    if (g_conv3_bm == 128 || (g_conv3_bm == 0 && (wgs256 < 256 || gn == 2 || gn >= 8))) wmm = 2;
  }
  const int S = 64 * wmm + 2 * p.Wi + 2;
  *a_need = cdiv(4 * S, 64 * wmm * wn);
  return wmm;
}

}  // namespace

void conv3_set_tuning(int key, int value) {
  if (key == 0) g_conv3 = value; else if (key == 1) g_conv3_bm = value; else if (key == 2) g_conv3_abl = value; else if (key == 4) g_conv3_m16 = value;
  else g_conv3_ls = value;
}

// can this launch run on the strip kernel?  (gran = rows per statistics partial the caller sized its buffer for)
bool conv3_applicable(const IgemmParams& p, int precision, int gran) {
  const bool f16 = precision == 4 && p.b_scale && p.amax_a && !p.wt16, b16 = precision == 2 && p.wt16;
  if (!g_conv3 || !(f16 || b16) || p.f8 || p.c4 || p.bmode != 0 || p.batch > 1 || p.row_scale || p.ncls) return false;
  if (p.ntaps != 9 || p.isy != 1 || p.isx != 1 || p.osy != 1 || p.osx != 1 || p.oy0 != 0 || p.ox0 != 0 || !p.dense_out) return false;
  if (p.Hs != p.Hi || p.Ws != p.Wi || p.Ho != p.Hi || p.Wo != p.Wi || p.M != p.N * p.Hi * p.Wi) return false;
  if (p.Ci % 32 != 0 || p.Co <= 64 || p.Wi < 2) return false;      // (64-filter layers only occur on the 104/208-wide maps: strip too long)
  if (gran != 128 && gran != 256) return false;
  unsigned seen = 0;
  for (int t = 0; t < 9; ++t) {
    const int dy = p.tap_dy[t], dx = p.tap_dx[t];
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
    seen |= 1u << ((dy + 1) * 3 + dx + 1);
  }
  if (seen != 0x1FF) return false;
  int a_need; const int wmm = conv3_tile(p, gran, &a_need);
  if (a_need > A_LD_MAX) return false;
  const int S = 64 * wmm + 2 * p.Wi + 2, bn = 128;
  if ((size_t)4 * (S + 2) * 32 + (size_t)6 * (2 * bn * 32 + 128) > 160 * 1024) return false;
  // 32-bit strip offsets: S pixels of ldi floats
  if ((long long)S * p.ldi * 4 >= 0x7FFFFFF0LL || (long long)p.Co * p.ldw * 4 >= 0x7FFFFFF0LL) return false;
  return true;
}

// ---- bf16 storage: the 3x3 stride-1 launches of conv1b_launch (conv1.hip) ------------------------------------------------------------
// OFF by default: built as the round-4 plan's "bf16 strip kernel" and measured against the gathered tiles it was meant to replace
// (tools/bench_b16.py --set 3h16=0 --ab 3h16=1, N = 64, forward / data gradient, ms): 128->256 @52 0.160 -> 0.187 / 0.162 -> 0.147, 256->512 @26 0.150 ->
// 0.148 / 0.139 -> 0.159, 512->1024 @13 0.137 -> 0.155 / 0.173 -> 0.172, 64->128 @104 0.184 -> 0.253 / 0.227 -> 0.229, 512->512 @52 0.851 -> 0.927 /
// 0.845 -> 0.906 — it LOSES: with one MFMA per product (not three) the strip loop's 12 MFMAs per wave between barriers and its register-staged
// loads cost more than the nine-fold gather of conv1b's LDS-DMA rings, which mostly hits L2.  Kept behind the knob with its exact-model tests.
int g_conv3b = 0;         // dcn_set_tuning("3h16", 1): bf16-storage 3x3 stride-1 launches on the strip kernel
void conv3b_set_tuning(int v) { g_conv3b = v; }

// M-tile (pixels) of a bf16-storage strip launch, 0 = not on this kernel.  A function of the shape and the knob only: the caller sizes its
// BatchNorm partial rows (one per M-tile) with it.  Wi = map width of the (stride-1, 3x3) launch.
int conv3b_bm(int M, int Co, int Wi) {
  if (!g_conv3b || Co <= 64 || Co % 32 != 0 || Wi < 2) return 0;
  const int gn = cdiv(Co, 128);
  const long long wgs256 = (long long)cdiv(M, 256) * gn;
  int bm = (wgs256 < 256 || gn == 2 || gn >= 8) ? 128 : 256;          // (conv3_tile's choice for the fp32 strip)
  const int S = bm + 2 * Wi + 2;
  if (2 * S > 4 * (bm * 2)) return 0;                                   // <= 4 strip pieces per thread (threads = 2 bm)
  if ((size_t)2 * (S + 2) * 32 + (size_t)6 * (128 * 32 + 128) > 160 * 1024) return 0;
  return bm;
}
bool conv3b_takes(const IgemmParams& p) {
  if (p.ntaps != 9 || p.isy != 1 || p.isx != 1 || p.osy != 1 || p.osx != 1 || p.oy0 != 0 || p.ox0 != 0 || !p.dense_out || p.ncls || p.batch > 1) return false;
  if (p.Hs != p.Hi || p.Ws != p.Wi || p.Ho != p.Hi || p.Wo != p.Wi || p.M != p.N * p.Hi * p.Wi || p.Ci % 32 != 0) return false;
  unsigned seen = 0;
  for (int t = 0; t < 9; ++t) {
    const int dy = p.tap_dy[t], dx = p.tap_dx[t];
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1) return false;
    seen |= 1u << ((dy + 1) * 3 + dx + 1);
  }
  if (seen != 0x1FF) return false;
  const int bm = conv3b_bm(p.M, p.Co, p.Wi);
  if (!bm) return false;
  const int S = bm + 2 * p.Wi + 2;
  return (long long)S * p.ldi * 2 < 0x7FFFFFF0LL && (long long)p.Co * p.ldw * 2 < 0x7FFFFFF0LL;
}
// p.in / p.residual / p.bt_y / p.wt point at bf16 data (p.wt: the bank [Co][9 Ci]), p.out at bf16 (out_f32 = 0) or fp32 data
int conv3b_launch(const IgemmParams& p0, int out_f32, hipStream_t stream) {
  IgemmParams p = p0;
  p.wt16 = p0.wt;                                                      // (the kernel's one-plane path reads the bank through wt16)
  const int bm = conv3b_bm(p.M, p.Co, p.Wi);
  const int S = bm + 2 * p.Wi + 2, need = cdiv(2 * S, 2 * bm);          // strip pieces per thread
  if (bm == 128) {
    if (need <= 2) return out_f32 ? launch3<2, 2, 2, 1, 0, 1, true, true>(p, bm, stream) : launch3<2, 2, 2, 1, 0, 1, true, false>(p, bm, stream);
    return out_f32 ? launch3<2, 2, 4, 1, 0, 1, true, true>(p, bm, stream) : launch3<2, 2, 4, 1, 0, 1, true, false>(p, bm, stream);
  }
  if (need <= 2) return out_f32 ? launch3<4, 2, 2, 1, 0, 1, true, true>(p, bm, stream) : launch3<4, 2, 2, 1, 0, 1, true, false>(p, bm, stream);
  return out_f32 ? launch3<4, 2, 4, 1, 0, 1, true, true>(p, bm, stream) : launch3<4, 2, 4, 1, 0, 1, true, false>(p, bm, stream);
}

int conv3_launch(const IgemmParams& p, int gran, hipStream_t stream) {
  int a_need; const int wmm = conv3_tile(p, gran, &a_need);
  if (g_conv3_m16 && !g_conv3_abl && conv3x_takes(p, gran)) {
    // conv3x.hip (16x16x32 MFMAs, 256 x 128 tile, one workgroup per CU) measured against the choice above per layer of the step at N = 64
    // (tools/bench_convs.py --strip --ab 3m16=2, forward / data gradient): 4-17 % faster on 12 of 14 launches; 8-9 % SLOWER on the two whose
    // grid is 1.32-1.34 rounds of 256 workgroups (512 -> 1024 @13 forward, 512 -> 256 @26 data gradient): the last third of a round costs a
    // whole one there, and half of one on the 128-row tile with two workgroups per CU.
    const long long wgs = (long long)cdiv(p.M, 256) * cdiv(p.Co, 128);
    const int cus = dcn_device_cus() > 0 ? dcn_device_cus() : 256;
    const bool short_tail = wgs > cus && wgs <= cus + cus / 2;
    if (g_conv3_m16 == 2 || !short_tail) return conv3x_launch(p, gran, stream);
  }
  if (wmm == 2) return launch3_ld<2, 2>(p, gran, stream);
  return launch3_ld<4, 2>(p, gran, stream);
}
