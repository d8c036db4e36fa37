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
#include <type_traits>

namespace {


typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr unsigned OOB = 0x80000000u;   // voffset sentinel: beyond every descriptor (num_records <= 0x7FFFFFF0) -> load returns 0

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, long long bytes) {
  const unsigned n = bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0u : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, n, 0x00020000);
}

// Loader design: every global access is a raw buffer load whose out-of-range lanes return zero, so the
// padding halo, the M / Cout / K tails and the stem's tap padding need no branches and no zero-fill moves.
// Per thread the row byte offsets and a per-row bit mask of in-bounds taps are computed ONCE; a K-step then
// costs ~4 VALU per activation load (add the wave-uniform tap delta, test the mask bit, select the sentinel)
// and none per weight load (constant voffset, wave-uniform soffset).  With 64 MFMAs per K-step per wave this
// keeps the vector ALU out of the matrix pipe's way (it was 3.4 VALU per MFMA with pointer arithmetic).
// SP ("split"): the same tile engine on the bf16 matrix pipe at fp32 accuracy.  Every fp32 operand is cut
// into three bf16 pieces x = h + m + l (truncation cuts, exact: 8 + 8 + 8 significant bits) when its tile
// is written to LDS, and a product a*b is accumulated as the six cross terms of weight >= 2^-16
// (l*h, h*l, m*m, m*h, h*m, h*h; the dropped m*l, l*m, l*l are <= 2^-23 relative, one fp32 rounding) with
// v_mfma_f32_32x32x16_bf16: 6 instructions of 32 cycles per 16 k against 8 of 64 cycles on the fp32 pipe
// (bf16 products are exact in the fp32 accumulator).  LDS holds three bf16 planes per operand,
// rows padded by 16 B so the ds_read_b128 fragments (8 consecutive k per lane) are conflict-free.
// NN split mode: the B tile keeps its HBM orientation [16 k][128 n] per bf16 plane (256-B rows, 16-B chunks
// XOR-swizzled) and the k-contiguous MFMA fragments come from the hardware transpose read (as in wgrad.hip).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
// amax bits -> the power of two that maps amax into [2^13, 2^14): amax = f * 2^e with f in [0.5, 1), s = 2^(14 - e).
// Zero / non-finite maxima give 1 (nothing to scale, or Inf/NaN data that stays Inf/NaN).
__device__ __forceinline__ float pow2_scale(unsigned amax_bits) {
  const int be = (int)((amax_bits >> 23) & 0xFF);            // biased exponent: amax in [2^(be-127), 2^(be-126))
  if (be == 0 || be == 255) return 1.f;
  int e = 14 - (be - 126);                                    // s = 2^e
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return __uint_as_float((unsigned)(e + 127) << 23);
}
__device__ __forceinline__ int nn_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int nn_off(int row, int c) { return 256 * row + 16 * ((c >> 3) ^ nn_swz(row)) + 8 * ((c >> 2) & 1); }
__device__ __forceinline__ bf16x8 nn_frag(const unsigned char* base, int byte0, int byte1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + byte0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + byte1));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// NP = 2: the fp32-accurate split on the FP16 pipe.  Each operand tensor carries its abs-max (IgemmParams::amax_a/b, kept
// up to date by the kernels that produce the tensor); the tile loader scales by the power of two that maps the maximum
// below 2^14, cuts x = h + l with two round-to-nearest f16 conversions (11 + 11 significant bits; elements far below the
// maximum lose bits only in ABSOLUTE terms, 2^-39 of the maximum) and accumulates l*h, h*l, h*h with
// v_mfma_f32_32x32x16_f16: three MFMAs per product instead of six, 3 vector-ALU operations per element instead of 5.5
// (v_pk_mul_f32, v_cvt_pk_f16_f32, v_cvt_f32_f16 x2, v_pk_add_f32, v_cvt_pk_f16_f32 per element pair).  The accumulator
// is rescaled by the exact 1/(sA*sB) before the epilogue.  Measured error against fp64: that of an fp32 GEMM.
// NP = number of bf16 pieces per operand: 3 = fp32-accurate split (six cross terms); 1 = plain bf16 operands
// (round-to-nearest-even, one MFMA per product, fp32 accumulate): the builder-defined bf16 mode of
// BASELINE.json configs[2] (dcn_set_tuning("precision", 2)); tensors in HBM stay fp32.
// OCC = waves per SIMD the register allocation must allow (__launch_bounds__): the split kernel needs 206 registers
// (2 waves/SIMD); forced to 168 (3 waves/SIMD, some loop-invariant state in scratch) it is 8-19 % faster on short-K
// launches (1x1 layers: more waves cover the prologue/epilogue and the barrier) and 2-5 % on the 3x3 layers
// (tools/bench_convs.py --ab occ3=0 --ab-default 1073741824: forward 34.4 -> 32.8 ms, data gradient 37.5 -> 35.7 ms).
// F8 (with NP = 1, BK = 32): operands scaled by a per-tensor power of two (IgemmParams::f8 = {sA, sB}, dcn_f8_scale),
// rounded to OCP fp8 e4m3 (v_cvt_pk_fp8_f32) when the tile is staged, multiplied by v_mfma_f32_32x32x16_fp8_fp8 (one
// 16-B LDS fragment = 16 fp8 feeds two MFMAs), accumulated in fp32 and rescaled by 1/(sA*sB) before the epilogue:
// the builder-defined fp8 conv path of BASELINE.json configs[4] (ops.set_precision("fp8")).
// BPRE (with NP = 2): the B operand (filter bank) arrives ALREADY split — dcn_presplit_f16 rewrote every 8 consecutive k of
// a row as [8 x f16 high | 8 x f16 low] (same 32 bytes), scaled by *b_scale — so its tile is a plain 16-B copy into the two
// LDS planes: the split of the weights happens once per layer and step instead of once per M-tile (338 times on a 26x26 map).
template <int BM, int BN, int WM, int WN, int BMODE, bool C4, int BK, bool SP = false, int ABL = 0, int NP = 3, int OCC = 1,
          bool F8 = false, bool BPRE = false>
__global__ __launch_bounds__(256, OCC) void igemm_kernel(const IgemmParams p) {
  static_assert(!BPRE || (SP && NP == 2 && BMODE == 0 && BK == 16), "pre-split B: f16 split, NT tiles, 16-deep K-step");
  static_assert(!F8 || (SP && NP == 1 && BK == 32 && BMODE == 0), "fp8 operands: NT tiles, 32-deep K-step");
  static_assert(!SP || !C4, "split mode: no stem path");
  static_assert(!SP || BMODE == 0 || (BN == 128 && BK == 16 && (NP == 3 || NP == 2)), "split NN mode: 128-wide tile, 16-deep K-step");
  constexpr bool H2 = SP && NP == 2;       // f16 two-piece split with per-tensor power-of-two scales
  constexpr int LDS_LD = BK + 4;           // padded LDS row: conflict-free ds_read_b128 fragments
  // SP: bf16 plane row in ushorts.  BK = 16: unpadded 32-B rows whose two 16-B halves swap places in rows
  // 8-15 (mod 16) — conflict-free for the ds_read_b128 fragments (16-lane groups see 16 distinct 16-B slots)
  // AND for the ds_write_b64 of the split pieces (a 16-lane group fills four whole rows = 128 contiguous bytes;
  // with 48-B padded rows a third of the LDS cycles were bank conflicts, SQ_LDS_BANK_CONFLICT).  BK = 32: padded.
  constexpr int LD16 = (BK == 16 || F8) ? 16 : BK + 8;      // (fp8, BK = 32: the same 32-B rows, 32 one-byte k)
  auto sp_w = [](int row, int chunk) {      // ushort offset of floats 4*chunk..4*chunk+3 of `row` inside a plane
    if (F8) return row * 16 + ((((chunk >> 2) ^ (row >> 3)) & 1) << 3) + (chunk & 3) * 2;
    return BK == 16 ? row * 16 + ((((chunk >> 1) ^ (row >> 3)) & 1) << 3) + (chunk & 1) * 4 : row * LD16 + chunk * 4;
  };
  auto sp_r = [](int row, int half) {       // ushort offset of the 16-B half `half` (K-step slice 0) of `row`
    return (BK == 16 || F8) ? row * 16 + (((half ^ (row >> 3)) & 1) << 3) : row * LD16 + half * 8;
  };
  // fp8: the scales either come ready (p.f8, from dcn_f8_scale's abs-max pass) or are derived here from the operands' tracked
  // abs-max words: the power of two that maps the maximum into [2^7, 2^8) — inside e4m3's finite range (448), no extra pass
  float f8_sa = 1.f, f8_sb = 1.f;
  if constexpr (F8) {
    if (p.f8) { f8_sa = p.f8[0]; f8_sb = p.f8[1]; }
    else { f8_sa = pow2_scale(amax_read(p.amax_a)) * (1.f / 64.f); f8_sb = pow2_scale(amax_read(p.amax_b)) * (1.f / 64.f); }
  }
  if constexpr (H2) { f8_sa = pow2_scale(amax_read(p.amax_a)); f8_sb = BPRE ? p.b_scale[0] : pow2_scale(amax_read(p.amax_b)); }
  // pre-split B: chunk c of a K-step row holds plane c & 1, k-half c >> 1 (ushort offset of the 16-B piece inside the plane)
  auto bpre_w = [](int row, int chunk) { return row * 16 + ((((chunk >> 1) ^ (row >> 3)) & 1) << 3); };
  constexpr int CPR = BK / 4;              // 16-B chunks per K-step row
  constexpr int RPP = 256 / CPR;           // rows staged per pass of the 256 threads
  constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
  constexpr int A_LD = BM / RPP;                                       // 16-B global loads per thread per K-step
  constexpr int B_LD = BMODE == 0 ? (BN >= RPP ? BN / RPP : 1) : BK * BN / 4 / 256;
  constexpr bool B_PART = BMODE == 0 && BN < RPP;                      // only threads with row0 < BN stage B
  constexpr int C4_STEPS = 64 / BK;
  static_assert(WM * WN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [2][BM][LDS_LD]
  float* Bs = smem + 2 * BM * LDS_LD;    // BMODE 0: [2][BN][LDS_LD]   BMODE 1: [2][BK][BN]
  constexpr int B_TILE = BMODE == 0 ? BN * LDS_LD : BK * BN;
  unsigned short* As16 = reinterpret_cast<unsigned short*>(smem);      // SP: [2][NP][BM][LD16]
  unsigned short* Bs16 = As16 + 2 * NP * BM * LD16;                    // SP: [2][NP][BN][LD16]   (NN: [2][NP][16][128])
  // (pre-split B: +64 B between the planes — a ds_write_b128 stores the high and the low chunk of a row from neighbouring
  //  lanes, and with planes 4096 B apart both fell on the same banks: two-way conflict on every filter store)
  constexpr int B_PLANE = (BMODE == 0 ? BN * LD16 : 2048) + (BPRE ? 32 : 0);   // ushorts per B plane

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int gn = (p.Co + BN - 1) / BN;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  // Sub-problem classes (ncls = 4: the parity classes of a stride-2 data gradient in ONE launch): block order (M-tile, class,
  // N-tile), so the four classes of a region run side by side on one XCD and share its L2 — as four launches every class
  // streamed dY from HBM again (9 tap reads of a tensor larger than the Infinity Cache).  Class c has its own output sub-grid
  // and its taps at [4c, 4c + ntaps).
  int bm, bn, M_ = p.M, Hs_ = p.Hs, Ws_ = p.Ws, oy0_ = p.oy0, ox0_ = p.ox0, ntaps_ = p.ntaps, tb_ = 0;
  if (p.ncls) {
    const int q = lin / gn, cls = q % p.ncls;
    bn = lin - q * gn; bm = q / p.ncls;
    M_ = p.cls_M[cls]; Hs_ = p.cls_Hs[cls]; Ws_ = p.cls_Ws[cls]; oy0_ = p.cls_oy0[cls]; ox0_ = p.cls_ox0[cls];
    ntaps_ = p.cls_ntaps[cls]; tb_ = 4 * cls;
  } else { bm = lin / gn; bn = lin - bm * gn; }
  const int kiters_ = p.c4 ? p.kiters : ntaps_ * p.cpt;
  const int chunk = tid & (CPR - 1);     // which 16-B piece of the K-step row
  const int row0 = tid / CPR;            // rows row0 + RPP*j
  const int hsws = Hs_ * Ws_;

  // ---- descriptors (wave-uniform) ---------------------------------------------------------------
  const int n0 = (bm * BM) / hsws;                                   // image of the block's first row
  const long long img = (long long)p.Hi * p.Wi * p.ldi;              // floats per image of the gathered tensor
  const float* a_base = p.in + (long long)blockIdx.y * p.in_bs + (long long)n0 * img;
  const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(a_base, ((long long)(p.N - n0) * img) * 4);
  const float* b_base = p.wt + (long long)blockIdx.y * p.wt_bs;
  const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(b_base, (long long)(BMODE == 0 ? p.Co : p.kvalid) * p.ldw * 4);
  float* __restrict__ gout = p.out + (long long)blockIdx.y * p.out_bs;

  // ---- per-thread row state, computed once ------------------------------------------------------
  unsigned a_off[A_LD];                  // byte offset of (row, tap 0,0, this chunk) from a_base
  unsigned a_msk[A_LD];                  // bit t set <=> tap t of this row reads inside the image
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    const int m = bm * BM + row0 + RPP * j;
    a_off[j] = 0; a_msk[j] = 0;
    if (ntaps_ == 1 && p.dense_out && p.isy == 1 && p.isx == 1 && !C4 && p.tap_dy[tb_ + 0] == 0 && p.tap_dx[tb_ + 0] == 0 && Ws_ == p.Wi && Hs_ == p.Hi) {
      // plain GEMM rows (1x1 convolutions, their data gradients, the co-attention products): row m IS pixel m — no divisions
      if (m < M_) { a_off[j] = (unsigned)((m - n0 * hsws) * p.ldi * 4 + chunk * 16); a_msk[j] = 1u; }
    } else
    if (m < M_) {
      const int n = m / hsws, rem = m - n * hsws;
      const int i = rem / Ws_, jx = rem - i * Ws_;
      const int iy0 = i * p.isy, ix0 = jx * p.isx;
      a_off[j] = (unsigned)((((n - n0) * p.Hi + iy0) * p.Wi + ix0) * p.ldi * 4 + chunk * 16 * (C4 ? 0 : 1));
      unsigned msk = 0;
      for (int t = 0; t < ntaps_; ++t) {
        int dy, dx;
        if (C4) { const int r = t / 3; dy = r - 1; dx = t - 3 * r - 1; }
        else { dy = p.tap_dy[tb_ + t]; dx = p.tap_dx[tb_ + t]; }
        if ((unsigned)(iy0 + dy) < (unsigned)p.Hi && (unsigned)(ix0 + dx) < (unsigned)p.Wi) msk |= 1u << t;
      }
      a_msk[j] = msk;
    }
  }
  unsigned b_off[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    if (BMODE == 0) {
      const int co = bn * BN + row0 + RPP * j;
      b_off[j] = (co < p.Co && (!B_PART || row0 < BN)) ? (unsigned)(co * p.ldw * 4 + chunk * 16) : OOB;
    } else {
      const int idx = tid + 256 * j;
      const int k = idx / (BN / 4), col = bn * BN + (idx - k * (BN / 4)) * 4;
      b_off[j] = col < p.Co ? (unsigned)((k * p.ldw + col) * 4) : OOB;
    }
  }
  // stem (C4): this thread's chunk IS a tap: taps chunk (K-step 0) and 8+chunk (K-step 1)
  int c4_delta[C4_STEPS]; unsigned c4_bit[C4_STEPS];
  if (C4) {
#pragma unroll
    for (int h = 0; h < C4_STEPS; ++h) {
      const int t = h * CPR + chunk, r = t / 3;
      c4_delta[h] = ((r - 1) * p.Wi + (t - 3 * r - 1)) * p.ldi * 4;
      c4_bit[h] = t < ntaps_ ? 1u << t : 0u;
    }
  }

  // ---- wave-uniform K iterator: (tap, channel step) -> activation byte delta, weight byte offset ----
  int k_tap = 0, k_c = 0;
  int a_delta = C4 ? 0 : (p.tap_dy[tb_ + 0] * p.Wi + p.tap_dx[tb_ + 0]) * p.ldi * 4;
  unsigned tap_bit = 1u;
  unsigned a_soff = 0;                                  // channel offset inside the tap (bytes)
  unsigned b_soff = C4 ? 0u : (unsigned)p.tap_w[tb_ + 0] * 4; // BMODE 0: K offset in the filter row; BMODE 1: k0*ldw
  int kbase = 0;                                        // BMODE 1: first k of the step (K tail masking)

  f32x4 a_reg[A_LD], b_reg[B_LD];
  f32x4 a_reg2[SP ? A_LD : 1], b_reg2[SP ? B_LD : 1];   // SP: second register stage (loads run two K-steps ahead)
  // live = false: the same instructions with every offset out of range (zeros, no memory traffic).  A load inside a
  // wave-uniform branch makes the s_waitcnt in front of the OLDER stage's LDS stores conservative — it then also waits for the
  // loads just issued, i.e. exposes a full L2/HBM latency per pair of K-steps — so the K loop never branches around a load.
  auto load_tiles_into = [&](f32x4* a_reg, f32x4* b_reg, int it, bool live = true) {
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      unsigned v;
      if (C4) {
        int dlt = c4_delta[0]; unsigned bit = c4_bit[0];
#pragma unroll
        for (int h = 1; h < C4_STEPS; ++h) if (it == h) { dlt = c4_delta[h]; bit = c4_bit[h]; }
        v = (a_msk[j] & bit) ? a_off[j] + (unsigned)dlt : OOB;
      }
      else v = (a_msk[j] & tap_bit) ? a_off[j] + (unsigned)a_delta : OOB;
      a_reg[j] = buf_load16(a_rs, live ? v : OOB, live ? a_soff : 0u);
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      unsigned v = b_off[j];
      if (BMODE == 1) { const int k = (tid + 256 * j) / (BN / 4); v = (kbase + k < p.kvalid) ? v : OOB; }
      b_reg[j] = buf_load16(b_rs, live ? v : OOB, live ? b_soff : 0u);
    }
    // advance to the next K-step (scalar unit)
    if (C4) { b_soff += BK * 4; }
    else {
      ++k_c; a_soff += BK * 4;
      if (BMODE == 0) b_soff += BK * 4; else { b_soff += BK * p.ldw * 4; kbase += BK; }
      if (k_c == p.cpt) {
        k_c = 0; a_soff = 0; ++k_tap;
        if (k_tap < ntaps_) {
          a_delta = (p.tap_dy[tb_ + k_tap] * p.Wi + p.tap_dx[tb_ + k_tap]) * p.ldi * 4;
          tap_bit = 1u << k_tap;
          if (BMODE == 0) b_soff = (unsigned)p.tap_w[tb_ + k_tap] * 4;
        }
      }
    }
  };
  auto split_store = [&](unsigned short* plane0, int plane_stride, int off, const f32x4 v, const float sc = 1.f) {
    if constexpr (F8) {                  // 4 floats -> 4 e4m3 bytes (scaled, clamped to the finite range, nearest even)
      float t[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = fminf(fmaxf(v[e] * sc, -448.f), 448.f);
      int w = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], 0, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], w, true);
      *reinterpret_cast<int*>(plane0 + off) = w;
      return;
    }
    if constexpr (NP == 2) {             // x*s = h + l, both f16 (round to nearest): v_pk_mul, v_cvt_pk_f16_f32, v_cvt_f32_f16, v_pk_add
      typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
      const f32x4 t = v * sc;
      const f16x4_t h = {(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
      const f16x4_t l = {(_Float16)(t[0] - (float)h[0]), (_Float16)(t[1] - (float)h[1]), (_Float16)(t[2] - (float)h[2]),
                         (_Float16)(t[3] - (float)h[3])};
      *reinterpret_cast<uint2*>(plane0 + off) = __builtin_bit_cast(uint2, h);
      *reinterpret_cast<uint2*>(plane0 + plane_stride + off) = __builtin_bit_cast(uint2, l);
      return;
    }
    if constexpr (NP == 1) {             // plain bf16 operands: v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN)
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      const bf16x4_t b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      *reinterpret_cast<uint2*>(plane0 + off) = __builtin_bit_cast(uint2, b);
      return;
    }
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = v[e];              // (a bit_cast straight from the vector element reads element 0)
      if (ABL == 1) { h[e] = m[e] = l[e] = __float_as_uint(x); continue; }
      h[e] = __float_as_uint(x) & 0xFFFF0000u;
      const float r1 = x - __uint_as_float(h[e]);
      m[e] = __float_as_uint(r1) & 0xFFFF0000u;
      l[e] = __float_as_uint(r1 - __uint_as_float(m[e]));
    }
    // pack the high halves of two dwords: bytes {lo.b2, lo.b3, hi.b2, hi.b3}
    uint2 ph = {__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u)};
    uint2 pm = {__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
    uint2 pl = {__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
    *reinterpret_cast<uint2*>(plane0 + off) = ph;
    *reinterpret_cast<uint2*>(plane0 + plane_stride + off) = pm;
    *reinterpret_cast<uint2*>(plane0 + 2 * plane_stride + off) = pl;
  };
  auto load_tiles = [&](int it) { load_tiles_into(a_reg, b_reg, it); };
  auto store_tiles_from = [&](const f32x4* a_reg, const f32x4* b_reg, int buf) {
    if constexpr (SP) {
      unsigned short* a16 = As16 + buf * NP * BM * LD16;
      unsigned short* b16 = Bs16 + buf * NP * B_PLANE;
#pragma unroll
      for (int j = 0; j < A_LD; ++j) split_store(a16, BM * LD16, sp_w(row0 + RPP * j, chunk), a_reg[j], f8_sa);
#pragma unroll
      for (int j = 0; j < B_LD; ++j) {
        if constexpr (BPRE) {
          if (!B_PART || row0 < BN) *reinterpret_cast<f32x4*>(b16 + (chunk & 1) * B_PLANE + bpre_w(row0 + RPP * j, chunk)) = b_reg[j];
        } else
        if constexpr (BMODE == 0) { if (!B_PART || row0 < BN) split_store(b16, B_PLANE, sp_w(row0 + RPP * j, chunk), b_reg[j], f8_sb); }
        else split_store(b16, B_PLANE, nn_off((tid + 256 * j) >> 5, ((tid + 256 * j) & 31) * 4) >> 1, b_reg[j], f8_sb);
      }
      return;
    }
    float* a = As + buf * BM * LDS_LD;
    float* b = Bs + buf * B_TILE;
#pragma unroll
    for (int j = 0; j < A_LD; ++j)
      *reinterpret_cast<f32x4*>(a + (row0 + RPP * j) * LDS_LD + chunk * 4) = a_reg[j];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      if (BMODE == 0) { if (!B_PART || row0 < BN) *reinterpret_cast<f32x4*>(b + (row0 + RPP * j) * LDS_LD + chunk * 4) = b_reg[j]; }
      else *reinterpret_cast<f32x4*>(b + (tid + 256 * j) * 4) = b_reg[j];
    }
  };
  auto store_tiles = [&](int buf) { store_tiles_from(a_reg, b_reg, buf); };

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

  int nn_tr[NI][2];
  if constexpr (SP && BMODE == 1) {
    // transposed-read addresses (wgrad.hip): 16-lane group (h, gg): k rows 8h + 4r + q, columns blk*32 + 16gg + 4pp
    const int g16 = lane >> 4, hh = g16 >> 1, gg = g16 & 1, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) nn_tr[ni][r2] = nn_off(8 * hh + 4 * r2 + q, wn * (BN / WN) + ni * 32 + 16 * gg + 4 * pp);
  }
  if constexpr (SP) {
    // Two K-steps of global loads in flight (a K-step of six bf16 MFMAs per block is shorter than the
    // L2/HBM latency): stage S of the registers holds step it+1 while stage S^1 receives step it+2.
    // One K-step: fragments of LDS buffer `cur`, six MFMA groups (one per cross term); the split + LDS
    // store of the NEXT step's tile (already in registers) is cut into pieces and issued between the
    // groups, so the vector ALU and the LDS write port work in the shadow of the matrix pipe (a wave
    // issues in order: anything placed after the last MFMA would wait for all of them).
    auto step = [&](int cur, const f32x4* ar, const f32x4* br, auto do_store) {
      const unsigned short* a16 = As16 + cur * NP * BM * LD16 + sp_r(wm * (BM / WM) + (lane & 31), lane >> 5);
      const unsigned short* b16 = Bs16 + cur * NP * B_PLANE + (BMODE == 0 ? sp_r(wn * (BN / WN) + (lane & 31), lane >> 5) : 0);
      unsigned short* na = As16 + (cur ^ 1) * NP * BM * LD16;
      unsigned short* nb = Bs16 + (cur ^ 1) * NP * B_PLANE;
      constexpr int PIECES = A_LD + B_LD;
      constexpr int TERMS = NP == 3 ? 6 : (NP == 2 ? 3 : 1);
      constexpr int KK = F8 ? 1 : BK / 16;          // fragment reads per K-step (fp8: one 16-B read covers 16 k)
      constexpr int GROUPS = TERMS * KK;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        bf16x8 af[MI][NP], bf[NI][NP];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int q = 0; q < NP; ++q)
            af[mi][q] = *reinterpret_cast<const bf16x8*>(a16 + q * BM * LD16 + mi * 32 * LD16 + kk * 16);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            if constexpr (BMODE == 0) bf[ni][q] = *reinterpret_cast<const bf16x8*>(b16 + q * B_PLANE + ni * 32 * LD16 + kk * 16);
            else bf[ni][q] = nn_frag(reinterpret_cast<const unsigned char*>(b16 + q * B_PLANE), nn_tr[ni][0], nn_tr[ni][1]);
          }
        // smallest terms first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
        // (f16 split: (l,h) (h,l) (h,h))
        constexpr int QA[6] = {NP == 3 ? 2 : (NP == 2 ? 1 : 0), 0, NP == 2 ? 0 : 1, 1, 0, 0}, QB[6] = {0, NP == 2 ? 1 : 2, NP == 2 ? 0 : 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < (ABL == 2 ? 1 : TERMS); ++t) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              if constexpr (F8) {
                typedef long l64x2 __attribute__((ext_vector_type(2)));
                const l64x2 a2 = __builtin_bit_cast(l64x2, af[mi][0]), b2 = __builtin_bit_cast(l64x2, bf[ni][0]);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a2[0], b2[0], acc[mi][ni], 0, 0, 0);
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a2[1], b2[1], acc[mi][ni], 0, 0, 0);
              } else if constexpr (NP == 2) {
                typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, af[mi][QA[t]]),
                                                                     __builtin_bit_cast(f16x8_t, bf[ni][QB[t]]), acc[mi][ni], 0, 0, 0);
              } else
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][QA[t]], bf[ni][QB[t]], acc[mi][ni], 0, 0, 0);
            }
          if constexpr (decltype(do_store)::value) {
            // pieces g, g + GROUPS, ... belong to group g = kk*6 + t
#pragma unroll
            for (int pc = kk * TERMS + t; pc < PIECES; pc += GROUPS) {
              if (pc < A_LD) split_store(na, BM * LD16, sp_w(row0 + RPP * pc, chunk), ar[pc], f8_sa);
              else if constexpr (BMODE == 1)
                split_store(nb, B_PLANE, nn_off((tid + 256 * (pc - A_LD)) >> 5, ((tid + 256 * (pc - A_LD)) & 31) * 4) >> 1, br[pc - A_LD], f8_sb);
              else if constexpr (BPRE) {
                if (!B_PART || row0 < BN) *reinterpret_cast<f32x4*>(nb + (chunk & 1) * B_PLANE + bpre_w(row0 + RPP * (pc - A_LD), chunk)) = br[pc - A_LD];
              }
              else if (!B_PART || row0 < BN) split_store(nb, B_PLANE, sp_w(row0 + RPP * (pc - A_LD), chunk), br[pc - A_LD], f8_sb);
            }
          }
        }
      }
      if constexpr (decltype(do_store)::value && (NP == 3 || NP == 2)) {
        // ask the scheduler for: 1 MFMA, then up to 5 VALU and an LDS write in its 32-cycle shadow, x24 (x12: f16 split)
#pragma unroll
        for (int g = 0; g < MI * NI * TERMS * (BK / 16); ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
          if (g & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      }
    };
    using T = std::true_type; using F = std::false_type;
    load_tiles_into(a_reg, b_reg, 0);
    if (kiters_ > 1) load_tiles_into(a_reg2, b_reg2, 1);
    store_tiles_from(a_reg, b_reg, 0);
    __syncthreads();
    int it = 0;
    using ST = std::conditional_t<ABL == 4, F, T>;      // (timing ablations: 3 = no global loads in the loop, 4 = no split / LDS stores)
    for (; it + 2 < kiters_; it += 2) {
      // even step: LDS buffer 0 is current, stage 2 holds step it+1, stage 1 is free for step it+2
      if (ABL != 3) load_tiles_into(a_reg, b_reg, it + 2);
      step(0, a_reg2, b_reg2, ST{});
      __syncthreads();
      if (ABL != 3) load_tiles_into(a_reg2, b_reg2, it + 3, it + 3 < kiters_);
      step(1, a_reg, b_reg, ST{});
      __syncthreads();
    }
    if (it + 1 < kiters_) {      // two steps left: buffer 0 current, stage 2 holds the last step
      step(0, a_reg2, b_reg2, T{});
      __syncthreads();
      step(1, a_reg, b_reg, F{});
    } else {
      step(0, a_reg, b_reg, F{});
    }
    __syncthreads();
  } else {
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int it = 0; it < kiters_; ++it) {
    const int cur = it & 1;
    if (it + 1 < kiters_) load_tiles(it + 1);
    const float* a = As + cur * BM * LDS_LD + a_frag;
    const float* b = Bs + cur * B_TILE + b_frag;
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
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
    if (it + 1 < kiters_) store_tiles(cur ^ 1);
    __syncthreads();
  }
  }

  if constexpr (F8 || H2) {
    const float dq = 1.f / (f8_sa * f8_sb);          // powers of two: exact
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] *= dq;
  }
  // ---- epilogue ----------------------------------------------------------------------------
  // (0) accumulate: the destination's current content joins the RAW accumulator, i.e. before the
  //     statistics and before scale/shift/activation (dX += ..., or a pre-filled per-image/per-position
  //     bias term of the fusion layer).
  if (p.accumulate) {
    const int hsws0 = Hs_ * Ws_;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (BM / WM) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int m = bm * BM + row;
        if (m >= M_) continue;
        size_t pix;
        if (p.dense_out) pix = (size_t)m;
        else {
          const int n = m / hsws0, rem = m - n * hsws0;
          const int i = rem / Ws_, jx = rem - i * Ws_;
          pix = ((size_t)n * p.Ho + oy0_ + i * p.osy) * p.Wo + ox0_ + jx * p.osx;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int co = bn * BN + wn * (BN / WN) + ni * 32 + (lane & 31);
          if (co < p.Co) acc[mi][ni][r] += gout[pix * p.ldo + co];
        }
      }
    }
  }
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
    for (int idx = tid; idx < 2 * BN; idx += 256) {
      const int which = idx / BN, col = idx - which * BN;
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
  float vmax = 0.f;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * (BM / WM) + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int m = bm * BM + row;
      if (m >= M_) continue;
      size_t pix;
      if (p.dense_out) {
        pix = (size_t)m;
      } else {
        const int n = m / hsws, rem = m - n * hsws;
        const int i = rem / Ws_, jx = rem - i * Ws_;
        pix = ((size_t)n * p.Ho + oy0_ + i * p.osy) * p.Wo + ox0_ + jx * p.osx;
      }
      const float rs = p.row_scale ? p.row_scale[(size_t)blockIdx.y * M_ + m] : 1.f;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        if (co_[ni] >= p.Co) continue;
        float v = acc[mi][ni][r] * rs * sc[ni] + sh[ni];
        if (p.act == DCN_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
        if (p.residual) v += p.residual[pix * p.ldr + co_[ni]];
        gout[pix * p.ldo + co_[ni]] = v;
        vmax = fmaxf(vmax, fabsf(v));
      }
    }
  }
  if (p.amax_out) {                       // abs-max of what was stored: the scale of the tensor's next consumer (order-independent)
    vmax = wave_max(vmax);
    if (lane == 0) amax_update(p.amax_out, vmax, blockIdx.x * 4 + wave);
  }
}

template <int BM, int BN, int WM, int WN, int BMODE, bool C4, int BK, bool SP = false, int ABL = 0, int NP = 3, int OCC = 1, bool F8 = false,
          bool BPRE = false>
int launch_bk(const IgemmParams& p0, hipStream_t stream) {
  IgemmParams p = p0;
  p.cpt = p.c4 ? 1 : p.Ci / BK;
  p.kiters = p.c4 ? 64 / BK : p.ntaps * p.cpt;
  constexpr int LDS_LD = BK + 4;
  const int gm = cdiv(p.M, BM) * (p.ncls ? p.ncls : 1), gn = cdiv(p.Co, BN);     // (classes: p.M = the largest class)
  constexpr int LD16 = (BK == 16 || F8) ? 16 : BK + 8;
  const size_t lds = SP ? (size_t)2 * NP * (BM * LD16 + (BMODE == 0 ? BN * LD16 : 2048) + (BPRE ? 32 : 0)) * sizeof(unsigned short)
                        : (size_t)2 * (BM * LDS_LD + (BMODE == 0 ? BN * LDS_LD : BK * BN)) * sizeof(float);
  static DcnPerDeviceFlag attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, WM, WN, BMODE, C4, BK, SP, ABL, NP, OCC, F8, BPRE>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int nb = p.batch > 0 ? p.batch : 1;
  // latency-bound little GEMMs (LSTM steps: 64 rows) are booked separately from the conv-stack tiles
  const int tag = SP ? (F8 ? 23 : NP == 1 ? 19 : NP == 2 ? (BMODE == 1 ? 27 : BN == 64 ? 26 : 24) : BMODE == 1 ? 21 : BN == 64 ? 18 : 16) : p.M < 1024 ? 13 : (BM == 128 && BN == 128 && BMODE == 0 && BK == 32) ? 15 : BM == 64 ? (BMODE == 1 ? 7 : 6) : (BMODE == 1 ? (BN == 128 ? 3 : 4) : (BN == 128 ? 0 : (BN == 64 ? 1 : 2)));
  double k_alg = p.c4 ? 27.0 : (double)p.ntaps * (p.bmode == 1 && p.kvalid > 0 ? p.kvalid : p.Ci);
  double mk = (double)p.M * k_alg;            // sum over classes of rows x K
  if (p.ncls) { mk = 0; for (int c = 0; c < p.ncls; ++c) mk += (double)p.cls_M[c] * p.cls_ntaps[c] * p.Ci; k_alg = 9.0 * p.Ci; }
  // algorithmic bytes: the gathered tensor once (rows actually addressed: N*Hi*Wi pixels of Ci), the filter bank, the output
  double m_out = p.M;
  if (p.ncls) { m_out = 0; for (int c = 0; c < p.ncls; ++c) m_out += p.cls_M[c]; }
  const double alg_bytes = 4.0 * nb * ((double)p.N * p.Hi * p.Wi * (p.c4 ? 4 : p.Ci) + (double)p.Co * k_alg + m_out * p.Co * epilogue_reads(p));
  const int pid = prof_begin(tag, 2.0 * nb * mk * p.Co, stream, alg_bytes);
  hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, BMODE, C4, BK, SP, ABL, NP, OCC, F8, BPRE>), dim3(gm * gn, nb), dim3(256), lds, stream, p);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("igemm");
  return DCN_OK;
}

// K-step choice (measured per layer, tools/bench_convs.py): 16 floats per step keeps the LDS footprint at
// ~40 KB so three workgroups share a CU — better latency hiding on the long-M layers (+3..+60 %, most on the
// narrow early layers) — while the short grids of the 13x13 maps (M <= 16 K rows) prefer fewer, longer steps.
int g_force_bk = 0;       // experiment knob (dcn_set_tuning("k", 16|32))
int g_split = 0;          // dcn_set_tuning("split", 16|32): force every NT tile onto the split-bf16 pipe (bench_convs A/B)
int g_nn_split = 1;       // dcn_set_tuning("nnsplit", 0): NN products back on the fp32 MFMA
int g_occ3 = 1 << 30;     // dcn_set_tuning("occ3", n): split launches of <= n K-steps use the 3-waves/SIMD build (A/B: 0 = never)
int g_abl = 0;            // dcn_set_tuning("abl", 1|2): timing-only ablations of the split kernel (results are wrong)
int g_precision = 4;      // dcn_set_tuning("precision", 0..4): 0 = fp32 MFMA everywhere; 1 = wide NT tiles of >= 1024 rows on the
                          // split-bf16 pipe (fp32 accuracy, six MFMAs per product); 2 = those tiles with plain bf16 operands
                          // (configs[2], reduced precision); 3 = fp8 operands (configs[4]); 4 (default) = the f16 two-piece split
                          // (fp32 accuracy, three MFMAs per product) wherever the operands carry their abs-max, else as 1
int g_h2_occ3 = 0;        // dcn_set_tuning("h2occ", 0): f16-split 128x128 tile built for 2 instead of 3 waves/SIMD
int g_h2_k32 = 128;       // dcn_set_tuning("gk32", n): the 256x32 tile takes the f16 split from this K on
int g_h2_narrow = 0;      // dcn_set_tuning("rnarrow", 1): the narrow NT tiles (128x64, 256x32, 64x128) on the f16 split as well
int g_h2_bk = 16;         // dcn_set_tuning("qbk", 32): K-step of the f16-split tiles
int g_h2_presplit = 1;    // dcn_set_tuning("ypresplit", 0): filter banks split inside every workgroup again

template <int BM, int BN, int WM, int WN, int BMODE, bool C4 = false>
int launch_variant(const IgemmParams& p, hipStream_t stream) {
  const long long rows = (long long)p.M * (p.batch > 0 ? p.batch : 1);       // batched GEMMs fill the chip like one long M
  const int bk = g_force_bk ? g_force_bk : (rows <= 16384 ? 32 : 16);
  if constexpr (BMODE == 0 && !C4) {
    // narrow tiles gain nothing from the split (its vector-ALU cost per MFMA grows as the tile shrinks:
    // measured 0.6-1.0x on the 128x64 / 256x32 tiles, 1.4-1.8x on 128x128)
    // (the 256x64 tile is the same 64x64-per-wave body as 128x128 with 25 % more split work per MFMA)
    if ((p.f8 || (g_precision == 3 && p.amax_a && p.amax_b)) && ((BM == 128 && BN == 128) || (BM == 256 && BN == 64)) && rows >= 1024)
      return launch_bk<BM, BN, WM, WN, BMODE, C4, 32, true, 0, 1, 1, true>(p, stream);  // fp8 e4m3 operands
    // (the 256x32 tile — data gradients towards 32 channels on the 416/208 maps — only where the K loop is long: 3x3 s1
    //  64->32 @208 1.10 -> 0.74 ms, the stride-2 classes @416 1.94 -> 1.62 in sum; a 1-tap K = 32..64 launch loses 20 %)
    if (g_precision == 4 && p.amax_a && (p.amax_b || p.b_scale) && rows >= 1024 &&
        ((BM == 128 && BN == 128) || (BM == 256 && BN == 64) || (BM == 256 && BN == 32 && p.ntaps * p.Ci >= g_h2_k32) || g_h2_narrow)) {
      // f16 two-piece split (fp32 accuracy, three MFMAs per product): launches whose operands carry their abs-max
      if constexpr (BM == 128 && BN == 128) {
        if (p.b_scale && g_abl == 2) return launch_bk<BM, BN, WM, WN, BMODE, C4, 16, true, 2, 2, 1, false, true>(p, stream);
        if (p.b_scale && g_abl == 3) return launch_bk<BM, BN, WM, WN, BMODE, C4, 16, true, 3, 2, 1, false, true>(p, stream);
        if (p.b_scale && g_abl == 4) return launch_bk<BM, BN, WM, WN, BMODE, C4, 16, true, 4, 2, 1, false, true>(p, stream);
      }
      if (p.b_scale) return launch_bk<BM, BN, WM, WN, BMODE, C4, 16, true, 0, 2, 1, false, true>(p, stream);    // pre-split filter bank
      if (g_h2_bk == 32) return launch_bk<BM, BN, WM, WN, BMODE, C4, 32, true, 0, 2>(p, stream);
      if (BM == 128 && BN == 128 && g_h2_occ3) return launch_bk<BM, BN, WM, WN, BMODE, C4, 16, true, 0, 2, 3>(p, stream);
      return launch_bk<BM, BN, WM, WN, BMODE, C4, 16, true, 0, 2>(p, stream);
    }
    if (g_precision >= 2 && g_precision <= 3 && ((BM == 128 && BN == 128) || (BM == 256 && BN == 64)) && rows >= 1024)
      return launch_bk<BM, BN, WM, WN, BMODE, C4, 32, true, 0, 1>(p, stream);     // bf16 operands: 8 MFMAs per 32-wide K-step
    if (g_split || ((g_precision == 1 || g_precision == 4) && ((BM == 128 && BN == 128) || (BM == 256 && BN == 64)) && rows >= 1024)) {
      if (g_split == 32) return launch_bk<BM, BN, WM, WN, BMODE, C4, 32, true>(p, stream);
      if (g_abl == 1) return launch_bk<BM, BN, WM, WN, BMODE, C4, 16, true, 1>(p, stream);   // ablation: no split arithmetic (wrong results)
      if (g_abl == 2) return launch_bk<BM, BN, WM, WN, BMODE, C4, 16, true, 2>(p, stream);   // ablation: 1 of 6 MFMA groups (wrong results)
      if (BM == 128 && BN == 128 && (p.c4 ? 4 : p.ntaps * (p.Ci / 16)) <= g_occ3)     // (256x64 tile: 10-25 % slower at 3)
        return launch_bk<BM, BN, WM, WN, BMODE, C4, 16, true, 0, 3, 3>(p, stream);
      return launch_bk<BM, BN, WM, WN, BMODE, C4, 16, true>(p, stream);
    }
  }
  if constexpr (BMODE == 1 && BM == 128 && BN == 128) {
    // NN products of the co-attention (E.f2, dA.f2): the same split body with transposed B fragments
    if (g_precision == 4 && p.amax_a && p.amax_b && g_nn_split && rows >= 1024) return launch_bk<BM, BN, WM, WN, BMODE, C4, 16, true, 0, 2, 3>(p, stream);
    if ((g_precision == 1 || g_precision == 4) && g_nn_split && rows >= 1024) return launch_bk<BM, BN, WM, WN, BMODE, C4, 16, true, 0, 3, 3>(p, stream);
  }
  if (bk == 32) return launch_bk<BM, BN, WM, WN, BMODE, C4, 32>(p, stream);
  return launch_bk<BM, BN, WM, WN, BMODE, C4, 16>(p, stream);
}

// Tile choice.  Narrow-N tiles for the 32/64-channel layers so no MFMA column is wasted; for wide layers
// the 128x128 tile unless the grid would leave most of a round of the workgroup slots (256 CUs x 2-3
// resident workgroups) empty: the 13x13 and 26x26 maps give 340-680 tiles of 128 rows, i.e. 66 % fill, and a
// 64x128 tile (measured ~0.9x the per-tile efficiency) doubles the workgroup count.
inline double fill(long long blocks, long long slots) { return (double)blocks / (double)(((blocks + slots - 1) / slots) * slots); }
int g_force_bm = 0;       // experiment knob, set through dcn_set_tuning (tools/bench_convs.py)
int g_tile64 = 1;         // dcn_set_tuning("tile64", 0): 64-channel layers back on the fp32-pipe 128x64 tile

// (A/B per layer, tools/bench_convs.py --ab tile64=0: the 256x64 split tile wins 10-30 % where the K loop is long —
//  3x3 layers and their data gradients — and loses on 1-tap problems with K <= 128, which stay on the fp32 pipe.)
inline int tile_bm(int M, int Co, int ntaps, int Ci) {
  if (Co <= 32) return 256;
  if (Co <= 64) return (g_tile64 && g_precision >= 1 && !g_force_bm && ntaps >= 2 && ntaps * Ci >= 256) ? 256 : 128;
  if (g_force_bm) return g_force_bm;
  // In-process A/B (tools/bench_convs.py --ab bm=64 / bm=128): with the current K-steps the 64-row tile only
  // wins by 3-4 % on the 3x3 layers of the 13x13 maps and loses 10-70 % everywhere else, so it is kept as a
  // knob (dcn_set_tuning("bm", 64)) but not selected automatically.
  (void)M;
  return 128;
}

}  // namespace

int igemm_grid_m(int M, int Co, int ntaps) { return cdiv(M, tile_bm(M, Co, ntaps, 32)); }   // Ci >= 32 on every multi-tap path

void wgrad_set_split(int v);
void wgrad_set_lds_pad(int kb);
void wgrad_set_abl(int v);
void wgrad_set_target(int v);
void wgrad_set_wide64(int v);
void wgrad_set_target_small(int v);
void wgrad_set_w3_b16(int v);
void wgrad_set_w9_b16(int v);
void wgrad_set_target_b16(int v, int small);
void wgrad3_set_tuning(int key, int value);
void wgrad_set_w1x(int v);
void bn_set_pc(int v);
void conv_set_merge(int v);
void conv_set_d2_b16(int v);
void conv_set_n1_b16(int v);
void score_set_tuning(int key, int value);
void bn_set_tuning(int v);
void wgrad9_set_tuning(int key, int value);
void wgrad_set_slab_fold(int v);

extern "C" int dcn_set_tuning(const char* key, int value) {
  const char k = key ? key[0] : 0;
  if (k == 'S' && key[1] == 'l') { wgrad_set_slab_fold(value); return DCN_OK; }         // "Slabfold": split-K slabs summed by the last-arriving workgroup (slabsum.h; 0 = reduce_slabs_kernel behind the launch)
  if (k == '2') { conv2b_set_tuning(value); return DCN_OK; }                             // "2btile": min 256 x 256 tiles for conv2b.hip (0 = off)
  if (k == 'H') { gemm3_set_h1(value); return DCN_OK; }                                  // "H1gemm3": one f16 piece per operand in the bf16 modes (gemm3.hip)
  if (k == 'q' && key[1] == 't') { wgrad_set_target_b16(value, 0); return DCN_OK; }     // "qtargetb16": workgroups a bf16-storage 3x3 stride-1 weight gradient aims for
  if (k == 'q' && key[1] == 's') { wgrad_set_target_b16(value, 1); return DCN_OK; }     // "qsmallb16": the same for its 1x1 / stride-2 layers
  if (k == 'w' && key[1] == '3') { wgrad_set_w3_b16(value); return DCN_OK; }           // "w3b16": bf16-storage 3x3 weight gradients by filter rows (wgrad3.hip)
  if (k == 'b' && key[1] == 'w') { conv1_set_tuning(4, value); return DCN_OK; }       // "bwide": conv1b 128 x 256 tiles from n workgroups on
  if (k == 'b' && key[1] == 't') { conv1_set_tuning(5, value); return DCN_OK; }       // "btall": conv1b 256 x 128 tiles from n workgroups on
  if (k == '1') { conv1_set_tuning(key[1] == 's' ? 1 : (key[1] == 'f' ? 2 : (key[1] == 'w' ? 3 : (key[1] == 't' ? 6 : 0))), value); return DCN_OK; }   // "1x1dma" (0/1/2), "1stages" (10 SA + SB), "1fill"
  if (k == '3') { if (key[1] == 'h') { conv3b_set_tuning(value); return DCN_OK; }
    if (key[1] == 'd') { conv3x_set_tuning(value); return DCN_OK; }      // "3dma": conv3x.hip's filter tiles by LDS-DMA (1) / through registers (0)      // "3h16": bf16-storage strip kernel (0 = gathered tiles)
    conv3_set_tuning(key[1] == 'b' ? 1 : key[1] == 'a' ? 2 : key[1] == 'l' ? 3 : key[1] == 'm' ? 4 : 0, value); return DCN_OK; }   // "3x3strip" (0/1), "3bm" (0/128/256), "3abl", "3ls", "3m16"
  if (k == 'N' && key[1] == 'b') { conv_set_n1_b16(value); return DCN_OK; }           // "Nb16": bf16-storage register-bank forward / data gradient 32 <-> 64 (nconv.hip)
  if (k == 'D') { conv_set_d2_b16(value); return DCN_OK; }                            // "Db16": bf16-storage register-bank stride-2 data gradient (nconv.hip)
  if (k == '9' && key[1] == 'b') { wgrad_set_w9_b16(value); return DCN_OK; }          // "9b16": bf16-storage nine-tap weight gradient (wgrad9.hip)
  if (k == '9') { wgrad9_set_tuning(key[1] == 't' && key[2] == 'a' && key[3] == 'r' ? 1 : 0, value); return DCN_OK; }   // "9tap" (0/1), "9target"
  if (k == 'G') { gemm3_set_tuning(value); return DCN_OK; }       // "Gemm3": the co-attention products on pre-split operands (gemm3.hip)
  if (k == 'N') { nconv_set_tuning(value); return DCN_OK; }       // "Nconv": register-bank kernels of the 32 <-> 64 channel layers (nconv.hip)
  if (k == 'd') { bn_set_tuning(value); return DCN_OK; }          // "dbnrev": sweep direction of the BatchNorm streaming passes (bn.hip)
  if (k == 'e') { score_set_tuning(0, value); return DCN_OK; }    // "e2rpw": rows per wave of l2norm_score_fwd
  if (k == 'f') { score_set_tuning(1, value); return DCN_OK; }    // "f2nt": non-temporal loads there
  if (k == 'l') { wgrad_set_lds_pad(value); return DCN_OK; }      // "lwgpad": KB of LDS a weight-gradient launch reserves at least (occupancy experiment)
  if (k == 'j') { stem_set_tuning(value); return DCN_OK; }        // "jstem": the stem directly on the vector ALU (stem.hip)
  if (k == 'm') { conv_set_merge(value); return DCN_OK; }         // "merge": parity classes of a stride-2 data gradient in one launch
  if (k == 'Y') { wgrad_set_w1x(value); return DCN_OK; }          // "Y1wide": 1x1 stride-1 weight gradients on the 256-wide tile (wgrad.hip wgrad1x_kernel; 0 = off, n > 1: workgroups aimed for)
  if (k == 'U') { wgrad3_set_tuning(2, value); return DCN_OK; }   // "U3m16": wgrad3.hip on 16x16x32 MFMAs (1) / 32x32x16 (0)
  if (k == 'u') { wgrad3_set_tuning(0, value); return DCN_OK; }   // "u3row": 3x3 stride-1 weight gradient by filter rows (wgrad3.hip)
  if (k == 'v') { wgrad3_set_tuning(1, value); return DCN_OK; }   // "v3target"
  if (k == 'B') { bn_set_pc(value); return DCN_OK; }              // "Bpc": BatchNorm apply passes with the channel fixed per thread (bn.hip; 0 = grid-stride forms)
  if (k == 'z') { wgrad_set_target_small(value); return DCN_OK; }   // "zwgsmall"
  if (k == 'c') { wgrad_set_wide64(value); return DCN_OK; }        // "cwide64"
  if (k == 'x') { wgrad_set_target(value); return DCN_OK; }   // "xwgtarget"
  if (k == 'w') { wgrad_set_split(value); return DCN_OK; }   // "wsplit": weight-gradient 128x128 tiles on the split-bf16 pipe
  if (k == 'p') { g_precision = value; wgrad_set_split(value == 3 ? 2 : value); return DCN_OK; }   // 3 (fp8): weight gradient with bf16 operands   // "precision": 0 native fp32 MFMA, 1 split-bf16 on the wide tiles
  if (k == 'b') g_force_bm = value;          // "bm": force the M tile (0 = automatic)
  else if (k == 'k') g_force_bk = value;     // "k": force the K-step (16 or 32, 0 = automatic)
  else if (k == 'n') g_nn_split = value;     // "nnsplit"
  else if (k == 'o') g_occ3 = value;         // "occ3"
  else if (k == 'h') g_h2_occ3 = value;      // "h2occ"
  else if (k == 'g') g_h2_k32 = value;       // "gk32"
  else if (k == 'r') g_h2_narrow = value;    // "rnarrow"
  else if (k == 'q') g_h2_bk = value;        // "qbk"
  else if (k == 'y') g_h2_presplit = value;  // "ypresplit"
  else if (k == 't') g_tile64 = value;       // "tile64"
  else if (k == 'a') { g_abl = value; wgrad_set_abl(value); }         // "abl"
  else if (k == 's') g_split = value;        // "split": 0 = fp32 MFMA, 16 / 32 = split-bf16 MFMA with that K-step
  else { dcn_set_error("set_tuning: unknown key"); return DCN_ERR_ARG; }
  return DCN_OK;
}

int igemm_precision() { return g_precision; }

bool igemm_will_presplit(long long rows, int Co, int ntaps, int Ci) {
  if (g_precision != 4 || !g_h2_presplit || g_split || rows < 1024 || Co <= 32) return false;
  if (Co <= 64) return tile_bm((int)rows, Co, ntaps, Ci) == 256;       // the 256x64 split tile
  return tile_bm((int)rows, Co, ntaps, Ci) == 128;                     // the 128x128 tile
}

bool igemm_tap_capable(const IgemmParams& p) {
  if (!p.stats || p.ncls || p.batch > 1 || !p.dense_out) return false;
  const int gran = tile_bm(p.M, p.Co, p.ntaps, 32);
  return conv3_applicable(p, g_precision, gran) || conv1_applicable(p, g_precision, gran);
}

int igemm_launch(const IgemmParams& p, hipStream_t stream) {
  DCN_CHECK_ARG(p.in && p.wt && p.out, "igemm: null pointer");
  DCN_CHECK_ARG(p.M > 0 && p.Co > 0 && p.Ci > 0, "igemm: empty problem (M=%d Co=%d Ci=%d)", p.M, p.Co, p.Ci);
  DCN_CHECK_ARG(p.c4 ? (p.Ci == 4) : (p.Ci % 32 == 0), "igemm: Ci=%d must be a multiple of 32 (or 4 in c4 mode)", p.Ci);
  DCN_CHECK_ARG(p.ntaps >= 1 && p.ntaps <= IGEMM_MAX_TAPS, "igemm: ntaps=%d", p.ntaps);
  DCN_CHECK_ARG(p.ldi % 4 == 0 && p.ldw % 4 == 0, "igemm: ldi=%d ldw=%d must be multiples of 4 floats", p.ldi, p.ldw);
  DCN_CHECK_ARG(((uintptr_t)p.in & 15) == 0 && ((uintptr_t)p.wt & 15) == 0, "igemm: in/wt must be 16-byte aligned");
  DCN_CHECK_ARG(p.stats == nullptr || p.batch <= 1, "igemm: stats are not supported on batched launches");
  DCN_CHECK_ARG(p.b_scale == nullptr || (p.amax_a && p.bmode == 0 && !p.c4 &&
                                         igemm_will_presplit((long long)p.M * (p.batch > 0 ? p.batch : 1), p.Co, p.ntaps, p.Ci)),
                "igemm: a pre-split filter bank on a launch whose tile cannot read it (M=%d Co=%d taps=%d)", p.M, p.Co, p.ntaps);
  // 32-bit offset windows: one M-tile spans at most ceil(BM/(Hs*Ws))+1 images of the gathered tensor
  {
    const long long img_bytes = (long long)p.Hi * p.Wi * p.ldi * 4;
    const long long span = (256 / (p.Hs * p.Ws) + 2) * img_bytes;
    DCN_CHECK_ARG(span < 0x7FFFFFF0LL, "igemm: image too large for 32-bit buffer offsets (%lld bytes per M-tile window)", span);
    DCN_CHECK_ARG((long long)(p.bmode == 1 ? p.kvalid : p.Co) * p.ldw * 4 < 0x7FFFFFF0LL, "igemm: filter bank exceeds 2 GB");
  }
  {
    // 3x3 stride-1 launches with a pre-split filter bank: the strip kernel of conv3.hip (activations staged once per 16 channels)
    const int gran = tile_bm(p.M, p.Co, p.ntaps, 32);
    if (conv3_applicable(p, g_precision, gran)) return conv3_launch(p, gran, stream);
    // 1x1 layers and their data gradients (plain GEMM rows, pre-split bank): both tiles by LDS-DMA, conv1.hip
    if (conv1_applicable(p, g_precision, gran)) return conv1_launch(p, gran, stream);
  }
  DCN_CHECK_ARG(!p.bt_y, "igemm: a BatchNorm tap on a launch outside conv1.hip / conv3.hip (ask igemm_tap_capable first)");
  if (p.bmode == 1) {
    DCN_CHECK_ARG(p.ntaps == 1 && !p.c4 && p.Co % 4 == 0, "igemm: NN mode needs one tap and Co %% 4 == 0 (Co=%d)", p.Co);
    if (p.Co <= 64) return launch_variant<128, 64, 2, 2, 1>(p, stream);
    if (tile_bm(p.M, p.Co, p.ntaps, p.Ci) == 64) return launch_variant<64, 128, 2, 2, 1>(p, stream);
    return launch_variant<128, 128, 2, 2, 1>(p, stream);
  }
  if (p.c4) {
    DCN_CHECK_ARG(p.Co <= 32 && p.ntaps == 9 && p.bmode == 0, "igemm: c4 (stem) path needs Co <= 32 and 9 taps");
    return launch_variant<256, 32, 4, 1, 0, true>(p, stream);
  }
  if (p.Co <= 32) return launch_variant<256, 32, 4, 1, 0>(p, stream);
  if (p.Co <= 64) {
    // 64 output channels: four waves stacked along M (each 64x64) on the split pipe; fp32-pipe 128x64 tile otherwise
    if (tile_bm(p.M, p.Co, p.ntaps, p.Ci) == 256) return launch_variant<256, 64, 4, 1, 0>(p, stream);
    return launch_variant<128, 64, 2, 2, 0>(p, stream);
  }
  if (tile_bm(p.M, p.Co, p.ntaps, p.Ci) == 64) return launch_variant<64, 128, 2, 2, 0>(p, stream);
  return launch_variant<128, 128, 2, 2, 0>(p, stream);
}
