// conv3.hip's 3x3 stride-1 strip convolution on v_mfma_f32_16x16x32_f16 (round 5).
//
// Why a second kernel for the same launches: the f16-split loops of this engine are POWER-bound before they are issue-bound — the chip holds
// its clock down under dense MFMA work on random data — and the clock it holds depends on the MFMA shape (MI355X_MICROARCH.md, 'DVFS
// give-back' item 7).  tools/mfma_shape_probe.hip, this engine's arithmetic (three f16 MFMAs per product, every fragment re-read from LDS,
// 64 x 64 per wave, no global traffic): 32x32x16 476-480 TFLOP/s at 1.49 GHz in-kernel, 16x16x32 543-548 at 1.69 — x1.14 at equal cycles.
//
// What K = 32 per MFMA means for a strip kernel whose unit of staging is 16 channels: the four k-groups of a lane quad are
//      lanes  0-31 (k 0-15):  tap j of filter row R0, 16 channels          lanes 32-63 (k 16-31): tap j of filter row R1, 16 channels
// — TWO TAPS of the same 16-channel step (or of two consecutive ones) per MFMA, chosen per lane by its LDS addresses alone: the strip and
// filter images in LDS are conv3.hip's (32-byte rows of 16 k, two f16 planes), nothing is staged twice.  A super-iteration is two filter rows
// (six taps, 144 MFMAs per wave between barriers); three of them cover two channel steps:
//      T0: (c0, row 0) + (c0, row 1)        T1: (c0, row 2) + (c1, row 0)        T2: (c1, row 1) + (c1, row 2)
// The strip of channel step c0 + 2 is loaded during T1 and split into c0's buffer during T2; that of c1 + 2 is loaded during T2 and stored
// during the next T0.  Filter tiles: six tap slots per stage, two stages, loaded two super-iterations ahead through registers.
// Tile: 256 pixels x 128 filters, eight waves (4 x 2) of 64 x 64 = 4 x 4 accumulator blocks of 16 x 16; one workgroup per CU (LDS: strip
// planes of a fixed 468 positions — 2 x 2 x 14.6 KB — + filters 2 x 6 x 8.1 KB = 156 KB).
// LDS images without swizzles: a ds_read_b128 lane group {0-3, 12-15, 20-27} of this shape reads rows r, r + 8 in DIFFERENT 16-byte halves.
// Filters are staged PERMUTED — LDS row 16 jn + c of a wave's 64 filters holds filter 4 c + jn — so that a lane's four accumulator blocks
// along N are four consecutive filters: every epilogue access (store, shortcut, accumulated-onto tensor, tapped BatchNorm input) is a
// 16-byte vector per lane, 256 contiguous bytes per 16 lanes.
// Order of the three terms: (l,h) (h,h) (h,l) — each fragment set dies one term-group before the next pair needs its registers, so ONE set
// of fragments (64 registers) rotates, every LDS read issued >= 16 MFMAs ahead of its use.
// Roofline: MFMA, 838.9 TFLOP/s (three f16 MFMAs per product), as conv3.hip.
#include "igemm.h"
#include "prof.h"
#include <type_traits>

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned OOBX = 0x80000000u;

constexpr int XBM = 256, XBN = 128, XNT = 512, XWM = 4;
constexpr int XPB = XBN * 32;            // bytes per filter plane of one tap
constexpr int XPB1 = XPB + 64;           // plane 1 sits 64 B past a multiple of 128 (conv3.hip: ds_write_b128 banks)
constexpr int XSLOT = 2 * XPB + 128;     // bytes per tap slot
constexpr int XSTAGE = 6 * XSLOT;        // six taps (two filter rows) per stage
constexpr int XA_LD = 4;                 // 16-byte strip pieces per thread and 16-channel step
// timing ablations (builds with -DC3X_ABL=bits only; results are WRONG): 1 = no filter loads in the loop, 2 = no strip loads, 4 = no LDS stores in
// the loop (the prologue's real data stays in LDS: zeros would raise the clock by themselves), 8 = no barrier in the loop
#ifndef C3X_ABL
#define C3X_ABL 0
#endif
constexpr int XSMAX = 466;               // longest strip (positions): 256 + 2 W + 2 with W <= 104; position XSMAX is the row of zeros, XSMAX + 1 the dump row
constexpr int XPA = (XSMAX + 2) * 32;    // bytes per strip plane — a constant, so that buffer / plane / zero-row offsets are instruction immediates

__device__ __forceinline__ f32x4 ldx16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrcx(const float* base, long long bytes) {
  const unsigned n = bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0u : (bytes < 0 ? 0u : (unsigned)bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, n, 0x00020000);
}
__device__ __forceinline__ float pow2_scalex(unsigned amax_bits) {      // = igemm.hip pow2_scale
  const int be = (int)((amax_bits >> 23) & 0xFF);
  if (be == 0 || be == 255) return 1.f;
  int e = 14 - (be - 126);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return __uint_as_float((unsigned)(e + 127) << 23);
}

// DMA: the filter tiles go global -> LDS by `buffer_load_dwordx4 ... lds` (no registers, no ds_write_b128: the ablations of the register-staged
// build, tools/ab_c3x.sh, priced its six LDS stores per thread and super-iteration at 9-10 % of the kernel and the loads at 7 %).  A wave
// brings one 1-KiB piece (32 filter rows of one plane) of every tap slot: tile k + 2 goes into the stage super-iteration k has just finished
// reading — slots 0, 1 behind its last two MFMA groups (after its barrier), slots 2-5 behind the first four groups of k + 1 — and has landed
// (counted s_waitcnt vmcnt in front of the raw s_barrier) before k + 1's barrier.
typedef __attribute__((address_space(3))) void lds_voidx;
template <bool DMA>
__global__ __launch_bounds__(XNT, 2) void conv3x_kernel(const IgemmParams p, const int S, const int gran) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemx[];
  constexpr int PA = XPA;
  unsigned char* const Abase = smemx;                      // [2 buffers][2 planes][PA]
  unsigned char* const Bbase = smemx + 4 * PA;             // [2 stages][6 taps][XSLOT]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c16 = lane & 15, g = lane >> 4, tsel = g >> 1, half = g & 1;
  const int gn = (p.Co + XBN - 1) / XBN;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int bm = lin / gn, bn = lin - bm * gn;
  const int W = p.Wi, H = p.Hi, M = p.M;
  const int m0 = bm * XBM;
  const float sa = pow2_scalex(amax_read(p.amax_a)), sb = p.b_scale[0];

  // ---- descriptors ---------------------------------------------------------------------------------
  const int lin0 = m0 - W - 1;                                     // pixel of strip position 0
  const int base_px = lin0 > 0 ? lin0 : 0;
  const float* a_base = p.in + (long long)base_px * p.ldi;
  const __amdgpu_buffer_rsrc_t a_rs = rsrcx(a_base, ((long long)(M - base_px - 1) * p.ldi + p.Ci) * 4);
  const __amdgpu_buffer_rsrc_t b_rs = rsrcx(p.wt, (long long)p.Co * p.ldw * 4);

  // ---- per-thread staging state --------------------------------------------------------------------
  // strip piece j of this thread: position pos0 + 128 j, 16-byte quarter q of its 64 bytes; global byte offset at channel step 0 / LDS byte
  // offset inside plane 0.  Only piece 0's values are kept: the others are + j x (a constant) where the piece exists (a lane mask per j, in
  // scalar registers), the out-of-range offset / the dump row where it does not.
  const int pos0 = tid >> 2, q0 = tid & 3;
  const unsigned a_off0 = (unsigned)((lin0 + pos0 - base_px) * p.ldi * 4 + q0 * 16);     // (meaningless where piece 0 does not exist: not used there)
  const int a_st0 = pos0 * 32 + q0 * 8;
  const unsigned a_pstride = (unsigned)(128 * p.ldi * 4);
  bool a_ok[XA_LD], a_in[XA_LD];
#pragma unroll
  for (int j = 0; j < XA_LD; ++j) {
    const int pos = pos0 + 128 * j, px = lin0 + pos;
    a_in[j] = pos < S;
    a_ok[j] = pos < S && px >= 0 && px < M;
  }
  auto a_off = [&](const int j) {
    unsigned o = a_off0;
    asm volatile("" : "+v"(o));                 // (loop-invariant otherwise: hoisted back into four registers)
    return a_ok[j] ? o + (unsigned)j * a_pstride : OOBX;
  };
  auto a_st = [&](const int j) {
    int o = a_st0;
    asm volatile("" : "+v"(o));
    return a_in[j] ? o + j * 4096 : (XSMAX + 1) * 32 + q0 * 8;                            // (beyond the strip -> dump row)
  };
  unsigned b_off; int b_st;
  if constexpr (DMA) {
    // this wave's piece of a tap slot: plane wave & 1, LDS rows 32 (wave >> 1) + (lane >> 1), k-half lane & 1 — lane-linear in LDS
    const int plane = wave & 1, rho = 32 * (wave >> 1) + (lane >> 1), kh = lane & 1;
    const int filt = (rho & ~63) + 4 * (rho & 15) + ((rho & 63) >> 4);               // the filter LDS row rho holds
    const int co = bn * XBN + filt;
    b_off = co < p.Co ? (unsigned)(co * p.ldw * 4 + (2 * kh + plane) * 16) : OOBX;
    b_st = __builtin_amdgcn_readfirstlane(plane * XPB1 + (wave >> 1) * 1024);       // (wave-uniform: the LDS base of the piece inside a slot)
  } else {
    const int rho = tid >> 2, chunk = tid & 3;                      // LDS row / 16-byte chunk of the pre-split 64 bytes
    const int filt = (rho & ~63) + 4 * (rho & 15) + ((rho & 63) >> 4);               // the filter LDS row rho holds
    const int co = bn * XBN + filt;
    b_off = co < p.Co ? (unsigned)(co * p.ldw * 4 + chunk * 16) : OOBX;
    b_st = (chunk & 1) * XPB1 + rho * 32 + (chunk >> 1) * 16;                        // plane = chunk & 1, k-half = chunk >> 1
  }
  // rows of this lane (one per accumulator block along M): in-image tap masks
  unsigned msk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + c16;
    unsigned v = 0;
    if (m < M) {
      const int rem = m % (H * W), y = rem / W, x = rem - y * W;
      for (int t = 0; t < 9; ++t)
        if ((unsigned)(y + p.tap_dy[t]) < (unsigned)H && (unsigned)(x + p.tap_dx[t]) < (unsigned)W) v |= 1u << t;
    }
    msk[i] = v;
  }
  const int a_lane0 = (wm * 64 + c16) * 32 + half * 16;                              // strip byte of (block 0, shift 0), plane 0
  const int b_lane0 = (wn * 64 + c16) * 32 + half * 16 + tsel * 3 * XSLOT;           // filter byte of block 0 in this lane's filter row

  const int nch = p.Ci >> 4;                  // 16-channel steps (even: Ci % 32 == 0)
  const int nper = nch >> 1;                  // periods of three super-iterations
  const int nsi = 3 * nper;

  f32x4 a_reg[XA_LD], b_reg[6];
  // DMA builds issue the two stores from inline assembly: an LDS store the compiler can see gets an s_waitcnt vmcnt(0) in front of it while
  // LDS-DMA pieces are in flight (it cannot tell that the piece and the store do not overlap), i.e. a wait for filter pieces issued moments ago
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_voidx*)smemx;
  auto store_a_piece = [&](const int buf_off, int j) {               // x*s = h + l, two f16 planes
    const f32x4 t = a_reg[j] * sa;
    const f16x4_t h = {(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
    const f16x4_t l = {(_Float16)(t[0] - (float)h[0]), (_Float16)(t[1] - (float)h[1]), (_Float16)(t[2] - (float)h[2]),
                       (_Float16)(t[3] - (float)h[3])};
    if constexpr (DMA) {
      const unsigned addr = lds0 + (unsigned)(buf_off + a_st(j));
      asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:%3"
                   :: "v"(addr), "v"(__builtin_bit_cast(unsigned long long, h)), "v"(__builtin_bit_cast(unsigned long long, l)), "n"(XPA));
    } else {
      unsigned char* abuf = Abase + buf_off;
      *reinterpret_cast<uint2*>(abuf + a_st(j)) = __builtin_bit_cast(uint2, h);
      *reinterpret_cast<uint2*>(abuf + PA + a_st(j)) = __builtin_bit_cast(uint2, l);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue --------------------------------------------------------------------------------------
  if (tid < 8) {                              // the zero rows: [buffer][plane] x two 16-B halves
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(Abase + (tid >> 1) * PA + XSMAX * 32 + (tid & 1) * 16) = z;
  }
#pragma unroll
  for (int j = 0; j < XA_LD; ++j) a_reg[j] = ldx16(a_rs, a_off(j), 0u);
  auto dma = [&](const int stage_off, const int slot, const bool live, const unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (lds_voidx*)(Bbase + stage_off + slot * XSLOT + b_st), 16, (int)(live ? b_off : OOBX),
                                             (int)(live ? soff : 0u), 0, 0);
  };
  if constexpr (DMA) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {             // super-iteration 0: rows (0, 0), (0, 1) -> stage 0
      const int G = q / 3, j = q - 3 * G;
      dma(0, q, true, (unsigned)p.tap_w[3 * G + j] * 4u);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)               // super-iteration 1, row (0, 2): slots 0, 1 -> stage 1 (slots 2-5 follow behind the first groups of the loop)
      dma(XSTAGE, q, true, (unsigned)p.tap_w[6 + q] * 4u);
#pragma unroll
    for (int j = 0; j < XA_LD; ++j) store_a_piece(0, j);
#pragma unroll
    for (int j = 0; j < XA_LD; ++j) a_reg[j] = ldx16(a_rs, a_off(j), 64u);          // channel step 1: stored during the first T0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                // (the pieces have landed)
  } else {
#pragma unroll
    for (int q = 0; q < 6; ++q) {             // super-iteration 0: rows (0, 0), (0, 1)
      const int G = q / 3, j = q - 3 * G;
      b_reg[q] = ldx16(b_rs, b_off, (unsigned)p.tap_w[3 * G + j] * 4u);
    }
#pragma unroll
    for (int j = 0; j < XA_LD; ++j) store_a_piece(0, j);
#pragma unroll
    for (int j = 0; j < XA_LD; ++j) a_reg[j] = ldx16(a_rs, a_off(j), 64u);          // channel step 1: stored during the first T0
#pragma unroll
    for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(Bbase + q * XSLOT + b_st) = b_reg[q];
#pragma unroll
    for (int q = 0; q < 6; ++q) {             // super-iteration 1: rows (0, 2), (1, 0) — stored during super-iteration 0
      const int G = q < 3 ? 2 : 0, cc = q < 3 ? 0 : 1, j = q % 3;
      b_reg[q] = ldx16(b_rs, b_off, (unsigned)(p.tap_w[3 * G + j] + cc * 16) * 4u);
    }
  }
  __syncthreads();

  // ---- main loop ---------------------------------------------------------------------------------------
  f16x8_t Al[4], Ah[4], Bh[4], Bl[4];
  int ad[4];
  // plane-0 strip addresses of pair J of a super-iteration of type T, one per accumulator block along M
  auto a_addr = [&](const int T, const int J) {
    const int Ga = T == 0 ? 0 : (T == 1 ? 2 : 1), Gb = T == 0 ? 1 : (T == 1 ? 0 : 2);
    const int ta = 3 * Ga + J, tb = 3 * Gb + J;
    const int sha = (p.tap_dy[ta] + 1) * W + p.tap_dx[ta] + 1, shb = (p.tap_dy[tb] + 1) * W + p.tap_dx[tb] + 1;
    // (everything below is loop-invariant per (T, J): hoisted, the shifts, addresses and tap masks of all nine pairs would live in registers
    //  across the loop and spill — the opaque copy of the lane's tap selector keeps the ~8 instructions here)
    int ts = tsel;
    asm volatile("" : "+v"(ts));
    const int sh = ts ? shb : sha;
    const int tl = ts ? tb : ta;
    const int bo = T == 0 ? 0 : (T == 2 ? 2 * PA : ts * (2 * PA));
    const int base = bo + a_lane0 + sh * 32;
    const int z = bo + XSMAX * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) ad[i] = ((msk[i] >> tl) & 1) ? base + i * 512 : z;
  };
  auto read_Al = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) Al[i] = *reinterpret_cast<const f16x8_t*>(Abase + PA + ad[i]);
  };
  auto read_Ah = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) Ah[i] = *reinterpret_cast<const f16x8_t*>(Abase + ad[i]);
  };
  // filter stage read by the current super-iteration, as a byte offset (0 | XSTAGE): a scalar that flips every super-iteration — a period has
  // three of them, so a compile-time stage would ask for two periods per loop trip (and a loop exit in the middle)
  int st_cur = 0;
  auto read_B = [&](f16x8_t (&B)[4], const int stage_off, const int J, const int plane) {
    const unsigned char* bb = Bbase + stage_off + b_lane0;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) B[jn] = *reinterpret_cast<const f16x8_t*>(bb + J * XSLOT + plane * XPB1 + jn * 512);
  };
  auto mm = [&](const f16x8_t (&A)[4], const f16x8_t (&B)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[i], B[jn], acc[i][jn], 0, 0, 0);
  };

  // One super-iteration, k = 3 u + T.  Nine MFMA groups (three pairs x three terms); behind group gi: the LDS store of filter tap gi of
  // super-iteration k + 1 (gi < 6), the load of tap gi - 3 of super-iteration k + 2 (gi >= 3), and the strip work of the type.
  // Every load and store is unconditional (conv3.hip: a load inside a branch makes the s_waitcnt in front of the older stage's stores
  // conservative); past the end of the K loop the loads take the out-of-range offset and the stores fill buffers nobody reads.
  auto body = [&](auto T_, const int u) {
    constexpr int T = decltype(T_)::value;
    constexpr int TN = (T + 1) % 3;
    const int k = 3 * u + T, c0 = 2 * u;
    const bool in_b = k + 2 < nsi;
    const int st_nxt = XSTAGE - st_cur;
    unsigned char* const bw = Bbase + st_nxt + b_st;
    auto side = [&](const int gi) {
      // the filter tile of super-iteration k + 1 (loaded behind groups 3-8 of the previous one) goes to the other stage behind groups 0-5;
      // register set q is reloaded three groups after its store with tap q of super-iteration k + 2
      if constexpr (DMA) {
        // slots 2-5 of tile k + 1 (rows: T0 -> (c0, 2), (c0 + 1, 0);  T1 -> (c0 + 1, 1), (c0 + 1, 2);  T2 -> (c0 + 2, 0), (c0 + 2, 1)) into the
        // other stage behind groups 0-3; slots 0, 1 of tile k + 2 into THIS stage behind groups 7, 8 (every wave is past the barrier)
        if (gi < 4 && !(C3X_ABL & 1)) {
          const int sl = gi + 2, second = sl >= 3, j = sl - 3 * second;
          const int G = T == 0 ? (second ? 0 : 2) : (T == 1 ? (second ? 2 : 1) : (second ? 1 : 0));
          const int dc = T == 0 ? (second ? 1 : 0) : (T == 1 ? 1 : 2);
          dma(st_nxt, sl, k + 1 < nsi, (unsigned)(p.tap_w[3 * G + j] + (c0 + dc) * 16) * 4u);
        }
        if (gi >= 7 && !(C3X_ABL & 1)) {
          const int sl = gi - 7;
          const int G = T == 0 ? 1 : (T == 1 ? 0 : 2), dc = T == 0 ? 1 : 2;
          dma(st_cur, sl, in_b, (unsigned)(p.tap_w[3 * G + sl] + (c0 + dc) * 16) * 4u);
        }
      } else {
      if (gi < 6 && !(C3X_ABL & 4)) *reinterpret_cast<f32x4*>(bw + gi * XSLOT) = b_reg[gi];
      if (gi >= 3 && !(C3X_ABL & 1)) {
        // rows of super-iteration k + 2: T0 -> (c0 + 1, 1), (c0 + 1, 2);  T1 -> (c0 + 2, 0), (c0 + 2, 1);  T2 -> (c0 + 2, 2), (c0 + 3, 0)
        const int q = gi - 3, second = q >= 3, j = q - 3 * second;
        const int G = T == 0 ? (second ? 2 : 1) : (T == 1 ? (second ? 1 : 0) : (second ? 0 : 2));
        const int dc = T == 0 ? 1 : (T == 1 ? 2 : (second ? 3 : 2));
        const unsigned soff = (unsigned)(p.tap_w[3 * G + j] + (c0 + dc) * 16) * 4u;
        b_reg[q] = ldx16(b_rs, in_b ? b_off : OOBX, in_b ? soff : 0u);
      }
      }
      if (T == 0 && gi >= 2 && gi < 2 + XA_LD && !(C3X_ABL & 4)) store_a_piece(2 * PA, gi - 2);            // strip c1 -> buffer 1
      if (T == 1 && gi < XA_LD && !(C3X_ABL & 2)) {                                                                     // strip c0 + 2
        const bool in_a = c0 + 2 < nch;
        a_reg[gi] = ldx16(a_rs, in_a ? a_off(gi) : OOBX, in_a ? (unsigned)(c0 + 2) * 64u : 0u);
      }
      if (T == 2 && gi < XA_LD && !(C3X_ABL & 4)) store_a_piece(0, gi);                                            // strip c0 + 2 -> buffer 0
      if (T == 2 && gi >= 4 && gi < 4 + XA_LD && !(C3X_ABL & 2)) {                                                     // strip c0 + 3
        const bool in_a = c0 + 3 < nch;
        a_reg[gi - 4] = ldx16(a_rs, in_a ? a_off(gi - 4) : OOBX, in_a ? (unsigned)(c0 + 3) * 64u : 0u);
      }
    };
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      // (l,h): Al, Bh of this pair were read during the previous pair's last two groups
      read_Ah();
      read_B(Bl, st_cur, j, 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(Al, Bh);
      side(3 * j);
      __builtin_amdgcn_sched_barrier(0);
      // every store of this super-iteration is done; the next pair is the next super-iteration's
      if (j == 2 && !(C3X_ABL & 8)) {
        if constexpr (DMA) {
          // this wave's pieces of tile k + 1 have landed: the only younger vector-memory operations are T2's strip loads behind groups 4-6
          if (T == 2 && !(C3X_ABL & 2)) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
          __syncthreads();
        }
      }
      if (j < 2) a_addr(T, j + 1); else a_addr(TN, 0);
      read_Al();
      __builtin_amdgcn_sched_barrier(0);
      mm(Ah, Bh);
      side(3 * j + 1);
      __builtin_amdgcn_sched_barrier(0);
      if (j < 2) read_B(Bh, st_cur, j + 1, 0); else read_B(Bh, st_nxt, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm(Ah, Bl);
      side(3 * j + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    st_cur = st_nxt;
  };
  a_addr(0, 0);
  read_Al();
  read_B(Bh, 0, 0, 0);
  {
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    for (int u = 0; u < nper; ++u) { body(I0{}, u); body(I1{}, u); body(I2{}, u); }
  }
  if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the pieces issued past the end may still be pending LDS writes)
  __syncthreads();                            // (the statistics reduction below reuses LDS)

  // ---- epilogue (conv3.hip's, on 16-byte vectors: this lane's rows m0 + wm 64 + 16 i + 4 g + r, filters co0 .. co0 + 3) --------------
  const float dq = 1.f / (sa * sb);           // powers of two: exact
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) acc[i][jn] *= dq;
  const int co0 = bn * XBN + wn * 64 + 4 * c16;
  const bool co_ok = co0 < p.Co;              // Co % 4 == 0 (launch side)
  const int mrow0 = m0 + wm * 64 + 4 * g;
  float* __restrict__ gout = p.out;
  if (p.accumulate && co_ok) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mrow0 + 16 * i + r;
        if (m >= M) continue;
        const f32x4 o = *reinterpret_cast<const f32x4*>(gout + (size_t)m * p.ldo + co0);
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[i][jn][r] += o[jn];
      }
  }
  if (p.stats) {
    float* red = reinterpret_cast<float*>(smemx);        // [2][XWM][XBN]  (LDS is free after the last barrier)
    float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bt_y && co_ok) {                                 // BatchNorm tap (igemm.h): channel_partials_kernel<1>'s terms (bn.hip)
      const f32x4 mu = *reinterpret_cast<const f32x4*>(p.bt_mean + co0), is = *reinterpret_cast<const f32x4*>(p.bt_invstd + co0);
      f32x4 ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
      if (p.bt_gamma) ga = *reinterpret_cast<const f32x4*>(p.bt_gamma + co0);
      if (p.bt_beta) be = *reinterpret_cast<const f32x4*>(p.bt_beta + co0);
      const float* yb = p.bt_y + co0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 yv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = mrow0 + 16 * i + r;
          yv[r] = *reinterpret_cast<const f32x4*>(yb + (size_t)(m < M ? m : M - 1) * p.Co);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = mrow0 + 16 * i + r;
#pragma unroll
          for (int jn = 0; jn < 4; ++jn) {
            const float xh = (yv[r][jn] - mu[jn]) * is[jn];
            float gq = acc[i][jn][r];
            if (p.bt_act == DCN_ACT_LEAKY) gq = (ga[jn] * xh + be[jn] <= 0.f) ? gq * p.bt_slope : gq;
            gq = m < M ? gq : 0.f;
            s[jn] += gq; ss[jn] += gq * xh;
          }
        }
      }
    } else if (!p.bt_y) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int jn = 0; jn < 4; ++jn) { const float v = acc[i][jn][r]; s[jn] += v; ss[jn] = __builtin_fmaf(v, v, ss[jn]); }
    }
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      s[jn] += __shfl_xor(s[jn], 16); ss[jn] += __shfl_xor(ss[jn], 16);
      s[jn] += __shfl_xor(s[jn], 32); ss[jn] += __shfl_xor(ss[jn], 32);
    }
    if (g == 0) {
      const int col = wn * 64 + 4 * c16;
      *reinterpret_cast<f32x4*>(red + (0 * XWM + wm) * XBN + col) = f32x4{s[0], s[1], s[2], s[3]};
      *reinterpret_cast<f32x4*>(red + (1 * XWM + wm) * XBN + col) = f32x4{ss[0], ss[1], ss[2], ss[3]};
    }
    __syncthreads();
    const int groups = XBM / gran, spg = XWM / groups;      // gran in {256, 128}: slabs (waves along M) per partial row
    const int rows_total = (M + gran - 1) / gran;
    for (int idx = tid; idx < 2 * XBN * groups; idx += XNT) {
      const int gq = idx / (2 * XBN), rest = idx - gq * 2 * XBN;
      const int which = rest / XBN, col = rest - which * XBN;
      float t = 0.f;
      for (int w = 0; w < spg; ++w) t += red[(which * XWM + gq * spg + w) * XBN + col];
      const int co = bn * XBN + col, srow = bm * groups + gq;
      if (co < p.Co && srow < rows_total) p.stats[((size_t)srow * 2 + which) * p.Co + co] = t;
    }
  }
  float vmax = 0.f;
  if (co_ok) {
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + co0);
    if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + co0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mrow0 + 16 * i + r;
        if (m >= M) continue;
        f32x4 v;
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
          float t = acc[i][jn][r] * sc[jn] + sh[jn];
          if (p.act == DCN_ACT_LEAKY) t = t > 0.f ? t : t * p.slope;
          v[jn] = t;
        }
        if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * p.ldr + co0);
        *reinterpret_cast<f32x4*>(gout + (size_t)m * p.ldo + co0) = v;
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) vmax = fmaxf(vmax, fabsf(v[jn]));
      }
  }
  if (p.amax_out) {
    vmax = wave_max(vmax);
    if (lane == 0) amax_update(p.amax_out, vmax, blockIdx.x * 8 + wave);
  }
}

}  // namespace

// can the launch (already accepted by conv3_applicable) run on this kernel?
bool conv3x_takes(const IgemmParams& p, int gran) {
  if (p.wt16 || (gran != 128 && gran != 256)) return false;
  if (p.Co % 4 != 0 || p.ldo % 4 != 0 || (p.residual && p.ldr % 4 != 0)) return false;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (!al16(p.out) || !al16(p.residual) || !al16(p.bt_y) || !al16(p.scale) || !al16(p.shift) || !al16(p.bt_mean) || !al16(p.bt_invstd) ||
      !al16(p.bt_gamma) || !al16(p.bt_beta)) return false;
  const int S = XBM + 2 * p.Wi + 2;
  if (S > XSMAX) return false;                             // (also: S positions x 4 pieces <= 512 threads x 4 pieces)
  return (long long)S * p.ldi * 4 < 0x7FFFFFF0LL && (long long)p.Co * p.ldw * 4 < 0x7FFFFFF0LL;
}

int g_conv3x_dma = 1;     // dcn_set_tuning("3dma", 0): filter tiles through registers again (A/B switch)
void conv3x_set_tuning(int v) { g_conv3x_dma = v; }

int conv3x_launch(const IgemmParams& p, int gran, hipStream_t stream) {
  const int S = XBM + 2 * p.Wi + 2;
  const size_t lds = (size_t)4 * XPA + (size_t)2 * XSTAGE;
  static_assert((size_t)4 * XPA + (size_t)2 * XSTAGE <= 160 * 1024, "LDS");
  static DcnPerDeviceFlag attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const int gm = cdiv(p.M, XBM), gn = cdiv(p.Co, XBN);
  const double k_alg = 9.0 * p.Ci;
  const double alg_bytes = 4.0 * ((double)p.N * p.Hi * p.Wi * p.Ci + (double)p.Co * k_alg + (double)p.M * p.Co * epilogue_reads(p));
  const int pid = prof_begin(51, 2.0 * (double)p.M * p.Co * k_alg, stream, alg_bytes);
  if (g_conv3x_dma) hipLaunchKernelGGL(conv3x_kernel<true>, dim3(gm * gn), dim3(XNT), lds, stream, p, S, gran);
  else hipLaunchKernelGGL(conv3x_kernel<false>, dim3(gm * gn), dim3(XNT), lds, stream, p, S, gran);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("conv3x");
  return DCN_OK;
}
