// Weight gradient of the 3x3 stride-1 layers, one filter ROW (three taps) per workgroup, f16 two-piece split.
//
//   dW[co][r][s][ci] = sum_px dY[px][co] * X[px + (r-1)*W + (s-1)][ci]        (terms whose tap leaves the image are absent)
//
// wgrad.hip gives every tap its own workgroups: the dY tile is loaded, split into f16 pieces and written to LDS nine times per
// (co, ci) tile, the X tile once per tap — 4 global loads and ~40 split operations per thread for 12 MFMAs per wave.  The three
// taps of a filter row read X at pixel shifts -1, 0, +1, so one strip of 16 + 2 pixels serves all three, and the dY tile is
// staged once for them: 3 loads per thread for 18 MFMAs per wave.  What stood in the way is the masking — the term (px, s) is
// absent when px sits in column 0 (s = 0) or W-1 (s = 2), a different set per tap, which a SHARED strip cannot express at load
// time.  It disappears in PADDED pixel coordinates: K runs over positions p of rows of W + 1 entries whose last entry is a
// pad (dY = 0, X = 0 there).  Then X_pad[p - 1] of a column-0 pixel and X_pad[p + 1] of a column-(W-1) pixel are pads, i.e.
// zero, for every row — no mask, no branch; the vertical taps (r = 0 on image row 0, r = 2 on row H-1) are the same for the
// three taps of the workgroup's filter row and are applied to dY when it is loaded.  Cost: (W+1)/W more K (2-8 %).
// The padded position is kept per load slot as (row index, column) and advanced by 16 per K-step: no division in the loop,
// no geometry table.  LDS planes keep the HBM orientation [position][128 channels] (256-B rows, XOR-swizzled 16-B chunks) and
// the MFMA operands come from ds_read_b64_tr_b16, as in wgrad.hip; the B fragment of tap s is the same read one row lower.
// 8 waves: wave (wm, wn) owns 64 filters x 32 channels x 3 taps (96 accumulator registers).  Split-K over padded positions
// into slabs summed in a fixed order (wgrad.hip reduce_slabs_kernel): bitwise reproducible.  Roofline: MFMA, 838.9 TFLOP/s.
#include "common.h"
#include "prof.h"
#include "slabsum.h"
#include <type_traits>

int wgrad_reduce_slabs(const float* ws, float* dw, int64_t n4, int splits, hipStream_t stream);

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned OOBW = 0x80000000u;

struct W3Params {
  const float* x; const float* dy; float* out;
  int N, H, W, Ci, ldx, Co, lddy;
  int Mp, kchunk, splits;          // padded positions N*H*(W+1); per split (multiple of 16)
  int tiles_co, tiles_ci, ld_out;
  const unsigned* amax_dy; const unsigned* amax_x;
  SlabFold fold;                   // slabsum.h
};

__device__ __forceinline__ f32x4 ldw16(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrcw(const float* base, long long bytes) {
  const unsigned n = bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0u : (unsigned)(bytes < 0 ? 0 : bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, n, 0x00020000);
}
__device__ __forceinline__ float pow2w(unsigned amax_bits) {
  const int be = (int)((amax_bits >> 23) & 0xFF);
  if (be == 0 || be == 255) return 1.f;
  int e = 14 - (be - 126);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return __uint_as_float((unsigned)(e + 127) << 23);
}
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
// byte offset of channels c..c+3 (c % 4 == 0) of row `row` inside one [rows][128] f16 plane
__device__ __forceinline__ int poff(int row, int c) { return 256 * row + 16 * ((c >> 3) ^ swz(row)) + 8 * ((c >> 2) & 1); }

constexpr int A_PLANE = 16 * 256, B_PLANE = 18 * 256;

// NP = 2: fp32-accurate f16 two-piece split (three MFMAs per product, per-tensor power-of-two scales from the abs-max words);
// NP = 1: plain bf16 operands (round to nearest even, one MFMA per product) — the weight gradient of the bf16- and fp8-operand
// modes (BASELINE.json configs[2] / [4]; ops.set_precision("bf16" | "fp8"))
// IN16 (NP = 1, bf16 storage): x and dy ARE bf16 tensors.  A 16-byte piece is 8 channels of a position and goes to its plane as loaded:
// waves 0-3 bring in the 256 pieces of the dY tile, waves 4-7 strip rows 0-15, the first 32 threads strip rows 16-17 as well.
template <int NP, bool IN16 = false>
__global__ __launch_bounds__(512, 2) void wgrad3_kernel(const W3Params p) {
  static_assert(!IN16 || NP == 1, "bf16 inputs: one plane");
  constexpr int ESZ = IN16 ? 2 : 4;
  constexpr int B_BASE = NP * A_PLANE, BUF = NP * A_PLANE + NP * B_PLANE;
  typedef typename std::conditional<NP == 2, f16x8_t, bf16x8_t>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm3[];       // [2 buffers][A: 2 planes x 16 rows | B: 2 planes x 18 rows]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;            // 64 filters x 32 channels per wave
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int per_split = p.tiles_ci * p.tiles_co * 3;
  const int split = b / per_split; b -= split * per_split;
  const int tci = b % p.tiles_ci; b /= p.tiles_ci;
  const int tco = b % p.tiles_co; b /= p.tiles_co;
  const int r = b;                                    // filter row of this workgroup
  const int co0 = tco * 128, ci0 = tci * 128;
  const int Wp = p.W + 1;
  const int p_begin = split * p.kchunk;
  const int p_end = min(p.Mp, p_begin + p.kchunk);
  const int iters = (p_end - p_begin + 15) / 16;
  float s_a = 1.f, s_b = 1.f;
  if constexpr (NP == 2) { s_a = pow2w(amax_read(p.amax_dy)); s_b = pow2w(amax_read(p.amax_x)); }

  const long long npix = (long long)p.N * p.H * p.W;
  const __amdgpu_buffer_rsrc_t a_rs = rsrcw(p.dy, ((npix - 1) * p.lddy + p.Co) * ESZ);
  const __amdgpu_buffer_rsrc_t b_rs = rsrcw(p.x, ((npix - 1) * p.ldx + p.Ci) * ESZ);
  const int nrows = p.N * p.H;                        // image rows in the tensor

  // ---- load slots: A = dY row (position p_begin + ra), B0 / B1 = strip rows s (position p_begin - 1 + s, filter row r-1 down) --
  // fp32 inputs: every thread loads an A piece and a B0 piece (4 channels each).  IN16: a piece is 8 channels — threads 0-255 (waves 0-3) own
  // the A pieces, 256-511 the B0 pieces (slot 0 = ar below), the first 32 threads the B1 pieces of strip rows 16, 17 (slot 1 = br[1]).
  const bool role_a = !IN16 || tid < 256;             // (wave-uniform)
  const int ra = IN16 ? ((tid & 255) >> 4) : (tid >> 5), cch = IN16 ? (tid & 15) * 8 : (tid & 31) * 4;
  const bool a_chan = co0 + cch < p.Co, b_chan = ci0 + cch < p.Ci;
  const __amdgpu_buffer_rsrc_t s0_rs = role_a ? a_rs : b_rs;
  // state (row index = n*H + y, column in [0, Wp)); A also keeps y for the vertical tap test
  int a_row, a_col, a_y;
  { const int q = p_begin + ra; a_row = q / Wp; a_col = q - a_row * Wp; a_y = a_row % p.H; }
  int b_row[2], b_col[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = p_begin - 1 + (IN16 && j == 1 ? (tid >> 4) : ra) + 16 * j;          // >= -1   (IN16, j = 1: rows 16 + tid / 16 of threads 0-31)
    if (q < 0) { b_row[j] = -1; b_col[j] = Wp + q; }
    else { b_row[j] = q / Wp; b_col[j] = q - b_row[j] * Wp; }
    b_row[j] += r - 1;
  }
  const bool b1_on = tid < (IN16 ? 32 : 64);          // strip rows 16, 17
  int pos_a = p_begin + ra;                           // padded position of the A row (end-of-split test)

  // 16 positions further: at most two row wraps (W >= 8), as selects — a loop here puts branches into the K loop
  auto advance = [&](int& row, int& col) -> int {
    col += 16;
    const int w1 = col >= Wp ? 1 : 0; col -= w1 ? Wp : 0;
    const int w2 = col >= Wp ? 1 : 0; col -= w2 ? Wp : 0;
    row += w1 + w2;
    return w1 + w2;
  };
  auto load_into = [&](f32x4& ar, f32x4* br) {
    if constexpr (IN16) {
      // slot 0: the A piece (waves 0-3) or the B0 piece (waves 4-7) — one load instruction, the descriptor and the offset chosen per wave
      const bool vert = (r == 0 && a_y == 0) || (r == 2 && a_y == p.H - 1);
      const bool ok_a = a_chan && pos_a < p_end && a_col < p.W && !vert;
      const unsigned off_a = (unsigned)(((a_row * p.W + a_col) * p.lddy + co0 + cch) * 2);
      const bool ok_b = b_chan && b_col[0] < p.W && (unsigned)b_row[0] < (unsigned)nrows;
      const unsigned off_b = (unsigned)(((b_row[0] * p.W + b_col[0]) * p.ldx + ci0 + cch) * 2);
      ar = ldw16(s0_rs, role_a ? (ok_a ? off_a : OOBW) : (ok_b ? off_b : OOBW));
      a_y += advance(a_row, a_col);
      a_y -= a_y >= p.H ? p.H : 0; a_y -= a_y >= p.H ? p.H : 0;
      pos_a += 16;
      advance(b_row[0], b_col[0]);
      // slot 1: strip rows 16, 17 (threads 0-31; the others read out of range: zeros, never stored)
      const int cc1 = (tid & 15) * 8;
      const bool ok1 = b1_on && ci0 + cc1 < p.Ci && b_col[1] < p.W && (unsigned)b_row[1] < (unsigned)nrows;
      br[1] = ldw16(b_rs, ok1 ? (unsigned)(((b_row[1] * p.W + b_col[1]) * p.ldx + ci0 + cc1) * 2) : OOBW);
      advance(b_row[1], b_col[1]);
      br[0] = f32x4{0.f, 0.f, 0.f, 0.f};
      return;
    }
    {
      const bool vert = (r == 0 && a_y == 0) || (r == 2 && a_y == p.H - 1);
      const bool ok = a_chan && pos_a < p_end && a_col < p.W && !vert;
      const unsigned off = (unsigned)(((a_row * p.W + a_col) * p.lddy + co0 + cch) * 4);
      ar = ldw16(a_rs, ok ? off : OOBW);
      a_y += advance(a_row, a_col);
      a_y -= a_y >= p.H ? p.H : 0; a_y -= a_y >= p.H ? p.H : 0;
      pos_a += 16;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const bool ok = b_chan && (j == 0 || b1_on) && b_col[j] < p.W && (unsigned)b_row[j] < (unsigned)nrows;
      const unsigned off = (unsigned)(((b_row[j] * p.W + b_col[j]) * p.ldx + ci0 + cch) * 4);
      br[j] = ldw16(b_rs, ok ? off : OOBW);
      advance(b_row[j], b_col[j]);
    }
  };
  auto split_store = [&](unsigned char* plane0, int plane_stride, int off, const f32x4 v, const float sc) {
    if constexpr (NP == 1) {
      const bf16x4_t b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      *reinterpret_cast<uint2*>(plane0 + off) = __builtin_bit_cast(uint2, b);
      return;
    }
    const f32x4 t = v * sc;
    const f16x4_t h = {(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
    const f16x4_t l = {(_Float16)(t[0] - (float)h[0]), (_Float16)(t[1] - (float)h[1]), (_Float16)(t[2] - (float)h[2]),
                       (_Float16)(t[3] - (float)h[3])};
    *reinterpret_cast<uint2*>(plane0 + off) = __builtin_bit_cast(uint2, h);
    *reinterpret_cast<uint2*>(plane0 + plane_stride + off) = __builtin_bit_cast(uint2, l);
  };
  const int a_st = poff(ra, cch), b_st0 = poff(ra, cch), b_st1 = IN16 ? poff(16 + (tid >> 4), (tid & 15) * 8) : poff(16 + (ra & 1), cch);   // (b1: rows 16, 17)
  auto store_a = [&](int buf, const f32x4& ar) {
    if constexpr (IN16) { *reinterpret_cast<f32x4*>(sm3 + buf * BUF + (role_a ? 0 : B_BASE) + a_st) = ar; return; }      // slot 0: 8 bf16 as loaded
    split_store(sm3 + buf * BUF, A_PLANE, a_st, ar, s_a);
  };
  auto store_b = [&](int buf, const f32x4* br) {
    if constexpr (IN16) { if (b1_on) *reinterpret_cast<f32x4*>(sm3 + buf * BUF + B_BASE + b_st1) = br[1]; return; }
    split_store(sm3 + buf * BUF + B_BASE, B_PLANE, b_st0, br[0], s_b);
    if (b1_on) split_store(sm3 + buf * BUF + B_BASE, B_PLANE, b_st1, br[1], s_b);      // (one wave of the eight: wave-uniform)
  };

  f32x16 acc[2][3];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mi][d][q] = 0.f;

  // transposed-read addresses: 16-lane group (hh, gg): k rows 8hh + 4r2 + q, channels blk*32 + 16gg + 4pp
  const int g16 = lane >> 4, hh = g16 >> 1, gg = g16 & 1, qq = (lane & 15) >> 2, pp = lane & 3;
  int a_tr[2][2], b_tr[3][2];
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) {
    const int row = 8 * hh + 4 * r2 + qq;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) a_tr[mi][r2] = poff(row, wm * 64 + mi * 32 + 16 * gg + 4 * pp);
#pragma unroll
    for (int d = 0; d < 3; ++d) b_tr[d][r2] = B_BASE + poff(row + d, wn * 32 + 16 * gg + 4 * pp);
  }
  auto tr_read = [&](int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm3 + byte_off));
  };
  auto frag = [&](int byte0, int byte1) {
    const s16x4 lo = tr_read(byte0), hi = tr_read(byte1);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(frag_t, v);
  };

  auto read_a = [&](int buf, frag_t (*af)[2]) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) af[mi][pl] = frag(buf * BUF + pl * A_PLANE + a_tr[mi][0], buf * BUF + pl * A_PLANE + a_tr[mi][1]);
  };
  auto read_b = [&](int buf, int d, frag_t* bf) {
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) bf[pl] = frag(buf * BUF + pl * B_PLANE + b_tr[d][0], buf * BUF + pl * B_PLANE + b_tr[d][1]);
  };
  auto mfma_tap = [&](int d, frag_t (*af)[2], const frag_t* bf) {      // smallest terms first: (l,h) (h,l) (h,h)
    if constexpr (NP == 1) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][0], bf[0], acc[mi][d], 0, 0, 0);
    } else {
#pragma unroll
      for (int term = 0; term < 3; ++term) {
        const int qa = term == 0 ? 1 : 0, qb = term == 1 ? 1 : 0;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[mi][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi][qa], bf[qb], acc[mi][d], 0, 0, 0);
      }
    }
  };

  // Issue order of one K-step, fixed with scheduling barriers (the compiler's own order issued the global loads at the END of
  // the step and read every fragment right in front of its MFMAs): global loads of step it+2 first | fragment of tap 1 |
  // 6 MFMAs of tap 0, split + store of dY(it+1) | fragment of tap 2 | 6 MFMAs of tap 1, split + store of X(it+1) | barrier |
  // fragments of step it+1 (dY and tap 0) | 6 MFMAs of tap 2.  Every LDS read is issued >= 6 MFMAs before its use; every
  // load of the loop is unconditional (past the end of the split: zeros, igemm.hip).
  f32x4 ar, br[2];
  frag_t af[2][2], bf0[2];
  if (iters > 0) {
    load_into(ar, br);
    store_a(0, ar); store_b(0, br);
    load_into(ar, br);
  }
  __syncthreads();
  read_a(0, af); read_b(0, 0, bf0);
  for (int it = 0; it < iters; ++it) {
    const int cur = it & 1;
    f32x4 an, bn[2];
    frag_t bf1[2], bf2[2], afn[2][2], bf0n[2];
    load_into(an, bn);                                  // step it+2
    read_b(cur, 1, bf1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_tap(0, af, bf0);
    store_a(cur ^ 1, ar);
    __builtin_amdgcn_sched_barrier(0);
    read_b(cur, 2, bf2);
    mfma_tap(1, af, bf1);
    store_b(cur ^ 1, br);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_a(cur ^ 1, afn); read_b(cur ^ 1, 0, bf0n);
    __builtin_amdgcn_sched_barrier(0);
    mfma_tap(2, af, bf2);
    __builtin_amdgcn_sched_barrier(0);
    ar = an; br[0] = bn[0]; br[1] = bn[1];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) af[mi][pl] = afn[mi][pl];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) bf0[pl] = bf0n[pl];
  }
  __syncthreads();

  const float dq = 1.f / (s_a * s_b);                   // powers of two: exact (NP = 1: 1)
  float* out = p.out + (size_t)split * p.Co * p.ld_out;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int co = co0 + wm * 64 + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
      if (co >= p.Co) continue;
      const int ci = ci0 + wn * 32 + (lane & 31);
      if (ci >= p.Ci) continue;
#pragma unroll
      for (int d = 0; d < 3; ++d) out[(size_t)co * p.ld_out + (3 * r + d) * p.Ci + ci] = acc[mi][d][q] * dq;
    }
  if (p.fold.counters)              // the three taps of filter row r: three segments of a dw row
    slab_fold<512>(p.fold, (r * p.tiles_co + tco) * p.tiles_ci + tci, p.splits, p.out, (size_t)p.Co * p.ld_out, co0, min(128, p.Co - co0),
                   p.ld_out, 3 * r * p.Ci + ci0, min(128, p.Ci - ci0), 3, p.Ci, reinterpret_cast<int*>(sm3));
}

// ---- the same weight gradient on v_mfma_f32_16x16x32_f16 (round 5) --------------------------------------------------------------------------
// The chip holds a higher clock on this MFMA shape under the f16-split loops (tools/mfma_shape_probe.hip: x1.14; conv3x.hip).  K = 32 padded
// positions per K-step (a strip of 32 + 2): twice the MFMAs between barriers (72 per wave), same LDS images (256-byte rows, the XOR of poff),
// same transposed reads — a 16-lane group of ds_read_b64_tr_b16 now takes k rows 8 g .. 8 g + 7 of ONE 16-channel block (the four groups of
// a wave: the same 16 channels, 32 k), so only the per-lane addresses differ.  Wave tile as before (64 filters x 32 channels x 3 taps =
// 4 x 2 x 3 accumulator blocks of 16 x 16).  One set of fragments rotates: per tap the terms run (h,l) (h,h) (l,h), so the l-plane
// of the strip tap dies after the first group, the h-plane after the second, and the next tap's (next K-step's) fragments are read into
// them one group (8 MFMAs) or more ahead; the dY fragments live for the whole K-step.
constexpr int AX_PLANE = 32 * 256, BX_PLANE = 34 * 256;
__global__ __launch_bounds__(512, 2) void wgrad3x_kernel(const W3Params p) {
  constexpr int B_BASE = 2 * AX_PLANE, BUF = 2 * AX_PLANE + 2 * BX_PLANE;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm3x[];      // [2 buffers][A: 2 planes x 32 rows | B: 2 planes x 34 rows]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;            // 64 filters x 32 channels per wave
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int per_split = p.tiles_ci * p.tiles_co * 3;
  const int split = b / per_split; b -= split * per_split;
  const int tci = b % p.tiles_ci; b /= p.tiles_ci;
  const int tco = b % p.tiles_co; b /= p.tiles_co;
  const int r = b;                                    // filter row of this workgroup
  const int co0 = tco * 128, ci0 = tci * 128;
  const int Wp = p.W + 1;
  const int p_begin = split * p.kchunk;
  const int p_end = min(p.Mp, p_begin + p.kchunk);
  const int iters = (p_end - p_begin + 31) / 32;
  const float s_a = pow2w(amax_read(p.amax_dy)), s_b = pow2w(amax_read(p.amax_x));

  const long long npix = (long long)p.N * p.H * p.W;
  const __amdgpu_buffer_rsrc_t a_rs = rsrcw(p.dy, ((npix - 1) * p.lddy + p.Co) * 4);
  const __amdgpu_buffer_rsrc_t b_rs = rsrcw(p.x, ((npix - 1) * p.ldx + p.Ci) * 4);
  const int nrows = p.N * p.H;

  // ---- load slots (4 channels each): A0 / A1 = dY rows ra, ra + 16 (positions p_begin + row); B0 / B1 = strip rows ra, ra + 16 (positions
  //      p_begin - 1 + row, one filter row up or down), B2 = strip rows 32, 33 (the first 64 threads).  State is kept for A0, B0 and B2: the
  //      +16 slots are the state advanced by 16, which advanced once more is the next K-step's.
  const int ra = tid >> 5, cch = (tid & 31) * 4;
  const bool a_chan = co0 + cch < p.Co, b_chan = ci0 + cch < p.Ci;
  int a_row, a_col, a_y;
  { const int q = p_begin + ra; a_row = q / Wp; a_col = q - a_row * Wp; a_y = a_row % p.H; }
  int pos_a = p_begin + ra;
  auto strip_state = [&](int q, int& row, int& col) {
    if (q < 0) { row = -1; col = Wp + q; } else { row = q / Wp; col = q - row * Wp; }
    row += r - 1;
  };
  int b_row, b_col, b2_row, b2_col;
  strip_state(p_begin - 1 + ra, b_row, b_col);
  strip_state(p_begin - 1 + 32 + (ra & 1), b2_row, b2_col);
  const bool b2_on = tid < 64;

  auto advance = [&](int& row, int& col) -> int {     // 16 positions further: at most two row wraps (W >= 8), as selects
    col += 16;
    const int w1 = col >= Wp ? 1 : 0; col -= w1 ? Wp : 0;
    const int w2 = col >= Wp ? 1 : 0; col -= w2 ? Wp : 0;
    row += w1 + w2;
    return w1 + w2;
  };
  auto wrap_y = [&](int y) { y -= y >= p.H ? p.H : 0; y -= y >= p.H ? p.H : 0; return y; };
  auto load_a = [&](int row, int col, int y, int pos) {
    const bool vert = (r == 0 && y == 0) || (r == 2 && y == p.H - 1);
    const bool ok = a_chan && pos < p_end && col < p.W && !vert;
    return ldw16(a_rs, ok ? (unsigned)(((row * p.W + col) * p.lddy + co0 + cch) * 4) : OOBW);
  };
  auto load_b = [&](int row, int col, bool on) {
    const bool ok = b_chan && on && col < p.W && (unsigned)row < (unsigned)nrows;
    return ldw16(b_rs, ok ? (unsigned)(((row * p.W + col) * p.ldx + ci0 + cch) * 4) : OOBW);
  };
  // the five pieces of one K-step; the states move on by 32 positions
  auto load_piece = [&](const int e) -> f32x4 {
    if (e == 0) return load_a(a_row, a_col, a_y, pos_a);
    if (e == 1) {
      a_y = wrap_y(a_y + advance(a_row, a_col));
      const f32x4 v = load_a(a_row, a_col, a_y, pos_a + 16);
      a_y = wrap_y(a_y + advance(a_row, a_col));
      pos_a += 32;
      return v;
    }
    if (e == 2) return load_b(b_row, b_col, true);
    if (e == 3) {
      advance(b_row, b_col);
      const f32x4 v = load_b(b_row, b_col, true);
      advance(b_row, b_col);
      return v;
    }
    const f32x4 v = load_b(b2_row, b2_col, b2_on);
    advance(b2_row, b2_col); advance(b2_row, b2_col);
    return v;
  };
  const int st0 = poff(ra, cch), st2 = poff(32 + (ra & 1), cch);        // (row + 16: + 4096 — swz(row + 16) = swz(row))
  auto store_piece = [&](const int buf, const int e, const f32x4 v) {    // x * s = h + l into the two planes of its tensor
    const bool is_a = e < 2;
    const f32x4 t = v * (is_a ? s_a : s_b);
    const f16x4_t h = {(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
    const f16x4_t l = {(_Float16)(t[0] - (float)h[0]), (_Float16)(t[1] - (float)h[1]), (_Float16)(t[2] - (float)h[2]),
                       (_Float16)(t[3] - (float)h[3])};
    const int off = buf * BUF + (is_a ? 0 : B_BASE) + (e == 4 ? st2 : st0 + ((e & 1) ? 4096 : 0));
    unsigned char* q = sm3x + off;
    if (e == 4 && !b2_on) return;                                       // (wave-uniform: wave 0 only)
    *reinterpret_cast<uint2*>(q) = __builtin_bit_cast(uint2, h);
    *reinterpret_cast<uint2*>(q + (is_a ? AX_PLANE : BX_PLANE)) = __builtin_bit_cast(uint2, l);
  };

  f32x4 acc[4][2][3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int d = 0; d < 3; ++d) acc[i][j][d] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed-read addresses: 16-lane group g16 takes k rows 8 g16 + 4 r2 + q of a 16-channel block, lane 4 q + pp the columns 4 pp .. + 3.
  // Block i / jb of the wave's channels: chunk index ^ (2 i), i.e. byte address ^ (i << 5) (16-byte chunks, bits 1-2 of the chunk index are
  // free of carries) — one base address per (tensor, tap, row pair), the blocks by an XOR in the loop.
  const int g16 = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  int a_tr0[2], b_tr0[3][2];
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) {
    const int row = 8 * g16 + 4 * r2 + qq;
    a_tr0[r2] = poff(row, wm * 64 + 4 * pp);
#pragma unroll
    for (int d = 0; d < 3; ++d) b_tr0[d][r2] = B_BASE + poff(row + d, wn * 32 + 4 * pp);
  }
  auto tr_read = [&](int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm3x + byte_off));
  };
  auto frag = [&](int byte0, int byte1) {
    const s16x4 lo = tr_read(byte0), hi = tr_read(byte1);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8_t, v);
  };
  f16x8_t Ah[4], Al[4], Bh[2], Bl[2];
  auto read_A = [&](f16x8_t (&A)[4], const int buf, const int plane) {
    int t0 = a_tr0[0], t1 = a_tr0[1];
    asm volatile("" : "+v"(t0), "+v"(t1));              // (the XORs below stay in the loop: hoisted they are 16 more registers)
#pragma unroll
    for (int i = 0; i < 4; ++i) A[i] = frag(buf * BUF + plane * AX_PLANE + (t0 ^ (i << 5)), buf * BUF + plane * AX_PLANE + (t1 ^ (i << 5)));
  };
  auto read_B = [&](f16x8_t (&B)[2], const int buf, const int d, const int plane) {
    int t0 = b_tr0[d][0], t1 = b_tr0[d][1];
    asm volatile("" : "+v"(t0), "+v"(t1));
#pragma unroll
    for (int j = 0; j < 2; ++j) B[j] = frag(buf * BUF + plane * BX_PLANE + (t0 ^ (j << 5)), buf * BUF + plane * BX_PLANE + (t1 ^ (j << 5)));
  };
  auto mm = [&](const f16x8_t (&A)[4], const f16x8_t (&B)[2], const int d) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[i], B[j], acc[i][j][d], 0, 0, 0);
  };

  // K-step `it` multiplies buffer it & 1; the pieces of step it + 1 (in `cur`, loaded one step ago) are split into the other buffer behind its
  // first MFMA groups, the pieces of step it + 2 are loaded into `nxt` behind them.  Nine groups of 8 MFMAs: tap d = (h,l) (h,h) (l,h).
  // Barrier behind the seventh group: every store of the step is done; the fragments of the next step are read behind it.
  f32x4 cur[5], nxt[5];
  if (iters > 0) {
#pragma unroll
    for (int e = 0; e < 5; ++e) cur[e] = load_piece(e);
#pragma unroll
    for (int e = 0; e < 5; ++e) store_piece(0, e, cur[e]);
#pragma unroll
    for (int e = 0; e < 5; ++e) cur[e] = load_piece(e);
  }
  __syncthreads();
  read_A(Ah, 0, 0); read_B(Bl, 0, 0, 1);
  for (int it = 0; it < iters; ++it) {
    const int cb = it & 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      // group 1: (h,l) — Ah and Bl_d were read one group or more ago
      read_B(Bh, cb, d, 0);
      if (d == 0) read_A(Al, cb, 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(Ah, Bl, d);
      if (d == 0) { nxt[0] = load_piece(0); store_piece(cb ^ 1, 0, cur[0]); }
      if (d == 1) { nxt[3] = load_piece(3); store_piece(cb ^ 1, 3, cur[3]); }
      __builtin_amdgcn_sched_barrier(0);
      if (d == 2) __syncthreads();
      // group 2: (h,h); the next tap's (next step's) l-plane strip fragment goes into the registers group 1 has just freed
      if (d < 2) read_B(Bl, cb, d + 1, 1); else read_B(Bl, cb ^ 1, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(Ah, Bh, d);
      if (d == 0) { nxt[1] = load_piece(1); store_piece(cb ^ 1, 1, cur[1]); }
      if (d == 1) { nxt[4] = load_piece(4); store_piece(cb ^ 1, 4, cur[4]); }
      __builtin_amdgcn_sched_barrier(0);
      // group 3: (l,h); behind the last one the next step's dY h-plane
      if (d == 2) read_A(Ah, cb ^ 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm(Al, Bh, d);
      if (d == 0) { nxt[2] = load_piece(2); store_piece(cb ^ 1, 2, cur[2]); }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int e = 0; e < 5; ++e) cur[e] = nxt[e];
  }
  __syncthreads();

  const float dq = 1.f / (s_a * s_b);                   // powers of two: exact
  float* out = p.out + (size_t)split * p.Co * p.ld_out;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = co0 + wm * 64 + i * 16 + 4 * g16 + q;
      if (co >= p.Co) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ci = ci0 + wn * 32 + j * 16 + (lane & 15);
        if (ci >= p.Ci) continue;
#pragma unroll
        for (int d = 0; d < 3; ++d) out[(size_t)co * p.ld_out + (3 * r + d) * p.Ci + ci] = acc[i][j][d][q] * dq;
      }
    }
  if (p.fold.counters)
    slab_fold<512>(p.fold, (r * p.tiles_co + tco) * p.tiles_ci + tci, p.splits, p.out, (size_t)p.Co * p.ld_out, co0, min(128, p.Co - co0),
                   p.ld_out, 3 * r * p.Ci + ci0, min(128, p.Ci - ci0), 3, p.Ci, reinterpret_cast<int*>(sm3x));
}

// OFF by default.  Alone it is the faster kernel (tools/bench_convs.py --strip --ab U3m16=0, N = 64, ms: 512->512 @52 2.505 -> 2.323, @26 0.632 -> 0.588,
// 128->256 @52 0.332 -> 0.312, 256->512 @26 0.326 -> 0.318, 512->1024 @13 0.346 -> 0.336: -4.5 % over the step's layers); in the replayed
// step, where it runs beside the data-gradient chain, the step is 0.45-0.6 ms LONGER with it on each of three boxes (bench.py --schedules 12
// --schedule-tunes "U3m16=0=1;...": 92.08 / 92.23 against 91.63 / 91.54; 93.27 against 92.79 / 92.88; 92.32 against 91.65).  What differs
// beside another queue: 219 registers against 190 (two of its waves leave a SIMD 64 registers instead of 128 for a wave of the chain's
// kernels) and 66 KB of LDS against 34.
int g_w3x = 0;            // dcn_set_tuning("U3m16", 1): 3x3 stride-1 weight gradients on the 16x16x32 build
int g_w3 = 1;             // dcn_set_tuning("u3row", 0): 3x3 stride-1 weight gradients back on the per-tap kernel
int g_w3_target = 512;    // dcn_set_tuning("v3target", n): workgroups a launch aims for (split-K sizing; 512 threads, 1-2 per CU)

struct Plan3 { int tiles_co, tiles_ci, splits, kchunk, Mp; };
Plan3 plan3(int n, int h, int wd, int cin, int cout) {
  Plan3 pl;
  pl.tiles_co = cdiv(cout, 128); pl.tiles_ci = cdiv(cin, 128);
  pl.Mp = n * h * (wd + 1);
  const int base = pl.tiles_co * pl.tiles_ci * 3;
  const int max_splits = pl.Mp / 256 > 0 ? pl.Mp / 256 : 1;
  int splits = g_w3_target / base;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  pl.kchunk = cdiv(cdiv(pl.Mp, splits), 16) * 16;
  pl.splits = cdiv(pl.Mp, pl.kchunk);
  return pl;
}

}  // namespace

void wgrad3_set_tuning(int key, int value) { if (key == 0) g_w3 = value; else if (key == 2) g_w3x = value; else g_w3_target = value > 0 ? value : 512; }

// shape test only (the workspace is sized without knowing whether the abs-max words will be there)
bool wgrad3_shape_ok(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
  if (!g_w3 || ksize != 3 || stride != 1 || cin < 64 || cout < 64 || (cin < 128 && cout < 128) || cin % 4 || cout % 4 || wd < 8 || h < 2) return false;
  // (a 64-channel side leaves half of the 128-wide tile empty and still beats the per-tap narrow tile on the fp32 pipe: 64->128 @104
  //  1.00 -> 0.58 ms, 128->64 @52 0.235 -> 0.154; with both sides below 128 it loses)
  const long long npix = (long long)n * h * wd;
  if (npix * (cin > cout ? cin : cout) * 4 >= 0x7FFFFFF0LL) return false;        // 32-bit byte offsets from the tensor base
  if ((long long)n * h * (wd + 1) >= 0x7FFFFFF0LL || npix < 1024) return false;
  return true;
}
int64_t wgrad3_ws(int n, int h, int wd, int cin, int cout) {
  const Plan3 pl = plan3(n, h, wd, cin, cout);
  return pl.splits > 1 ? (int64_t)pl.splits * cout * 9 * cin : 0;
}

int wgrad_lds_pad();

// bf16 storage: x, dy bf16 tensors (strides in elements), dw / slabs fp32
int wgrad_slab_fold();
int wgrad3_launch_b16(const void* x, int ldx, const void* dy, int lddy, float* dw, float* ws, uint32_t* counters, int n, int h, int wd, int cin, int cout,
                      hipStream_t stream) {
  const Plan3 pl = plan3(n, h, wd, cin, cout);
  DCN_CHECK_ARG(pl.splits == 1 || ws, "conv2d_bwd_weight_b16: workspace required (%d splits)", pl.splits);
  W3Params p{};
  p.x = (const float*)x; p.dy = (const float*)dy; p.out = pl.splits > 1 ? ws : dw;
  p.N = n; p.H = h; p.W = wd; p.Ci = cin; p.ldx = ldx; p.Co = cout; p.lddy = lddy;
  p.Mp = pl.Mp; p.kchunk = pl.kchunk; p.splits = pl.splits;
  p.tiles_co = pl.tiles_co; p.tiles_ci = pl.tiles_ci; p.ld_out = 9 * cin;
  const bool fold = slab_fold_ok(counters, pl.tiles_co * pl.tiles_ci * 3, pl.splits, 128LL * 3 * 128 * 4, wgrad_slab_fold());
  if (fold) p.fold = SlabFold{counters, dw};
  size_t lds = (size_t)2 * (A_PLANE + B_PLANE);
  static DcnPerDeviceFlag attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = pl.tiles_co * pl.tiles_ci * 3 * pl.splits;
  const int pid = prof_begin(46, 2.0 * (double)n * h * wd * cout * 9.0 * cin, stream);
  hipLaunchKernelGGL((wgrad3_kernel<1, true>), dim3(grid), dim3(512), lds, stream, p);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("wgrad3 b16");
  if (pl.splits > 1 && !fold) return wgrad_reduce_slabs(ws, dw, (int64_t)cout * 9 * cin / 4, pl.splits, stream);
  return DCN_OK;
}
bool wgrad3_b16_ok(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
  return wgrad3_shape_ok(n, h, wd, cin, cout, ksize, stride) && cin % 8 == 0 && cout % 8 == 0;
}

int wgrad3_launch(const float* x, int ldx, const float* dy, int lddy, float* dw, float* ws, uint32_t* counters, int n, int h, int wd, int cin, int cout,
                  const uint32_t* amax_x, const uint32_t* amax_dy, int np, hipStream_t stream) {
  const Plan3 pl = plan3(n, h, wd, cin, cout);
  DCN_CHECK_ARG(pl.splits == 1 || ws, "conv2d_bwd_weight: workspace required (%d splits)", pl.splits);
  W3Params p{};
  p.x = x; p.dy = dy; p.out = pl.splits > 1 ? ws : dw;
  p.N = n; p.H = h; p.W = wd; p.Ci = cin; p.ldx = ldx; p.Co = cout; p.lddy = lddy;
  p.Mp = pl.Mp; p.kchunk = pl.kchunk; p.splits = pl.splits;
  p.tiles_co = pl.tiles_co; p.tiles_ci = pl.tiles_ci; p.ld_out = 9 * cin;
  const bool fold = slab_fold_ok(counters, pl.tiles_co * pl.tiles_ci * 3, pl.splits, 128LL * 3 * 128 * 4, wgrad_slab_fold());
  if (fold) p.fold = SlabFold{counters, dw};
  p.amax_dy = amax_dy; p.amax_x = amax_x;
  size_t lds = (size_t)2 * np * (A_PLANE + B_PLANE);
  if ((size_t)wgrad_lds_pad() > lds) lds = (size_t)wgrad_lds_pad();       // (occupancy experiment: "lwgpad")
  static DcnPerDeviceSize attr_lds;
  if (attr_lds.raise(lds)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds > 2 * 2 * (A_PLANE + B_PLANE) ? lds : 2 * 2 * (A_PLANE + B_PLANE)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds > 2 * (A_PLANE + B_PLANE) ? lds : 2 * (A_PLANE + B_PLANE)));
  }
  const int grid = pl.tiles_co * pl.tiles_ci * 3 * pl.splits;
  const int pid = prof_begin(np == 2 ? 32 : 20, 2.0 * (double)n * h * wd * cout * 9.0 * cin, stream);
  if (np == 2 && g_w3x) {
    const size_t ldsx = (size_t)2 * (2 * AX_PLANE + 2 * BX_PLANE);
    static DcnPerDeviceFlag attr_x;
    if (attr_x.first()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsx);
    hipLaunchKernelGGL(wgrad3x_kernel, dim3(grid), dim3(512), ldsx, stream, p);
  } else if (np == 2) hipLaunchKernelGGL(wgrad3_kernel<2>, dim3(grid), dim3(512), lds, stream, p);
  else hipLaunchKernelGGL(wgrad3_kernel<1>, dim3(grid), dim3(512), lds, stream, p);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("wgrad3");
  if (pl.splits > 1 && !fold) return wgrad_reduce_slabs(ws, dw, (int64_t)cout * 9 * cin / 4, pl.splits, stream);
  return DCN_OK;
}
