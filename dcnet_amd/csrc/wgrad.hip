// Weight gradient of the convolution as a split-K GEMM on v_mfma_f32_32x32x2_f32.
//
//   dw[co][t][ci] = sum_m dy[m][co] * x[src(m,t)][ci]        m = (n,ho,wo), t = (r,s)
//
// GEMM view: rows = Cout, cols = Cin (one tap per workgroup), K = the N*Ho*Wo output pixels.
// Both operands are channel-contiguous in HBM, i.e. K is the STRIDED dimension, so the LDS tiles
// are [k = 32 pixels][channels] straight copies (512-B coalesced rows) and MFMA fragments are
// ds_read_b32 (lanes 0-31 consecutive channels, lanes 32-63 the next pixel: conflict-free).
// K is split over workgroups (the early layers have 2.7 M pixels and a 64x32 filter bank);
// partial slabs go to a workspace and are summed in a fixed order by reduce_slabs_kernel, so
// the result is bitwise reproducible (no float atomics).
// Roofline: MFMA (same 157.3 TFLOP/s fp32 peak as the forward).
#include "igemm.h"
#include "prof.h"
#include "slabsum.h"
#include <type_traits>

#ifndef WG_OCC
#define WG_OCC 2            // waves/SIMD the split weight-gradient kernel must fit (2: 205 registers; 3: 168 + 312 B scratch, measured 1.5x slower)
#endif
#ifndef WGRAD_KP
#define WGRAD_KP 16          // pixels per K-step (16: ~32 KB LDS, three workgroups per CU; measured +6 % over 32)
#endif

namespace {

struct WgradParams {
  const float* x; const float* dy; float* out;   // out = dw or the slab workspace
  int N, H, W, Ci, ldx;          // x dims (NHWC)
  int Ho, Wo, Co, lddy;
  int ksize, stride, pad, T;
  int M, kchunk, splits;         // pixels per split (multiple of 32)
  int tiles_co, tiles_ci;
  int c4;                        // stem: x has 4 channels, "ci" axis = 16 taps x 4 (9 real)
  int ld_out;                    // floats per co row of out = T*Ci (c4: 64)
  // batched plain TN GEMM use (co-attention): grid.y = batch
  long long x_bs, dy_bs, out_bs;
  const float* row_scale;        // per output row (index batch*Co + co), may be null
  int Co_ld;                     // dy columns that may be LOADED (>= Co, zero padded by the producer)
  int accumulate;
  const unsigned* geom;          // per output pixel: (centre input pixel << 5) | edge flags; null = identity (plain GEMM)
  const unsigned* amax_dy;       // abs-max words (float bits) of dy / x: the f16 two-piece split derives its scales from them
  const unsigned* amax_x;
  SlabFold fold;                 // split-K: the last-arriving workgroup of a tile sums the slabs (slabsum.h); counters == null: a second launch does
};

// amax bits -> the power of two that maps amax into [2^13, 2^14) (igemm.hip); zero / non-finite maxima give 1
__device__ __forceinline__ float pow2_scale(unsigned amax_bits) {
  const int be = (int)((amax_bits >> 23) & 0xFF);
  if (be == 0 || be == 255) return 1.f;
  int e = 14 - (be - 126);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return __uint_as_float((unsigned)(e + 127) << 23);
}

// geometry table entry flags (dcn_conv2d_geom)
constexpr unsigned GEOM_TOP = 1, GEOM_BOTTOM = 2, GEOM_LEFT = 4, GEOM_RIGHT = 8, GEOM_INVALID = 16;
constexpr int GEOM_SLACK = 128;  // entries past M (flagged invalid): the loader prefetches one K-step ahead

__global__ __launch_bounds__(256) void geom_kernel(unsigned* __restrict__ t, int N, int H, int W, int Ho, int Wo,
                                                   int stride, int pad, int M) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M + GEOM_SLACK) return;
  if (m >= M) { t[m] = GEOM_INVALID; return; }
  const int n = m / (Ho * Wo), rem = m - n * (Ho * Wo);
  const int ho = rem / Wo, wo = rem - ho * Wo;
  const int hy = ho * stride, wx = wo * stride;            // centre tap: always inside the image
  unsigned f = 0;
  if (hy - pad < 0) f |= GEOM_TOP;
  if (hy + pad >= H) f |= GEOM_BOTTOM;
  if (wx - pad < 0) f |= GEOM_LEFT;
  if (wx + pad >= W) f |= GEOM_RIGHT;
  t[m] = ((unsigned)((n * H + hy) * W + wx) << 5) | f;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;
__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, long long bytes) {
  const unsigned n = bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0u : (unsigned)(bytes < 0 ? 0 : bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, n, 0x00020000);
}

// KP = pixels (K) per step.  Loads are raw buffer loads (out-of-range -> 0): the dY rows use a constant
// per-thread voffset plus a wave-uniform soffset that advances by KP rows; the gathered X rows keep their
// (n, ho, wo) coordinates and byte offset incrementally (no division, no multiply in the loop).
// SP ("split", 128x128 tiles only): the bf16 matrix pipe at fp32 accuracy, as in igemm.hip — operands are cut
// into three bf16 pieces when the tile goes to LDS and six cross terms are accumulated with
// v_mfma_f32_32x32x16_bf16.  K (pixels) is the strided dimension of both operands, so the LDS planes
// keep the HBM orientation [pixel][128 channels] (256-B rows, 16-B chunks XOR-swizzled) and the MFMA
// operands (8 consecutive k per lane) come from the hardware transpose read ds_read_b64_tr_b16.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int sp_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
// byte offset of channels c..c+3 (c % 4 == 0) of pixel row `row` inside one [16][128] bf16 plane
__device__ __forceinline__ int sp_off(int row, int c) { return 256 * row + 16 * ((c >> 3) ^ sp_swz(row)) + 8 * ((c >> 2) & 1); }

// NP: 16-bit pieces per operand (3 = fp32-accurate bf16 split, 1 = plain bf16 operands, 2 = fp32-accurate f16 split with
// per-tensor power-of-two scales and three MFMAs per product; igemm.hip)
// IN16 (bf16 storage, NP = 1 only): x and dy ARE bf16 tensors — a 16-byte piece is 8 channels and goes to its LDS plane as it is
// (no conversion, half the loads); strides stay in elements.
template <int TM, int TN, int KP, bool SP = false, int ABL = 0, int NP = 3, bool IN16 = false>
__global__ __launch_bounds__(256, (SP && NP == 3) ? WG_OCC : 2) void wgrad_kernel(const WgradParams p) {
  static_assert(!SP || (TM == 128 && TN == 128 && KP == 16), "split mode: 128x128x16 tiles");
  static_assert(!IN16 || (SP && NP == 1), "bf16 inputs: the one-plane 128x128 tile");
  constexpr int EPP = IN16 ? 8 : 4;                                 // elements per 16-byte piece
  constexpr int ESZ = IN16 ? 2 : 4;                                 // bytes per element
  constexpr int WM = TM >= 64 ? 2 : 1, WN = TN >= 64 ? 2 : 1, WK = 4 / (WM * WN);
  constexpr int MI = TM / (32 * WM), NI = TN / (32 * WN);
  constexpr int A_N = KP * TM / EPP, B_N = KP * TN / EPP;           // 16-B pieces per tile
  constexpr int A_LD = (A_N + 255) / 256, B_LD = (B_N + 255) / 256; // per thread per K-step
  constexpr int KS = KP / 2 / WK;                                   // k2-steps per wave per K-step
  static_assert(KS >= 1, "K-step too small for the wave split");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                 // [2][KP][TM]
  float* Bs = smem + 2 * KP * TM;   // [2][KP][TN]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn / WN, wn = wmn % WN;

  // Block order: split-major, and each XCD gets a contiguous run of it (xcd_remap).  The tiles_co x tiles_ci x T
  // workgroups of one split walk the SAME pixel range at the same pace, so they are served from one XCD's L2
  // (with the split index fastest, the sharers sat 'splits' blocks apart, i.e. on all eight XCDs).
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int per_split = p.tiles_ci * p.tiles_co * p.T;
  const int split = b / per_split; b -= split * per_split;
  const int tci = b % p.tiles_ci; b /= p.tiles_ci;
  const int tco = b % p.tiles_co; b /= p.tiles_co;
  const int t = b;                                   // tap (c4: always 0)
  const int r = t / p.ksize, s = t - r * p.ksize;
  const int co0 = tco * TM, ci0 = tci * TN;
  const int m_begin = split * p.kchunk;
  const int m_end = min(p.M, m_begin + p.kchunk);
  const int iters = (m_end - m_begin + KP - 1) / KP;
  const int howo = p.Ho * p.Wo;

  // descriptors: dY window starts at this split's first row; X window at the first image it touches
  const float* a_base = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.dy) +
                                                       ((long long)blockIdx.y * p.dy_bs + (long long)m_begin * p.lddy) * ESZ);
  const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(a_base, (long long)(p.M - m_begin) * p.lddy * ESZ);
  const int n_first = m_begin / howo;
  const long long ximg = (long long)p.H * p.W * p.ldx;
  const float* b_base = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.x) +
                                                       ((long long)blockIdx.y * p.x_bs + (long long)n_first * ximg) * ESZ);
  const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(b_base, (long long)(p.N - n_first) * ximg * ESZ);

  unsigned a_voff[A_LD]; int a_pix[A_LD];
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    const int idx = tid + 256 * j;
    const int pix = idx / (TM / EPP), c = (idx - pix * (TM / EPP)) * EPP;
    a_pix[j] = pix;
    a_voff[j] = (idx < A_N && co0 + c < p.Co_ld) ? (unsigned)((pix * p.lddy + co0 + c) * ESZ) : OOB;
  }
  // gathered operand: the (n, ho, wo) decomposition of an output pixel never happens in this kernel — the
  // geometry table (dcn_conv2d_geom, built once per conv geometry) holds per output pixel m the index of its
  // centre input pixel and four edge flags; a tap is a wave-uniform pixel delta plus a flag mask.  The entry
  // for the next K-step is fetched while the current one is consumed (one 4-B load per slot, L1/L2 resident).
  int b_pix[B_LD], b_tdelta[B_LD]; unsigned b_ent[B_LD], b_tmask[B_LD], b_coff[B_LD]; bool b_ok[B_LD];
  const int ldx4 = p.ldx * ESZ;
  const int pixbase = n_first * p.H * p.W;
  // (a buffer load whose offset goes out of range when there is no table: no branch around the load — a load inside a
  //  wave-uniform branch makes every later s_waitcnt conservative, here a vmcnt(0) in front of each K-step's tile loads)
  const bool has_geom = p.geom != nullptr;
  const __amdgpu_buffer_rsrc_t g_rs = make_rsrc(has_geom ? reinterpret_cast<const float*>(p.geom) : p.x,
                                                has_geom ? ((long long)p.M + GEOM_SLACK) * 4 : 0);
  auto entry = [&](int m) -> unsigned {
    const unsigned v = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(g_rs, has_geom ? (unsigned)m * 4u : OOB, 0, 0);
    return has_geom ? v : (unsigned)m << 5;
  };
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int idx = tid + 256 * j;
    const int pix = idx / (TN / EPP), c = (idx - pix * (TN / EPP)) * EPP;
    b_pix[j] = pix;
    int rr = r, ss = s, ci = ci0 + c;
    bool ok = idx < B_N && ci < p.Ci;
    if (p.c4) { const int tap = c >> 2; rr = tap / 3; ss = tap - 3 * rr; ci = 0; ok = idx < B_N && tap < 9; }
    b_ok[j] = ok;
    b_tdelta[j] = (rr - p.pad) * p.W + (ss - p.pad) - pixbase;
    b_tmask[j] = (rr < p.pad ? GEOM_TOP : 0u) | (rr > p.pad ? GEOM_BOTTOM : 0u) | (ss < p.pad ? GEOM_LEFT : 0u) |
                 (ss > p.pad ? GEOM_RIGHT : 0u) | GEOM_INVALID;
    b_coff[j] = (unsigned)ci * (unsigned)ESZ;
    b_ent[j] = entry(m_begin + pix);
  }
  int m_cur = m_begin;

  f32x4 a_reg[A_LD], b_reg[B_LD];
  f32x4 a_reg2[SP ? A_LD : 1], b_reg2[SP ? B_LD : 1];    // SP: second register stage (two K-steps of loads in flight)
  unsigned a_soff = 0; int rows_left = m_end - m_begin;
  // Past the end of the split (rows_left <= 0) every offset is out of range: the K loop calls this unconditionally (a load
  // inside a wave-uniform branch makes the s_waitcnt in front of the older stage's LDS stores wait for the loads just
  // issued as well, igemm.hip).  The geometry entries of the NEXT step are requested before this step's tiles: vmcnt
  // retires in order, so waiting for an entry that was issued behind the tiles meant waiting for the tiles.
  auto load_tiles_into = [&](f32x4* a_reg, f32x4* b_reg) {
    m_cur += KP;
    unsigned e_cur[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      e_cur[j] = b_ent[j];
      b_ent[j] = entry(m_cur + b_pix[j]);          // next K-step's pixel (the table is padded past M by GEOM_SLACK)
    }
#pragma unroll
    for (int j = 0; j < A_LD; ++j)
      a_reg[j] = buf_load16(a_rs, a_pix[j] < rows_left ? a_voff[j] : OOB, a_soff);
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const unsigned e = e_cur[j];
      const bool ok = b_ok[j] && b_pix[j] < rows_left && !(e & b_tmask[j]);
      const unsigned voff = (unsigned)(((int)(e >> 5) + b_tdelta[j]) * ldx4) + b_coff[j];
      b_reg[j] = buf_load16(b_rs, ok ? voff : OOB, 0);
    }
    a_soff += (unsigned)(KP * p.lddy * ESZ); rows_left -= KP;
  };
  auto load_tiles = [&]() { load_tiles_into(a_reg, b_reg); };
  unsigned char* sp_base = reinterpret_cast<unsigned char*>(smem);   // SP: [2 buf][A,B][NP planes][16][256 B]
  constexpr int SP_OPND = NP * 4096, SP_BUF = 2 * SP_OPND;
  float s_a = 1.f, s_b = 1.f;
  if constexpr (SP && NP == 2) { s_a = pow2_scale(amax_read(p.amax_dy)); s_b = pow2_scale(amax_read(p.amax_x)); }
  auto split_store = [&](unsigned char* plane0, int off, const f32x4 v, const float sc) {
    if constexpr (NP == 2) {
      typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
      const f32x4 t = v * sc;
      const f16x4_t h = {(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
      const f16x4_t l = {(_Float16)(t[0] - (float)h[0]), (_Float16)(t[1] - (float)h[1]), (_Float16)(t[2] - (float)h[2]),
                         (_Float16)(t[3] - (float)h[3])};
      *reinterpret_cast<uint2*>(plane0 + off) = __builtin_bit_cast(uint2, h);
      *reinterpret_cast<uint2*>(plane0 + 4096 + off) = __builtin_bit_cast(uint2, l);
      return;
    }
    if constexpr (NP == 1) {
      typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
      const bf16x4_t b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      *reinterpret_cast<uint2*>(plane0 + off) = __builtin_bit_cast(uint2, b);
      return;
    }
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x = v[e];
      if (ABL == 1) { h[e] = m[e] = l[e] = __float_as_uint(x); continue; }   // timing ablation: no split arithmetic
      h[e] = __float_as_uint(x) & 0xFFFF0000u;
      const float r1 = x - __uint_as_float(h[e]);
      m[e] = __float_as_uint(r1) & 0xFFFF0000u;
      l[e] = __float_as_uint(r1 - __uint_as_float(m[e]));
    }
    uint2 ph = {__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u)};
    uint2 pm = {__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
    uint2 pl = {__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
    *reinterpret_cast<uint2*>(plane0 + off) = ph;
    *reinterpret_cast<uint2*>(plane0 + 4096 + off) = pm;
    *reinterpret_cast<uint2*>(plane0 + 8192 + off) = pl;
  };
  // SP: piece j of the A / B tile (pixel, 4 channels) -> the three planes of LDS buffer `buf`
  auto sp_store_piece = [&](int buf, int pc, const f32x4* ar, const f32x4* br) {
    const bool isb = pc >= A_LD;
    const int j = isb ? pc - A_LD : pc;
    const int idx = tid + 256 * j;
    if constexpr (IN16) {               // 8 bf16 channels of a pixel: one 16-byte chunk of the plane, as loaded
      const int pix = idx / 16, c = (idx - pix * 16) * 8;
      *reinterpret_cast<f32x4*>(sp_base + buf * SP_BUF + (isb ? SP_OPND : 0) + sp_off(pix, c)) = isb ? br[j] : ar[j];
      return;
    }
    const int pix = idx / 32, c = (idx - pix * 32) * 4;
    split_store(sp_base + buf * SP_BUF + (isb ? SP_OPND : 0), sp_off(pix, c), isb ? br[j] : ar[j], isb ? s_b : s_a);
  };
  auto store_tiles = [&](int buf) {
    if constexpr (SP) {
#pragma unroll
      for (int pc = 0; pc < A_LD + B_LD; ++pc) sp_store_piece(buf, pc, a_reg, b_reg);
      return;
    }
#pragma unroll
    for (int j = 0; j < A_LD; ++j)
      if (A_N % 256 == 0 || tid + 256 * j < A_N) *reinterpret_cast<f32x4*>(As + buf * KP * TM + (tid + 256 * j) * 4) = a_reg[j];
#pragma unroll
    for (int j = 0; j < B_LD; ++j)
      if (B_N % 256 == 0 || tid + 256 * j < B_N) *reinterpret_cast<f32x4*>(Bs + buf * KP * TN + (tid + 256 * j) * 4) = b_reg[j];
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mi][ni][q] = 0.f;

  if (iters > 0) {
    load_tiles();
    store_tiles(0);
  }
  __syncthreads();
  // fragment element for k2-step ks: pixel row 2*ks + (lane>>5), channel (lane&31)
  const int a_off = (wk * KS * 2 + (lane >> 5)) * TM + wm * (TM / WM) + (lane & 31);
  const int b_off = (wk * KS * 2 + (lane >> 5)) * TN + wn * (TN / WN) + (lane & 31);
  if constexpr (SP) {
    // transposed-read addresses: 16-lane group g16 = (h, gg): k rows 8h + 4r + q, channels blk*32 + 16gg + 4pp
    const int g16 = lane >> 4, hh = g16 >> 1, gg = g16 & 1, q = (lane & 15) >> 2, pp = lane & 3;
    int a_tr[MI][2], b_tr[NI][2];
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      const int row = 8 * hh + 4 * r2 + q;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a_tr[mi][r2] = sp_off(row, wm * 64 + mi * 32 + 16 * gg + 4 * pp);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b_tr[ni][r2] = SP_OPND + sp_off(row, wn * 64 + ni * 32 + 16 * gg + 4 * pp);
    }
    auto tr_read = [&](int byte_off) {
      return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(sp_base + byte_off));
    };
    auto frag = [&](int byte0, int byte1) {
      const s16x4 lo = tr_read(byte0), hi = tr_read(byte1);
      s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(bf16x8, v);
    };
    auto step = [&](int cur, const f32x4* ar, const f32x4* br, auto do_store) {
      bf16x8 af[MI][NP], bf[NI][NP];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          af[mi][pl] = frag(cur * SP_BUF + pl * 4096 + a_tr[mi][0], cur * SP_BUF + pl * 4096 + a_tr[mi][1]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          bf[ni][pl] = frag(cur * SP_BUF + pl * 4096 + b_tr[ni][0], cur * SP_BUF + pl * 4096 + b_tr[ni][1]);
      constexpr int TERMS = NP == 3 ? 6 : (NP == 2 ? 3 : 1);
      constexpr int QA[6] = {NP == 3 ? 2 : (NP == 2 ? 1 : 0), 0, NP == 2 ? 0 : 1, 1, 0, 0}, QB[6] = {0, NP == 2 ? 1 : 2, NP == 2 ? 0 : 1, 0, 1, 0};
#pragma unroll
      for (int t6 = 0; t6 < TERMS; ++t6) {
        if (ABL != 2 || t6 == 5)        // timing ablation 2: one of the six MFMA groups
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if constexpr (NP == 2) {
              typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, af[mi][QA[t6]]),
                                                                   __builtin_bit_cast(f16x8_t, bf[ni][QB[t6]]), acc[mi][ni], 0, 0, 0);
            } else
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi][QA[t6]], bf[ni][QB[t6]], acc[mi][ni], 0, 0, 0);
          }
        if constexpr (decltype(do_store)::value) {
          if constexpr (NP == 3) { if (t6 < A_LD + B_LD) sp_store_piece(cur ^ 1, t6, ar, br); }
          else if constexpr (NP == 2) {          // four pieces over three MFMA groups
#pragma unroll
            for (int pc = t6; pc < A_LD + B_LD; pc += 3) sp_store_piece(cur ^ 1, pc, ar, br);
          }
          else {
#pragma unroll
            for (int pc = 0; pc < A_LD + B_LD; ++pc) sp_store_piece(cur ^ 1, pc, ar, br);
          }
        }
      }
      if constexpr (decltype(do_store)::value && (NP == 3 || NP == 2)) {
#pragma unroll
        for (int g = 0; g < MI * NI * TERMS; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
          if (g & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      }
    };
    // stage 2 holds step it+1 while step it is multiplied and stage 1 receives step it+2; the split + LDS
    // store of step it+1 rides between the MFMAs of step it
    using T = std::true_type; using F = std::false_type;
    if (iters > 1) load_tiles_into(a_reg2, b_reg2);
    int it = 0;
    for (; it + 2 < iters; it += 2) {
      load_tiles_into(a_reg, b_reg);                       // step it+2
      step(0, a_reg2, b_reg2, T{});
      __syncthreads();
      load_tiles_into(a_reg2, b_reg2);                     // step it+3 (all zeros past the end)
      step(1, a_reg, b_reg, T{});
      __syncthreads();
    }
    if (it + 1 < iters) {
      step(0, a_reg2, b_reg2, T{});
      __syncthreads();
      step(1, a_reg, b_reg, F{});
    } else if (it < iters) {
      step(0, a_reg, b_reg, F{});
    }
    __syncthreads();
  } else
  for (int it = 0; it < iters; ++it) {
    const int cur = it & 1;
    if (it + 1 < iters) load_tiles();
    const float* a = As + cur * KP * TM + a_off;
    const float* bb = Bs + cur * KP * TN + b_off;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = a[ks * 2 * TM + mi * 32];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[ni] = bb[ks * 2 * TN + ni * 32];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    }
    if (it + 1 < iters) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- combine the WK wave-level K-slices through LDS, then store ---------------------------
  if (WK > 1) {
    float* red = smem;                                 // [(WK-1)][WM*WN][MI*NI*16][64]
    if (wk > 0) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int q = 0; q < 16; ++q)
            red[((((wk - 1) * (WM * WN) + wmn) * (MI * NI) + mi * NI + ni) * 16 + q) * 64 + lane] = acc[mi][ni][q];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int k = 1; k < WK; ++k)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int q = 0; q < 16; ++q)
              acc[mi][ni][q] += red[((((k - 1) * (WM * WN) + wmn) * (MI * NI) + mi * NI + ni) * 16 + q) * 64 + lane];
    }
  }
  if (wk != 0 && !p.fold.counters) return;
  if constexpr (SP && NP == 2) {
    const float dq = 1.f / (s_a * s_b);              // powers of two: exact
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] *= dq;
  }
  float* out = p.out + (size_t)split * p.Co * p.ld_out + (long long)blockIdx.y * p.out_bs;
  if (wk == 0)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int co = co0 + wm * (TM / WM) + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
      if (co >= p.Co) continue;
      const float rs = p.row_scale ? p.row_scale[(size_t)blockIdx.y * p.Co + co] : 1.f;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int ci = ci0 + wn * (TN / WN) + ni * 32 + (lane & 31);
        if (p.c4) { if (ci < 64) out[(size_t)co * 64 + ci] = acc[mi][ni][q]; }
        else if (ci < p.Ci) {
          float* o = out + (size_t)co * p.ld_out + t * p.Ci + ci;
          float v = acc[mi][ni][q] * rs;
          if (p.accumulate) v += *o;
          *o = v;
        }
      }
    }
  if (p.fold.counters) {          // (wave-uniform; conv2d weight gradients only: one batch, no row scale, no accumulate)
    const int group = (t * p.tiles_co + tco) * p.tiles_ci + tci;
    const int ncol = min(TN, (p.c4 ? 64 : p.Ci) - ci0);
    slab_fold<256>(p.fold, group, p.splits, p.out, (size_t)p.Co * p.ld_out, co0, min(TM, p.Co - co0), p.ld_out,
                   (p.c4 ? 0 : t * p.Ci) + ci0, ncol, 1, 0, reinterpret_cast<int*>(smem));
  }
}

// ---- 1x1 stride-1 layers on a (64 MI) x 256 tile (round 5) -----------------------------------------------------------------------------------
// The 128 x 128 tile stages 32 KB per 12 MFMAs per wave; on the 1x1 layers (no taps to share a tile between) it runs at 0.19-0.33 of the
// ceiling.  tools/wgrad_tile_probe.py — the TN form of gemm3.hip on these shapes, operands split beforehand: an upper bound — measured a
// 256 x 256 tile 24-37 % faster than this kernel on every 1x1 layer of the step.  Here: MI = 4: 256 filters x 256 channels, MI = 2: 128 x 256
// (the 256 -> 128 layers); eight waves (2 x 4), each (32 MI) x 64 = MI x 2 accumulator blocks of 32 x 32; K-step = 16 pixels; both tiles
// staged through registers with the f16 split (two K-steps of loads in flight), the LDS planes of the 128 x 128 build ([16][128 channels],
// same XOR: sp_off) side by side, the same transposed reads.  One rotating fragment set: terms (l,h) (h,h) (h,l), the next K-step's
// fragments read behind the barrier into the registers the running term has freed.  Split-K slabs as above.
template <int MI>
__global__ __launch_bounds__(512, 2) void wgrad1x_kernel(const WgradParams p) {
  typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
  typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
  constexpr int TCO = 64 * MI, ASUB = TCO / 128;          // filters per tile; 128-channel sub-planes of the dY tile
  constexpr int A_PL = ASUB * 4096, B_PL = 2 * 4096;      // bytes per plane (h | l) of the dY / X tile of a K-step
  constexpr int B_BASE = 2 * A_PL, STAGE = 2 * A_PL + 2 * B_PL;
  constexpr int A_LD = 16 * TCO / 4 / 512, B_LD = 2;      // 16-byte pieces per thread and K-step
  constexpr int NPC = A_LD + B_LD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int per_split = p.tiles_ci * p.tiles_co;
  const int split = b / per_split; b -= split * per_split;
  const int tci = b % p.tiles_ci, tco = b / p.tiles_ci;
  const int co0 = tco * TCO, ci0 = tci * 256;
  const int m_begin = split * p.kchunk;
  const int m_end = min(p.M, m_begin + p.kchunk);
  const int iters = (m_end - m_begin + 15) / 16;
  const float s_a = pow2_scale(amax_read(p.amax_dy)), s_b = pow2_scale(amax_read(p.amax_x));
  const __amdgpu_buffer_rsrc_t a_rs = make_rsrc(p.dy + (long long)m_begin * p.lddy, ((long long)(m_end - m_begin - 1) * p.lddy + p.Co) * 4);
  const __amdgpu_buffer_rsrc_t b_rs = make_rsrc(p.x + (long long)m_begin * p.ldx, ((long long)(m_end - m_begin - 1) * p.ldx + p.Ci) * 4);

  // pieces of a K-step: A piece j: index tid + 512 j over [16 pixels][TCO / 4]; B piece j over [16 pixels][64]
  unsigned voff[NPC]; int lds_off[NPC];
#pragma unroll
  for (int j = 0; j < NPC; ++j) {
    const bool isb = j >= A_LD;
    const int idx = tid + 512 * (isb ? j - A_LD : j);
    const int per_row = isb ? 64 : TCO / 4;
    const int pix = idx / per_row, c = (idx - pix * per_row) * 4;
    const bool ok = isb ? ci0 + c < p.Ci : co0 + c < p.Co;
    voff[j] = ok ? (unsigned)((pix * (isb ? p.ldx : p.lddy) + (isb ? ci0 : co0) + c) * 4) : OOB;
    lds_off[j] = (isb ? B_BASE : 0) + (c >> 7) * 4096 + sp_off(pix, c & 127);
  }
  const unsigned a_step = (unsigned)(16 * p.lddy * 4), b_step = (unsigned)(16 * p.ldx * 4);
  int rows_left = m_end - m_begin;              // rows of the split not yet requested
  unsigned a_soff = 0, b_soff = 0;
  auto load_step = [&](f32x4 (&r)[NPC]) {      // every load unconditional: past the end of the split the rows are out of range (zeros)
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const bool isb = j >= A_LD;
      const int idx = tid + 512 * (isb ? j - A_LD : j);
      const int pix = idx / (isb ? 64 : TCO / 4);
      r[j] = buf_load16(isb ? b_rs : a_rs, pix < rows_left ? voff[j] : OOB, isb ? b_soff : a_soff);
    }
    a_soff += a_step; b_soff += b_step; rows_left -= 16;
  };
  auto store_piece = [&](const int buf, const int j, const f32x4 v) {
    const bool isb = j >= A_LD;
    const f32x4 t = v * (isb ? s_b : s_a);
    const f16x4_t h = {(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
    const f16x4_t l = {(_Float16)(t[0] - (float)h[0]), (_Float16)(t[1] - (float)h[1]), (_Float16)(t[2] - (float)h[2]),
                       (_Float16)(t[3] - (float)h[3])};
    unsigned char* q = smw + buf * STAGE + lds_off[j];
    *reinterpret_cast<uint2*>(q) = __builtin_bit_cast(uint2, h);
    *reinterpret_cast<uint2*>(q + (isb ? B_PL : A_PL)) = __builtin_bit_cast(uint2, l);
  };

  f32x16 acc[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mi][ni][q] = 0.f;

  // transposed-read addresses (wgrad_kernel's): 16-lane group (hh, gg): k rows 8 hh + 4 r2 + q, channels blk 32 + 16 gg + 4 pp
  const int g16 = lane >> 4, hh = g16 >> 1, gg = g16 & 1, qq = (lane & 15) >> 2, pp = lane & 3;
  int a_tr[MI][2], b_tr[2][2];
  {
    const int a_ch0 = wm * (TCO / 2), b_ch0 = wn * 64;       // first channel of this wave inside the tile
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      const int row = 8 * hh + 4 * r2 + qq;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int c = a_ch0 + mi * 32 + 16 * gg + 4 * pp;
        a_tr[mi][r2] = (c >> 7) * 4096 + sp_off(row, c & 127);
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int c = b_ch0 + ni * 32 + 16 * gg + 4 * pp;
        b_tr[ni][r2] = B_BASE + (c >> 7) * 4096 + sp_off(row, c & 127);
      }
    }
  }
  auto tr_read = [&](int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smw + byte_off));
  };
  auto frag = [&](int byte0, int byte1) {
    const s16x4 lo = tr_read(byte0), hi = tr_read(byte1);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8_t, v);
  };
  f16x8_t Ah[MI], Al[MI], Bh[2], Bl[2];
  auto read_A = [&](f16x8_t (&A)[MI], const int buf, const int plane) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) A[mi] = frag(buf * STAGE + plane * A_PL + a_tr[mi][0], buf * STAGE + plane * A_PL + a_tr[mi][1]);
  };
  auto read_B = [&](f16x8_t (&B)[2], const int buf, const int plane) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) B[ni] = frag(buf * STAGE + plane * B_PL + b_tr[ni][0], buf * STAGE + plane * B_PL + b_tr[ni][1]);
  };
  auto mm = [&](const f16x8_t (&A)[MI], const f16x8_t (&B)[2]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[mi], B[ni], acc[mi][ni], 0, 0, 0);
  };

  // K-step `it` multiplies buffer it & 1; the pieces of step it + 1 (`cur`, loaded one step ago) are split into the other buffer behind the
  // first two MFMA groups, the pieces of step it + 2 are loaded meanwhile (`nxt`); barrier; behind the third group the fragments of it + 1.
  f32x4 r0[NPC], r1[NPC];
  if (iters > 0) {
    load_step(r0);
#pragma unroll
    for (int j = 0; j < NPC; ++j) store_piece(0, j, r0[j]);
    load_step(r0);
  }
  __syncthreads();
  read_A(Al, 0, 1); read_B(Bh, 0, 0);
  auto step = [&](const int cb, f32x4 (&cur)[NPC], f32x4 (&nxt)[NPC]) {
    read_A(Ah, cb, 0); read_B(Bl, cb, 1);
    load_step(nxt);
    __builtin_amdgcn_sched_barrier(0);
    mm(Al, Bh);
#pragma unroll
    for (int j = 0; j < NPC / 2; ++j) store_piece(cb ^ 1, j, cur[j]);
    __builtin_amdgcn_sched_barrier(0);
    mm(Ah, Bh);
#pragma unroll
    for (int j = NPC / 2; j < NPC; ++j) store_piece(cb ^ 1, j, cur[j]);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_A(Al, cb ^ 1, 1); read_B(Bh, cb ^ 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mm(Ah, Bl);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int it = 0; it < iters; it += 2) {
    step(0, r0, r1);
    if (it + 1 < iters) step(1, r1, r0);
  }
  __syncthreads();

  const float dq = 1.f / (s_a * s_b);                   // powers of two: exact
  float* out = p.out + (size_t)split * p.Co * p.ld_out;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int co = co0 + wm * (TCO / 2) + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
      if (co >= p.Co) continue;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int ci = ci0 + wn * 64 + ni * 32 + (lane & 31);
        if (ci < p.Ci) out[(size_t)co * p.ld_out + ci] = acc[mi][ni][q] * dq;
      }
    }
  if (p.fold.counters)
    slab_fold<512>(p.fold, tco * p.tiles_ci + tci, p.splits, p.out, (size_t)p.Co * p.ld_out, co0, min(TCO, p.Co - co0), p.ld_out, ci0,
                   min(256, p.Ci - ci0), 1, 0, reinterpret_cast<int*>(smw));
}

// Sums the split-K slabs in a fixed order (bitwise reproducible).  Eight independent 16-B loads are in
// flight per thread: with one dependent load per split the pass ran at a fifth of the HBM rate.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                           int64_t n4, int splits) {
  const f32x4* __restrict__ w = reinterpret_cast<const f32x4*>(ws);
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 s = w[i];
    int k = 1;
    for (; k + 8 <= splits; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = w[i + (int64_t)(k + u) * n4];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < splits; ++k) s += w[i + (int64_t)k * n4];
    reinterpret_cast<f32x4*>(out)[i] = s;
  }
}

struct Plan { int tm, tn, tiles_co, tiles_ci, T, splits, kchunk, M, ld_out; };
int g_wide64 = 1;           // dcn_set_tuning("cwide64", 0): 64-channel sides back on the narrow fp32-pipe tiles
int g_wg_target = 1024;    // dcn_set_tuning("xwgtarget", n): workgroups a 3x3 stride-1 weight-gradient launch aims for (split-K sizing)
int g_wg_target_small_b16 = 512;   // ("qsmallb16")
int g_wg_target_b16 = 768;     // dcn_set_tuning("qtargetb16", n): the same for the bf16-storage 3x3 stride-1 layers (make_plan_b16)
int g_wg_target_small = 512;   // dcn_set_tuning("zwgsmall", n): the same for 1x1 and stride-2 layers
int dispatch_wgrad(const WgradParams& p, int tm, int tn, int grid, int batch, hipStream_t stream);

Plan make_plan(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
  Plan pl;
  const int pad = (ksize - 1) / 2;
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (wd + 2 * pad - ksize) / stride + 1;
  pl.M = n * ho * wo;
  const bool c4 = cin == 4;
  const int ci_axis = c4 ? 64 : cin;
  pl.tm = cout >= 128 ? 128 : (cout >= 64 ? 64 : 32);
  pl.tn = ci_axis >= 128 ? 128 : (ci_axis >= 64 ? 64 : 32);
  // one side at 64 channels: the 128x128 tile, half empty, has the f16-split build (the narrow tiles run on the fp32 pipe)
  if (g_wide64 && !c4 && cout >= 64 && ci_axis >= 64 && (cout >= 128 || ci_axis >= 128)) pl.tm = pl.tn = 128;
  pl.tiles_co = cdiv(cout, pl.tm); pl.tiles_ci = cdiv(ci_axis, pl.tn);
  pl.T = c4 ? 1 : ksize * ksize;
  pl.ld_out = c4 ? 64 : pl.T * cin;
  const int base = pl.tiles_co * pl.tiles_ci * pl.T;
  // Workgroup count = base * splits.  256 CUs x 2 resident workgroups = 512 slots: aim for at most 1024
  // workgroups (two full rounds) and never for "a round plus a few" — 1026 workgroups cost a third round
  // for 2 of them (measured: -20 % on the 3x3 layers with the naive ceil(1024/base) choice).
  const int max_splits = pl.M / 256 > 0 ? pl.M / 256 : 1;   // at least 8 K-steps per split
  // 1x1 and stride-2 layers on the 128x128 tile (2 workgroups per CU = 512 slots) have few tiles, so the slab traffic of many
  // splits (splits x |W| written and read again) outweighs the finer balance: one full round of 512 workgroups beats 1024 by
  // 5-20 % there (256->128 @52: 0.105 -> 0.088 ms, 1024->512 @13: 0.079 -> 0.064), 256 loses again; the 3x3 stride-1 layers
  // lose 15-90 % at 512, and the narrow tiles of the 32/64-channel layers (more workgroups per CU, 11 M pixels) want all 1024.
  const int target = ((ksize == 1 || stride == 2) && pl.tm == 128 && pl.tn == 128) ? g_wg_target_small : g_wg_target;
  int splits = target / base;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  for (;;) {
    pl.kchunk = cdiv(cdiv(pl.M, splits), 32) * 32;
    pl.splits = cdiv(pl.M, pl.kchunk);
    if (base * pl.splits <= target || splits == 1 || base > target) break;
    --splits;
  }
  return pl;
}

int g_wabl = 0;            // timing-only ablations of the split kernel (wrong results): dcn_set_tuning("abl", v)
int g_wsplit = 4;          // 128x128 weight-gradient / TN tiles: 0 native fp32 MFMA, 1 bf16 three-piece split, 2 bf16 operands,
                           // 4 f16 two-piece split where the operands carry their abs-max (else as 1)   (dcn_set_tuning("precision"|"wsplit"))

int g_slab_fold = 0;       // dcn_set_tuning("Slabfold", KB): split-K slabs are summed by the last-arriving workgroup of a tile (slabsum.h) when that workgroup
                           // has at most this much to read; 0 = always reduce_slabs_kernel behind the launch.  OFF: measured in the replayed step
                           // (profiles/r06_experiments.md) 91.15 ms without, 91.70 / 91.80 at 2.5 / 4.2 MB, 100.06 with every launch folded
int g_wg_lds_pad = 0;      // dcn_set_tuning("lwgpad", KB): dynamic LDS the weight-gradient launches ask for at least (81+ = one workgroup per CU)
}
int wgrad_lds_pad() { return g_wg_lds_pad; }
int wgrad_slab_fold() { return g_slab_fold; }
void wgrad_set_slab_fold(int v) { g_slab_fold = v; }
void wgrad_set_lds_pad(int kb) { g_wg_lds_pad = kb * 1024; }
namespace {
template <int TM, int TN, bool SP = false, int ABL = 0, int NP = 3, bool IN16 = false>
int launch_wgrad(const WgradParams& p, int grid, int batch, hipStream_t stream) {
  constexpr int WM = TM >= 64 ? 2 : 1, WN = TN >= 64 ? 2 : 1, WK = 4 / (WM * WN);
  constexpr int KP = SP ? 16 : (WGRAD_KP < 2 * WK ? 2 * WK : WGRAD_KP);
  size_t lds = SP ? (size_t)2 * 2 * NP * 4096 : (size_t)2 * KP * (TM + TN) * sizeof(float);
  const size_t red = (size_t)(WK - 1) * TM * TN * sizeof(float);
  if (red > lds) lds = red;
  if ((size_t)wgrad_lds_pad() > lds) lds = (size_t)wgrad_lds_pad();      // (occupancy experiment: "lwgpad")
  static DcnPerDeviceSize attr_lds;
  if (attr_lds.raise(lds)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<TM, TN, KP, SP, ABL, NP, IN16>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const double n_alg = p.c4 ? 27.0 : (double)p.T * p.Ci;
  const int pid = prof_begin(IN16 ? 42 : SP ? (NP == 1 ? 20 : NP == 2 ? 25 : 17) : p.M < 1024 ? 14 : 5, 2.0 * batch * (double)p.M * p.Co * n_alg, stream);
  hipLaunchKernelGGL((wgrad_kernel<TM, TN, KP, SP, ABL, NP, IN16>), dim3(grid, batch), dim3(256), lds, stream, p);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("wgrad");
  return DCN_OK;
}

int dispatch_wgrad(const WgradParams& p, int tm, int tn, int grid, int batch, hipStream_t stream) {
  if (tm == 128 && tn == 128 && g_wabl && g_wsplit && !p.c4 && p.M >= 1024)
    return g_wabl == 1 ? launch_wgrad<128, 128, true, 1>(p, grid, batch, stream)
         : launch_wgrad<128, 128, true, 2>(p, grid, batch, stream);
  if (tm == 128 && tn == 128 && g_wsplit == 2 && !p.c4 && p.M >= 1024) return launch_wgrad<128, 128, true, 0, 1>(p, grid, batch, stream);
  if (tm == 128 && tn == 128 && g_wsplit == 4 && p.amax_dy && p.amax_x && !p.c4 && p.M >= 1024)
    return launch_wgrad<128, 128, true, 0, 2>(p, grid, batch, stream);                  // f16 two-piece split
  if (tm == 128 && tn == 128) return (g_wsplit && !p.c4 && p.M >= 1024) ? launch_wgrad<128, 128, true>(p, grid, batch, stream)
                                                         : launch_wgrad<128, 128>(p, grid, batch, stream);
  if (tm == 128 && tn == 64) return launch_wgrad<128, 64>(p, grid, batch, stream);
  if (tm == 128 && tn == 32) return launch_wgrad<128, 32>(p, grid, batch, stream);
  if (tm == 64 && tn == 128) return launch_wgrad<64, 128>(p, grid, batch, stream);
  if (tm == 64 && tn == 64) return launch_wgrad<64, 64>(p, grid, batch, stream);
  if (tm == 64 && tn == 32) return launch_wgrad<64, 32>(p, grid, batch, stream);
  if (tm == 32 && tn == 128) return launch_wgrad<32, 128>(p, grid, batch, stream);
  if (tm == 32 && tn == 64) return launch_wgrad<32, 64>(p, grid, batch, stream);
  return launch_wgrad<32, 32>(p, grid, batch, stream);
}

}  // namespace

int wgrad_reduce_slabs(const float* ws, float* dw, int64_t n4, int splits, hipStream_t stream) {
  if (g_slab_fold < 0) return DCN_OK;          // timing-only ablation (dcn_set_tuning("Slabfold", -1)): no slab pass at all, dw is NOT written
  const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  const int pid = prof_begin(30, (double)(splits + 1) * n4 * 16.0, stream);     // HBM-priced: slabs read + dw written
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, stream, ws, dw, n4, splits);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("reduce_slabs");
  return DCN_OK;
}
int wgrad_split_mode() { return g_wsplit; }

void wgrad_set_split(int v) { g_wsplit = v; }
void wgrad_set_abl(int v) { g_wabl = v; }
void wgrad_set_wide64(int v) { g_wide64 = v; }
void wgrad_set_target(int v) { g_wg_target = v > 0 ? v : 1024; }
void wgrad_set_target_small(int v) { g_wg_target_small = v > 0 ? v : 512; }
void wgrad_set_w3_b16(int v);

// C[b][m][n] (+)= row_scale[b][m] * sum_k A[b][k][m] * B[b][k][n]   ("TN" GEMM: K is the strided dim of
// both operands).  A may be loaded up to column m_ld (zero padded by its producer).  No split-K.
int tn_gemm_batched(const float* A, int lda, long long a_bs, const float* B, int ldb, long long b_bs,
                    float* C, int ldc, long long c_bs, const float* row_scale,
                    int M, int m_ld, int N, int K, int batch, int accumulate, hipStream_t stream,
                    const unsigned* amax_a, const unsigned* amax_b) {
  DCN_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0, "tn_gemm: bad argument");
  DCN_CHECK_ARG(lda % 4 == 0 && ldb % 4 == 0 && N % 4 == 0 && m_ld % 4 == 0 && m_ld >= M && m_ld <= lda,
                "tn_gemm: alignment (lda=%d ldb=%d N=%d m_ld=%d)", lda, ldb, N, m_ld);
  WgradParams p{};
  p.dy = A; p.lddy = lda; p.dy_bs = a_bs; p.Co = M; p.Co_ld = m_ld;
  p.x = B; p.ldx = ldb; p.x_bs = b_bs; p.Ci = N;
  p.out = C; p.ld_out = ldc; p.out_bs = c_bs; p.row_scale = row_scale; p.accumulate = accumulate;
  p.amax_dy = amax_a; p.amax_x = amax_b;      // both present: the 128x128 tile takes the f16 two-piece split
  p.N = 1; p.H = 1; p.W = K; p.Ho = 1; p.Wo = K; p.ksize = 1; p.stride = 1; p.pad = 0; p.T = 1;
  p.M = K; p.kchunk = cdiv(K, 32) * 32; p.splits = 1;
  const int tm = M >= 128 ? 128 : (M >= 64 ? 64 : 32), tn = N >= 128 ? 128 : (N >= 64 ? 64 : 32);
  p.tiles_co = cdiv(M, tm); p.tiles_ci = cdiv(N, tn);
  return dispatch_wgrad(p, tm, tn, p.tiles_co * p.tiles_ci, batch, stream);
}

// wgrad3.hip: 3x3 stride-1 layers, one filter row per workgroup (f16 split)
bool wgrad3_shape_ok(int n, int h, int wd, int cin, int cout, int ksize, int stride);
int64_t wgrad3_ws(int n, int h, int wd, int cin, int cout);
int wgrad3_launch(const float* x, int ldx, const float* dy, int lddy, float* dw, float* ws, uint32_t* counters, int n, int h, int wd, int cin, int cout,
                  const uint32_t* amax_x, const uint32_t* amax_dy, int np, hipStream_t stream);

// wgrad9.hip: the 3x3 layers with 32 input channels and 64 filters, nine taps per workgroup (f16 split)
bool wgrad9_shape_ok(int n, int h, int wd, int cin, int cout, int ksize, int stride);
int64_t wgrad9_ws(int n, int h, int wd, int cin, int cout, int stride);
int wgrad9_launch(const float* x, int ldx, const float* dy, int lddy, float* dw, float* ws, uint32_t* counters, int n, int h, int wd, int cin, int cout, int stride,
                  const uint32_t* amax_x, const uint32_t* amax_dy, const DcnPreAct* pre, hipStream_t stream);

// wgrad1x_kernel: which layers, and its split-K plan
int g_w1x = 1;            // dcn_set_tuning("Y1wide", 0): 1x1 stride-1 weight gradients back on the 128 x 128 tile; n > 1: workgroups a launch aims for
struct Plan1x { int mi, tiles_co, tiles_ci, splits, kchunk; };
static bool wgrad1x_shape_ok(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
  // (128 filters — the 128 x 256 build, MI = 2 — measured level with the 128 x 128 tile: 256 -> 128 @52 0.080-0.086 against 0.080 ms; left there)
  if (!g_w1x || ksize != 1 || stride != 1 || cin < 256 || cout <= 128 || cin % 4 || cout % 4) return false;
  const long long m = (long long)n * h * wd;
  return m >= 4096 && m * (cin > cout ? cin : cout) * 4 < 0x7FFFFFF0LL;
}
static Plan1x plan1x(int m, int cin, int cout) {
  Plan1x pl;
  pl.mi = cout <= 128 ? 2 : 4;
  pl.tiles_co = cdiv(cout, 64 * pl.mi); pl.tiles_ci = cdiv(cin, 256);
  const int base = pl.tiles_co * pl.tiles_ci;
  const int target = g_w1x > 1 ? g_w1x : 256;            // one workgroup (eight waves) per CU
  int splits = target / base;
  const int max_splits = m / 256 > 0 ? m / 256 : 1;       // at least 16 K-steps per split
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  pl.kchunk = cdiv(cdiv(m, splits), 16) * 16;
  pl.splits = cdiv(m, pl.kchunk);
  return pl;
}
static int wgrad1x_launch(const float* x, int ldx, const float* dy, int lddy, float* dw, float* ws, uint32_t* counters, int m, int cin, int cout,
                          const uint32_t* amax_x, const uint32_t* amax_dy, hipStream_t stream) {
  const Plan1x pl = plan1x(m, cin, cout);
  DCN_CHECK_ARG(pl.splits == 1 || ws, "conv2d_bwd_weight: workspace required (%d splits)", pl.splits);
  WgradParams p{};
  p.x = x; p.dy = dy; p.out = pl.splits > 1 ? ws : dw;
  p.Ci = cin; p.ldx = ldx; p.Co = cout; p.lddy = lddy; p.M = m; p.kchunk = pl.kchunk; p.splits = pl.splits;
  p.tiles_co = pl.tiles_co; p.tiles_ci = pl.tiles_ci; p.ld_out = cin; p.amax_x = amax_x; p.amax_dy = amax_dy;
  const bool fold = slab_fold_ok(counters, pl.tiles_co * pl.tiles_ci, pl.splits, 64LL * pl.mi * 256 * 4, g_slab_fold);
  if (fold) p.fold = SlabFold{counters, dw};
  const int grid = pl.tiles_co * pl.tiles_ci * pl.splits;
  const size_t lds4 = (size_t)2 * (2 * 2 * 4096 + 2 * 2 * 4096), lds2 = (size_t)2 * (2 * 1 * 4096 + 2 * 2 * 4096);
  static DcnPerDeviceFlag attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad1x_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad1x_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
  }
  const int pid = prof_begin(25, 2.0 * (double)m * cout * cin, stream);
  if (pl.mi == 4) hipLaunchKernelGGL(wgrad1x_kernel<4>, dim3(grid), dim3(512), lds4, stream, p);
  else hipLaunchKernelGGL(wgrad1x_kernel<2>, dim3(grid), dim3(512), lds2, stream, p);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("wgrad1x");
  if (pl.splits > 1 && !fold) return wgrad_reduce_slabs(ws, dw, (int64_t)cout * cin / 4, pl.splits, stream);
  return DCN_OK;
}
void wgrad_set_w1x(int v) { g_w1x = v; }

extern "C" int64_t dcn_conv2d_bwd_weight_ws(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
  const Plan pl = make_plan(n, h, wd, cin, cout, ksize, stride);
  int64_t ws = pl.splits > 1 ? (int64_t)pl.splits * cout * pl.ld_out : 0;
  if (wgrad9_shape_ok(n, h, wd, cin, cout, ksize, stride)) {
    const int64_t w9 = wgrad9_ws(n, h, wd, cin, cout, stride);
    if (w9 > ws) ws = w9;
  }
  if (wgrad3_shape_ok(n, h, wd, cin, cout, ksize, stride)) {       // (which kernel runs depends on the abs-max words: size for both)
    const int64_t w3 = wgrad3_ws(n, h, wd, cin, cout);
    if (w3 > ws) ws = w3;
  }
  if (wgrad1x_shape_ok(n, h, wd, cin, cout, ksize, stride)) {
    const Plan1x p1 = plan1x(n * h * wd, cin, cout);
    const int64_t w1 = p1.splits > 1 ? (int64_t)p1.splits * cout * cin : 0;
    if (w1 > ws) ws = w1;
  }
  return ws;
}

// Weight gradient of a convolution whose input is the RAW output of the conv + BatchNorm layer in front (dcn_conv2d_fwd_pre): the
// activation is formed where X is loaded.  Shapes: those dcn_conv2d_pre_supported accepts.
extern "C" int dcn_conv2d_bwd_weight_pre(const float* x, int ldx, const float* dy, int lddy, float* dw, float* ws, uint32_t* counters,
                                         int n, int h, int wd, int cin, int cout, int ksize, int stride,
                                         const float* pre_scale, const float* pre_shift, int pre_act, float pre_slope,
                                         const uint32_t* amax_x, const uint32_t* amax_dy, void* stream_) {
  DCN_CHECK_ARG(x && dy && dw && pre_scale && pre_shift && amax_x && amax_dy, "conv2d_bwd_weight_pre: null pointer (the abs-max words are required)");
  DCN_CHECK_ARG(g_wsplit == 4 && !g_wabl && cin == 32 && wgrad9_shape_ok(n, h, wd, cin, cout, ksize, stride),
                "conv2d_bwd_weight_pre: no loader-side activation for this shape / precision");
  DCN_CHECK_ARG(pre_act == DCN_ACT_NONE || pre_act == DCN_ACT_LEAKY, "conv2d_bwd_weight_pre: pre_act=%d", pre_act);
  const int lx = ldx > 0 ? ldx : cin, ly = lddy > 0 ? lddy : cout;
  DCN_CHECK_ARG(lx % 4 == 0 && ly % 4 == 0, "conv2d_bwd_weight_pre: pixel strides must be multiples of 4 floats");
  const DcnPreAct pre{pre_scale, pre_shift, pre_act, pre_slope};
  return wgrad9_launch(x, lx, dy, ly, dw, ws, counters, n, h, wd, cin, cout, stride, amax_x, amax_dy, &pre, (hipStream_t)stream_);
}
extern "C" int dcn_conv2d_bwd_weight_pre_supported(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
  return (g_wsplit == 4 && !g_wabl && cin == 32 && wgrad9_shape_ok(n, h, wd, cin, cout, ksize, stride)) ? 1 : 0;
}

extern "C" int64_t dcn_conv2d_geom_size(int n, int h, int wd, int ksize, int stride) {
  const int pad = (ksize - 1) / 2;
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (wd + 2 * pad - ksize) / stride + 1;
  return (int64_t)n * ho * wo + GEOM_SLACK;
}

extern "C" int dcn_conv2d_geom(uint32_t* table, int n, int h, int wd, int ksize, int stride, void* stream_) {
  DCN_CHECK_ARG(table && n > 0 && h > 0 && wd > 0, "conv2d_geom: bad argument");
  DCN_CHECK_ARG((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "conv2d_geom: ksize=%d stride=%d", ksize, stride);
  DCN_CHECK_ARG((int64_t)n * h * wd < (1LL << 27), "conv2d_geom: %lld input pixels exceed the 27-bit index", (long long)n * h * wd);
  const int pad = (ksize - 1) / 2;
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (wd + 2 * pad - ksize) / stride + 1;
  const int M = n * ho * wo;
  hipLaunchKernelGGL(geom_kernel, dim3(cdiv(M + GEOM_SLACK, 256)), dim3(256), 0, (hipStream_t)stream_,
                     table, n, h, wd, ho, wo, stride, pad, M);
  DCN_CHECK_LAUNCH("conv2d_geom");
  return DCN_OK;
}

extern "C" int dcn_conv2d_bwd_weight(const float* x, int ldx, const float* dy, int lddy, float* dw, float* ws, uint32_t* counters,
                                     const uint32_t* geom,
                                     int n, int h, int wd, int cin, int cout, int ksize, int stride,
                                     const uint32_t* amax_x, const uint32_t* amax_dy, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DCN_CHECK_ARG(ksize == 1 || ksize == 3, "conv2d_bwd_weight: ksize=%d", ksize);
  DCN_CHECK_ARG(stride == 1 || stride == 2, "conv2d_bwd_weight: stride=%d", stride);
  DCN_CHECK_ARG(cin == 4 || cin % 4 == 0, "conv2d_bwd_weight: cin=%d must be a multiple of 4", cin);
  DCN_CHECK_ARG(cin != 4 || (ksize == 3 && stride == 1), "conv2d_bwd_weight: cin=4 path is the 3x3 stride-1 stem only");
  DCN_CHECK_ARG(cout % 4 == 0, "conv2d_bwd_weight: cout=%d must be a multiple of 4", cout);
  DCN_CHECK_ARG(x && dy && dw && geom, "conv2d_bwd_weight: null pointer (geom = table of dcn_conv2d_geom for this geometry)");
  {
    const int lx = ldx > 0 ? ldx : cin, ly = lddy > 0 ? lddy : cout;
    const long long npix = (long long)n * h * wd;
    const bool f16 = g_wsplit == 4 && amax_x && amax_dy, b16 = g_wsplit == 2;      // (2: the bf16- and fp8-operand modes)
    if (f16 && !g_wabl && wgrad9_shape_ok(n, h, wd, cin, cout, ksize, stride) && lx % 4 == 0 && ly % 4 == 0)
      return wgrad9_launch(x, lx, dy, ly, dw, ws, counters, n, h, wd, cin, cout, stride, amax_x, amax_dy, nullptr, stream);
    if ((f16 || b16) && !g_wabl && wgrad3_shape_ok(n, h, wd, cin, cout, ksize, stride) &&
        npix * lx * 4 < 0x7FFFFFF0LL && npix * ly * 4 < 0x7FFFFFF0LL && lx % 4 == 0 && ly % 4 == 0)
      return wgrad3_launch(x, lx, dy, ly, dw, ws, counters, n, h, wd, cin, cout, amax_x, amax_dy, f16 ? 2 : 1, stream);
    if (f16 && !g_wabl && wgrad1x_shape_ok(n, h, wd, cin, cout, ksize, stride) && lx % 4 == 0 && ly % 4 == 0 &&
        npix * lx * 4 < 0x7FFFFFF0LL && npix * ly * 4 < 0x7FFFFFF0LL)
      return wgrad1x_launch(x, lx, dy, ly, dw, ws, counters, (int)npix, cin, cout, amax_x, amax_dy, stream);
  }
  const Plan pl = make_plan(n, h, wd, cin, cout, ksize, stride);
  DCN_CHECK_ARG(pl.splits == 1 || ws, "conv2d_bwd_weight: workspace required (%d splits)", pl.splits);
  WgradParams p{};
  p.x = x; p.dy = dy; p.out = pl.splits > 1 ? ws : dw;
  p.N = n; p.H = h; p.W = wd; p.Ci = cin; p.ldx = ldx > 0 ? ldx : cin;
  p.ksize = ksize; p.stride = stride; p.pad = (ksize - 1) / 2; p.T = pl.T;
  p.Ho = (h + 2 * p.pad - ksize) / stride + 1; p.Wo = (wd + 2 * p.pad - ksize) / stride + 1;
  p.Co = cout; p.lddy = lddy > 0 ? lddy : cout;
  p.M = pl.M; p.kchunk = pl.kchunk; p.splits = pl.splits;
  p.tiles_co = pl.tiles_co; p.tiles_ci = pl.tiles_ci; p.c4 = cin == 4; p.ld_out = pl.ld_out;
  p.Co_ld = cout; p.geom = geom; p.amax_x = amax_x; p.amax_dy = amax_dy;
  const bool fold = slab_fold_ok(counters, pl.tiles_co * pl.tiles_ci * pl.T, pl.splits, (long long)pl.tm * pl.tn * 4, g_slab_fold);
  if (fold) p.fold = SlabFold{counters, dw};
  const int grid = pl.tiles_co * pl.tiles_ci * pl.T * pl.splits;
  int rc = dispatch_wgrad(p, pl.tm, pl.tn, grid, 1, stream);
  if (rc != DCN_OK) return rc;
  if (pl.splits > 1 && !fold) {
    return wgrad_reduce_slabs(ws, dw, (int64_t)cout * pl.ld_out / 4, pl.splits, stream);
  }
  return DCN_OK;
}

// ---- bf16 storage: x and dy are bf16 NHWC tensors, dw is fp32 [Cout][k][k][Cin] --------------------------------------------------
// Always the 128 x 128 one-plane tile of wgrad_kernel with bf16 loads (narrower layers leave part of it empty: those launches are
// bound by their 9-fold gather, not by the matrix pipe).  Split-K over the pixels into fp32 slabs, summed in a fixed order.
namespace {
Plan make_plan_b16(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
  Plan pl;
  const int pad = (ksize - 1) / 2;
  const int ho = (h + 2 * pad - ksize) / stride + 1, wo = (wd + 2 * pad - ksize) / stride + 1;
  pl.M = n * ho * wo; pl.tm = pl.tn = 128;
  pl.tiles_co = cdiv(cout, 128); pl.tiles_ci = cdiv(cin, 128);
  pl.T = ksize * ksize; pl.ld_out = pl.T * cin;
  const int base = pl.tiles_co * pl.tiles_ci * pl.T;
  const int max_splits = pl.M / 256 > 0 ? pl.M / 256 : 1;
  // (3x3 stride-1 layers: with one MFMA per product the kernel is twice as fast as the f16-split tile while a slab costs the same, so fewer,
  //  longer splits win — tools/bench_b16.py --ab qtargetb16=..., N = 64: 1024 -> 768 workgroups 128->256 @52 0.169 -> 0.146 ms, 512->512 @52
  //  1.25 -> 1.02, 64->128 @104 0.311 -> 0.243; 512 loses again on the 13-wide maps; the 1x1 / stride-2 layers keep 512: 256 costs 15-70 %)
  // (1x1 / stride-2 layers: one full round of 768 as well once the launch is large — 1024->512 @52 0.303 -> 0.249 ms, 128->256 s2 @104
  //  0.177 -> 0.151 —, 512 for the small residual 1x1 layers, whose slabs weigh more: 256->128 @52 0.048 vs 0.052)
  const bool small_cls = ksize == 1 || stride == 2;
  const int target = small_cls ? (2.0 * pl.M * (double)cin * cout * pl.T >= 3e10 ? g_wg_target_b16 : g_wg_target_small_b16) : g_wg_target_b16;
  int splits = target / base;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  for (;;) {
    pl.kchunk = cdiv(cdiv(pl.M, splits), 32) * 32;
    pl.splits = cdiv(pl.M, pl.kchunk);
    if (base * pl.splits <= target || splits == 1 || base > target) break;
    --splits;
  }
  return pl;
}
}  // namespace

bool wgrad3_b16_ok(int n, int h, int wd, int cin, int cout, int ksize, int stride);
int wgrad3_launch_b16(const void* x, int ldx, const void* dy, int lddy, float* dw, float* ws, uint32_t* counters, int n, int h, int wd, int cin, int cout,
                      hipStream_t stream);
int g_w3_b16 = 0;         // dcn_set_tuning("w3b16", 1): bf16-storage 3x3 stride-1 weight gradients by filter rows (wgrad3.hip, bf16 inputs).  Measured
                          // (tools/bench_b16.py --set w3b16=0 --ab w3b16=1, N = 64): it LOSES to the per-tap tile here — 128->256 @52 0.170 -> 0.207 ms,
                          // 256->512 @26 0.167 -> 0.204, 512->512 @52 1.22 -> 1.50: with one MFMA per product both are bound by the bytes they stage per
                          // FLOP, and the filter-row form stages more (a 16-position step per 3 x 8 MFMAs); off

int wgrad9_launch_b16(const void* x, int ldx, const void* dy, int lddy, float* dw, float* ws, uint32_t* counters, int n, int h, int wd, int cin, int cout, int stride,
                      hipStream_t stream);
int g_w9_b16 = 1;         // dcn_set_tuning("9b16", 0): the 32 -> 64 / 64 -> 128 3x3 layers of the bf16-storage mode back on the per-tap tile

extern "C" int64_t dcn_conv2d_bwd_weight_ws_b16(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
  const Plan pl = make_plan_b16(n, h, wd, cin, cout, ksize, stride);
  int64_t ws = pl.splits > 1 ? (int64_t)pl.splits * cout * pl.ld_out : 0;
  if (wgrad9_shape_ok(n, h, wd, cin, cout, ksize, stride)) { const int64_t w9 = wgrad9_ws(n, h, wd, cin, cout, stride); if (w9 > ws) ws = w9; }
  if (wgrad3_b16_ok(n, h, wd, cin, cout, ksize, stride)) { const int64_t w3 = wgrad3_ws(n, h, wd, cin, cout); if (w3 > ws) ws = w3; }
  return ws;
}

extern "C" int dcn_conv2d_bwd_weight_b16(const void* x, int ldx, const void* dy, int lddy, float* dw, float* ws, uint32_t* counters, const uint32_t* geom,
                                         int n, int h, int wd, int cin, int cout, int ksize, int stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DCN_CHECK_ARG((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "conv2d_bwd_weight_b16: ksize=%d stride=%d", ksize, stride);
  DCN_CHECK_ARG(cin % 8 == 0 && cout % 8 == 0, "conv2d_bwd_weight_b16: cin=%d / cout=%d must be multiples of 8", cin, cout);
  DCN_CHECK_ARG(x && dy && dw && geom, "conv2d_bwd_weight_b16: null pointer (geom = table of dcn_conv2d_geom for this geometry)");
  const int lx = ldx > 0 ? ldx : cin, ly = lddy > 0 ? lddy : cout;
  DCN_CHECK_ARG(lx % 8 == 0 && ly % 8 == 0, "conv2d_bwd_weight_b16: pixel strides must be multiples of 8 elements");
  // the 32 -> 64 (either stride) and 64 -> 128 (stride 1) 3x3 layers: all nine taps per workgroup (wgrad9.hip), dY and X read once
  if (g_w9_b16 && wgrad9_shape_ok(n, h, wd, cin, cout, ksize, stride))
    return wgrad9_launch_b16(x, lx, dy, ly, dw, ws, counters, n, h, wd, cin, cout, stride, stream);
  // 3x3 stride-1 layers with >= 128 channels on a side: one filter row per workgroup (wgrad3.hip), the dY tile staged once for three taps
  if (g_w3_b16 && wgrad3_b16_ok(n, h, wd, cin, cout, ksize, stride) && (long long)n * h * wd * lx * 2 < 0x7FFFFFF0LL &&
      (long long)n * h * wd * ly * 2 < 0x7FFFFFF0LL)
    return wgrad3_launch_b16(x, lx, dy, ly, dw, ws, counters, n, h, wd, cin, cout, stream);
  const Plan pl = make_plan_b16(n, h, wd, cin, cout, ksize, stride);
  DCN_CHECK_ARG(pl.M >= 16, "conv2d_bwd_weight_b16: fewer than 16 output pixels");
  DCN_CHECK_ARG(pl.splits == 1 || ws, "conv2d_bwd_weight_b16: workspace required (%d splits)", pl.splits);
  WgradParams p{};
  p.x = (const float*)x; p.dy = (const float*)dy; p.out = pl.splits > 1 ? ws : dw;
  p.N = n; p.H = h; p.W = wd; p.Ci = cin; p.ldx = lx;
  p.ksize = ksize; p.stride = stride; p.pad = (ksize - 1) / 2; p.T = pl.T;
  p.Ho = (h + 2 * p.pad - ksize) / stride + 1; p.Wo = (wd + 2 * p.pad - ksize) / stride + 1;
  p.Co = cout; p.lddy = ly;
  p.M = pl.M; p.kchunk = pl.kchunk; p.splits = pl.splits;
  p.tiles_co = pl.tiles_co; p.tiles_ci = pl.tiles_ci; p.c4 = 0; p.ld_out = pl.ld_out;
  p.Co_ld = cout; p.geom = geom;
  const bool fold = slab_fold_ok(counters, pl.tiles_co * pl.tiles_ci * pl.T, pl.splits, (long long)pl.tm * pl.tn * 4, g_slab_fold);
  if (fold) p.fold = SlabFold{counters, dw};
  const int grid = pl.tiles_co * pl.tiles_ci * pl.T * pl.splits;
  int rc = launch_wgrad<128, 128, true, 0, 1, true>(p, grid, 1, stream);
  if (rc != DCN_OK) return rc;
  if (pl.splits > 1 && !fold) return wgrad_reduce_slabs(ws, dw, (int64_t)cout * pl.ld_out / 4, pl.splits, stream);
  return DCN_OK;
}

void wgrad_set_w3_b16(int v) { g_w3_b16 = v; }
void wgrad_set_w9_b16(int v) { g_w9_b16 = v; }
void wgrad_set_target_b16(int v, int small) { if (small) g_wg_target_small_b16 = v > 0 ? v : 512; else g_wg_target_b16 = v > 0 ? v : 768; }
