// Weight gradient of the 3x3 layers with 32 input channels and 64 filters — the 32 -> 64 stride-2 layer on the 416x416 map and the
// 32 -> 64 layer of the first residual block on the 208x208 map (yolov3.cfg) — ALL NINE taps per workgroup, f16 two-piece split.
//
//   dW[co][r][s][ci] = sum_q dY[q][co] * X[S*oy + r - 1][S*ox + s - 1][ci]          q = (n, oy, ox), S = stride
//
// wgrad.hip runs these layers as nine per-tap launches' worth of 64 x 32 tiles on the fp32 MFMA pipe: dY (0.7 GB) and X (0.35 / 1.4 GB)
// are read nine times through L2 and the pipe runs at 1/16 of the f16 rate (1.4 ms per layer against 0.2-0.4 ms of HBM traffic).
// With 64 x 32 x 9 = 18 432 accumulators the whole filter bank fits one workgroup: six waves, wave (cb, r) owns 32 filters x 32
// channels x the three taps of filter row r.  K runs over PADDED positions as in wgrad3.hip: q counts rows of Wo + 1 entries whose
// last entry is a pad (dY = 0), and X is addressed as X_pad[S*q + s] in rows of S*(Wo + 1) entries whose entry u is image column
// u - 1 (u = 0 and u > W are zero) — then the left / right taps of the border columns read zeros without a mask, for both strides,
// and the top / bottom filter rows are zeroed when their image row is loaded.  Per K-step of 32 positions a workgroup stages dY
// (32 x 64) and, per filter row, the S*31 + 3 entries of X_pad its three taps touch (stride 2: even and odd entries in separate
// planes, so that every tap reads consecutive rows); both are split into f16 pieces on the way to LDS and the MFMA operands come
// from ds_read_b64_tr_b16 (wgrad.hip).  Split-K over positions into slabs summed in a fixed order (reduce_slabs_kernel):
// bitwise reproducible.  Roofline: HBM (dY + X once: 1.06 / 2.13 GB per launch at N = 64) — the MFMA work is 0.12 ms at 838.9.
#include "igemm.h"
#include "prof.h"

int wgrad_reduce_slabs(const float* ws, float* dw, int64_t n4, int splits, hipStream_t stream);

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned OOB9 = 0x80000000u;

struct W9Params {
  const float* x; const float* dy; float* out;
  int N, H, W, Ho, Wo, ldx, lddy;
  int Mp, kchunk, splits;          // padded positions N*Ho*(Wo+1); per split (multiple of 32)
  const unsigned* amax_dy; const unsigned* amax_x;
  // PRE: x is the RAW output of the layer in front; its BatchNorm scale / shift and activation are applied when a piece of X is staged
  // (pads stay zero); amax_x is then the abs-max word of the activation
  const float* pre_scale; const float* pre_shift; int pre_act; float pre_slope;
};

__device__ __forceinline__ f32x4 ld9(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
// a piece of four channels: 16 bytes of fp32, or (IN16) 8 bytes of bf16 carried in the first two lanes of the register set
template <bool IN16> __device__ __forceinline__ f32x4 ld9x(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  if constexpr (IN16) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 w = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0));
    return f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), 0.f, 0.f};
  } else {
    return ld9(r, voff);
  }
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc9(const float* base, long long bytes) {
  const unsigned n = bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0u : (unsigned)(bytes < 0 ? 0 : bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, n, 0x00020000);
}
__device__ __forceinline__ float pow2_9(unsigned amax_bits) {
  const int be = (int)((amax_bits >> 23) & 0xFF);
  if (be == 0 || be == 255) return 1.f;
  int e = 14 - (be - 126);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return __uint_as_float((unsigned)(e + 127) << 23);
}

constexpr int KS = 32;                    // dY positions per K-step (two MFMA K-steps per barrier)

// LDS rows hold a position's channels as f16: 64 | 128 | 256 bytes.  A half-wave of the transposed read takes 4 consecutive rows x one
// 64-byte segment (32 channels); the segment index is XORed with a function of the row so that the four pieces sit in four different
// 64-byte bank groups: 64-byte rows need nothing, 128-byte rows swap their halves on rows 2, 3 (mod 4), 256-byte rows rotate by row % 4.
template <int ROWB> __device__ __forceinline__ int seg_swz(int seg, int row) {
  if constexpr (ROWB == 64) return seg;
  else if constexpr (ROWB == 128) return seg ^ ((row >> 1) & 1);
  else return seg ^ (row & 3);
}

// Geometry of one launch form: CO filters x CI channels, stride S
template <int S, int CO, int CI> struct G9 {
  static constexpr int NW = CO == 64 ? 6 : 8;            // waves: (64, 32): (filter half, filter row); (128, 64): (filter quarter, channel half), all rows
  static constexpr int NR = CO == 64 ? 1 : 3;            // filter rows per wave
  static constexpr int NT = 64 * NW;
  static constexpr int A_ROWB = 2 * CO, X_ROWB = 2 * CI;
  static constexpr int A_PLANE = KS * A_ROWB;
  static constexpr int NU = S == 1 ? KS + 2 : 2 * KS + 1;             // staged X entries per filter row: 34 | 65
  // S = 2: even entries (33 rows) then the odd ones, starting 128 bytes (mod 256) further so that a store of two even and two odd
  // rows spreads over all banks
  static constexpr int ODD = X_ROWB == 64 ? 34 : 33;
  static constexpr int FROWS = S == 1 ? NU : ODD + KS;
  static constexpr int FROWB = FROWS * X_ROWB;                         // one filter row's X plane
  static constexpr int B_PLANE = 3 * FROWB;
  static constexpr int B_BASE = 2 * A_PLANE, BUF = 2 * A_PLANE + 2 * B_PLANE;
  static constexpr int NA = KS * (CO / 4), NX = 3 * NU * (CI / 4), NSLOT = (NA + NX + NT - 1) / NT;
  static_assert(NA % 64 == 0, "a staging slot is one kind per wave");
  static __device__ __forceinline__ int a_off(int row, int c) { return A_ROWB * row + 64 * seg_swz<A_ROWB>(c >> 5, row) + 2 * (c & 31); }
  static __device__ __forceinline__ int x_row(int s) { return S == 1 ? s : ((s & 1) ? ODD + (s >> 1) : (s >> 1)); }                 // LDS row of entry s
  static __device__ __forceinline__ int tap_row(int k, int d) { return S == 1 ? k + d : (d == 1 ? ODD + k : k + (d >> 1)); }        // ... of position k, tap d
  static __device__ __forceinline__ int x_off(int fr, int row, int c) { return fr * FROWB + X_ROWB * row + 64 * seg_swz<X_ROWB>(c >> 5, row) + 2 * (c & 31); }
};

// IN16 (bf16 storage, BASELINE.json configs[2]): x and dy ARE bf16 tensors — a piece of four channels is an 8-byte load that goes to LDS
// as it is (one plane, no scales, no split), the transposed reads hand the MFMA its bf16 operands, one v_mfma_f32_32x32x16_bf16 per product.
template <int S, int CO, int CI, bool PRE = false, bool IN16 = false>
__global__ __launch_bounds__(64 * (CO == 64 ? 6 : 8)) __attribute__((amdgpu_waves_per_eu(CO == 64 ? 3 : 2, CO == 64 ? 3 : 2)))
void wgrad9_kernel(const W9Params p) {
  static_assert(!(PRE && IN16), "the loader-side activation exists for fp32 tensors only");
  constexpr int ESZ = IN16 ? 2 : 4;
  typedef G9<S, CO, CI> G;
  constexpr int NT = G::NT, NR = G::NR, NU = G::NU, A_PLANE = G::A_PLANE, B_PLANE = G::B_PLANE, B_BASE = G::B_BASE, BUF = G::BUF;
  constexpr int NA = G::NA, NX = G::NX, NSLOT = G::NSLOT;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm9[];     // [2 buffers][dY: 2 planes | X: 2 planes x 3 filter rows]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = CO == 64 ? (wave & 1) : (wave & 3);                      // 32-filter block
  const int ib = CO == 64 ? 0 : (wave >> 2);                              // 32-channel block
  const int r0 = CO == 64 ? (wave >> 1) : 0;                              // first filter row of the wave
  const int split = xcd_remap(blockIdx.x, gridDim.x);
  const int Wp = p.Wo + 1, RL = S * Wp;
  const int p_begin = split * p.kchunk;
  const int p_end = min(p.Mp, p_begin + p.kchunk);
  const int iters = (p_end - p_begin + KS - 1) / KS;
  float s_a = 1.f, s_b = 1.f;
  if constexpr (!IN16) { s_a = pow2_9(amax_read(p.amax_dy)); s_b = pow2_9(amax_read(p.amax_x)); }

  const __amdgpu_buffer_rsrc_t a_rs = rsrc9(p.dy, (((long long)p.N * p.Ho * p.Wo - 1) * p.lddy + CO) * ESZ);
  const __amdgpu_buffer_rsrc_t b_rs = rsrc9(p.x, (((long long)p.N * p.H * p.W - 1) * p.ldx + CI) * ESZ);

  // ---- load slots: element e = slot*NT + tid of the K-step's staging list (dY pieces first, then X pieces) ------------------
  // dY piece: position q0 + idx, filters c..c+3;  X piece: filter row fr, entry idx of the staged range, channels c..c+3.
  // NA is a multiple of 64, so a slot is one kind for a whole wave (is_a: scalar — the two kinds use different buffer
  // descriptors and must not meet in one load); elements past the end of the list repeat the last X piece (same bytes, same place).
  // meta = fr << 2 | c << 4 | idx << 11.  The step's first position q0 = (row g_row = n*Ho + oy, column g_col) is the same for
  // the whole workgroup (scalar registers); a slot adds its idx and wraps into the next row at most once.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  int meta[NSLOT], st_off[NSLOT];
  bool is_a[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const int e = j * NT + tid;
    is_a[j] = j * NT + 64 * wave_u < NA;
    if (is_a[j]) {
      const int pos = e / (CO / 4), c = (e % (CO / 4)) * 4;
      meta[j] = (c << 4) | (pos << 11); st_off[j] = G::a_off(pos, c);
    } else {
      const int x = min(e - NA, NX - 1);
      const int fr = x / (NU * (CI / 4)), rem = x - fr * (NU * (CI / 4));
      const int s_ = rem / (CI / 4), c = (rem % (CI / 4)) * 4;
      meta[j] = (fr << 2) | (c << 4) | (s_ << 11); st_off[j] = B_BASE + G::x_off(fr, G::x_row(s_), c);
    }
  }
  // PRE: every X piece of a thread holds the same four channels (NT and NA are multiples of CI/4); the repeats of the list's last
  // piece are switched off instead (dead_last)
  static_assert(NT % (CI / 4) == 0 && NA % (CI / 4) == 0, "one channel group per thread");
  const bool dead_last = PRE && ((NSLOT - 1) * NT + tid - NA > NX - 1);
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  if constexpr (PRE) {
    const int c = (tid % (CI / 4)) * 4;
    psc = *reinterpret_cast<const f32x4*>(p.pre_scale + c) * s_b; psh = *reinterpret_cast<const f32x4*>(p.pre_shift + c) * s_b;
  }
  const bool leaky_max = p.pre_slope >= 0.f && p.pre_slope <= 1.f;
  unsigned vmask = 0;                                     // PRE: which X pieces of the loaded step are real pixels
  int q_step = p_begin;                                   // first position of the step being LOADED
  int g_row = p_begin / Wp, g_col = p_begin - g_row * Wp;
  int g_nb = (g_row / p.Ho) * p.H, g_oy = g_row % p.Ho;
  const int rows_x = p.N * p.H;

  auto load_step = [&](f32x4* v) {
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      unsigned off = OOB9;
      const int c = (meta[j] >> 4) & 127, idx = meta[j] >> 11;
      if (is_a[j]) {
        int col = g_col + idx, row = g_row;
        if (col >= Wp) { col -= Wp; ++row; }
        if (col < p.Wo && q_step + idx < p_end) off = (unsigned)(((row * p.Wo + col) * p.lddy + c) * ESZ);      // (< 2^31: checked by the launcher)
        v[j] = ld9x<IN16>(a_rs, off);
      } else {
        int u = S * g_col + idx, oy = g_oy, nb = g_nb;
        if (u >= RL) { u -= RL; if (++oy == p.Ho) { oy = 0; nb += p.H; } }
        const int iy = S * oy + ((meta[j] >> 2) & 3) - 1, ix = u - 1;
        if ((unsigned)ix < (unsigned)p.W && (unsigned)iy < (unsigned)p.H && nb < rows_x)
          off = (unsigned)((((nb + iy) * p.W + ix) * p.ldx + c) * ESZ);
        v[j] = ld9x<IN16>(b_rs, off);
        if constexpr (PRE) vmask = off != OOB9 ? (vmask | (1u << j)) : (vmask & ~(1u << j));
      }
    }
    q_step += KS; g_col += KS;
    if (g_col >= Wp) { g_col -= Wp; ++g_row; if (++g_oy == p.Ho) { g_oy = 0; g_nb += p.H; } }
  };
  auto store_step = [&](int buf, const f32x4* v) {
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      if (PRE && j == NSLOT - 1 && !is_a[j] && dead_last) continue;
      if constexpr (IN16) {                                  // four bf16 channels, as loaded (pads and rows past the end loaded zeros)
        *reinterpret_cast<uint2*>(sm9 + buf * BUF + st_off[j]) = uint2{__float_as_uint(v[j][0]), __float_as_uint(v[j][1])};
        continue;
      }
      f32x4 t;
      if (PRE && !is_a[j]) {
        // scale_act_kernel's arithmetic (bn.hip) on operands that carry the power-of-two operand scale already (psc, psh = s_b * scale,
        // s_b * shift: exact, and LeakyReLU commutes with a positive factor), then zero for the pads
        t = v[j] * psc + psh;
        if (p.pre_act == DCN_ACT_LEAKY) {
          if (leaky_max) {
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = fmaxf(t[k], t[k] * p.pre_slope);
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = t[k] > 0.f ? t[k] : t[k] * p.pre_slope;
          }
        }
        if (!((vmask >> j) & 1u)) t = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        t = v[j] * (is_a[j] ? s_a : s_b);
      }
      const f16x4_t h = {(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
      const f16x4_t l = {(_Float16)(t[0] - (float)h[0]), (_Float16)(t[1] - (float)h[1]), (_Float16)(t[2] - (float)h[2]),
                         (_Float16)(t[3] - (float)h[3])};
      unsigned char* dst = sm9 + buf * BUF + st_off[j];
      *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, h);
      *reinterpret_cast<uint2*>(dst + (is_a[j] ? A_PLANE : B_PLANE)) = __builtin_bit_cast(uint2, l);
    }
  };

  f32x16 acc[NR * 3];
#pragma unroll
  for (int d = 0; d < NR * 3; ++d)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[d][q] = 0.f;

  // transposed-read addresses (wgrad.hip): 16-lane group (hh, gg): K rows 8hh + 4r2 + q, channels 16gg + 4pp of the wave's block.
  // X: filter row r0's plane; the wave's other filter rows are FROWB further each.
  const int g16 = lane >> 4, hh = g16 >> 1, gg = g16 & 1, qq = (lane & 15) >> 2, pp = lane & 3;
  int a_tr[2][2], b_tr[2][3][2];                           // [MFMA K-step][..][r2]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      const int k = 16 * ks + 8 * hh + 4 * r2 + qq;
      a_tr[ks][r2] = G::a_off(k, cb * 32 + 16 * gg + 4 * pp);
#pragma unroll
      for (int d = 0; d < 3; ++d) b_tr[ks][d][r2] = B_BASE + G::x_off(r0, G::tap_row(k, d), ib * 32 + 16 * gg + 4 * pp);
    }
  auto tr_read = [&](int byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sm9 + byte_off));
  };
  auto frag = [&](int byte0, int byte1) {
    const s16x4 lo = tr_read(byte0), hi = tr_read(byte1);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8_t, v);
  };
  auto k_step = [&](int buf, int ks) {
    if constexpr (IN16) {
      typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
      const bf16x8_t a16 = __builtin_bit_cast(bf16x8_t, frag(buf * BUF + a_tr[ks][0], buf * BUF + a_tr[ks][1]));
#pragma unroll
      for (int rr = 0; rr < NR; ++rr)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const bf16x8_t b16 = __builtin_bit_cast(bf16x8_t, frag(buf * BUF + rr * G::FROWB + b_tr[ks][d][0], buf * BUF + rr * G::FROWB + b_tr[ks][d][1]));
          acc[rr * 3 + d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a16, b16, acc[rr * 3 + d], 0, 0, 0);
        }
      return;
    }
    f16x8_t af[2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) af[pl] = frag(buf * BUF + pl * A_PLANE + a_tr[ks][0], buf * BUF + pl * A_PLANE + a_tr[ks][1]);
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
      for (int d = 0; d < 3; ++d) {                         // fragments per tap: 8 live registers (smallest terms first: (l,h) (h,l) (h,h))
        f16x8_t bf[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
          bf[pl] = frag(buf * BUF + pl * B_PLANE + rr * G::FROWB + b_tr[ks][d][0], buf * BUF + pl * B_PLANE + rr * G::FROWB + b_tr[ks][d][1]);
        acc[rr * 3 + d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bf[0], acc[rr * 3 + d], 0, 0, 0);
        acc[rr * 3 + d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[1], acc[rr * 3 + d], 0, 0, 0);
        acc[rr * 3 + d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[0], acc[rr * 3 + d], 0, 0, 0);
      }
  };

  // One register set for the staged pieces: step it+1 is split and stored between the two MFMA K-steps of step it, and the loads
  // of step it+2 are issued right behind the stores — they fly under the second K-step, the barrier and the first K-step of
  // the next iteration (a second set cost 50 registers and the second workgroup per CU with them).
  f32x4 cur[NSLOT];
  if (iters > 0) {
    load_step(cur);
    store_step(0, cur);
    load_step(cur);                                          // step 1 (past the end of the split: dY masked to zero)
  }
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    const int b = it & 1;
    k_step(b, 0);
    store_step(b ^ 1, cur);                                  // step it + 1
    load_step(cur);                                          // step it + 2
    k_step(b, 1);
    __syncthreads();
  }

  const float dq = 1.f / (s_a * s_b);                        // powers of two: exact
  float* out = p.out + (size_t)split * CO * 9 * CI;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int co = cb * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
    const int ci = ib * 32 + (lane & 31);
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
      for (int d = 0; d < 3; ++d) out[(size_t)co * 9 * CI + (3 * (r0 + rr) + d) * CI + ci] = acc[rr * 3 + d][q] * dq;
  }
}

int g_w9 = 2;             // dcn_set_tuning("9tap", 0): these layers back on the kernels of wgrad.hip / wgrad3.hip; 1: 32 -> 64 only;
                          // 2: + 64 -> 128 at stride 1; 3: + 64 -> 128 at stride 2 (54 spilled registers)
int g_w9_target = 512;    // dcn_set_tuning("9target", n): workgroups (= split-K slabs) per launch

struct Plan9 { int splits, kchunk, Mp; };
Plan9 plan9(int n, int ho, int wo) {
  Plan9 pl;
  pl.Mp = n * ho * (wo + 1);
  int splits = g_w9_target;
  const int max_splits = pl.Mp / 256 > 0 ? pl.Mp / 256 : 1;
  if (splits > max_splits) splits = max_splits;
  pl.kchunk = cdiv(cdiv(pl.Mp, splits), KS) * KS;
  pl.splits = cdiv(pl.Mp, pl.kchunk);
  return pl;
}

template <int S, int CO, int CI> size_t lds9() { return (size_t)2 * G9<S, CO, CI>::BUF; }

template <int S, int CO, int CI, bool PRE = false, bool IN16 = false>
int launch9(const W9Params& p, int splits, hipStream_t stream) {
  static DcnPerDeviceFlag attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad9_kernel<S, CO, CI, PRE, IN16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds9<S, CO, CI>());
  }
  const int threads = G9<S, CO, CI>::NT;
  const size_t lds = lds9<S, CO, CI>();
  hipLaunchKernelGGL((wgrad9_kernel<S, CO, CI, PRE, IN16>), dim3(splits), dim3(threads), lds, stream, p);
  return DCN_OK;
}

}  // namespace

void wgrad9_set_tuning(int key, int value) { if (key == 0) g_w9 = value; else g_w9_target = value > 0 ? value : 512; }

// shape test only (the workspace is sized without knowing whether the abs-max words will be there): 32 -> 64 (stride 1 | 2) and
// 64 -> 128 (g_w9 = 2: stride 1 only; 1: neither — the half-empty 128 x 128 tiles of wgrad3.hip / wgrad.hip keep them)
bool wgrad9_shape_ok(int n, int h, int wd, int cin, int cout, int ksize, int stride) {
  if (!g_w9 || ksize != 3 || (stride != 1 && stride != 2)) return false;
  const bool small = cin == 32 && cout == 64, big = cin == 64 && cout == 128 && (g_w9 == 3 || (g_w9 == 2 && stride == 1));
  if (!small && !big) return false;
  if (h % stride || wd % stride || wd / stride < KS || h / stride < 2) return false;       // (one row wrap per K-step at most)
  const long long npix = (long long)n * h * wd, opix = (long long)n * (h / stride) * (wd / stride);
  if (npix * cin * 4 >= 0x7FFFFFF0LL || opix * cout * 4 >= 0x7FFFFFF0LL) return false;     // 32-bit byte offsets from the tensor bases (dense)
  if ((long long)n * (h / stride) * (wd / stride + 1) * stride >= 0x7FFFFFF0LL || npix < 4096) return false;
  return true;
}
int64_t wgrad9_ws(int n, int h, int wd, int cin, int cout, int stride) {
  const Plan9 pl = plan9(n, h / stride, wd / stride);
  return pl.splits > 1 ? (int64_t)pl.splits * cout * 9 * cin : 0;
}

int wgrad9_launch(const float* x, int ldx, const float* dy, int lddy, float* dw, float* ws, uint32_t* /*counters: one tile, 512 splits — a second launch sums them (slabsum.h)*/, int n, int h, int wd, int cin, int cout, int stride,
                  const uint32_t* amax_x, const uint32_t* amax_dy, const DcnPreAct* pre, hipStream_t stream) {
  const int ho = h / stride, wo = wd / stride;
  const Plan9 pl = plan9(n, ho, wo);
  DCN_CHECK_ARG(pl.splits == 1 || ws, "conv2d_bwd_weight: workspace required (%d splits)", pl.splits);
  DCN_CHECK_ARG((long long)n * ho * wo * lddy * 4 < 0x7FFFFFF0LL && (long long)n * h * wd * ldx * 4 < 0x7FFFFFF0LL,
                "conv2d_bwd_weight: a sliced operand of %lld bytes exceeds the 32-bit byte offsets of the nine-tap kernel",
                (long long)n * h * wd * ldx * 4);
  W9Params p{};
  p.x = x; p.dy = dy; p.out = pl.splits > 1 ? ws : dw;
  p.N = n; p.H = h; p.W = wd; p.Ho = ho; p.Wo = wo; p.ldx = ldx; p.lddy = lddy;
  p.Mp = pl.Mp; p.kchunk = pl.kchunk; p.splits = pl.splits;
  p.amax_dy = amax_dy; p.amax_x = amax_x;
  if (pre) {
    DCN_CHECK_ARG(cin == 32 && pre->scale && pre->shift && (((uintptr_t)pre->scale | (uintptr_t)pre->shift) & 15) == 0,
                  "conv2d_bwd_weight: the loader-side activation exists for the 32-channel form only");
    p.pre_scale = pre->scale; p.pre_shift = pre->shift; p.pre_act = pre->act; p.pre_slope = pre->slope;
  }
  // HBM-priced: dY and X once, the slabs written
  const double bytes = 4.0 * ((double)n * ho * wo * cout + (double)n * h * wd * cin + (double)pl.splits * cout * 9 * cin);
  const int pid = prof_begin(36, 2.0 * (double)n * ho * wo * cout * 9.0 * cin, stream, bytes);
  if (cin == 32 && pre) { if (stride == 1) launch9<1, 64, 32, true>(p, pl.splits, stream); else launch9<2, 64, 32, true>(p, pl.splits, stream); }
  else if (cin == 32) { if (stride == 1) launch9<1, 64, 32>(p, pl.splits, stream); else launch9<2, 64, 32>(p, pl.splits, stream); }
  else { if (stride == 1) launch9<1, 128, 64>(p, pl.splits, stream); else launch9<2, 128, 64>(p, pl.splits, stream); }
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("wgrad9");
  if (pl.splits > 1) return wgrad_reduce_slabs(ws, dw, (int64_t)cout * 9 * cin / 4, pl.splits, stream);
  return DCN_OK;
}

// bf16 storage: the same two layer forms on bf16 tensors (x, dy bf16 NHWC; dw fp32 [Cout][3][3][Cin]); the per-tap tile of wgrad.hip reads
// dY and X nine times for them (32 -> 64 @208: 0.85 ms against 0.11 ms of HBM traffic).
int wgrad9_launch_b16(const void* x, int ldx, const void* dy, int lddy, float* dw, float* ws, uint32_t* /*counters*/, int n, int h, int wd, int cin, int cout, int stride,
                      hipStream_t stream) {
  const int ho = h / stride, wo = wd / stride;
  const Plan9 pl = plan9(n, ho, wo);
  DCN_CHECK_ARG(pl.splits == 1 || ws, "conv2d_bwd_weight_b16: workspace required (%d splits)", pl.splits);
  DCN_CHECK_ARG((long long)n * ho * wo * lddy * 2 < 0x7FFFFFF0LL && (long long)n * h * wd * ldx * 2 < 0x7FFFFFF0LL,
                "conv2d_bwd_weight_b16: a sliced operand exceeds the 32-bit byte offsets of the nine-tap kernel");
  W9Params p{};
  p.x = (const float*)x; p.dy = (const float*)dy; p.out = pl.splits > 1 ? ws : dw;
  p.N = n; p.H = h; p.W = wd; p.Ho = ho; p.Wo = wo; p.ldx = ldx; p.lddy = lddy;
  p.Mp = pl.Mp; p.kchunk = pl.kchunk; p.splits = pl.splits;
  const double bytes = 2.0 * ((double)n * ho * wo * cout + (double)n * h * wd * cin) + 4.0 * (double)pl.splits * cout * 9 * cin;
  const int pid = prof_begin(36, 2.0 * (double)n * ho * wo * cout * 9.0 * cin, stream, bytes);
  if (cin == 32) { if (stride == 1) launch9<1, 64, 32, false, true>(p, pl.splits, stream); else launch9<2, 64, 32, false, true>(p, pl.splits, stream); }
  else { if (stride == 1) launch9<1, 128, 64, false, true>(p, pl.splits, stream); else launch9<2, 128, 64, false, true>(p, pl.splits, stream); }
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("wgrad9 (bf16)");
  if (pl.splits > 1) return wgrad_reduce_slabs(ws, dw, (int64_t)cout * 9 * cin / 4, pl.splits, stream);
  return DCN_OK;
}
