// Register-bank kernels for the 3x3 layers of the 416x416 / 208x208 maps (yolov3.cfg blocks 2-6: 32 -> 64 stride 2, the 32 -> 64 layer
// of the first residual block, 64 -> 128 stride 2), f16 two-piece split.  Their tensors are the largest of the network (1.4 GB at
// 416 x 416 x 32 channels, N = 64) and their filter banks the smallest (73-295 KB), so the filter bank lives in REGISTERS and one
// persistent workgroup per CU (four waves, one per SIMD, up to 512 registers each) streams the activations past it:
//
//   dgrad2_kernel<CK, CN>        data gradient of a stride-2 layer: dY (CK = 64 | 128 channels) -> dX (CN = 32 | 64 channels)
//   nconv1_kernel<S, CK, CN, F>  forward 32 -> 64 at stride S = 1 | 2 (+ BatchNorm partial sums), data gradient 64 -> 32 at stride 1
//
// dgrad2:  dX[2r+a][2c+b][ci] = sum over the taps (ky, kx) of parity class (a, b), over co:  dY[r + (ky==0)][c + (kx==0)][co] * W[co][ky][kx][ci]
//          class (a, b): ky = 1 for a = 0, ky in {0, 2} for a = 1 (kx likewise): 1 / 2 / 2 / 4 taps
// igemm.hip runs the four classes as implicit GEMMs on 256 x 32 tiles: K loops of 4-16 steps, dY gathered from L2 once per tap, the
// 1-tap class on the fp32 pipe — 1.8 ms (32 <- 64) and 1.0 ms (64 <- 128) against 0.4 / 0.2 ms of HBM traffic.  Here a workgroup walks
// a contiguous range of dY positions in chunks of 64 (CN = 32) or 32 (CN = 64); a wave owns 32 positions x 32 output channels and the
// taps of two classes (role 0: the 4-tap class and the 1-tap class, role 1: the two 2-tap classes — 5 / 4 taps x CK/16 MFMA
// triples per chunk), whose B fragments it split once (160 | 320 registers).  Per chunk the workgroup stages two row strips of dY
// (rows r and r+1, chunk + 1 entries x CK channels, split into f16 pieces on the way to LDS, double buffered); every A fragment is
// one ds_read_b128 at (strip, entry + 0|1).  Positions are PADDED as in wgrad3.hip (rows of Wo + 1 entries, the last one a pad that
// is staged as zero), so the right neighbour of the last column and the row below the last row read zeros without masks in the
// loop.  All four classes of a chunk are stored by the workgroup that computed them: full 128-byte pixels.  Buffer descriptors start
// at the workgroup's first row, so byte offsets are 32-bit whatever the tensor size.  Optional BatchNorm tap: the partial sums of the
// backward of the layer in front, formed on dX in the epilogue.  Roofline: HBM.  (nconv1: see its own comment below.)
#include "igemm.h"
#include "prof.h"
#include <type_traits>

#ifndef NCONV_ABL
#define NCONV_ABL 0      // timing-only ablations of nconv1_kernel (wrong results): 1 no output stores, 2 no MFMAs, 3 no staging loads
#endif

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned OOBN = 0x80000000u;

struct D2Params {
  const float* dy; const float* wt; float* dx;      // wt: transposed bank [CN ci][9 taps][CK co] fp32
  int N, Ho, Wo, lddy, ldo, accumulate;
  int Mp, nchunks, per_wg;                          // padded positions N*Ho*(Wo+1); chunks of 64; chunks per workgroup
  const unsigned* amax_dy; const unsigned* amax_w;
  // optional tap: dX is the gradient w.r.t. act(bn(y_prev)) — the output of the layer in front; its BatchNorm backward starts with
  // sum(g) and sum(g * xhat) per channel, g = dX * act'(.), which this kernel can form while it still holds dX in registers
  const float* tap_y; const float* tap_mean; const float* tap_invstd; const float* tap_gamma; const float* tap_beta;
  int tap_act; float tap_slope; float* tap_stats;   // tap_stats: [gridDim.x][2][CN]
};

__device__ __forceinline__ f32x4 ldn(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
// bytes a buffer descriptor may span (its range check returns zero / drops the store beyond)
__device__ __forceinline__ unsigned span32(long long bytes) { return bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0u : (unsigned)(bytes < 0 ? 0 : bytes); }
__device__ __forceinline__ float pow2n(unsigned amax_bits) {
  const int be = (int)((amax_bits >> 23) & 0xFF);
  if (be == 0 || be == 255) return 1.f;
  int e = 14 - (be - 126);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return __uint_as_float((unsigned)(e + 127) << 23);
}

// LDS geometry for CK dY channels and CN dX channels
template <int CK, int CN> struct Geo {
  static constexpr int MBS = CN == 32 ? 2 : 1;         // 32-position blocks per chunk (the four waves: MBS x 2 roles x CN/32 channel blocks)
  static constexpr int CH = 32 * MBS;                  // positions per chunk
  static constexpr int NE = CH + 1;                    // entries per strip (the right neighbour of the last position)
  static constexpr int ROW = 2 * CK;                   // bytes of an entry in one f16 plane
  static constexpr int STRIP = NE * ROW;
  static constexpr int PLANE = 2 * STRIP;              // both strips of one f16 piece
  static constexpr int BUFB = 2 * PLANE;               // high + low pieces
  static constexpr int PIECES = 2 * NE * (CK / 4);     // 16-byte pieces staged per chunk
  static constexpr int NSLOT = (PIECES + 255) / 256;
  // the 16-byte chunk index of an entry is XORed with a function of the entry so that the 16 lanes of a ds_read_b128 pass
  // (consecutive entries, one chunk) land in 16 different 16-byte bank groups: 128-byte rows: bank group = 8 (e & 1) + chunk,
  // 256-byte rows: bank group = chunk
  static __device__ __forceinline__ int off(int strip, int e, int co) {
    const int x = CK == 64 ? ((e >> 1) & 7) : (e & 15);
    return strip * STRIP + ROW * e + 16 * ((co >> 3) ^ x) + 2 * (co & 7);
  }
};

// taps of a role: (ky, kx, accumulator).  role 0: class (1,1) <- (0,0) (0,2) (2,0) (2,2), class (0,0) <- (1,1);
// role 1: class (0,1) <- (1,0) (1,2), class (1,0) <- (0,1) (2,1)
template <int ROLE> struct Taps;
template <> struct Taps<0> {
  static constexpr int N = 5;
  static constexpr int ky[5] = {0, 0, 2, 2, 1}, kx[5] = {0, 2, 0, 2, 1}, ac[5] = {0, 0, 0, 0, 1};
  static constexpr int ca[2] = {1, 0}, cb[2] = {1, 0};        // parity class (a, b) of each accumulator
};
template <> struct Taps<1> {
  static constexpr int N = 4;
  static constexpr int ky[4] = {1, 1, 0, 2}, kx[4] = {0, 2, 1, 1}, ac[4] = {0, 0, 1, 1};
  static constexpr int ca[2] = {0, 1}, cb[2] = {1, 0};
};

// scalar walker over padded positions: (row = n*Ho + r, column in [0, Wp), r)
struct Walk { int row, col, r; };
__device__ __forceinline__ void walk_to(Walk& w, int q, int Wp, int Ho) { w.row = q / Wp; w.col = q - w.row * Wp; w.r = w.row % Ho; }
__device__ __forceinline__ void walk_step(Walk& w, int step, int Wp, int Ho) {      // + step (< Wp: one wrap at most)
  w.col += step;
  if (w.col >= Wp) { w.col -= Wp; ++w.row; if (++w.r == Ho) w.r = 0; }
}

// IN16 (bf16 storage): dY, the transposed bank and dX are bf16 — 8-byte pieces staged as they are (one plane), the bank's 16-byte runs
// are MFMA operands, one v_mfma_f32_32x32x16_bf16 per product, dX rounded to bf16 where it is stored (no tap in this form).
template <int ROLE, int CK, int CN, bool TAP, bool IN16 = false>
__device__ __forceinline__ void dgrad2_body(const D2Params& p, unsigned char* sm) {
  static_assert(!(TAP && IN16), "no BatchNorm tap in the bf16 form");
  constexpr int ESZ = IN16 ? 2 : 4;
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  typedef Taps<ROLE> T;
  typedef Geo<CK, CN> G;
  constexpr int NT = T::N, KS = CK / 16, CH = G::CH, NE = G::NE, PLANE = G::PLANE, BUFB = G::BUFB, NSLOT = G::NSLOT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int mb = CN == 32 ? (wave & 1) : 0, nb = CN == 32 ? 0 : (wave & 1);
  const int m = lane & 31, kg = lane >> 5;
  const int Wp = p.Wo + 1, NR = p.N * p.Ho, W = 2 * p.Wo;
  float s_a = 1.f, s_b = 1.f;
  if constexpr (!IN16) { s_a = pow2n(amax_read(p.amax_dy)); s_b = pow2n(amax_read(p.amax_w)); }
  // Buffer descriptors start at the first dY row of THIS workgroup's range (row0): byte offsets stay 32-bit whatever the tensor size
  const int c_begin = blockIdx.x * p.per_wg, c_end = min(p.nchunks, c_begin + p.per_wg);
  const int row0 = (c_begin * CH) / Wp;
  const long long a_skip = (long long)row0 * p.Wo * p.lddy * ESZ, a_all = (((long long)NR * p.Wo - 1) * p.lddy + CK) * ESZ;
  const long long o_skip = (long long)row0 * 4 * p.Wo * p.ldo * ESZ, o_all = (((long long)NR * 4 * p.Wo - 1) * p.ldo + CN) * ESZ;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.dy + a_skip), 0, span32(a_all - a_skip), 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.dx + o_skip), 0, span32(o_all - o_skip), 0x00020000);

  // ---- the wave's filter fragments, split once: B[k = co][n = ci = lane % 32], 8 consecutive co per lane ------------------
  f16x8_t bh[NT][KS], bl[IN16 ? 1 : NT][IN16 ? 1 : KS];
  auto load_b = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      if constexpr (IN16) {
        const unsigned short* src16 = reinterpret_cast<const unsigned short*>(p.wt) + ((size_t)((nb * 32 + m) * 9 + T::ky[i] * 3 + T::kx[i]) * CK + kk * 16 + 8 * kg);
        bh[i][kk] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const f32x4*>(src16));
        continue;
      }
      const float* src = p.wt + ((size_t)((nb * 32 + m) * 9 + T::ky[i] * 3 + T::kx[i]) * CK + kk * 16 + 8 * kg);
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(src) * s_b, v1 = *reinterpret_cast<const f32x4*>(src + 4) * s_b;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bh[i][kk][e] = (_Float16)v0[e]; bh[i][kk][4 + e] = (_Float16)v1[e];
        if constexpr (!IN16) { bl[i][kk][e] = (_Float16)(v0[e] - (float)bh[i][kk][e]); bl[i][kk][4 + e] = (_Float16)(v1[e] - (float)bh[i][kk][4 + e]); }
      }
    }
  };
  load_b(std::integral_constant<int, 0>{}); load_b(std::integral_constant<int, 1>{}); load_b(std::integral_constant<int, 2>{});
  load_b(std::integral_constant<int, 3>{});
  if constexpr (NT > 4) load_b(std::integral_constant<int, 4>{});

  // ---- staging list of a chunk: 2 strips x NE entries x CK/4 pieces of 4 channels (2080 | 2112): 8 per thread + a rest for wave 0
  static_assert(G::PIECES > 8 * 256 && G::PIECES <= 8 * 256 + 64, "the ninth slot is one wave");
  const bool last_on = __builtin_amdgcn_readfirstlane(wave) == 0;           // slot 8: lanes past the list repeat its last piece
  // per piece: meta = strip | entry << 1, its LDS offset, and the chunk-independent part of its global byte offset: the chunk's
  // first entry sits at dY pixel (row + strip)*Wo + col + e, minus one when the entry wrapped into the next padded row
  int meta[NSLOT], st_off[NSLOT], k_off[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    int idx = j * 256 + tid;
    if (idx > G::PIECES - 1) idx = G::PIECES - 1;
    const int strip = idx / (NE * (CK / 4)), rem = idx - strip * (NE * (CK / 4));
    const int e = rem / (CK / 4), co = (rem % (CK / 4)) * 4;
    meta[j] = strip | (e << 1);
    st_off[j] = G::off(strip, e, co);
    k_off[j] = ((strip * p.Wo + e) * p.lddy + co) * ESZ;
  }

  Walk wl, wc;                                       // chunk being LOADED / being computed; rows count from row0
  walk_to(wl, c_begin * CH, Wp, p.Ho); wl.row -= row0; wc = wl;

  auto load_chunk = [&](f32x4* v) {
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      if (j == NSLOT - 1 && !last_on) continue;
      const int strip = meta[j] & 1, e = meta[j] >> 1;
      const int t = wl.col + e;                                       // == Wo: the pad entry; beyond: the next padded row
      const int r = t > p.Wo ? (wl.r + 1 == p.Ho ? 0 : wl.r + 1) : wl.r;
      const int base = (wl.row * p.Wo + wl.col) * p.lddy * ESZ;         // (scalar)
      unsigned off = (unsigned)(base + k_off[j] - (t > p.Wo ? p.lddy * ESZ : 0));
      if (t == p.Wo || (strip == 1 && r + 1 >= p.Ho)) off = OOBN;     // (rows past the tensor: out of the descriptor's range, read as zero)
      if constexpr (IN16) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 w2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(a_rs, off, 0, 0));
        v[j] = f32x4{__uint_as_float(w2[0]), __uint_as_float(w2[1]), 0.f, 0.f};
      } else {
        v[j] = ldn(a_rs, off);
      }
    }
    walk_step(wl, CH, Wp, p.Ho);
  };
  auto store_chunk = [&](int buf, const f32x4* v) {
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      if (j == NSLOT - 1 && !last_on) continue;
      if constexpr (IN16) {
        *reinterpret_cast<uint2*>(sm + buf * BUFB + st_off[j]) = uint2{__float_as_uint(v[j][0]), __float_as_uint(v[j][1])};
        continue;
      }
      const f32x4 t = v[j] * s_a;
      const f16x4_t h = {(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
      const f16x4_t l = {(_Float16)(t[0] - (float)h[0]), (_Float16)(t[1] - (float)h[1]), (_Float16)(t[2] - (float)h[2]),
                         (_Float16)(t[3] - (float)h[3])};
      unsigned char* dst = sm + buf * BUFB + st_off[j];
      *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, h);
      *reinterpret_cast<uint2*>(dst + PLANE) = __builtin_bit_cast(uint2, l);
    }
  };

  // A fragment of (strip dyo, entry mb*32 + m + dxo, K-step kk): 8 channels from kk*16 + 8*kg.  The swizzle only depends on the
  // entry, so a K-step moves the 16-byte chunk index by two: chunk (2 kk + kg) ^ x
  int a_base[2][2], a_x[2];                          // [dyo][dxo]: row base; [dxo]: the entry's XOR term
#pragma unroll
  for (int dxo = 0; dxo < 2; ++dxo) {
    const int e = mb * 32 + m + dxo;
    a_x[dxo] = CK == 64 ? ((e >> 1) & 7) : (e & 15);
#pragma unroll
    for (int dyo = 0; dyo < 2; ++dyo) a_base[dyo][dxo] = dyo * G::STRIP + G::ROW * e;
  }

  const float dq = 1.f / (s_a * s_b);                // powers of two: exact
  constexpr bool tapped = TAP;                        // (a build of its own: the constants and sums cost registers the 128-channel form does not have)
  float t_mu = 0.f, t_is = 1.f, t_ga = 1.f, t_be = 0.f, t_s = 0.f, t_ss = 0.f;      // this lane's channel nb*32 + m of the layer in front
  if constexpr (tapped) {
    t_mu = p.tap_mean[nb * 32 + m]; t_is = p.tap_invstd[nb * 32 + m];
    if (p.tap_gamma) t_ga = p.tap_gamma[nb * 32 + m];
    if (p.tap_beta) t_be = p.tap_beta[nb * 32 + m];
  }
  const long long y_skip = (long long)row0 * 4 * p.Wo * CN * 4;
  const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(tapped ? p.tap_y : p.dx) + (tapped ? y_skip : 0)), 0,
      span32((long long)NR * 4 * p.Wo * CN * 4 - y_skip), 0x00020000);
  f32x4 stage[NSLOT];
  if (c_begin < c_end) {
    load_chunk(stage);
    store_chunk(0, stage);
    load_chunk(stage);                                // chunk c_begin + 1 (past the range: unused)
  }
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
    auto tap = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int dyo = T::ky[i] == 0 ? 1 : 0, dxo = T::kx[i] == 0 ? 1 : 0;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const int ao = a_base[dyo][dxo] + 16 * ((2 * kk + kg) ^ a_x[dxo]);
        const f16x8_t ah = *reinterpret_cast<const f16x8_t*>(sm + buf * BUFB + ao);
        if constexpr (IN16) {
          acc[T::ac[i]] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bh[i][kk]), acc[T::ac[i]], 0, 0, 0);
        } else {
          const f16x8_t al = *reinterpret_cast<const f16x8_t*>(sm + buf * BUFB + PLANE + ao);
          acc[T::ac[i]] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[i][kk], acc[T::ac[i]], 0, 0, 0);      // smallest terms first
          acc[T::ac[i]] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[i][kk], acc[T::ac[i]], 0, 0, 0);
          acc[T::ac[i]] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[i][kk], acc[T::ac[i]], 0, 0, 0);
          // (320 filter registers at CK = 128: keep the scheduler from reading all eight K-steps' fragments ahead of the MFMAs)
          if constexpr (KS > 4) { if (kk & 1) __builtin_amdgcn_sched_barrier(0); }
        }
      }
    };
    const int q0 = c * CH;
    auto pix_of = [&](int q, int a, bool& ok) {
      const int pos = mb * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
      const int t = wc.col + pos;
      ok = t != p.Wo && q0 + pos < p.Mp;
      return 2 * wc.row * W + 2 * t + (t > p.Wo ? 2 * W - 2 * Wp : 0) + T::ca[a] * W + T::cb[a];
    };
    float ty[tapped ? 2 : 1][tapped ? 16 : 1];
    if constexpr (tapped) {                         // all 32 loads of y issued in front of the chunk's MFMAs (one by one, behind each
#pragma unroll                                      // store, every load's latency was exposed: the tap cost more than the pass it replaces)
      for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          bool ok; const int pix = pix_of(q, a, ok);
          ty[a][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(y_rs, ok ? (pix * CN + nb * 32 + m) * 4 : (int)OOBN, 0, 0));
        }
    }
    tap(std::integral_constant<int, 0>{}); tap(std::integral_constant<int, 1>{});
    store_chunk(buf ^ 1, stage);                      // chunk c + 1
    load_chunk(stage);                                // chunk c + 2
    tap(std::integral_constant<int, 2>{}); tap(std::integral_constant<int, 3>{});
    if constexpr (NT > 4) tap(std::integral_constant<int, 4>{});

    // ---- store: D[m = position][n = ci]; register q <-> position mb*32 + (q & 3) + 8 (q >> 2) + 4 kg ---------------------------
    // dX pixel of position pos, class (a, b): (2 row + a)*W + 2 (col + pos) + b; behind the pad entry the position sits in the next
    // padded row: +2W - 2 Wp = -2 pixels ... +2W
#pragma unroll
    for (int q = 0; q < 16; ++q) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {               // (32-bit byte offsets through a buffer descriptor: one address register per store)
        bool ok; const int pix = pix_of(q, a, ok);
        if (!ok) continue;
        const int off = (pix * p.ldo + nb * 32 + m) * ESZ;
        float v = acc[a][q] * dq;
        if constexpr (IN16) {
          if (p.accumulate) v += __uint_as_float((unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(o_rs, off, 0, 0) << 16);
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (__bf16)v), o_rs, off, 0, 0);
          continue;
        }
        if (p.accumulate) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(o_rs, off, 0, 0));
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), o_rs, off, 0, 0);
        if constexpr (tapped) {                   // the terms of channel_partials_kernel<1> (bn.hip), on the value just stored
          const float xh = (ty[a][q] - t_mu) * t_is;
          const float g = (p.tap_act == DCN_ACT_LEAKY && t_ga * xh + t_be <= 0.f) ? v * p.tap_slope : v;
          t_s += g; t_ss += g * xh;
        }
      }
    }
    walk_step(wc, CH, Wp, p.Ho);
  }
  if constexpr (tapped) {                             // one partial row per workgroup: [2][CN]
    t_s += __shfl_xor(t_s, 32); t_ss += __shfl_xor(t_ss, 32);
    __syncthreads();
    float* red = reinterpret_cast<float*>(sm);       // [wave][2][32]
    if (kg == 0) { red[(wave * 2 + 0) * 32 + m] = t_s; red[(wave * 2 + 1) * 32 + m] = t_ss; }
    __syncthreads();
    if (tid < 2 * CN) {
      const int which = tid / CN, ch = tid - which * CN;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (CN == 32 || (w & 1) == (ch >> 5)) t += red[(w * 2 + which) * 32 + (ch & 31)];      // (CN = 64: wave & 1 is its channel block)
      p.tap_stats[((size_t)blockIdx.x * 2 + which) * CN + ch] = t;
    }
  }
}

template <int CK, int CN, bool TAP, bool IN16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void dgrad2_kernel(const D2Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smn[];       // [2 buffers][high | low][strip r | strip r+1]
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 7) == 0) dgrad2_body<0, CK, CN, TAP, IN16>(p, smn);
  else dgrad2_body<1, CK, CN, TAP, IN16>(p, smn);
}

// =====================================================================================================================
// 3x3 layers between 32 and 64 channels with the filter bank in registers: the 32 -> 64 layer of the first residual block on the
// 208x208 map (S = 1, forward, and its data gradient 64 -> 32) and the forward of the 32 -> 64 stride-2 layer on the 416x416 map
// (S = 2).  igemm.hip runs them on 256 x 64 / 256 x 32 tiles that gather every tap from L2: 0.75-0.93 ms against 0.2-0.4 ms of
// HBM traffic.  Same scheme as dgrad2_kernel: one persistent workgroup per CU walks padded OUTPUT positions (rows of Wo + 1 entries)
// in chunks of 64 and stages, per chunk, the three input rows the taps touch as strips of X_pad[S*q + kx] entries (rows of
// S*(Wo+1) entries whose entry u is image column u - 1: the left / right borders read zeros without masks; the top / bottom rows
// are zeroed when staged; S = 2: even and odd entries in separate planes, so that every tap reads consecutive rows).  The four
// waves are (position block, channel block): CN = 64: wave (mb, nb) owns 32 positions x 32 filters and all of K = 32;
// CN = 32 (the data gradient): wave (mb, kh) owns 32 positions x 32 channels and HALF of K = 64 — the two halves meet in LDS
// before the store.  Either way a wave keeps 9 taps x 2 K-steps of split filter fragments (144 registers).  The forward also
// produces the BatchNorm partial sums: one row per workgroup (accumulated over its chunks), the caller's remaining rows are zeroed.
struct N1Params {
  const float* x; const float* w; float* y; float* stats;   // w: [CN][9][CK] fp32 (OHWI bank, or the transposed bank of the data gradient)
  int N, Ho, Wo, ldi, ldo;                                    // Ho x Wo: OUTPUT grid; input (S*Ho) x (S*Wo)
  int Mp, nchunks, per_wg;
  const unsigned* amax_x; const unsigned* amax_w;
  // PRE: the input is the RAW output of the layer in front; its BatchNorm scale / shift and activation are applied when a piece is
  // staged (pad pieces stay zero) — that layer's scale_act pass and its activation tensor do not exist.  amax_x: of the activation.
  const float* pre_scale; const float* pre_shift; int pre_act; float pre_slope;
};

template <int S, int CK> struct Geo1 {
  static constexpr int CH = 64;
  static constexpr int NE = S * (CH - 1) + 3;            // entries per strip: 66 | 129
  static constexpr int NEV = S == 1 ? NE : CH + 1;       // S = 2: rows of the even plane (65), then CH rows of the odd plane
  static constexpr int ROW = 2 * CK;                     // bytes of an entry in one f16 plane: 64 | 128
  static constexpr int STRIP = NE * ROW;
  static constexpr int PLANE = 3 * STRIP;                // three input rows
  static constexpr int BUFB = 2 * PLANE;                 // high + low pieces
  static constexpr int PIECES = 3 * NE * (CK / 4);
  static constexpr int NSLOT = (PIECES + 255) / 256;
  static __device__ __forceinline__ int row_of(int e) { return S == 1 ? e : ((e & 1) ? NEV + (e >> 1) : (e >> 1)); }    // LDS row of entry e
  // 16-byte chunk swizzle: 64-byte rows: chunk ^ (row / 4) % 4; 128-byte rows: chunk ^ (row / 2) % 8 (16 consecutive rows of one
  // chunk -> 16 different 16-byte bank groups)
  static __device__ __forceinline__ int off(int strip, int row, int c) {
    const int x = CK == 32 ? ((row >> 2) & 3) : ((row >> 1) & 7);
    return strip * STRIP + ROW * row + 16 * ((c >> 3) ^ x) + 2 * (c & 7);
  }
};

// Eight waves: waves 0-3 compute (MFMAs + epilogue, the filter registers), waves 4-7 stage (global loads, split, LDS stores) — two
// waves per SIMD, one of each kind, so the staging's vector work and memory latency run under the other wave's MFMAs.  (With four
// waves doing both, timing ablations showed the three parts adding up: 0.57 ms = 0.38 without MFMAs + 0.19 of MFMAs.)
// IN16 (bf16 storage): x, the bank [CN][9][CK] and y are bf16 — 8-byte pieces staged as loaded (one plane), the bank's 16-byte runs are
// MFMA operands, one v_mfma_f32_32x32x16_bf16 per product, y rounded to bf16 where it is stored and the BatchNorm partial sums taken of
// the ROUNDED values (the statistics of the stored tensor, as in conv1.hip's bf16 kernel).
template <int S, int CK, int CN, bool FLIP, bool PRE = false, bool IN16 = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void nconv1_kernel(const N1Params p) {
  static_assert(!(PRE && IN16), "the loader-side activation exists for fp32 tensors only");
  constexpr int ESZ = IN16 ? 2 : 4;
  typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
  typedef Geo1<S, CK> G;
  constexpr int CH = G::CH, NE = G::NE, PLANE = G::PLANE, BUFB = G::BUFB, NSLOT = G::NSLOT;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];       // [2 buffers][high | low][3 strips]  (+ CN = 32: the K-halves' exchange)
  const int lane = threadIdx.x & 63, wave8 = threadIdx.x >> 6;
  const bool loader = __builtin_amdgcn_readfirstlane(wave8) >= 4;
  const int tid = threadIdx.x & 255, wave = wave8 & 3;   // index within the role
  const int mb = wave & 1, hb = wave >> 1;               // hb: channel block (CN = 64) or K half (CN = 32)
  const int nb = CN == 64 ? hb : 0, kofs = CN == 64 ? 0 : hb * 32;
  const int m = lane & 31, kg = lane >> 5;
  const int Wp = p.Wo + 1, RL = S * Wp, NR = p.N * p.Ho, W = S * p.Wo, H = S * p.Ho;
  float s_a = 1.f, s_b = 1.f;
  if constexpr (!IN16) { s_a = pow2n(amax_read(p.amax_x)); s_b = pow2n(amax_read(p.amax_w)); }
  // Buffer descriptors start at this workgroup's first output row (row0) and, for the input, one input row above S*row0 (the top
  // strip; `bias` is that row in bytes — zero for row0 = 0, where the row above does not exist and its offsets go negative = out of
  // range): byte offsets stay 32-bit whatever the tensor size
  const int c_begin = blockIdx.x * p.per_wg, c_end = min(p.nchunks, c_begin + p.per_wg);
  const int row0 = (c_begin * CH) / Wp;
  const int bias = row0 > 0 ? W * p.ldi * ESZ : 0;
  const long long a_skip = (long long)S * row0 * W * p.ldi * ESZ - bias, a_all = (((long long)p.N * H * W - 1) * p.ldi + CK) * ESZ;
  const long long o_skip = (long long)row0 * p.Wo * p.ldo * ESZ, o_all = (((long long)NR * p.Wo - 1) * p.ldo + CN) * ESZ;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x + a_skip), 0, span32(a_all - a_skip), 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.y + o_skip), 0, span32(o_all - o_skip), 0x00020000);

  // ---- the wave's filter fragments (tap t = 3 j + kx of strip j, entry offset kx; FLIP: the data gradient reads bank tap 8 - t) ----
  f16x8_t bh[9][2], bl[IN16 ? 1 : 9][IN16 ? 1 : 2];
  if (!loader)
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if constexpr (IN16) {
        const unsigned short* src16 = reinterpret_cast<const unsigned short*>(p.w) + ((size_t)((nb * 32 + m) * 9 + (FLIP ? 8 - t : t)) * CK + kofs + kk * 16 + 8 * kg);
        bh[t][kk] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const f32x4*>(src16));
        continue;
      }
      const float* src = p.w + ((size_t)((nb * 32 + m) * 9 + (FLIP ? 8 - t : t)) * CK + kofs + kk * 16 + 8 * kg);
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(src) * s_b, v1 = *reinterpret_cast<const f32x4*>(src + 4) * s_b;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bh[t][kk][e] = (_Float16)v0[e]; bh[t][kk][4 + e] = (_Float16)v1[e];
        if constexpr (!IN16) { bl[t][kk][e] = (_Float16)(v0[e] - (float)bh[t][kk][e]); bl[t][kk][4 + e] = (_Float16)(v1[e] - (float)bh[t][kk][4 + e]); }
      }
    }

  // ---- staging list of a chunk: 3 strips x NE entries x CK/4 pieces; the last slot is a partial one (whole waves) --------------
  constexpr int FULL = G::PIECES / 256, REST = G::PIECES - FULL * 256;            // REST pieces in slot FULL: waves 0 .. ceil(REST/64)-1
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bool last_on = REST > 0 && wave_u * 64 < REST;
  // per piece: meta = strip | entry << 2, its LDS offset, and the part of its global byte offset that does not depend on the chunk:
  // the chunk's first entry sits at input pixel (S*row + strip - 1)*W + S*col - 1 + e, plus S*W - RL when the entry wrapped into the
  // next padded row
  int meta[NSLOT], st_off[NSLOT], k_off[NSLOT];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    int idx = j * 256 + tid;
    if (idx > G::PIECES - 1) idx = G::PIECES - 1;                              // (lanes past the list repeat its last piece)
    const int strip = idx / (NE * (CK / 4)), rem = idx - strip * (NE * (CK / 4));
    const int e = rem / (CK / 4), c = (rem % (CK / 4)) * 4;
    meta[j] = strip | (e << 2);
    st_off[j] = G::off(strip, G::row_of(e), c);
    k_off[j] = ((((strip - 1) * W + e - 1) * p.ldi) + c) * ESZ;
  }
  // PRE: every piece of a thread holds the same four channels (256 and the piece counts per row are multiples of CK/4) — except
  // the repeats of the list's last piece, which are switched off instead (dead_last)
  static_assert(256 % (CK / 4) == 0, "one channel group per thread");
  const bool dead_last = PRE && (FULL * 256 + tid > G::PIECES - 1);
  f32x4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
  if constexpr (PRE) {
    const int c = (tid % (CK / 4)) * 4;
    psc = *reinterpret_cast<const f32x4*>(p.pre_scale + c) * s_a; psh = *reinterpret_cast<const f32x4*>(p.pre_shift + c) * s_a;
  }
  const bool leaky_max = p.pre_slope >= 0.f && p.pre_slope <= 1.f;
  const int wrap_delta = (S * W - RL) * p.ldi * ESZ;        // an output row further: S input rows on, one padded row of entries back

  Walk wl, wc;                                       // chunk being LOADED / being computed: (row n*Ho + oy - row0, column, oy)
  walk_to(wl, c_begin * CH, Wp, p.Ho); wl.row -= row0; wc = wl;

  // scalar state of the chunk being loaded (all derived from the walker): byte offset of its first entry, the entry index from which
  // entries wrap into the next padded row, and whether strip 0 / strip 2 lie outside the image before / after the wrap
  auto load_slot = [&](int j, f32x4* v, unsigned& vmask) {
    if (j >= FULL && !last_on) return;
    const int strip = meta[j] & 3, e = meta[j] >> 2;
    const int u0 = S * wl.col;
    const bool wrapped = e >= RL - u0;
    const int u = u0 + e - (wrapped ? RL : 0);                      // entry within its padded row: image column u - 1
    const int oy = wrapped ? (wl.r + 1 == p.Ho ? 0 : wl.r + 1) : wl.r;
    const bool row_out = (strip == 0 && oy == 0) || (S == 1 && strip == 2 && oy == p.Ho - 1);
    const int base = (S * wl.row * W + u0) * p.ldi * ESZ + bias;      // (scalar)
    unsigned off = (unsigned)(base + k_off[j] + (wrapped ? wrap_delta : 0));
    if ((unsigned)(u - 1) >= (unsigned)W || row_out || NCONV_ABL == 3) off = OOBN;    // (rows past the tensor: out of the descriptor's range, read as zero)
    if constexpr (IN16) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 w2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(a_rs, off, 0, 0));
      v[j] = f32x4{__uint_as_float(w2[0]), __uint_as_float(w2[1]), 0.f, 0.f};
    } else {
      v[j] = ldn(a_rs, off);
    }
    if constexpr (PRE) {                             // (a piece past the tensor's end is a pad as well: its rows do not exist)
      const bool in = off != OOBN && (S * (wl.row + row0 + (wrapped ? 1 : 0)) + strip - 1) < p.N * H;
      vmask = in ? (vmask | (1u << j)) : (vmask & ~(1u << j));
    }
  };
  auto store_slot = [&](int j, int buf, const f32x4* v, unsigned vmask) {
    if (j >= FULL && !last_on) return;
    if (PRE && j >= FULL && dead_last) return;
    if constexpr (IN16) {
      *reinterpret_cast<uint2*>(sm + buf * BUFB + st_off[j]) = uint2{__float_as_uint(v[j][0]), __float_as_uint(v[j][1])};
      return;
    }
    f32x4 t;
    if constexpr (PRE) {
      // scale_act_kernel's arithmetic (bn.hip) on operands that carry the power-of-two operand scale already (psc, psh = s_a * scale,
      // s_a * shift: exact, and LeakyReLU commutes with a positive factor), then zero for the pads.  max(z, slope z) IS z > 0 ? z : slope z
      // for 0 <= slope <= 1
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
      t = v[j] * s_a;
    }
    const f16x4_t h = {(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
    const f16x4_t l = {(_Float16)(t[0] - (float)h[0]), (_Float16)(t[1] - (float)h[1]), (_Float16)(t[2] - (float)h[2]),
                       (_Float16)(t[3] - (float)h[3])};
    unsigned char* dst = sm + buf * BUFB + st_off[j];
    *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, h);
    *reinterpret_cast<uint2*>(dst + PLANE) = __builtin_bit_cast(uint2, l);
  };
  auto load_chunk = [&](f32x4* v, unsigned& vmask) {
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) load_slot(j, v, vmask);
    walk_step(wl, CH, Wp, p.Ho);
  };
  auto store_chunk = [&](int buf, const f32x4* v, unsigned vmask) {
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) store_slot(j, buf, v, vmask);
  };

  // A fragment of (strip j, tap column kx, K-step kk): LDS row of entry S*(mb*32 + m) + kx, 8 channels from kofs + kk*16 + 8*kg
  int a_addr[3][2];                                  // [kx][kk], strip 0
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) a_addr[kx][kk] = G::off(0, G::row_of(S * (mb * 32 + m) + kx), kofs + kk * 16 + 8 * kg);

  const float dq = 1.f / (s_a * s_b);                // powers of two: exact
  float st_s = 0.f, st_ss = 0.f;                     // BatchNorm partial sums of this lane's filter over the workgroup's positions
  // ---- the staging waves: a loop of their own (so that their registers are not live beside the filter fragments) -----------------
  if (loader) {
    // two register sets: the loads of chunk c + 3 are issued when chunk c + 1 is stored, i.e. two chunks of pieces are in flight per
    // CU (with one set the four staging waves kept 28-52 KB in flight and the kernel ran at the latency of its loads)
    f32x4 st0[NSLOT], st1[NSLOT];
    unsigned vm0 = 0, vm1 = 0;                        // PRE: which pieces of a set are real pixels
    if (c_begin < c_end) {
      load_chunk(st0, vm0);
      store_chunk(0, st0, vm0);
      load_chunk(st1, vm1);                           // chunk c_begin + 1 (past the range: unused)
      load_chunk(st0, vm0);                           // chunk c_begin + 2
    }
    for (int c = c_begin; c < c_end; c += 2) {        // chunk c + 1 waits in st1, chunk c + 2 in st0
      __syncthreads();
      store_chunk(1, st1, vm1);
      load_chunk(st1, vm1);                           // chunk c + 3
      if constexpr (CN == 32) __syncthreads();        // (the compute waves' exchange)
      if (c + 1 >= c_end) break;
      __syncthreads();
      store_chunk(0, st0, vm0);
      load_chunk(st0, vm0);                           // chunk c + 4
      if constexpr (CN == 32) __syncthreads();
    }
    if (p.stats) { __syncthreads(); __syncthreads(); }
    return;
  }

  // ---- the compute waves ---------------------------------------------------------------------------------------------------------
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    __syncthreads();
    // Two accumulators (one per K-step of a tap): 54 MFMAs chained through ONE accumulator wait for each other's results
    f32x16 acc, acc1;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acc1[q] = 0.f; }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int j = t / 3, kx = t - 3 * j;
      {
        const int ao = buf * BUFB + j * G::STRIP + a_addr[kx][0];
        const f16x8_t ah = *reinterpret_cast<const f16x8_t*>(sm + ao);
        if constexpr (IN16) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bh[t][0]), acc, 0, 0, 0);
        } else {
        const f16x8_t al = *reinterpret_cast<const f16x8_t*>(sm + ao + PLANE);
        if (NCONV_ABL == 2) { acc[0] += (float)al[0] + (float)ah[0]; } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[t][0], acc, 0, 0, 0);      // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[t][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[t][0], acc, 0, 0, 0); }
        }
      }
      {
        const int ao = buf * BUFB + j * G::STRIP + a_addr[kx][1];
        const f16x8_t ah = *reinterpret_cast<const f16x8_t*>(sm + ao);
        if constexpr (IN16) {
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bh[t][1]), acc1, 0, 0, 0);
        } else {
        const f16x8_t al = *reinterpret_cast<const f16x8_t*>(sm + ao + PLANE);
        if (NCONV_ABL == 2) { acc1[0] += (float)al[0] + (float)ah[0]; } else {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[t][1], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[t][1], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[t][1], acc1, 0, 0, 0); }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] += acc1[q];

    if constexpr (CN == 32) {                         // the two K halves of a position block meet in LDS (behind the staging buffers)
      float* xch = reinterpret_cast<float*>(sm + 2 * BUFB) + mb * 16 * 64;
      if (hb == 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) xch[q * 64 + lane] = acc[q];
      }
      __syncthreads();
      if (hb == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += xch[q * 64 + lane];
      }
    }
    // ---- store: D[m = position][n = channel]; register q <-> position mb*32 + (q & 3) + 8 (q >> 2) + 4 kg -------------------
    if (CN == 64 || hb == 0) {
      // output pixel of position pos: row*Wo + col + pos, minus one behind the pad entry (col + pos == Wo: the pad itself, no pixel)
      const int q0 = c * CH;
      const int obase = ((wc.row * p.Wo + wc.col) * p.ldo + nb * 32 + m) * ESZ;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int pos = mb * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
        const int t = wc.col + pos;
        if (t == p.Wo || q0 + pos >= p.Mp) continue;
        float v = acc[q] * dq;
        if constexpr (IN16) {
          const __bf16 v16 = (__bf16)v;
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, v16), o_rs, obase + (pos - (t > p.Wo ? 1 : 0)) * p.ldo * ESZ, 0, 0);
          v = (float)v16;
        } else {
          if (NCONV_ABL != 1 || v == 123.f)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), o_rs, obase + (pos - (t > p.Wo ? 1 : 0)) * p.ldo * 4, 0, 0);
        }
        st_s += v; st_ss += v * v;
      }
    }
    walk_step(wc, CH, Wp, p.Ho);
  }
  if (p.stats) {                                      // one partial row per workgroup: [2][CN]
    st_s += __shfl_xor(st_s, 32); st_ss += __shfl_xor(st_ss, 32);
    __syncthreads();
    float* red = reinterpret_cast<float*>(sm);       // [mb][2][CN]
    if (kg == 0 && (CN == 64 || hb == 0)) { red[(mb * 2 + 0) * CN + nb * 32 + m] = st_s; red[(mb * 2 + 1) * CN + nb * 32 + m] = st_ss; }
    __syncthreads();
    if (tid < 2 * CN) p.stats[(size_t)blockIdx.x * 2 * CN + tid] = red[tid] + red[2 * CN + tid];
  }
}

int g_nconv = 1;          // dcn_set_tuning("Nconv", 0): these layers back on the implicit-GEMM tiles

}  // namespace

void nconv_set_tuning(int v) { g_nconv = v; }

// data gradients of the 32 -> 64 and 64 -> 128 3x3 stride-2 layers (dY 64 | 128 channels -> dX 32 | 64 channels), dense dX
bool dgrad2_applicable(int n, int h, int wd, int cin, int cout, int ksize, int stride, int accumulate) {
  if (!g_nconv || ksize != 3 || stride != 2 || !((cin == 32 && cout == 64) || (cin == 64 && cout == 128 && g_nconv != 2))) return false;
  (void)accumulate;
  if ((h & 1) || (wd & 1) || wd / 2 < 64 || h < 4) return false;                       // (one row wrap per chunk at most)
  if ((long long)n * (h / 2) * (wd / 2 + 1) >= 0x7FFFFFF0LL / 64 || (long long)n * h >= 0x7FFFFFF0LL / 8) return false;      // (position / row indices in 32 bits)
  return (long long)n * h * wd >= 65536;                                               // (a persistent grid wants work for every CU)
}

template <int CK, int CN, bool TAP, bool IN16 = false>
int launch_d2(D2Params& p, int grid, double flop, double bytes, hipStream_t stream) {
  typedef Geo<CK, CN> G;
  static DcnPerDeviceFlag attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad2_kernel<CK, CN, TAP, IN16>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G::BUFB);
  }
  const int pid = prof_begin(37, flop, stream, bytes);
  hipLaunchKernelGGL((dgrad2_kernel<CK, CN, TAP, IN16>), dim3(grid), dim3(256), 2 * G::BUFB, stream, p);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("dgrad2");
  return DCN_OK;
}

int dgrad2_grid(int n, int h, int wd, int cin) {
  const int g_ncus = dcn_device_cus();
  if (!g_ncus) return -1;
  const int nchunks = cdiv((long long)n * (h / 2) * (wd / 2 + 1), cin == 32 ? 64 : 32);
  const int grid = g_ncus < nchunks ? g_ncus : nchunks;
  return cdiv(nchunks, cdiv(nchunks, grid));
}

int dgrad2_launch(const float* dy, int lddy, const float* wt, float* dx, int n, int h, int wd, int cin, int accumulate,
                  const uint32_t* amax_dy, const uint32_t* amax_w, const DcnBnTap* tap, hipStream_t stream) {
  const int g_ncus = dcn_device_cus();
  if (!g_ncus) { dcn_set_error("dgrad2: device query failed"); return DCN_ERR_LAUNCH; }
  const int ch = cin == 32 ? 64 : 32, ck = 2 * cin;
  D2Params p{};
  p.dy = dy; p.wt = wt; p.dx = dx; p.N = n; p.Ho = h / 2; p.Wo = wd / 2; p.lddy = lddy; p.ldo = cin; p.accumulate = accumulate;
  p.Mp = n * p.Ho * (p.Wo + 1);
  p.nchunks = cdiv(p.Mp, ch);
  int grid = g_ncus < p.nchunks ? g_ncus : p.nchunks;
  p.per_wg = cdiv(p.nchunks, grid);
  grid = cdiv(p.nchunks, p.per_wg);
  DCN_CHECK_ARG(((long long)p.per_wg * ch / (p.Wo + 1) + 4) * 4 * p.Wo * (lddy > cin ? lddy : cin) * 4 < 0x7FFFFFF0LL,
                "conv2d_bwd_data: the rows of one workgroup exceed 32-bit byte offsets");
  p.amax_dy = amax_dy; p.amax_w = amax_w;
  if (tap) {
    DCN_CHECK_ARG(cin == 32, "conv2d_bwd_data: the BatchNorm tap exists for the 32-channel form only");
    DCN_CHECK_ARG(tap->y && tap->mean && tap->invstd && tap->stats && tap->stats_rows >= grid,
                  "conv2d_bwd_data: BatchNorm tap needs y, mean, invstd and %d statistics rows (%d given)", grid, tap->stats_rows);
    p.tap_y = tap->y; p.tap_mean = tap->mean; p.tap_invstd = tap->invstd; p.tap_gamma = tap->gamma; p.tap_beta = tap->beta;
    p.tap_act = tap->act; p.tap_slope = tap->slope; p.tap_stats = tap->stats;
  }
  const double bytes = 4.0 * ((double)n * p.Ho * p.Wo * ck + (double)n * h * wd * cin * ((accumulate ? 2 : 1) + (tap ? 1 : 0)) + (double)cin * 9 * ck);
  const double flop = 2.0 * (double)n * p.Ho * p.Wo * ck * 9.0 * cin;
  if (cin == 32) return tap ? launch_d2<64, 32, true>(p, grid, flop, bytes, stream) : launch_d2<64, 32, false>(p, grid, flop, bytes, stream);
  return launch_d2<128, 64, false>(p, grid, flop, bytes, stream);
}

// bf16 storage: dY, the transposed bank [Cin][9][Cout] and dX are bf16 (dense dX, no tap)
int dgrad2_launch_b16(const void* dy, int lddy, const void* wt16, void* dx, int n, int h, int wd, int cin, int accumulate, hipStream_t stream) {
  const int g_ncus = dcn_device_cus();
  if (!g_ncus) { dcn_set_error("dgrad2: device query failed"); return DCN_ERR_LAUNCH; }
  const int ch = cin == 32 ? 64 : 32, ck = 2 * cin;
  D2Params p{};
  p.dy = (const float*)dy; p.wt = (const float*)wt16; p.dx = (float*)dx; p.N = n; p.Ho = h / 2; p.Wo = wd / 2; p.lddy = lddy; p.ldo = cin;
  p.accumulate = accumulate;
  p.Mp = n * p.Ho * (p.Wo + 1);
  p.nchunks = cdiv(p.Mp, ch);
  int grid = g_ncus < p.nchunks ? g_ncus : p.nchunks;
  p.per_wg = cdiv(p.nchunks, grid);
  grid = cdiv(p.nchunks, p.per_wg);
  DCN_CHECK_ARG(((long long)p.per_wg * ch / (p.Wo + 1) + 4) * 4 * p.Wo * (lddy > cin ? lddy : cin) * 2 < 0x7FFFFFF0LL,
                "conv2d_bwd_data_b16: the rows of one workgroup exceed 32-bit byte offsets");
  const double bytes = 2.0 * ((double)n * p.Ho * p.Wo * ck + (double)n * h * wd * cin * (accumulate ? 2 : 1) + (double)cin * 9 * ck);
  const double flop = 2.0 * (double)n * p.Ho * p.Wo * ck * 9.0 * cin;
  if (cin == 32) return launch_d2<64, 32, false, true>(p, grid, flop, bytes, stream);
  return launch_d2<128, 64, false, true>(p, grid, flop, bytes, stream);
}

// ---- 3x3 layers between 32 and 64 channels (forward S = 1 | 2, data gradient S = 1) ---------------------------------------------
// mode 0: forward 32 -> 64 (stride 1 | 2);  mode 1: data gradient of the stride-1 layer (dY 64 channels -> dX 32 channels)
bool nconv1_applicable(int mode, int n, int h, int wd, int cin, int cout, int ksize, int stride) {
  if (!g_nconv || g_nconv == 3 || ksize != 3 || cin != 32 || cout != 64) return false;
  if (mode == 1 && stride != 1) return false;
  if (stride != 1 && stride != 2) return false;
  if ((h % stride) || (wd % stride) || wd / stride < 64 || h / stride < 2) return false;      // (one row wrap per chunk at most)
  if ((long long)n * (h / stride) * (wd / stride + 1) >= 0x7FFFFFF0LL / 64 || (long long)n * h >= 0x7FFFFFF0LL / 8) return false;   // (position / row indices in 32 bits)
  return (long long)n * (h / stride) * (wd / stride) >= 65536;                                // (a persistent grid wants work for every CU)
}

template <int S, int CK, int CN, bool FLIP, bool PRE = false, bool IN16 = false>
int launch_n1(N1Params& p, int grid, double flop, double bytes, hipStream_t stream) {
  typedef Geo1<S, CK> G;
  const int lds = 2 * G::BUFB + (CN == 32 ? 2 * 16 * 64 * 4 : 0);
  static DcnPerDeviceFlag attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nconv1_kernel<S, CK, CN, FLIP, PRE, IN16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  const int pid = prof_begin(38, flop, stream, bytes);
  hipLaunchKernelGGL((nconv1_kernel<S, CK, CN, FLIP, PRE, IN16>), dim3(grid), dim3(512), lds, stream, p);
  prof_end(pid, stream);
  DCN_CHECK_LAUNCH("nconv1");
  return DCN_OK;
}

// x: the gathered tensor (forward: input (n, h, wd, 32); data gradient: dY (n, h, wd, 64)), w: fp32 bank [CN][9][CK], y: the output
// (forward: (n, h/s, wd/s, 64) with pixel stride ldo; data gradient: dX (n, h, wd, 32)), stats: [stats_rows][2][64] or null
// pre (optional, forward only): x is the RAW output of the layer in front; its per-channel scale / shift and activation are applied
// in the loader (DcnPreAct), amax_x is then the abs-max word of the ACTIVATION (dcn_bn_act_amax_bound)
int nconv1_launch(int mode, const float* x, int ldi, const float* w, float* y, int ldo, float* stats, int stats_rows,
                  int n, int h, int wd, int stride, const uint32_t* amax_x, const uint32_t* amax_w, const DcnPreAct* pre, hipStream_t stream) {
  const int g_ncus = dcn_device_cus();
  if (!g_ncus) { dcn_set_error("nconv1: device query failed"); return DCN_ERR_LAUNCH; }
  const int ck = mode == 0 ? 32 : 64, cn = mode == 0 ? 64 : 32;
  N1Params p{};
  p.x = x; p.w = w; p.y = y; p.stats = stats; p.N = n; p.Ho = h / stride; p.Wo = wd / stride; p.ldi = ldi; p.ldo = ldo;
  p.Mp = n * p.Ho * (p.Wo + 1);
  p.nchunks = cdiv(p.Mp, 64);
  int grid = g_ncus < p.nchunks ? g_ncus : p.nchunks;
  p.per_wg = cdiv(p.nchunks, grid);
  grid = cdiv(p.nchunks, p.per_wg);
  DCN_CHECK_ARG(((long long)p.per_wg * 64 / (p.Wo + 1) + 4) * stride * wd * (ldi > ldo ? ldi : ldo) * 4 < 0x7FFFFFF0LL,
                "conv2d: the rows of one workgroup exceed 32-bit byte offsets");
  p.amax_x = amax_x; p.amax_w = amax_w;
  if (pre) {
    DCN_CHECK_ARG(mode == 0 && pre->scale && pre->shift && (((uintptr_t)pre->scale | (uintptr_t)pre->shift) & 15) == 0 && ldi == 32,
                  "conv2d_fwd: the loader-side activation exists for the dense 32-channel forward only");
    p.pre_scale = pre->scale; p.pre_shift = pre->shift; p.pre_act = pre->act; p.pre_slope = pre->slope;
  }
  if (stats) {
    DCN_CHECK_ARG(stats_rows >= grid, "conv2d_fwd: %d statistics rows for %d workgroups", stats_rows, grid);
    if (stats_rows > grid && hipMemsetAsync(stats + (size_t)grid * 2 * cn, 0, (size_t)(stats_rows - grid) * 2 * cn * sizeof(float), stream) != hipSuccess) {
      dcn_set_error("conv2d_fwd: memset of the statistics rows failed"); return DCN_ERR_LAUNCH;
    }
  }
  const double bytes = 4.0 * ((double)n * h * wd * ck + (double)n * p.Ho * p.Wo * cn + 64.0 * 9 * 32);
  const double flop = 2.0 * (double)n * p.Ho * p.Wo * 64 * 9.0 * 32;
  if (mode == 1) return launch_n1<1, 64, 32, true>(p, grid, flop, bytes, stream);
  if (pre) return stride == 1 ? launch_n1<1, 32, 64, false, true>(p, grid, flop, bytes, stream) : launch_n1<2, 32, 64, false, true>(p, grid, flop, bytes, stream);
  return stride == 1 ? launch_n1<1, 32, 64, false>(p, grid, flop, bytes, stream) : launch_n1<2, 32, 64, false>(p, grid, flop, bytes, stream);
}

// bf16 storage: x, the bank and y are bf16 (mode 0: forward 32 -> 64 with the OHWI bank [64][9][32]; mode 1: data gradient 64 -> 32 of the
// stride-1 layer with the transposed bank [32][9][64]); stats as in nconv1_launch
int nconv1_launch_b16(int mode, const void* x, int ldi, const void* w16, void* y, int ldo, float* stats, int stats_rows,
                      int n, int h, int wd, int stride, hipStream_t stream) {
  const int g_ncus = dcn_device_cus();
  if (!g_ncus) { dcn_set_error("nconv1: device query failed"); return DCN_ERR_LAUNCH; }
  const int ck = mode == 0 ? 32 : 64, cn = mode == 0 ? 64 : 32;
  N1Params p{};
  p.x = (const float*)x; p.w = (const float*)w16; p.y = (float*)y; p.stats = stats; p.N = n; p.Ho = h / stride; p.Wo = wd / stride; p.ldi = ldi; p.ldo = ldo;
  p.Mp = n * p.Ho * (p.Wo + 1);
  p.nchunks = cdiv(p.Mp, 64);
  int grid = g_ncus < p.nchunks ? g_ncus : p.nchunks;
  p.per_wg = cdiv(p.nchunks, grid);
  grid = cdiv(p.nchunks, p.per_wg);
  DCN_CHECK_ARG(((long long)p.per_wg * 64 / (p.Wo + 1) + 4) * stride * wd * (ldi > ldo ? ldi : ldo) * 2 < 0x7FFFFFF0LL,
                "conv2d (bf16): the rows of one workgroup exceed 32-bit byte offsets");
  if (stats) {
    DCN_CHECK_ARG(stats_rows >= grid, "conv2d_fwd_b16: %d statistics rows for %d workgroups", stats_rows, grid);
    if (stats_rows > grid && hipMemsetAsync(stats + (size_t)grid * 2 * cn, 0, (size_t)(stats_rows - grid) * 2 * cn * sizeof(float), stream) != hipSuccess) {
      dcn_set_error("conv2d_fwd_b16: memset of the statistics rows failed"); return DCN_ERR_LAUNCH;
    }
  }
  const double bytes = 2.0 * ((double)n * h * wd * ck + (double)n * p.Ho * p.Wo * cn + 64.0 * 9 * 32);
  const double flop = 2.0 * (double)n * p.Ho * p.Wo * 64 * 9.0 * 32;
  if (mode == 1) return launch_n1<1, 64, 32, true, false, true>(p, grid, flop, bytes, stream);
  return stride == 1 ? launch_n1<1, 32, 64, false, false, true>(p, grid, flop, bytes, stream) : launch_n1<2, 32, 64, false, false, true>(p, grid, flop, bytes, stream);
}
