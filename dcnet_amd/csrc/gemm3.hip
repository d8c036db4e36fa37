// Batched GEMM on PRE-SPLIT operands: C[b] (+)= diag(row_scale) . op(A[b]) . op(B[b])^T in fp32 accuracy, for the products of the
// inter-frame co-attention (coattn.hip: model/DCNet_model.py:449-459) — both operands are activations there, no filter bank, so the
// implicit-GEMM tiles split BOTH tiles in their loaders (8.5 vector instructions per MFMA) and ran these nine products per scale at
// 0.25 of the MFMA ceiling.
//
// Here every operand arrives in the split form of dcn_prepare_filters' banks: same bytes and strides as the fp32 tensor, each run of
// 8 consecutive elements of a row replaced by [8 f16 high | 8 f16 low] of s*x (s = the tensor's power-of-two scale).  Written ONCE
// per tensor (gemm3_presplit, or by the producing pass: exp_sums_kernel writes E that way), read by every product that uses it, in
// either orientation:
//   * K along the row ("R": A of the NT / NN forms, B of NT): a lane's MFMA fragment — 8 consecutive k — is one 16-byte piece.
//   * K across rows ("T": B of NN, A and B of TN): the LDS tile keeps the HBM orientation [k][columns] and the fragment comes from
//     two ds_read_b64_tr_b16 (as in wgrad3.hip); 4 consecutive columns never leave an 8-element run, so the same bytes serve.
// Both tiles of a K-slice (16 k: 16 KB + 16 KB for the 256 x 256 tile) go global -> LDS by `buffer_load_dwordx4 ... lds`: no vector
// registers, no ds_write, no split in the loop; the swizzles that make the fragment reads conflict-free live in the per-lane SOURCE
// address (the LDS image of an LDS-DMA is lane-linear).  Ring of four slices, three in flight across the barriers (counted
// `s_waitcnt vmcnt`, raw `s_barrier`: cdna_hip_programming.md section 5), one workgroup of 8 waves per CU, wave tile 128 x 64: 12
// 16-byte fragment reads per 24 MFMAs (conv1.hip's 32 x 128 wave tile needs 20).  Rows / k beyond a tensor are out-of-range DMA
// lanes (zeros).  Roofline: MFMA, 838.9 TFLOP/s (three f16 MFMAs per product: (l,h) + (h,l) + (h,h)).
#include "igemm.h"
#include "prof.h"

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

#define G3_ABL 0                                 // 1: the timing-only ablations of dcn_set_tuning("Gemm3", 1 + 16 * bits) are compiled in
constexpr unsigned G3_OOB = 0x80000000u;
constexpr int G3_BM = 256, G3_BN = 256, G3_STAGES = 4;
constexpr int G3_TILE = 256 * 64;                 // bytes of one operand tile of a K-slice (R: [256 rows][64 B]; T: [16 k][1024 B])
constexpr int G3_STAGE = 2 * G3_TILE;
constexpr int G3_LDS = G3_STAGES * G3_STAGE;      // 128 KB

__device__ __forceinline__ float g3_pow2_scale(unsigned amax_bits) {      // = igemm.hip pow2_scale
  const int be = (int)((amax_bits >> 23) & 0xFF);
  if (be == 0 || be == 255) return 1.f;
  int e = 14 - (be - 126);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return __uint_as_float((unsigned)(e + 127) << 23);
}

struct G3Params {
  const float* A; const float* B; float* C;       // A, B in split form
  long long a_bs, b_bs, c_bs;                     // batch strides (floats)
  int lda, ldb, ldc;                              // row strides (floats)
  int M, N, K;                                    // K: valid k (k >= K reads zeros from the T operands; R operands must hold zeros in [K, ceil16 K))
  int a_rows, b_rows;                             // rows of A / B that exist per batch (R: M / N; T: K)
  int a_cols, b_cols;                             // floats of a row that exist (R: >= ceil16 K; T: M / N)
  int tiles_m, tiles_n;
  const float* row_scale; long long rs_bs;        // optional [batch][M]
  int accumulate;
  const unsigned* amax_a; const unsigned* amax_b;
  unsigned* amax_out;                             // optional: abs-max word of what is stored
  int var;                                        // schedule variant (dcn_set_tuning("Gemm3", 1 + 256 * v))
  int abl;                                        // timing-only ablations (wrong results): 1 no DMA, 2 no fragment reads, 4 no MFMAs
};

// T-form swizzle of a k-row's 16-byte chunks: rows k, k+1, k+2, k+3 of a transposed read land in four disjoint bank sets
__device__ __forceinline__ int g3_tswz(int krow) { return (krow & 1) | ((krow & 2) << 1); }

// H1 (the bf16 precision modes): only the HIGH piece of each operand is read and multiplied — f16 operands with the tensor's power-of-two
// scale, 11 significant bits (three more than bf16), ONE MFMA per product instead of three; same tiles, same DMA, same schedule.
template <bool AT, bool BT, bool H1 = false>
__global__ __launch_bounds__(512, 1) void gemm3_kernel(const G3Params p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem3g[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int per_batch = p.tiles_m * p.tiles_n;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  // (readfirstlane: the divisions run on the vector ALU; descriptors built from their results would otherwise sit in vector
  //  registers and every LDS-DMA would be wrapped in a waterfall loop)
  const int batch = __builtin_amdgcn_readfirstlane(lin / per_batch), t_ = lin - batch * per_batch;
  const int bm = __builtin_amdgcn_readfirstlane(t_ / p.tiles_n), bn = t_ - bm * p.tiles_n;
  const int m0 = bm * G3_BM, n0 = bn * G3_BN;
  const int kslices = (p.K + 15) >> 4;

  // ---- descriptors: one per operand, based at this batch (and, for R operands, at the tile's first row) -----------------------
  const float* a_base = p.A + batch * p.a_bs + (AT ? 0 : (long long)m0 * p.lda);
  const float* b_base = p.B + batch * p.b_bs + (BT ? 0 : (long long)n0 * p.ldb);
  const int a_rows_left = AT ? p.a_rows : p.a_rows - m0, b_rows_left = BT ? p.b_rows : p.b_rows - n0;
  const long long a_bytes = ((long long)(a_rows_left - 1) * p.lda + p.a_cols) * 4, b_bytes = ((long long)(b_rows_left - 1) * p.ldb + p.b_cols) * 4;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, a_bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0 : (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, b_bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0 : (int)b_bytes, 0x00020000);

  // ---- per-lane source offsets of this wave's four 1-KiB pieces per K-slice (pieces 2 wave, 2 wave + 1 of each tile) ------------
  // R tile: piece j = rows 16 j .. 16 j + 15; LDS position (row, c') = (16 j + lane / 4, lane % 4) holds chunk c = c' ^ ((row >> 2) & 3)
  //         of the row's 64 bytes (chunks: h of k 0-7, l of k 0-7, h of k 8-15, l of k 8-15).
  // T tile: piece j = k-row j; LDS position c' = lane holds chunk c = c' ^ tswz(j) of the row's 1024 bytes (256 columns).
  unsigned voff[4];
  int sstep_a, sstep_b;                            // scalar advance per K-slice (bytes)
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int j = 2 * wave + e;
    if (AT) {
      voff[e] = (unsigned)((long long)j * p.lda * 4 + m0 * 4 + ((lane ^ g3_tswz(j)) << 4));
      if (m0 + ((lane ^ g3_tswz(j)) << 2) >= p.a_cols) voff[e] = G3_OOB;          // (a chunk = 4 columns; columns past the row: zeros)
    } else {
      const int row = 16 * j + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
      voff[e] = row < a_rows_left ? (unsigned)((long long)row * p.lda * 4 + c * 16) : G3_OOB;
    }
    if (BT) {
      voff[2 + e] = (unsigned)((long long)j * p.ldb * 4 + n0 * 4 + ((lane ^ g3_tswz(j)) << 4));
      if (n0 + ((lane ^ g3_tswz(j)) << 2) >= p.b_cols) voff[2 + e] = G3_OOB;
    } else {
      const int row = 16 * j + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
      voff[2 + e] = row < b_rows_left ? (unsigned)((long long)row * p.ldb * 4 + c * 16) : G3_OOB;
    }
  }
  sstep_a = AT ? 16 * p.lda * 4 : 64;
  sstep_b = BT ? 16 * p.ldb * 4 : 64;
  const int my_dst = 2 * wave * 1024;

  int k_done = 0;
  // this wave's pieces of its next K-slice: the two of the A tile, then the two of the B tile (past the end: no-ops that still count in vmcnt)
  auto issue_a = [&]() {
    const bool live = k_done < kslices && !(G3_ABL && (p.abl & 1));
    unsigned char* st = smem3g + (k_done & (G3_STAGES - 1)) * G3_STAGE + my_dst;
    int sa_ = live ? k_done * sstep_a : 0;
    if (G3_ABL && (p.abl & 8)) sa_ = 0;                     // every slice re-reads the first one (cache-hot)
#pragma unroll
    for (int e = 0; e < 2; ++e)
      // (explicit int casts: with unsigned arguments hipcc 7.2 silently drops the instantiation of the whole kernel template)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_void*)(st + e * 1024), 16, (int)(live ? voff[e] : G3_OOB), sa_, 0, 0);
  };
  auto issue_b = [&]() {
    const bool live = k_done < kslices && !(G3_ABL && (p.abl & 1));
    unsigned char* st = smem3g + (k_done & (G3_STAGES - 1)) * G3_STAGE + my_dst;
    int sb_ = live ? k_done * sstep_b : 0;
    if (G3_ABL && (p.abl & 8)) sb_ = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (lds_void*)(st + G3_TILE + e * 1024), 16, (int)(live ? voff[2 + e] : G3_OOB), sb_, 0, 0);
    ++k_done;
  };
  auto issue = [&]() { issue_a(); issue_b(); };

  // ---- fragment addresses (bytes inside a stage) ---------------------------------------------------------------------------------
  // R: lane (r = lane & 31, g = lane >> 5): h = chunk 2 g, l = chunk 2 g + 1 of row r.
  // T: 16-lane group (hh, gg), lane (qq, pp) inside it: k rows 8 hh + 4 r2 + qq, columns 16 gg + 4 pp .. + 3 (wgrad3.hip's map).
  const int g = lane >> 5;
  const int hh = lane >> 5, gg = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
  int a_rd[4][2], b_rd[2][2];                     // R: [tile][plane];  T: [tile][r2] of the h plane (l: + 16 bytes before the swizzle -> ^ 16)
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    if (AT) {
      const int c = wm * 128 + mi * 32 + 16 * gg + 4 * pp;
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        const int kr = 8 * hh + 4 * r2 + qq;
        a_rd[mi][r2] = kr * 1024 + ((((c >> 3) * 2) ^ g3_tswz(kr)) << 4) + (c & 7) * 2;
      }
    } else {
      const int ar = wm * 128 + mi * 32 + (lane & 31);
      a_rd[mi][0] = ar * 64 + (((2 * g) ^ ((ar >> 2) & 3)) << 4);
      a_rd[mi][1] = ar * 64 + (((2 * g + 1) ^ ((ar >> 2) & 3)) << 4);
    }
  }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    if (BT) {
      const int c = wn * 64 + ni * 32 + 16 * gg + 4 * pp;
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        const int kr = 8 * hh + 4 * r2 + qq;
        b_rd[ni][r2] = G3_TILE + kr * 1024 + ((((c >> 3) * 2) ^ g3_tswz(kr)) << 4) + (c & 7) * 2;
      }
    } else {
      const int br = wn * 64 + ni * 32 + (lane & 31);
      b_rd[ni][0] = G3_TILE + br * 64 + (((2 * g) ^ ((br >> 2) & 3)) << 4);
      b_rd[ni][1] = G3_TILE + br * 64 + (((2 * g + 1) ^ ((br >> 2) & 3)) << 4);
    }
  }
  auto tr_read = [&](const unsigned char* ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr));
  };
  auto tfrag = [&](const unsigned char* st, int off0, int off1, int plane) {      // plane 1: the chunk beside (^ 16 bytes)
    const s16x4 lo = tr_read(st + (off0 ^ (plane << 4))), hi = tr_read(st + (off1 ^ (plane << 4)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8_t, v);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  for (int s = 0; s < G3_STAGES - 1; ++s) issue();

  // Two groups of four waves (wm = 0 | 1: one wave of each per SIMD) run the same loop ONE BARRIER APART, so that in every barrier
  // interval one group streams its 24 MFMAs while the other issues the next slice's DMA and reads its fragments:
  //   group 0:  b1 [issue, read s] b2 [MFMA s, wait s+1] b1 [issue, read s+1] b2 ...
  //   group 1:     b0              b1 [issue, read s, wait s+1]  b2 [MFMA s]  b1 ...
  // A slice is read only after every wave's counted vmcnt for it AND a barrier behind that wait (group 0 waits behind its MFMAs, group 1
  // behind its reads: both in front of the barrier that opens group 0's reads of slice s+1); a stage is refilled only behind a barrier
  // that the last reader reached with its reads retired (lgkmcnt(0) in front of b2).
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"((G3_STAGES - 2) * 4) : "memory");       // slice 0 of this wave
  if (wm == 1) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < kslices; ++it) {
    __builtin_amdgcn_s_barrier();                                                     // b1
    asm volatile("" ::: "memory");
    const unsigned char* st = smem3g + (it & (G3_STAGES - 1)) * G3_STAGE;
    f16x8_t bh[2], bl[2];
    f16x8_t ah[4], al[4];
    if (G3_ABL && (p.abl & 2)) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) { bh[ni] = f16x8_t{(_Float16)it, 0, 0, 0, 0, 0, 0, 0}; bl[ni] = bh[ni]; }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) { ah[mi] = bh[0]; al[mi] = bh[1]; }
    } else {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (BT) { bh[ni] = tfrag(st, b_rd[ni][0], b_rd[ni][1], 0); if constexpr (!H1) bl[ni] = tfrag(st, b_rd[ni][0], b_rd[ni][1], 1); }
        else { bh[ni] = *reinterpret_cast<const f16x8_t*>(st + b_rd[ni][0]); if constexpr (!H1) bl[ni] = *reinterpret_cast<const f16x8_t*>(st + b_rd[ni][1]); }
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        if (AT) { ah[mi] = tfrag(st, a_rd[mi][0], a_rd[mi][1], 0); if constexpr (!H1) al[mi] = tfrag(st, a_rd[mi][0], a_rd[mi][1], 1); }
        else { ah[mi] = *reinterpret_cast<const f16x8_t*>(st + a_rd[mi][0]); if constexpr (!H1) al[mi] = *reinterpret_cast<const f16x8_t*>(st + a_rd[mi][1]); }
      }
    }
    // slice it + 3, into the stage of slice it - 1.  BEHIND the fragment reads: the memory path is the narrow one here (a CU gets
    // ~40 GB/s from L2 into LDS, measured with the loop stripped to its DMA), so a DMA instruction can sit in the issue stage until
    // the queue has room — in front of the reads it kept this group's fragments, hence its MFMA phase, waiting (0.92 -> 0.70 ms)
    if (p.var >= 1) {                        // (variants: the B pieces behind (1) / between (2) the MFMAs)
      issue_a();
      if (wm == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((G3_STAGES - 2) * 4 - 2) : "memory");
    } else {
      issue();
      if (wm == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((G3_STAGES - 2) * 4) : "memory");      // slice it + 1 of this wave
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                                     // b2
    asm volatile("" ::: "memory");
    if (G3_ABL && (p.abl & 4)) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) { asm volatile("" :: "v"(ah[mi]), "v"(al[mi])); }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) { asm volatile("" :: "v"(bh[ni]), "v"(bl[ni])); }
    } else {
      // smallest terms first: (l,h) (h,l) (h,h); eight independent accumulators between two uses of one
      __builtin_amdgcn_s_setprio(1);
      if constexpr (!H1) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
        if (p.var == 2) issue_b();
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
      } else if (p.var == 2) issue_b();
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    if (p.var == 1) issue_b();
    if (wm == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((G3_STAGES - 2) * 4) : "memory");      // slice it + 1 of this wave
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();
  // the no-op pieces issued past the end may still be pending LDS writes: drain before the workgroup ends
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: 1 / (s_a s_b), row scale, accumulate, store --------------------------------------------------------------------
  const float dq = 1.f / (g3_pow2_scale(amax_read(p.amax_a)) * g3_pow2_scale(amax_read(p.amax_b)));      // powers of two: exact
  float* __restrict__ cb = p.C + batch * p.c_bs;
  const float* rs = p.row_scale ? p.row_scale + batch * p.rs_bs : nullptr;
  float vmax = 0.f;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 128 + mi * 32 + 4 * g + (r & 3) + 8 * (r >> 2);
      if (m >= p.M) continue;
      const float f = rs ? dq * rs[m] : dq;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wn * 64 + ni * 32 + (lane & 31);
        if (n >= p.N) continue;
        float v = acc[mi][ni][r] * f;
        float* dst = cb + (long long)m * p.ldc + n;
        if (p.accumulate) v += *dst;
        *dst = v;
        vmax = fmaxf(vmax, fabsf(v));
      }
    }
  if (p.amax_out) {
    vmax = wave_max(vmax);
    if (lane == 0) amax_update(p.amax_out, vmax, blockIdx.x * 8 + wave);
  }
}

// x (fp32 [batch][rows][c], row stride ld) -> its split form, dense or in place: 8 elements per thread
__global__ __launch_bounds__(256) void presplit_kernel(const float* __restrict__ src, int ld, long long bs, float* __restrict__ dst, int ldd,
                                                        long long bsd, int rows, int c, long long total8, const unsigned* __restrict__ amax) {
  const float s = g3_pow2_scale(amax_read(amax));
  const int c8 = c >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long long)gridDim.x * 256) {
    const long long row_all = i / c8; const int o = (int)(i - row_all * c8);
    const long long b = row_all / rows; const int row = (int)(row_all - b * rows);
    const float* sp = src + b * bs + (long long)row * ld + o * 8;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(sp) * s, x1 = *reinterpret_cast<const f32x4*>(sp + 4) * s;
    f16x8_t h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h[e] = (_Float16)x0[e]; h[4 + e] = (_Float16)x1[e]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { l[e] = (_Float16)(x0[e] - (float)h[e]); l[4 + e] = (_Float16)(x1[e] - (float)h[4 + e]); }
    float* dp = dst + b * bsd + (long long)row * ldd + o * 8;
    *reinterpret_cast<f16x8_t*>(dp) = h;
    *reinterpret_cast<f16x8_t*>(dp + 4) = l;
  }
}

int g_gemm3 = 1;          // dcn_set_tuning("Gemm3", 0): the co-attention products back on the implicit-GEMM / weight-gradient tiles

int g_gemm3_h1 = 1;       // dcn_set_tuning("H1gemm3", 0): the bf16 precision modes multiply both pieces as well (fp32-accurate co-attention)

template <bool AT, bool BT, bool H1>
int launch3g_(const G3Params& p, int grid, hipStream_t stream) {
  static DcnPerDeviceFlag attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel<AT, BT, H1>), hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
  }
  hipLaunchKernelGGL((gemm3_kernel<AT, BT, H1>), dim3(grid), dim3(512), G3_LDS, stream, p);
  return DCN_OK;
}
template <bool AT, bool BT>
int launch3g(const G3Params& p, int grid, hipStream_t stream) {
  // precision 2 = the bf16 modes: one f16 piece per operand (11 significant bits: finer than the bf16 those modes are defined by)
  if (g_gemm3_h1 && igemm_precision() == 2) return launch3g_<AT, BT, true>(p, grid, stream);
  return launch3g_<AT, BT, false>(p, grid, stream);
}

}  // namespace

void gemm3_set_tuning(int v) { g_gemm3 = v; }
void gemm3_set_h1(int v) { g_gemm3_h1 = v; }

// shapes the kernel takes: rows of 16-byte granularity everywhere, the T operands' tile columns inside their rows
bool gemm3_applicable(int M, int N, int K, int batch) {
  // (precision 2, the bf16 modes: the co-attention keeps its fp32-accurate f16 split — its operands are fp32 tensors either way, and
  //  these tiles are twice as fast as the fp32-pipe tiles the mode would fall back to)
  return g_gemm3 && (igemm_precision() == 4 || igemm_precision() == 2) && M >= 512 && N >= 256 && K >= 64 && batch >= 1;
}

int gemm3_presplit(const float* src, int ld, long long bs, float* dst, int ldd, long long bsd, int batch, int rows, int c,
                   const unsigned* amax, hipStream_t stream) {
  DCN_CHECK_ARG(src && dst && amax && c % 8 == 0 && ld % 4 == 0 && ldd % 4 == 0 && bs % 4 == 0 && bsd % 4 == 0, "gemm3_presplit: bad argument");
  const long long total8 = (long long)batch * rows * (c / 8);
  long long g = (total8 + 255) / 256; if (g > 16384) g = 16384;
  hipLaunchKernelGGL(presplit_kernel, dim3((int)g), dim3(256), 0, stream, src, ld, bs, dst, ldd, bsd, rows, c, total8, amax);
  DCN_CHECK_LAUNCH("gemm3_presplit");
  return DCN_OK;
}

// C[b][M][N] (+)= diag(row_scale[b]) op(A[b]) op(B[b])^T on split operands.  at / bt: the operand is stored [K][M] / [K][N] (K across
// rows).  a_cols / b_cols: floats of a row that exist (R operands: zeros in [K, ceil16 K) required; T operands: the M / N columns).
int gemm3_launch(const float* A, int lda, long long a_bs, int at, const float* B, int ldb, long long b_bs, int bt,
                 float* C, int ldc, long long c_bs, const float* row_scale, long long rs_bs,
                 int M, int N, int K, int batch, int accumulate, const unsigned* amax_a, const unsigned* amax_b, hipStream_t stream,
                 unsigned* amax_out) {
  DCN_CHECK_ARG(A && B && C && amax_a && amax_b, "gemm3: null pointer");
  DCN_CHECK_ARG(lda % 4 == 0 && ldb % 4 == 0 && a_bs % 4 == 0 && b_bs % 4 == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0, "gemm3: 16-byte rows");
  G3Params p{};
  p.A = A; p.B = B; p.C = C; p.a_bs = a_bs; p.b_bs = b_bs; p.c_bs = c_bs; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  const int k16 = (K + 15) / 16 * 16;
  p.a_rows = at ? K : M; p.b_rows = bt ? K : N;
  p.a_cols = at ? (M + 7) / 8 * 8 : k16; p.b_cols = bt ? (N + 7) / 8 * 8 : k16;
  DCN_CHECK_ARG(p.a_cols <= lda && p.b_cols <= ldb, "gemm3: rows shorter than the tile reads (lda=%d ldb=%d)", lda, ldb);
  p.tiles_m = cdiv(M, G3_BM); p.tiles_n = cdiv(N, G3_BN);
  p.row_scale = row_scale; p.rs_bs = rs_bs; p.accumulate = accumulate; p.amax_a = amax_a; p.amax_b = amax_b; p.amax_out = amax_out; p.abl = (g_gemm3 >> 4) & 15; p.var = g_gemm3 >> 8;
  const int grid = p.tiles_m * p.tiles_n * batch;
  const int pid = prof_begin(40, 2.0 * batch * (double)M * N * K, stream);
  int rc;
  if (at) rc = bt ? launch3g<true, true>(p, grid, stream) : DCN_ERR_ARG;
  else rc = bt ? launch3g<false, true>(p, grid, stream) : launch3g<false, false>(p, grid, stream);
  prof_end(pid, stream);
  DCN_CHECK_ARG(rc == DCN_OK, "gemm3: no A^T . B^T form");
  DCN_CHECK_LAUNCH("gemm3");
  return DCN_OK;
}

// ---- C face (tests / callers with their own operands) ---------------------------------------------------------------------------
extern "C" int dcn_gemm3_supported(int m, int n, int k, int batch) { return gemm3_applicable(m, n, k, batch) ? 1 : 0; }

extern "C" int dcn_gemm3_presplit(const float* src, int ld, int64_t bs, float* dst, int ldd, int64_t bsd, int batch, int rows, int c,
                                  const uint32_t* amax, void* stream) {
  return gemm3_presplit(src, ld, bs, dst, ldd, bsd, batch, rows, c, amax, (hipStream_t)stream);
}

extern "C" int dcn_gemm3(const float* a, int lda, int64_t a_bs, int a_t, const float* b, int ldb, int64_t b_bs, int b_t,
                         float* c, int ldc, int64_t c_bs, const float* row_scale, int64_t rs_bs,
                         int m, int n, int k, int batch, int accumulate, const uint32_t* amax_a, const uint32_t* amax_b, void* stream) {
  DCN_CHECK_ARG(m > 0 && n > 0 && k > 0 && batch > 0, "gemm3: empty problem");
  return gemm3_launch(a, lda, a_bs, a_t, b, ldb, b_bs, b_t, c, ldc, c_bs, row_scale, rs_bs, m, n, k, batch, accumulate, amax_a, amax_b,
                      (hipStream_t)stream, nullptr);
}
