// Probe: does the f16 two-piece split (three MFMAs per product, the fp32 path's arithmetic) run faster on v_mfma_f32_16x16x32_f16 than on
// v_mfma_f32_32x32x16_f16?  Both have the same cycles per FLOP; the microarchitecture guide measured the chip holding a HIGHER CLOCK on the
// 16x16x32 bf16 loop on random data (x1.12-1.15 FLOP/s, 'DVFS give-back' item 7).  Every convolution kernel of the step is built on 32x32x16.
//
// One workgroup = four waves (2 x 2), each a 64 x 64 output tile; operands live in LDS as two f16 planes (h, l) of 64-byte rows (32 k) and
// EVERY fragment is re-read from LDS by ds_read_b128 for each K-step (conflict-free swizzles for either lane layout), as in conv3.hip's loop:
//    32x32x16: per 32 k and wave 16 fragment reads, 24 MFMAs of 32 cycles         16x16x32: 16 fragment reads, 48 MFMAs of 16 cycles
// No global traffic in the loop.  Timing only, random operands (zeros raise the clock by themselves).  Optional: in-kernel clock stamps.
//    hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o /tmp/mfma_shape_probe && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 128;                     // rows of A and of B per workgroup
constexpr int PLANE = ROWS * 64;              // bytes per plane per K-step
constexpr int STEP = 4 * PLANE;               // A h, A l, B h, B l
constexpr int NSTEP = 2;                      // K-steps resident in LDS (the loop alternates between them)

// SHAPE 0: 32x32x16, 1: 16x16x32.  TERMS: 3 = split, 1 = plain f16.
template <int SHAPE, int TERMS>
__global__ __launch_bounds__(256, 2) void probe(const uint4* __restrict__ src, float* __restrict__ out, int iters, long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  for (int i = tid; i < NSTEP * STEP / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = src[(size_t)(blockIdx.x & 7) * (NSTEP * STEP / 16) + i];
  __syncthreads();
  long long t0 = 0, r0 = 0;
  if (stamps) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

  if constexpr (SHAPE == 0) {
    // lane (row = lane & 31, k-half = lane >> 5); K16 sub-step q: chunk 2 q + half; chunk' = chunk ^ ((row >> 2) & 3)
    int a_rd[2][2], b_rd[2][2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ar = wm * 64 + blk * 32 + (lane & 31), br = wn * 64 + blk * 32 + (lane & 31);
        a_rd[blk][q] = ar * 64 + (((2 * q + (lane >> 5)) ^ ((ar >> 2) & 3)) << 4);
        b_rd[blk][q] = 2 * PLANE + br * 64 + (((2 * q + (lane >> 5)) ^ ((br >> 2) & 3)) << 4);
      }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f16x8_t af[2][2][2], bf[2][2][2];         // [stage][block][plane]
    auto rd = [&](int st, const unsigned char* base, int q) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int pl = 0; pl < (TERMS == 3 ? 2 : 1); ++pl) {
          af[st][blk][pl] = *reinterpret_cast<const f16x8_t*>(base + pl * PLANE + a_rd[blk][q]);
          bf[st][blk][pl] = *reinterpret_cast<const f16x8_t*>(base + pl * PLANE + b_rd[blk][q]);
        }
    };
    auto mm = [&](int st) {
#pragma unroll
      for (int term = 0; term < TERMS; ++term) {
        const int qa = (TERMS == 3 && term == 0) ? 1 : 0, qb = (TERMS == 3 && term == 1) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[st][i][qa], bf[st][j][qb], acc[i][j], 0, 0, 0);
      }
    };
    rd(0, smem, 0);
    for (int it = 0; it < iters; ++it) {
      const unsigned char* base = smem + (it & (NSTEP - 1)) * STEP;
      const unsigned char* nbase = smem + ((it + 1) & (NSTEP - 1)) * STEP;
      rd(1, base, 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(0);
      __builtin_amdgcn_sched_barrier(0);
      rd(0, nbase, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm(1);
      __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
  } else {
    // lane (row = lane & 15, k-group g = lane >> 4): chunk g; chunk' = g ^ f(row >> 2 & 3), f = (0, 2, 3, 1): conflict-free for the
    // ds_read_b128 lane groups {0-3, 12-15, 20-27} ...
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    int a_rd[4], b_rd[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      const int ar = wm * 64 + blk * 16 + (lane & 15), br = wn * 64 + blk * 16 + (lane & 15);
      const int fa = (0x1320 >> (4 * ((ar >> 2) & 3))) & 3, fb = (0x1320 >> (4 * ((br >> 2) & 3))) & 3;
      a_rd[blk] = ar * 64 + ((((lane >> 4)) ^ fa) << 4);
      b_rd[blk] = 2 * PLANE + br * 64 + ((((lane >> 4)) ^ fb) << 4);
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    f16x8_t af[2][4][2], bf[2][4][2];
    auto rd_half = [&](int st, const unsigned char* base, int lo) {           // blocks lo, lo + 1 of A and of B
#pragma unroll
      for (int blk = lo; blk < lo + 2; ++blk)
#pragma unroll
        for (int pl = 0; pl < (TERMS == 3 ? 2 : 1); ++pl) {
          af[st][blk][pl] = *reinterpret_cast<const f16x8_t*>(base + pl * PLANE + a_rd[blk]);
          bf[st][blk][pl] = *reinterpret_cast<const f16x8_t*>(base + pl * PLANE + b_rd[blk]);
        }
    };
    auto mm_half = [&](int st, int ilo) {     // A blocks ilo, ilo + 1 against all four B blocks
#pragma unroll
      for (int term = 0; term < TERMS; ++term) {
        const int qa = (TERMS == 3 && term == 0) ? 1 : 0, qb = (TERMS == 3 && term == 1) ? 1 : 0;
#pragma unroll
        for (int i = ilo; i < ilo + 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[st][i][qa], bf[st][j][qb], acc[i][j], 0, 0, 0);
      }
    };
    rd_half(0, smem, 0); rd_half(0, smem, 2);
    for (int it = 0; it < iters; it += 2) {
      const unsigned char* b1 = smem + ((it + 1) & (NSTEP - 1)) * STEP;
      const unsigned char* b2 = smem + ((it + 2) & (NSTEP - 1)) * STEP;
      rd_half(1, b1, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm_half(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      rd_half(1, b1, 2);
      __builtin_amdgcn_sched_barrier(0);
      mm_half(0, 2);
      __builtin_amdgcn_sched_barrier(0);
      rd_half(0, b2, 0);
      __builtin_amdgcn_sched_barrier(0);
      mm_half(1, 0);
      __builtin_amdgcn_sched_barrier(0);
      rd_half(0, b2, 2);
      __builtin_amdgcn_sched_barrier(0);
      mm_half(1, 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
  }
  if (stamps && tid == 0) {
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SHAPE, int TERMS>
static void run(const char* name, const uint4* src, float* out, long long* stamps, int grid, int iters) {
  const size_t lds = (size_t)NSTEP * STEP;
  CK(hipFuncSetAttribute((const void*)probe<SHAPE, TERMS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // >= 2 s of back-to-back launches before the timed ones (the clock settles under load)
  float ms = 0.f;
  int reps = 0;
  CK(hipEventRecord(e0));
  do {
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<SHAPE, TERMS>), dim3(grid), dim3(256), lds, 0, src, out, iters, (long long*)nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    reps += 20;
  } while (ms < 2000.f);
  CK(hipEventRecord(e0));
  const int N = 50;
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL((probe<SHAPE, TERMS>), dim3(grid), dim3(256), lds, 0, src, out, iters, (long long*)nullptr);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  const double per = ms / N;
  // per workgroup and iteration: 4 waves x 64 x 64 x 32 k x 2 FLOP (algorithmic: one product, whatever the number of MFMA terms)
  const double flop = (double)grid * iters * 4 * 64 * 64 * 32 * 2;
  // the in-kernel clock, stamped in a launch of its own right behind the timed ones
  hipLaunchKernelGGL((probe<SHAPE, TERMS>), dim3(grid), dim3(256), lds, 0, src, out, iters, stamps);
  CK(hipDeviceSynchronize());
  std::vector<long long> st(2 * grid);
  CK(hipMemcpy(st.data(), stamps, sizeof(long long) * 2 * grid, hipMemcpyDeviceToHost));
  std::vector<double> mhz;
  for (int b = 0; b < grid; ++b) if (st[2 * b + 1] > 0) mhz.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100.0);
  std::sort(mhz.begin(), mhz.end());
  printf("%-28s %8.3f ms  %7.1f TFLOP/s algorithmic  (%6.1f TFLOP/s of MFMA work)  in-kernel clock %.0f MHz\n", name, per, flop / per * 1e-9,
         flop * TERMS / per * 1e-9, mhz.empty() ? 0.0 : mhz[mhz.size() / 2]);
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 4096;
  const size_t n16 = (size_t)8 * NSTEP * STEP / 16;
  std::vector<uint16_t> h(n16 * 8);
  uint32_t s = 12345u;
  for (auto& v : h) {                          // random f16 values in +-[2^-3, 2^3): sign, exponent 12..17, random mantissa
    s = s * 1664525u + 1013904223u;
    const uint32_t r = s >> 8;
    v = (uint16_t)(((r & 1) << 15) | ((12 + (r >> 1) % 6) << 10) | ((r >> 8) & 0x3FF));
  }
  uint4* src; float* out; long long* stamps;
  CK(hipMalloc(&src, n16 * 16)); CK(hipMalloc(&out, (size_t)grid * 256 * 4)); CK(hipMalloc(&stamps, sizeof(long long) * 2 * grid));
  CK(hipMemcpy(src, h.data(), n16 * 16, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 3>("32x32x16 f16 split (3 MFMA)", src, out, stamps, grid, iters);
    run<1, 3>("16x16x32 f16 split (3 MFMA)", src, out, stamps, grid, iters);
    run<0, 1>("32x32x16 f16 (1 MFMA)", src, out, stamps, grid, iters);
    run<1, 1>("16x16x32 f16 (1 MFMA)", src, out, stamps, grid, iters);
  }
  return 0;
}
