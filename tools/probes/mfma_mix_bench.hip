// Micro-benchmark: the split kernel's per-K-step mix (24 x 32x32x16 or 48 x 16x16x32 bf16 MFMAs + ~100 VALU "split" ops
// + LDS fragment reads) on random data, 3 waves/SIMD-like occupancy (256 threads x 3 blocks/CU via LDS size), to see
// whether the MFMA shape changes the sustained rate (DVFS) under this instruction mix.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, int VALU>
__global__ __launch_bounds__(256, 3) void k(const float* __restrict__ in, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  // fill 24 KB of LDS "planes" with random bf16 from global
  for (int i = tid; i < 6144; i += 256) reinterpret_cast<float*>(smem)[i] = in[(blockIdx.x * 6144 + i) & 0xFFFFF];
  __syncthreads();
  f32x4 x = *reinterpret_cast<const f32x4*>(in + tid * 4);
  float vs = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[2][2] = {};
    for (int it = 0; it < iters; ++it) {
      bf16x8 af[2][3], bf[2][3];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          af[m][q] = *reinterpret_cast<const bf16x8*>(smem + ((q * 2 + m) * 1024 + lane * 16 + (it & 7) * 16) % 24560 / 16 * 16);
          bf[m][q] = *reinterpret_cast<const bf16x8*>(smem + ((q * 2 + m) * 1024 + 12288 + lane * 16 + (it & 7) * 16) % 24560 / 16 * 16);
        }
      constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][QA[t]], bf[n][QB[t]], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < VALU / 6 / 4; ++v) {      // fake split arithmetic: and, sub chains on 4 values
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float h = __uint_as_float(__float_as_uint(x[e]) & 0xFFFF0000u); x[e] = x[e] - h * 0.5f; }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) vs += acc[m][n][r];
  } else {
    f32x4 acc[4][4] = {};
    for (int it = 0; it < iters; ++it) {
      bf16x8 af[4][2], bf[4][3];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int q = 0; q < 2; ++q) af[m][q] = *reinterpret_cast<const bf16x8*>(smem + ((q * 4 + m) * 1024 + lane * 16 + (it & 7) * 16) % 24560 / 16 * 16);
#pragma unroll
        for (int q = 0; q < 3; ++q) bf[m][q] = *reinterpret_cast<const bf16x8*>(smem + ((q * 4 + m) * 1024 + 8192 + lane * 16 + (it & 7) * 16) % 24560 / 16 * 16);
      }
      constexpr int QA[3] = {0, 0, 1}, QB[3] = {0, 1, 2};
#pragma unroll
      for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[m][QA[t]], bf[n][QB[t]], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < VALU / 3 / 4; ++v) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float h = __uint_as_float(__float_as_uint(x[e]) & 0xFFFF0000u); x[e] = x[e] - h * 0.5f; }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) vs += acc[m][n][r];
  }
  out[blockIdx.x * 256 + tid] = vs + x[0] + x[1] + x[2] + x[3];
}

template <int SHAPE, int VALU>
float run(const float* in, float* out, int blocks, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<SHAPE, VALU>), hipFuncAttributeMaxDynamicSharedMemorySize, 49152);
  hipLaunchKernelGGL((k<SHAPE, VALU>), dim3(blocks), dim3(256), 49152, 0, in, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<SHAPE, VALU>), dim3(blocks), dim3(256), 49152, 0, in, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
  const int blocks = 256 * 3 * 4, iters = 2000;
  std::vector<float> h(1 << 20); srand(1); for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  float *in, *out; hipMalloc(&in, h.size() * 4); hipMalloc(&out, blocks * 256 * 4);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const double flop = (double)blocks * 4 * iters * 24 * 32768;   // MFMA flops (both shapes: 24 x 32x32x16 == 48 x 16x16x32)
  for (int rep = 0; rep < 2; ++rep) {
    float t;
    t = run<32, 96>(in, out, blocks, iters);  printf("32x32x16  valu96  %.3f ms  %.0f TF/s (bf16 MFMA)\n", t, flop / t / 1e9);
    t = run<16, 96>(in, out, blocks, iters);  printf("16x16x32  valu96  %.3f ms  %.0f TF/s\n", t, flop / t / 1e9);
    t = run<32, 0>(in, out, blocks, iters);   printf("32x32x16  valu0   %.3f ms  %.0f TF/s\n", t, flop / t / 1e9);
    t = run<16, 0>(in, out, blocks, iters);   printf("16x16x32  valu0   %.3f ms  %.0f TF/s\n", t, flop / t / 1e9);
    t = run<32, 48>(in, out, blocks, iters);  printf("32x32x16  valu48  %.3f ms  %.0f TF/s\n", t, flop / t / 1e9);
    t = run<16, 48>(in, out, blocks, iters);  printf("16x16x32  valu48  %.3f ms  %.0f TF/s\n", t, flop / t / 1e9);
  }
  return 0;
}
