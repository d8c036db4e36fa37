// Probe for BASELINE.json configs[4] (fp8 conv path): what would fp8 STORAGE with the block-scaled MFMA buy the forward / data-gradient
// launches of the bf16-storage mode?  conv1.hip's conv1b ring kernel — 128 x 128 tile, four waves, both operand tiles by LDS-DMA into
// rings of 3 + 2 K-steps of 64-byte rows, counted s_waitcnt vmcnt, raw s_barrier — as a plain NT GEMM, in two builds that stage the SAME
// bytes per K-step:
//    bf16:  a K-step is 32 k, eight v_mfma_f32_32x32x16_bf16 per wave                       (the kernel the product runs)
//    fp8 :  a K-step is 64 k, four v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3, e8m0 block scales of 1.0) per wave
// Same MFMA cycles and the same staged bytes per K-step, twice the k: the fp8 build's time for a given (M, N, K) against the bf16 build's
// is the kernel-level gain of 1-byte storage (operand rounding, quantisation passes and scale traffic NOT included: an upper bound).
// Timing only (random operands, scale bytes 127 = 1.0; results are not checked against a reference).
//    hipcc --offload-arch=gfx950 -O3 tools/fp8_probe.hip -o /tmp/fp8_probe && /tmp/fp8_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
constexpr unsigned OOB = 0x80000000u;
constexpr int SA = 3, SB = 2, NI = 4, BM = 128, BN = 128, ASTAGE = BM * 64, BSTAGE = BN * 64;

template <bool F8>
__global__ __launch_bounds__(256, 2) void gemm_ring(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B, __bf16* __restrict__ C,
                                                    int M, int N, int kb /* bytes per row of A and B */) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gn = N / BN;
  const int bm = blockIdx.x / gn, bn = blockIdx.x - bm * gn;
  const int m0 = bm * BM;
  const int kiters = kb >> 6;
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)m0 * kb), 0, (int)((size_t)(M - m0) * kb > 0x7FFFFFF0u ? 0x7FFFFFF0u : (size_t)(M - m0) * kb), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (int)((size_t)N * kb), 0x00020000);
  const bool loads_a = wave < 2;
  unsigned voff[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int j = 4 * (wave & 1) + e;
    const int row = 16 * j + (lane >> 2), cp = lane & 3, c = cp ^ ((row >> 2) & 3);
    if (loads_a) voff[e] = m0 + row < M ? (unsigned)(row * kb + c * 16) : OOB;
    else voff[e] = (unsigned)((bn * BN + row) * kb + c * 16);
  }
  const __amdgpu_buffer_rsrc_t my_rs = loads_a ? a_rs : b_rs;
  const int my_dst = loads_a ? 4 * (wave & 1) * 1024 : SA * ASTAGE + 4 * (wave & 1) * 1024;
  int k_done = 0;
  auto issue = [&]() {
    const bool live = k_done < kiters;
    unsigned char* st = smem + (loads_a ? (k_done % SA) * ASTAGE : (k_done % SB) * BSTAGE) + my_dst;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(my_rs, (lds_void*)(st + e * 1024), 16, (int)(live ? voff[e] : OOB), (int)(live ? k_done * 64 : 0), 0, 0);
    ++k_done;
  };
  const int kh = lane >> 5;
  // bf16: MFMA k-block q (16 k) of the K-step, lane half kh: chunk 2 q + kh.  fp8: lane half kh owns bytes [32 kh, 32 kh + 32): chunks 2 kh, 2 kh + 1.
  int a_rd[2], b_rd[NI][2];
  {
    const int ar = wave * 32 + (lane & 31);
#pragma unroll
    for (int q = 0; q < 2; ++q) a_rd[q] = ar * 64 + ((((F8 ? 2 * kh + q : 2 * q + kh)) ^ ((ar >> 2) & 3)) << 4);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int row = ni * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 2; ++q) b_rd[ni][q] = SA * ASTAGE + row * 64 + ((((F8 ? 2 * kh + q : 2 * q + kh)) ^ ((row >> 2) & 3)) << 4);
    }
  }
  f32x16 acc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
  for (int s = 0; s < (loads_a ? SA : SB) - 1; ++s) issue();
  for (int it = 0; it < kiters; ++it) {
    if (loads_a) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((SA - 2) * 4) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((SB - 2) * 4) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue();
    const unsigned char* st = smem + (it % SA) * ASTAGE;
    const unsigned char* sb_ = smem + (it % SB) * BSTAGE;
    v4i a0 = *reinterpret_cast<const v4i*>(st + a_rd[0]), a1 = *reinterpret_cast<const v4i*>(st + a_rd[1]);
    v4i b0[NI], b1[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { b0[ni] = *reinterpret_cast<const v4i*>(sb_ + b_rd[ni][0]); b1[ni] = *reinterpret_cast<const v4i*>(sb_ + b_rd[ni][1]); }
    if constexpr (F8) {
      const v8i af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const v8i bf = {b0[ni][0], b0[ni][1], b0[ni][2], b0[ni][3], b1[ni][0], b1[ni][1], b1[ni][2], b1[ni][3]};
        acc[ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af, bf, acc[ni], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a0), __builtin_bit_cast(bf16x8_t, b0[ni]), acc[ni], 0, 0, 0);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a1), __builtin_bit_cast(bf16x8_t, b1[ni]), acc[ni], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wave * 32 + 4 * kh + (r & 3) + 8 * (r >> 2);
    if (m >= M) continue;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) C[(size_t)m * N + bn * BN + ni * 32 + (lane & 31)] = (__bf16)acc[ni][r];
  }
}

template <bool F8>
static double run(const uint8_t* A, const uint8_t* B, __bf16* C, int M, int N, int K) {
  const int kb = F8 ? K : 2 * K;
  const size_t lds = (size_t)SA * ASTAGE + (size_t)SB * BSTAGE;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring<F8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = ((M + BM - 1) / BM) * (N / BN);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(gemm_ring<F8>, dim3(grid), dim3(256), lds, 0, A, B, C, M, N, kb);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(gemm_ring<F8>, dim3(grid), dim3(256), lds, 0, A, B, C, M, N, kb);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms / 10 < best) best = ms / 10;
  }
  return best;
}

int main() {
  struct Sh { int M, N, K; const char* what; };
  const Sh shapes[] = {{43264, 512, 2304, "256->512 3x3 @26 (K = 9 x 256)"}, {173056, 256, 1152, "128->256 3x3 @52"}, {10816, 1024, 4608, "512->1024 3x3 @13"},
                       {173056, 512, 4608, "512->512 3x3 @52"}, {173056, 512, 1024, "1024->512 1x1 @52"}, {173056, 512, 512, "512->512 1x1 @52"},
                       {43264, 256, 512, "512->256 1x1 @26"}, {10816, 512, 1024, "1024->512 1x1 @13"}, {173056, 128, 256, "256->128 1x1 @52"}};
  size_t amax = 0, bmax = 0, cmax = 0;
  for (const Sh& s : shapes) {
    if ((size_t)s.M * s.K * 2 > amax) amax = (size_t)s.M * s.K * 2;
    if ((size_t)s.N * s.K * 2 > bmax) bmax = (size_t)s.N * s.K * 2;
    if ((size_t)s.M * s.N * 2 > cmax) cmax = (size_t)s.M * s.N * 2;
  }
  uint8_t *A, *B; __bf16* C;
  (void)hipMalloc(&A, amax + 4096); (void)hipMalloc(&B, bmax + 4096); (void)hipMalloc(&C, cmax);
  // random bytes that are finite small numbers in both readings: bf16 high bytes 0x3B..0x3F / 0xBB..0xBF, e4m3 exponents below the top
  std::vector<uint8_t> h(amax > bmax ? amax : bmax);
  uint32_t st = 12345u;
  for (size_t i = 0; i < h.size(); ++i) { st = st * 1664525u + 1013904223u; uint8_t v = (uint8_t)(st >> 24); h[i] = (i & 1) ? (uint8_t)((v & 0x80) | 0x38 | (v & 0x07)) : (uint8_t)(v & 0xB7); }
  (void)hipMemcpy(A, h.data(), amax, hipMemcpyHostToDevice); (void)hipMemcpy(B, h.data(), bmax, hipMemcpyHostToDevice);
  printf("%-34s %10s %10s %8s\n", "shape (M x N x K)", "bf16 ms", "fp8 ms", "ratio");
  for (const Sh& s : shapes) {
    const double tb = run<false>(A, B, C, s.M, s.N, s.K), tf = run<true>(A, B, C, s.M, s.N, s.K);
    const double gf = 2.0 * s.M * s.N * (double)s.K / 1e9;
    printf("%-34s %7.3f (%5.0f TF/s) %7.3f (%5.0f TF/s) %5.2fx\n", s.what, tb, gf / tb, tf, gf / tf, tb / tf);
  }
  return 0;
}
