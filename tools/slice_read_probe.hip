// Probe: how fast can 128-row workgroup tiles of an [M][K] fp32 tensor be streamed from HBM by LDS-DMA when a K-step takes a SLICE of W bytes
// of every row (conv1.hip: W = 64, rows 1 KB apart for K = 256) — against wider slices at the same bytes in flight?  No compute; each
// workgroup reads its 128 x K tile in K*4/W steps through a ring of `depth` stages (counted s_waitcnt vmcnt + s_barrier per step, as the
// kernels do), touches one dword per lane of every landed stage (ds_read) and writes one float.
//    hipcc --offload-arch=gfx950 -O3 tools/slice_read_probe.hip -o /tmp/slice_read_probe && /tmp/slice_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// W: bytes of a row per K-step; DEPTH: ring stages; 256 threads, 128 rows per workgroup.  A stage = 128 rows x W bytes = 128 W / 1024
// wave-instructions of 1 KiB, 4 waves -> PW = 32 W / 1024 pieces per wave and stage.
template <int W, int DEPTH>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ x, float* __restrict__ out, int M, int Kbytes) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int STAGE = 128 * W, PW = STAGE / 1024 / 4, CPR = W / 16;      // chunks (16 B) per row and step
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 128;
  const long long bytes = (long long)(M - m0) * Kbytes;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)x + (long long)m0 * Kbytes), 0,
                                                                     (int)(bytes > 0x7FFFFFF0LL ? 0x7FFFFFF0LL : bytes), 0x00020000);
  unsigned voff[PW];
#pragma unroll
  for (int e = 0; e < PW; ++e) {
    const int chunk = (wave * PW + e) * 64 + lane;      // 16-byte chunk of the stage, row-major [128 rows][CPR]
    const int row = chunk / CPR, c = chunk - row * CPR;
    voff[e] = (unsigned)(row * Kbytes + c * 16);
  }
  const int steps = Kbytes / W;
  int issued = 0;
  auto issue = [&]() {
    unsigned char* st = smem + (issued % DEPTH) * STAGE + wave * PW * 1024;
    const bool live = issued < steps;
#pragma unroll
    for (int e = 0; e < PW; ++e)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(st + e * 1024), 16, (int)(live ? voff[e] : 0x80000000u), (int)(live ? issued * W : 0), 0, 0);
    ++issued;
  };
  for (int s = 0; s < DEPTH - 1; ++s) issue();
  float acc = 0.f;
  for (int it = 0; it < steps; ++it) {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DEPTH - 2) * PW) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue();
    acc += *reinterpret_cast<const float*>(smem + (it % DEPTH) * STAGE + tid * 16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * 256 + tid] = acc;
}

template <int W, int DEPTH>
static void run(const float* x, float* out, int M, int K) {
  const size_t lds = (size_t)DEPTH * 128 * W;
  CK(hipFuncSetAttribute((const void*)probe<W, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = M / 128;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<W, DEPTH>), dim3(grid), dim3(256), lds, 0, x, out, M, K * 4);
  CK(hipEventRecord(e0));
  const int N = 20;
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL((probe<W, DEPTH>), dim3(grid), dim3(256), lds, 0, x, out, M, K * 4);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double per = ms / N;
  printf("  slice %4d B x depth %d (%3zu KB in flight per workgroup): %7.4f ms  %6.2f TB/s\n", W, DEPTH, (size_t)(DEPTH - 1) * 128 * W / 1024, per,
         (double)M * K * 4 / per * 1e-9);
}

int main() {
  // the step's 1x1 layers: 64 images of 52x52 / 104x104 / 26x26, K input channels
  const int shapes[][2] = {{64 * 52 * 52, 256}, {64 * 52 * 52, 512}, {64 * 104 * 104, 128}, {64 * 26 * 26, 512}, {64 * 26 * 26, 1024}};
  float* x; float* out;
  const size_t maxb = (size_t)64 * 104 * 104 * 128 * 4 * 2;
  CK(hipMalloc(&x, maxb)); CK(hipMalloc(&out, (size_t)8192 * 256 * 4));
  CK(hipMemset(x, 0x3c, maxb));
  for (auto& s : shapes) {
    const int M = s[0] / 128 * 128, K = s[1];
    printf("M = %d rows of %d fp32 (%d B), %.0f MB, %d workgroups\n", M, K, K * 4, (double)M * K * 4e-6, M / 128);
    run<64, 4>(x, out, M, K);     // conv1.hip: 3 + 1 stages of 8 KB
    run<64, 7>(x, out, M, K);
    run<128, 4>(x, out, M, K);
    if (K * 4 >= 256) run<256, 3>(x, out, M, K);
    if (K * 4 >= 512) { run<512, 2>(x, out, M, K); }
  }
  return 0;
}
