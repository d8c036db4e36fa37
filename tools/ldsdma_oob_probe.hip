// Probe: what does `buffer_load_dwordx4 ... lds` leave in LDS for a lane whose offset is out of the descriptor's range?
// (conv1.hip relies on: in-range lanes land lane-linear; this asks whether out-of-range lanes write ZEROS or nothing.)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/ldsdma_oob_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const float* a, float* o, int nbytes) {
  __shared__ __attribute__((aligned(1024))) unsigned char sm[1024];
  f32x4 fill = {7.f, 7.f, 7.f, 7.f};
  *reinterpret_cast<f32x4*>(sm + threadIdx.x * 16) = fill;            // pre-fill: a dropped write leaves 7s
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, nbytes, 0x00020000);
  const unsigned voff = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16u;   // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)sm, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  *reinterpret_cast<f32x4*>(o + threadIdx.x * 4) = *reinterpret_cast<f32x4*>(sm + threadIdx.x * 16);
}
int main() {
  float *a, *o; float h[256], r[256];
  for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
  hipMalloc(&a, 1024); hipMalloc(&o, 1024);
  hipMemcpy(a, h, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, o, 1024);
  hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
  int zeros = 0, sevens = 0, ok = 0;
  for (int l = 0; l < 64; ++l) {
    if (l & 1) { if (r[4 * l] == 0.f && r[4 * l + 3] == 0.f) ++zeros; else if (r[4 * l] == 7.f) ++sevens; }
    else if (r[4 * l] == 100.f + 4 * l) ++ok;
  }
  printf("in-range lanes correct: %d/32; out-of-range lanes: %d wrote zeros, %d left the old LDS content (first odd lane: %g %g %g %g)\n",
         ok, zeros, sevens, r[4], r[5], r[6], r[7]);
  return 0;
}
