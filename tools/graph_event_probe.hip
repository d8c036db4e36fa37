// Probe: can HIP events recorded INSIDE a captured graph (hipEventRecord on a capturing stream -> event-record nodes) be read with
// hipEventElapsedTime after a replay?  bench.py wants per-kernel durations of the replayed step (real stream concurrency).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/graph_event_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(float* o, int n) { float v = o[threadIdx.x]; for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f; o[threadIdx.x] = v; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  float* d; CK(hipMalloc(&d, 1024)); CK(hipMemset(d, 0, 1024));
  hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
  hipEvent_t a, b, c, f, j; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); CK(hipEventCreate(&c)); CK(hipEventCreate(&f)); CK(hipEventCreate(&j));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  CK(hipEventRecord(a, s));
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 200000);
  CK(hipEventRecord(b, s));
  CK(hipEventRecord(f, s)); CK(hipStreamWaitEvent(s2, f, 0));          // fork
  hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, d + 64, 400000);
  CK(hipEventRecord(c, s2));
  CK(hipEventRecord(j, s2)); CK(hipStreamWaitEvent(s, j, 0));          // join
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int it = 0; it < 3; ++it) {
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    float t1 = -1, t2 = -1;
    hipError_t e1 = hipEventElapsedTime(&t1, a, b), e2 = hipEventElapsedTime(&t2, b, c);
    printf("replay %d: kernel 1 %.3f ms (%s), side-stream kernel %.3f ms (%s)\n", it, t1, hipGetErrorString(e1), t2, hipGetErrorString(e2));
  }
  return 0;
}
