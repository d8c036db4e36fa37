// Shared helpers for the libdcnet_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dcnet_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void dcn_set_error(const char* fmt, ...);

#define DCN_CHECK_ARG(cond, ...)                       \
  do {                                                 \
    if (!(cond)) {                                     \
      dcn_set_error(__VA_ARGS__);                      \
      return DCN_ERR_ARG;                              \
    }                                                  \
  } while (0)

#define DCN_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      dcn_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return DCN_ERR_LAUNCH;                                                \
    }                                                                       \
  } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Function attributes (the dynamic-LDS limit) and the CU count belong to a DEVICE, not to the process: launch-site caches are
// indexed by the calling thread's current device, so a process that drives a second GPU sets its limits there too.
#define DCN_MAX_DEVICES 32
static inline int dcn_device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
  return d % DCN_MAX_DEVICES;
}
struct DcnPerDeviceFlag {      // `static DcnPerDeviceFlag f; if (f.first()) { ...once per device... }`
  bool done[DCN_MAX_DEVICES] = {};
  bool first() { const int d = dcn_device_slot(); const bool was = done[d]; done[d] = true; return !was; }
};
struct DcnPerDeviceSize {      // `static DcnPerDeviceSize s; if (s.raise(lds)) { ...limit grows on this device... }`
  size_t v[DCN_MAX_DEVICES] = {};
  bool raise(size_t want) { size_t& c = v[dcn_device_slot()]; if (c >= want) return false; c = want; return true; }
};
// CUs of the current device (0 on a failed query), cached per device
static inline int dcn_device_cus() {
  static int cus[DCN_MAX_DEVICES] = {};
  const int d = dcn_device_slot();
  if (!cus[d]) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) != hipSuccess) return 0;
    cus[d] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus[d];
}

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous chunk of
// the logical tile order so that neighbouring tiles (which share operand panels) hit the
// same 4 MiB L2.  Bijective for any grid size.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// An abs-max "word" is DCN_AMAX_WORDS words (float bits of non-negative values) whose maximum is the tensor's abs-max:
// the waves of a producing kernel finish together, and thousands of atomics on ONE address serialise in L2 (measured
// +0.15 ms on a 0.06 ms streaming kernel), so each wave updates word (its index mod 64).  The stale-tolerant read in
// front skips the atomic once the word is large enough (max is idempotent).  Readers take the maximum over the words.
#define DCN_AMAX_WORDS 64
__device__ __forceinline__ void amax_update(unsigned* amax, float v, unsigned spread) {
  unsigned* w = amax + (spread & (DCN_AMAX_WORDS - 1));
  if (v > 0.f && __float_as_uint(v) > __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(w, __float_as_uint(v));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// One atomic per WORKGROUP (256 threads): the waves' maxima meet in LDS first.  `red` = 4 floats of shared memory.
__device__ __forceinline__ void amax_update_block(unsigned* amax, float vmax, float* red) {
  vmax = wave_max(vmax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
  __syncthreads();
  if (threadIdx.x == 0) amax_update(amax, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), blockIdx.x);
}
// the abs-max held by a DCN_AMAX_WORDS-word vector, as float bits (wave-uniform; every lane of the wave must call)
__device__ __forceinline__ unsigned amax_read(const unsigned* amax) {
  return __float_as_uint(wave_max(__uint_as_float(amax[threadIdx.x & (DCN_AMAX_WORDS - 1)])));
}
