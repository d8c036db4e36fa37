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
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
