// Filter-bank preparation for a whole network in three launches per step.
//
// Every convolution of a training step needs its OIHW parameter as (a) the OHWI bank of the forward GEMM, (b) that bank
// cut into f16 pieces (dcn_presplit_f16 layout: 8 consecutive k -> [8 high | 8 low], igemm.hip BPRE), (c) the channel-
// transposed bank [Ci][T][Co] of the data gradient and (d) its split form, plus the bank's abs-max word for the power-of-two
// scale.  Done per layer this was ~420 tiny launches per step (layout transposes, abs-max, pre-split; 3 ms of kernel time
// and as much again in launch gaps).  Here a table of jobs (one per layer) is walked by two kernels: abs-max of every bank,
// then one LDS-tile pass that writes all four forms.  Roofline: HBM, ~0.2 GB per step — negligible; the point is launches.
#include "common.h"

namespace {

struct FilterJob {
  const float* src;     // OIHW [co][ci][T]
  float* ohwi;          // [co][T][ci] fp32, or null (T == 1: src already is this)
  float* ohwi_split;    // [co][T][ci] split, + 16 floats (scale at [numel]), or null
  float* t;             // [ci][T][co] fp32, or null
  float* t_split;       // [ci][T][co] split, + 16 floats, or null
  unsigned* amax;       // DCN_AMAX_WORDS words (zeroed by the caller's memset)
  void* ohwi_b16;       // [co][T][ci] bf16 (round to nearest even): the bank of the bf16-operand mode, or null
  void* t_b16;          // [ci][T][co] bf16, or null
  int co, ci, T;
  int blk0;             // first block of this job in xform_kernel
  int ablk0;            // first block of this job in amax_kernel
  int pad_;
};

__device__ __forceinline__ int find_job(const FilterJob* jobs, int njobs, int b, bool amax) {
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((amax ? jobs[mid].ablk0 : jobs[mid].blk0) <= b) lo = mid; else hi = mid - 1;
  }
  return lo;
}

constexpr int AMAX_CHUNK = 4096;

__global__ __launch_bounds__(256) void filters_amax_kernel(const FilterJob* __restrict__ jobs, int njobs) {
  __shared__ float red[4];
  const int j = find_job(jobs, njobs, blockIdx.x, true);
  const FilterJob job = jobs[j];
  const long long numel = (long long)job.co * job.ci * job.T;
  const long long base = (long long)(blockIdx.x - job.ablk0) * AMAX_CHUNK;
  float vmax = 0.f;
#pragma unroll
  for (int k = 0; k < AMAX_CHUNK / 1024; ++k) {
    const long long i = base + k * 1024 + threadIdx.x * 4;
    if (i + 3 < numel) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(job.src + i);
      vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    } else {
      for (long long q = i; q < numel; ++q) vmax = fmaxf(vmax, fabsf(job.src[q]));
    }
  }
  vmax = wave_max(vmax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
  __syncthreads();
  if (threadIdx.x == 0) amax_update(job.amax, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), blockIdx.x);
}

__device__ __forceinline__ float scale_of(unsigned bits) {      // = igemm.hip pow2_scale / conv.hip presplit_kernel
  const int be = (int)((bits >> 23) & 0xFF);
  int e = (be == 0 || be == 255) ? 0 : 14 - (be - 126);
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return __uint_as_float((unsigned)(e + 127) << 23);
}

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split_store8(float* dst, const float* v, float s) {
  f16x8_t h, l;
#pragma unroll
  for (int k = 0; k < 8; ++k) { const float t = v[k] * s; h[k] = (_Float16)t; l[k] = (_Float16)(t - (float)h[k]); }
  *reinterpret_cast<f16x8_t*>(dst) = h;
  *reinterpret_cast<f16x8_t*>(dst + 4) = l;
}

// one block = one (job, tap, 32 filters, 32 channels) tile
__global__ __launch_bounds__(256) void filters_xform_kernel(const FilterJob* __restrict__ jobs, int njobs) {
  __shared__ float tile[32][33];
  const int j = find_job(jobs, njobs, blockIdx.x, false);
  const FilterJob job = jobs[j];
  const int tci = job.ci / 32, tco = job.co / 32;
  int b = blockIdx.x - job.blk0;
  const int ci0 = (b % tci) * 32; b /= tci;
  const int co0 = (b % tco) * 32; b /= tco;
  const int tap = b;
  const float s = scale_of(amax_read(job.amax));
  const long long numel = (long long)job.co * job.ci * job.T;
  const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
  {
    const float* sp = job.src + ((long long)(co0 + r) * job.ci + ci0 + c4) * job.T + tap;
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[r][c4 + e] = sp[(long long)e * job.T];
  }
  __syncthreads();
  if (job.ohwi) {
    const f32x4 v = {tile[r][c4], tile[r][c4 + 1], tile[r][c4 + 2], tile[r][c4 + 3]};
    *reinterpret_cast<f32x4*>(job.ohwi + ((long long)(co0 + r) * job.T + tap) * job.ci + ci0 + c4) = v;
  }
  if (job.t) {
    const f32x4 v = {tile[c4][r], tile[c4 + 1][r], tile[c4 + 2][r], tile[c4 + 3][r]};
    *reinterpret_cast<f32x4*>(job.t + ((long long)(ci0 + r) * job.T + tap) * job.co + co0 + c4) = v;
  }
  if (threadIdx.x < 128) {
    const int rr = threadIdx.x >> 2, g = (threadIdx.x & 3) * 8;
    float v[8];
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    if (job.ohwi_b16) {
      bf16x8_t b;
#pragma unroll
      for (int e = 0; e < 8; ++e) b[e] = (__bf16)tile[rr][g + e];
      *reinterpret_cast<bf16x8_t*>(reinterpret_cast<__bf16*>(job.ohwi_b16) + ((long long)(co0 + rr) * job.T + tap) * job.ci + ci0 + g) = b;
    }
    if (job.t_b16) {
      bf16x8_t b;
#pragma unroll
      for (int e = 0; e < 8; ++e) b[e] = (__bf16)tile[g + e][rr];
      *reinterpret_cast<bf16x8_t*>(reinterpret_cast<__bf16*>(job.t_b16) + ((long long)(ci0 + rr) * job.T + tap) * job.co + co0 + g) = b;
    }
    if (job.ohwi_split) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[rr][g + e];
      split_store8(job.ohwi_split + ((long long)(co0 + rr) * job.T + tap) * job.ci + ci0 + g, v, s);
    }
    if (job.t_split) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[g + e][rr];
      split_store8(job.t_split + ((long long)(ci0 + rr) * job.T + tap) * job.co + co0 + g, v, s);
    }
  }
  if (blockIdx.x == job.blk0 && threadIdx.x == 0) {
    if (job.ohwi_split) job.ohwi_split[numel] = s;
    if (job.t_split) job.t_split[numel] = s;
  }
}

}  // namespace

extern "C" int dcn_filter_job_bytes(void) { return (int)sizeof(FilterJob); }

// jobs: device array of njobs FilterJob records (layout above; dcn_filter_job_bytes() each), blk0 / ablk0 ascending prefix
// sums of (T * co/32 * ci/32) and ceil(co*ci*T / 4096); amax_all: the contiguous region holding every job's abs-max words.
extern "C" int dcn_prepare_filters(const void* jobs, int njobs, int total_blocks, int total_amax_blocks,
                                   uint32_t* amax_all, int64_t amax_words, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DCN_CHECK_ARG(jobs && njobs > 0 && total_blocks > 0 && total_amax_blocks > 0 && amax_all && amax_words > 0, "prepare_filters: bad argument");
  if (hipMemsetAsync(amax_all, 0, (size_t)amax_words * 4, stream) != hipSuccess) { dcn_set_error("prepare_filters: memset failed"); return DCN_ERR_LAUNCH; }
  hipLaunchKernelGGL(filters_amax_kernel, dim3(total_amax_blocks), dim3(256), 0, stream, (const FilterJob*)jobs, njobs);
  DCN_CHECK_LAUNCH("filters_amax");
  hipLaunchKernelGGL(filters_xform_kernel, dim3(total_blocks), dim3(256), 0, stream, (const FilterJob*)jobs, njobs);
  DCN_CHECK_LAUNCH("filters_xform");
  return DCN_OK;
}
