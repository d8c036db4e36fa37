// Split-K slabs summed by the LAST-ARRIVING workgroup of a tile's split group (round 6).
//
// The weight-gradient kernels split K (the pixels) over workgroups; every split writes its partial tile into its own slab of the workspace
// and the slabs are summed in slab order, so the result does not depend on which workgroup ran when (no float atomics).  Until round 5 a
// second launch (reduce_slabs_kernel) did the sum: 88 launches per training step, 4.1 ms inside the replayed step against 1.8 alone,
// because each sat between two weight-gradient launches of the same queue.  Here the sum moves into the producing kernel:
//
//   every workgroup:  store its partial tile -> s_waitcnt vmcnt(0) -> workgroup barrier -> one lane: agent-scope release fence,
//                     atomic add on the tile's arrival counter
//   the workgroup that reads splits - 1 from the counter (the last to arrive; every other slab of the tile is complete and released):
//                     resets the counter for the next launch, agent-scope acquire fence, sums the tile's slabs 0 .. splits-1 IN SLAB ORDER
//                     with the very expression tree of reduce_slabs_kernel, writes dw.
//
// MEASURED AND OFF BY DEFAULT (dcn_set_tuning("Slabfold", KB) turns it on for launches whose last workgroup reads at most KB): the step's
// launches have FEW tiles and MANY splits (2-8 tiles x 30-250 splits on the 1x1 layers, 6-24 x 21-85 on most 3x3 layers — 5.6 GB of slabs per
// step, of which 17 % sit in launches where a tile's slabs are <= 2.5 MB), so the sum that 1024 workgroups of the separate pass share
// lands on 2-96 single workgroups at ~0.1 TB/s each, behind the kernel's last round.  Replayed step, same process: 91.15 ms with the
// separate pass, 91.70 folding <= 2.5 MB, 91.80 <= 4.2 MB, 100.06 folding everything (profiles/r06_experiments.md).
//
// Which workgroup does the sum varies from run to run; what it computes does not: bitwise the results of the separate pass
// (tests/test_ops_gpu.py::test_slab_fold_is_bitwise_the_separate_pass).  The counters are caller memory (DCN_SLAB_COUNTERS zeroed words per
// stream, see include/dcnet_hip.h): launches that share them are ordered on one stream exactly as launches that share the slab workspace.
#pragma once
#include "common.h"

struct SlabFold {
  unsigned* counters;      // one word per tile group; nullptr: the slabs are summed by reduce_slabs_kernel behind the launch
  float* dw;               // where the sum goes (the layout of one slab)
};

// host side: may this launch fold?  (groups = tile groups of the launch, each with `splits` workgroups; tile_bytes = one workgroup's part of a slab)
// ONE workgroup sums a tile's slabs, at what one CU draws from L2 / Infinity Cache (~0.1 TB/s): that beats a second launch while
// splits x tile stays in the low megabytes (many tiles, few splits: the wide 3x3 layers) and loses badly on the launches with 2-8 tiles
// and 30-250 splits (the 1x1 layers: 8-33 MB through one workgroup behind a 50-microsecond kernel).  limit_kb = dcn_set_tuning("Slabfold", KB).
static inline bool slab_fold_ok(const uint32_t* counters, int groups, int splits, long long tile_bytes, int limit_kb) {
  return limit_kb > 0 && counters != nullptr && splits > 1 && groups <= DCN_SLAB_COUNTERS && (long long)splits * tile_bytes <= (long long)limit_kb * 1024;
}

// All NT threads of the workgroup call this behind their slab stores.  The tile: rows [row0, row0 + nrows) of a [.][ld] matrix, in each row
// `nseg` segments of `ncols` floats starting at col0 + s * seg_stride (ncols, col0, seg_stride, ld multiples of 4: 16-byte vectors).
// `flag` = one int of LDS nobody else uses any more (the K loop is over).
template <int NT>
__device__ __forceinline__ void slab_fold(const SlabFold f, const int group, const int splits, const float* __restrict__ ws,
                                          const size_t slab_floats, const int row0, const int nrows, const int ld, const int col0,
                                          const int ncols, const int nseg, const int seg_stride, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's slab stores have left
  __syncthreads();                                           // ... and every wave's
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (ROCm 7.2 can drop the fence's own wait: lstm.hip)
    const unsigned prev = __hip_atomic_fetch_add(f.counters + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev + 1u == (unsigned)splits;
    if (last) __hip_atomic_store(f.counters + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // clean for the next launch
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");         // every wave: nothing stale of the other workgroups' slabs in L1 / this XCD's L2
  const int c4n = ncols >> 2;
  const int per_row = c4n * nseg;
  const int total = nrows * per_row;
  for (int i = threadIdx.x; i < total; i += NT) {
    const int r = i / per_row, rem = i - r * per_row;
    const int s = rem / c4n, c = rem - s * c4n;
    const size_t off = (size_t)(row0 + r) * ld + col0 + (size_t)s * seg_stride + 4 * c;
    const f32x4* __restrict__ w = reinterpret_cast<const f32x4*>(ws + off);
    const size_t n4 = slab_floats >> 2;
    f32x4 acc = w[0];                                        // the order of reduce_slabs_kernel (wgrad.hip), term for term
    int k = 1;
    for (; k + 8 <= splits; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = w[(size_t)(k + u) * n4];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; k < splits; ++k) acc += w[(size_t)k * n4];
    *reinterpret_cast<f32x4*>(f.dw + off) = acc;
  }
}
