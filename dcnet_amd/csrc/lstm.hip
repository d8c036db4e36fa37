// Persistent bidirectional LSTM recurrence (model/DCNet_model.py:134-137,172-183: nn.LSTM(512, 512, 1, batch_first,
// bidirectional) over a packed batch).  The input projections of all time steps are two plain GEMMs outside; this file
// runs the 2 x L dependent steps as ONE launch per pass instead of 2 x L x (GEMM + cell) launches.
//
// Grid: (H/8 workgroups) x (2 directions), 256 threads.  A workgroup owns 8 hidden units of one direction, wave w the
// units 2w, 2w+1 (their 4 gate rows each).  Lanes are BATCH ROWS (n = lane, lane+64, ...): every thread computes, for
// its row, the 8 gate pre-activations of its wave's two units as 8 running dot products over the 512 previous hidden
// values, so there is no cross-lane reduction; the recurrent weights of the 8 units (32 rows x 512 = 64 KB) live in LDS
// for all steps and are read as wave-uniform (broadcast) ds_read_b128.  The cell (sigmoid/tanh, packed-sequence masking)
// runs in the same thread, the cell state stays in registers for the whole sequence.
// Between steps the workgroups of a direction exchange h_t through global memory: plain stores, every wave drains
// (s_waitcnt vmcnt(0)), workgroup barrier, one lane's agent-scope release fence + counter add; the next step's readers
// poll the counter relaxed, then one agent-scope acquire, workgroup barrier, plain loads (cdna_hip_programming.md §6
// Guideline 16).  The counters are zeroed by a memset node ahead of every launch; every spin is bounded and reports
// through an error word.  All 2*H/8 = 128 workgroups must be resident together: 64 KB of LDS and 256 threads each, on 256 CUs.
// Backward: the same structure with the roles of W_hh transposed (a workgroup owns 8 COLUMNS: 8 x 2048 floats in LDS),
// dgates of step s+1 exchanged between the workgroups; dW_hh, dW_ih, dx are plain GEMMs over all (row, time) pairs after.
// Roofline: latency (2 x 20 dependent steps of ~10 us); 5.4 GFLOP forward at N = 64.
#include "common.h"

namespace {

constexpr int LS_H = 512;         // hidden size the kernels are built for
constexpr int LS_UNITS = 8;       // hidden units per workgroup
constexpr int LS_MAXC = 8;        // row chunks of 64: batch rows <= 512
constexpr unsigned LS_SPIN_LIMIT = 1u << 24;

typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// all workgroups of a direction have arrived `target` times.  Returns false on timeout (uniform over the workgroup).
__device__ __forceinline__ bool grid_wait(unsigned* counter, unsigned target, unsigned* err, volatile int* flag_s) {
  if (threadIdx.x == 0) {
    unsigned spins = 0; int ok = 1;
    while (__hip_atomic_load((gu32*)counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > LS_SPIN_LIMIT) { ok = 0; __hip_atomic_store((gu32*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *flag_s = ok;
  }
  __syncthreads();
  return *flag_s != 0;
}

__device__ __forceinline__ void grid_arrive(unsigned* counter) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (ROCm 7.2 can drop the fence's own wait)
    __hip_atomic_fetch_add((gu32*)counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// xg   [2][N][L][4H]  input projections x.W_ih^T + b_ih (gate order i,f,g,o)
// whh  [2][4H][H], bhh [2][4H], lens [N] (may be null)
// out  [N][L][2H] (zeros where t >= len), hprev [2][N][L][H] = h BEFORE time t (the t-first slice must be zeroed by the caller),
// cprev [2][N][L][H] = c before time t, acts [2][N][L][5H] = i,f,g,o,tanh(c_t)
__global__ __launch_bounds__(256) void bilstm_fwd_kernel(const float* __restrict__ xg, const float* __restrict__ whh0,
                                                         const float* __restrict__ whh1, const float* __restrict__ bhh0,
                                                         const float* __restrict__ bhh1, const int64_t* __restrict__ lens,
                                                         float* __restrict__ out, float* hprev, float* __restrict__ cprev,
                                                         float* __restrict__ acts, unsigned* sync, int N, int L) {
  extern __shared__ __attribute__((aligned(16))) float Wl[];      // [32][H]: row w*8 + q*2 + j = gate q of unit k0 + 2w + j
  constexpr int H = LS_H;
  int* flag_p = reinterpret_cast<int*>(Wl + 32 * H);              // (no static LDS: it would misalign the dynamic base, Guideline 17)
  const int d = blockIdx.y, k0 = blockIdx.x * LS_UNITS, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int G = gridDim.x;
  unsigned* counter = sync + d * 4;                                // [dir][4] (16-byte lines of their own), error word at sync[8]
  const float* W = d == 0 ? whh0 : whh1;
  const float* bhh = d == 0 ? bhh0 : bhh1;
  for (int i = tid; i < 32 * H / 4; i += 256) {
    const int rr = i / (H / 4), c4 = i - rr * (H / 4);
    const int ww = rr >> 3, q = (rr >> 1) & 3, j = rr & 1;
    reinterpret_cast<f32x4*>(Wl)[i] = *reinterpret_cast<const f32x4*>(W + ((size_t)q * H + k0 + 2 * ww + j) * H + c4 * 4);
  }
  __syncthreads();
  float c_state[LS_MAXC][2];
#pragma unroll
  for (int ch = 0; ch < LS_MAXC; ++ch) c_state[ch][0] = c_state[ch][1] = 0.f;
  const float* wrow = Wl + (size_t)w * 8 * H;
  const int u0 = k0 + 2 * w;
  float bias[8];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int j = 0; j < 2; ++j) bias[q * 2 + j] = bhh[q * H + u0 + j];
  for (int s = 0; s < L; ++s) {
    const int t = d == 0 ? s : L - 1 - s;
    const int tn = d == 0 ? t + 1 : t - 1;
    if (s > 0 && !grid_wait(counter, (unsigned)(s * G), sync + 8, flag_p)) return;
#pragma unroll
    for (int ch = 0; ch < LS_MAXC; ++ch) {
      const int n = lane + 64 * ch;
      if (n >= N) break;
      const size_t nt = (size_t)n * L + t;
      const float* xrow = xg + ((size_t)d * N * L + nt) * 4 * H;
      float acc[8];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[q * 2 + j] = xrow[q * H + u0 + j] + bias[q * 2 + j];
      const float* hrow = hprev + ((size_t)d * N * L + nt) * H;
      float hp[2] = {0.f, 0.f};
      if (s > 0) {
        for (int k = 0; k < H; k += 4) {
          const f32x4 hv = *reinterpret_cast<const f32x4*>(hrow + k);
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + r * H + k);
            acc[r] = fmaf(hv[0], wv[0], fmaf(hv[1], wv[1], fmaf(hv[2], wv[2], fmaf(hv[3], wv[3], acc[r]))));
          }
        }
        hp[0] = hrow[u0]; hp[1] = hrow[u0 + 1];
      }
      const bool live = lens == nullptr || t < lens[n];
      float* a = acts + ((size_t)d * N * L + nt) * 5 * H;
      float* cp = cprev + ((size_t)d * N * L + nt) * H;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float ig = sigmoidf_(acc[j]), fg = sigmoidf_(acc[2 + j]), gg = tanhf(acc[4 + j]), og = sigmoidf_(acc[6 + j]);
        const float cprev_v = c_state[ch][j];
        const float c = fg * cprev_v + ig * gg, tc = tanhf(c), h = og * tc;
        const int u = u0 + j;
        a[u] = ig; a[H + u] = fg; a[2 * H + u] = gg; a[3 * H + u] = og; a[4 * H + u] = tc;
        cp[u] = cprev_v;
        c_state[ch][j] = live ? c : cprev_v;
        const float hn = live ? h : hp[j];
        out[nt * 2 * H + (size_t)d * H + u] = live ? h : 0.f;
        if (tn >= 0 && tn < L) hprev[((size_t)d * N * L + (size_t)n * L + tn) * H + u] = hn;
      }
    }
    if (s + 1 < L) grid_arrive(counter);
  }
}

// dout [N][L][2H]; dxg [2][N][L][4H] receives the gate gradients (also the exchange buffer between the workgroups)
__global__ __launch_bounds__(256) void bilstm_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ whh0,
                                                         const float* __restrict__ whh1, const float* __restrict__ acts, const float* __restrict__ cprev,
                                                         const int64_t* __restrict__ lens, float* dxg, unsigned* sync, int N, int L) {
  extern __shared__ __attribute__((aligned(16))) float Wt[];      // [8 units][4H]: Wt[u][r] = W_hh[r][k0 + u]
  constexpr int H = LS_H;
  int* flag_p = reinterpret_cast<int*>(Wt + LS_UNITS * 4 * H);
  const int d = blockIdx.y, k0 = blockIdx.x * LS_UNITS, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int G = gridDim.x;
  unsigned* counter = sync + d * 4;
  const float* W = d == 0 ? whh0 : whh1;
  for (int i = tid; i < 4 * H * LS_UNITS; i += 256) {
    const int r = i / LS_UNITS, u = i - r * LS_UNITS;          // 32-byte pieces of consecutive rows
    Wt[(size_t)u * 4 * H + r] = W[(size_t)r * H + k0 + u];
  }
  __syncthreads();
  float dc_next[LS_MAXC][2], dh_pass[LS_MAXC][2];
#pragma unroll
  for (int ch = 0; ch < LS_MAXC; ++ch) dc_next[ch][0] = dc_next[ch][1] = dh_pass[ch][0] = dh_pass[ch][1] = 0.f;
  const int u0 = k0 + 2 * w;
  const float* w0 = Wt + (size_t)(2 * w) * 4 * H;
  const float* w1 = w0 + 4 * H;
  for (int s = L - 1; s >= 0; --s) {
    const int t = d == 0 ? s : L - 1 - s;
    const int tl = d == 0 ? t + 1 : t - 1;                        // the time processed one step LATER in the forward
    if (s < L - 1 && !grid_wait(counter, (unsigned)((L - 1 - s) * G), sync + 8, flag_p)) return;
#pragma unroll
    for (int ch = 0; ch < LS_MAXC; ++ch) {
      const int n = lane + 64 * ch;
      if (n >= N) break;
      const size_t nt = (size_t)n * L + t;
      float dhr[2] = {dh_pass[ch][0], dh_pass[ch][1]};
      if (s < L - 1) {
        const float* g = dxg + ((size_t)d * N * L + (size_t)n * L + tl) * 4 * H;
        float a0 = 0.f, a1 = 0.f;
        for (int r = 0; r < 4 * H; r += 4) {
          const f32x4 gv = *reinterpret_cast<const f32x4*>(g + r);
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(w0 + r), x1 = *reinterpret_cast<const f32x4*>(w1 + r);
          a0 = fmaf(gv[0], x0[0], fmaf(gv[1], x0[1], fmaf(gv[2], x0[2], fmaf(gv[3], x0[3], a0))));
          a1 = fmaf(gv[0], x1[0], fmaf(gv[1], x1[1], fmaf(gv[2], x1[2], fmaf(gv[3], x1[3], a1))));
        }
        dhr[0] += a0; dhr[1] += a1;
      }
      const bool live = lens == nullptr || t < lens[n];
      const float* a = acts + ((size_t)d * N * L + nt) * 5 * H;
      const float* cp = cprev + ((size_t)d * N * L + nt) * H;
      float* dg = dxg + ((size_t)d * N * L + nt) * 4 * H;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int u = u0 + j;
        if (!live) {               // the state was copied through: gradients pass straight to the earlier step
          dg[u] = 0.f; dg[H + u] = 0.f; dg[2 * H + u] = 0.f; dg[3 * H + u] = 0.f;
          dh_pass[ch][j] = dhr[j];
          continue;
        }
        const float ig = a[u], fg = a[H + u], gg = a[2 * H + u], og = a[3 * H + u], tc = a[4 * H + u];
        const float dh = dout[nt * 2 * H + (size_t)d * H + u] + dhr[j];
        const float dc = dc_next[ch][j] + dh * og * (1.f - tc * tc);
        dg[u] = dc * gg * ig * (1.f - ig);
        dg[H + u] = dc * cp[u] * fg * (1.f - fg);
        dg[2 * H + u] = dc * ig * (1.f - gg * gg);
        dg[3 * H + u] = dh * tc * og * (1.f - og);
        dc_next[ch][j] = dc * fg;
        dh_pass[ch][j] = 0.f;
      }
    }
    if (s > 0) grid_arrive(counter);
  }
}

}  // namespace

extern "C" int64_t dcn_bilstm_sync_bytes(void) { return 64; }

namespace {
// All 128 workgroups of a pass must be resident at once (they hand h_t to each other through global memory): checked once per
// process against the device's occupancy answer for this kernel and LDS size, so that a partitioned / smaller GPU fails with a
// message instead of spinning every hand-off to its timeout (round-2 advisor finding).  (-1: the query itself failed.)
template <typename K>
int resident_blocks(K kernel, size_t lds) {
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds) != hipSuccess) return -1;
  return per_cu * prop.multiProcessorCount;
}
}  // namespace

// sync: 64 bytes of device memory (zeroed here by a memset node on the stream before the launch).
extern "C" int dcn_bilstm_fwd(const float* xg, const float* whh_fwd, const float* whh_rev, const float* bhh_fwd, const float* bhh_rev,
                              const int64_t* lens, float* out, float* hprev, float* cprev, float* acts, void* sync,
                              int n, int l, int hidden, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DCN_CHECK_ARG(xg && whh_fwd && whh_rev && bhh_fwd && bhh_rev && out && hprev && cprev && acts && sync, "bilstm_fwd: null pointer");
  DCN_CHECK_ARG(hidden == LS_H, "bilstm_fwd: hidden size %d (built for %d)", hidden, LS_H);
  DCN_CHECK_ARG(n > 0 && n <= 64 * LS_MAXC && l > 0, "bilstm_fwd: %d rows (1..%d), %d steps", n, 64 * LS_MAXC, l);
  if (hipMemsetAsync(sync, 0, 32, stream) != hipSuccess) { dcn_set_error("bilstm_fwd: memset failed"); return DCN_ERR_LAUNCH; }   // the counters; the error word sync[8] is sticky
  const size_t lds = (size_t)32 * LS_H * sizeof(float) + 16;
  static DcnPerDeviceFlag attr_once;
  static int resident = 0;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    resident = resident_blocks(bilstm_fwd_kernel, lds);
  }
  DCN_CHECK_ARG(resident < 0 || resident >= 2 * (LS_H / LS_UNITS), "bilstm_fwd: the device keeps %d workgroups of this kernel resident, the "
                "persistent recurrence needs all %d at once (use the per-step dcn_lstm_cell_* path)", resident, 2 * (LS_H / LS_UNITS));
  hipLaunchKernelGGL(bilstm_fwd_kernel, dim3(LS_H / LS_UNITS, 2), dim3(256), lds, stream, xg, whh_fwd, whh_rev, bhh_fwd, bhh_rev, lens, out, hprev, cprev, acts,
                     (unsigned*)sync, n, l);
  DCN_CHECK_LAUNCH("bilstm_fwd");
  return DCN_OK;
}

extern "C" int dcn_bilstm_bwd(const float* dout, const float* whh_fwd, const float* whh_rev, const float* acts, const float* cprev,
                              const int64_t* lens, float* dxg, void* sync, int n, int l, int hidden, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  DCN_CHECK_ARG(dout && whh_fwd && whh_rev && acts && cprev && dxg && sync, "bilstm_bwd: null pointer");
  DCN_CHECK_ARG(hidden == LS_H, "bilstm_bwd: hidden size %d (built for %d)", hidden, LS_H);
  DCN_CHECK_ARG(n > 0 && n <= 64 * LS_MAXC && l > 0, "bilstm_bwd: %d rows (1..%d), %d steps", n, 64 * LS_MAXC, l);
  if (hipMemsetAsync(sync, 0, 32, stream) != hipSuccess) { dcn_set_error("bilstm_bwd: memset failed"); return DCN_ERR_LAUNCH; }
  const size_t lds = (size_t)LS_UNITS * 4 * LS_H * sizeof(float) + 16;
  static DcnPerDeviceFlag attr_once;
  static int resident = 0;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bilstm_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    resident = resident_blocks(bilstm_bwd_kernel, lds);
  }
  DCN_CHECK_ARG(resident < 0 || resident >= 2 * (LS_H / LS_UNITS), "bilstm_bwd: the device keeps %d workgroups of this kernel resident, the "
                "persistent recurrence needs all %d at once", resident, 2 * (LS_H / LS_UNITS));
  hipLaunchKernelGGL(bilstm_bwd_kernel, dim3(LS_H / LS_UNITS, 2), dim3(256), lds, stream, dout, whh_fwd, whh_rev, acts, cprev, lens, dxg,
                     (unsigned*)sync, n, l);
  DCN_CHECK_LAUNCH("bilstm_bwd");
  return DCN_OK;
}
