// Host-side negative sampling for the two correspondence heads, bit-exact with the reference.
//
// The reference draws its negatives with Python's global `random.sample` inside nested Python
// loops: 30 calls per frame pair for the inter-frame head (model/DCNet_model.py:394-420) and
// N*N*HW0 calls for the cross-modal head (model/DCNet_model.py:62-96; only the last of every N
// draws is kept, but each one advances the generator).  At N=64, 416x416 that is 692 k
// interpreter-level calls per step — seconds of host time against a ~0.4 s GPU step.
// Here the same MT19937 stream is advanced natively: the caller passes random.getstate()[1]
// (624 words + position), gets the sampled indices, and writes the state back with
// random.setstate, so a run seeded with random.seed(k) reproduces the reference's index tensors
// exactly while the loop costs milliseconds.
//
// CPython semantics restated (Lib/random.py, 3.9-3.12):
//   sample(pop, k): setsize = 21 (+ 4**ceil(log(3k, 4)) if k > 5);
//     n <= setsize: pool algorithm (j = randbelow(n-i); take pool[j]; pool[j] = pool[n-i-1]);
//     else        : rejection into a set (j = randbelow(n) until unseen; take pop[j]).
//   randbelow(n): b = n.bit_length(); r = getrandbits(b) until r < n;  getrandbits(b<=32) = genrand32 >> (32-b).
#include <stdint.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include "common.h"

namespace {

// One regeneration of the 624-word state ("twist"), written as three dependence-free loops so that the
// compiler vectorises them: word kk reads kk+1 and kk+397 of the OLD block in the first loop and
// kk-227 of the NEW block in the others (227 apart: chunks of any vector width are independent).
inline void mt_twist(uint32_t* mt) {
  for (int kk = 0; kk < 227; ++kk) {
    const uint32_t y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
    mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((0U - (y & 1U)) & 0x9908b0dfU);
  }
  for (int kk = 227; kk < 454; ++kk) {
    const uint32_t y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
    mt[kk] = mt[kk - 227] ^ (y >> 1) ^ ((0U - (y & 1U)) & 0x9908b0dfU);
  }
  for (int kk = 454; kk < 623; ++kk) {
    const uint32_t y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
    mt[kk] = mt[kk - 227] ^ (y >> 1) ^ ((0U - (y & 1U)) & 0x9908b0dfU);
  }
  const uint32_t y = (mt[623] & 0x80000000U) | (mt[0] & 0x7fffffffU);
  mt[623] = mt[396] ^ (y >> 1) ^ ((0U - (y & 1U)) & 0x9908b0dfU);
}

inline uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680U; y ^= (y << 15) & 0xefc60000U; y ^= (y >> 18);
  return y;
}

inline int bit_length(uint32_t n) { int b = 0; for (; n; n >>= 1) ++b; return b; }

struct MT {
  uint32_t* mt;   // 624 words
  uint32_t* pos;  // index word
  uint32_t next() {
    if (*pos >= 624) { mt_twist(mt); *pos = 0; }
    return mt_temper(mt[(*pos)++]);
  }
  uint32_t randbelow(uint32_t n, int bits) {
    uint32_t r = next() >> (32 - bits);
    while (r >= n) r = next() >> (32 - bits);
    return r;
  }
};

int set_size(int k) {
  int s = 21;
  if (k > 5) s += (int)pow(4.0, ceil(log((double)k * 3.0) / log(4.0)));
  return s;
}

// indices (positions in the population list) of random.sample(population of length n, k).  `stamp` (>= n entries,
// zero before the first call) and `tick` replace CPython's `selected` set: an entry is taken when it holds this call's tick.
void sample_positions(MT& g, int n, int k, bool pool_path, std::vector<int>& pool, std::vector<uint32_t>& stamp, uint32_t& tick,
                      int* out) {
  if (pool_path) {
    pool.resize(n);
    for (int i = 0; i < n; ++i) pool[i] = i;
    for (int i = 0; i < k; ++i) {
      const uint32_t j = g.randbelow((uint32_t)(n - i), bit_length((uint32_t)(n - i)));
      out[i] = pool[j];
      pool[j] = pool[n - i - 1];
    }
  } else {
    const int bits = bit_length((uint32_t)n);
    if (++tick == 0) { std::fill(stamp.begin(), stamp.end(), 0u); tick = 1; }
    for (int i = 0; i < k; ++i) {
      uint32_t j = g.randbelow((uint32_t)n, bits);
      while (stamp[j] == tick) j = g.randbelow((uint32_t)n, bits);
      stamp[j] = tick;
      out[i] = (int)j;
    }
  }
}

// The rejection branch of random.sample on a stream of ACCEPTED draws.  Every randbelow(n) of that branch takes
// the top `bits` bits of one generator output and discards it when it is >= n; which outputs survive r < limit
// does not depend on what was drawn before, so a whole 624-word block is tempered, shifted and compacted at once
// (branch-free), and the sample loop — duplicates, the one-smaller population of the (ii == index) draw — walks
// the survivors.  idx[] remembers where each survivor sat in the block: the stream position after the last draw
// is the position behind the last survivor USED, exactly where CPython's generator stands.
struct Accepted {
  uint32_t* mt; uint32_t* pos;
  int shift; uint32_t limit;       // survivors: (y >> shift) < limit
  uint16_t val[624], idx[624];
  int cn = 0, ci = 0;
  void fill(int from) {
    int c = 0;
    for (int i = from; i < 624; ++i) {
      const uint32_t r = mt_temper(mt[i]) >> shift;
      val[c] = (uint16_t)r; idx[c] = (uint16_t)(i + 1);
      c += r < limit;
    }
    cn = c; ci = 0;
  }
  void begin() { if (*pos < 624) fill((int)*pos); }
  inline uint32_t next() {
    while (ci == cn) { mt_twist(mt); *pos = 0; fill(0); }
    return val[ci++];
  }
  void end() { if (ci > 0) *pos = idx[ci - 1]; }
};

}  // namespace

// state: uint32[625].  kpos: [pairs][top_k] matched frame-2 positions.  out: [pairs][top_k][neg_n].
extern "C" int dcn_mt_sample_interframe(uint32_t* state, const int64_t* kpos, int pairs, int top_k, int hw, int neg_n,
                                        int64_t* out) {
  DCN_CHECK_ARG(state && out && pairs > 0 && top_k > 0 && hw > 1 && neg_n > 0 && neg_n <= hw - 1 && neg_n <= 64,
                "mt_sample_interframe: bad argument");
  MT g{state, state + 624};
  std::vector<int> pool; std::vector<uint32_t> stamp((size_t)hw, 0u); uint32_t tick = 0; int tmp[64];
  const bool pool_path = hw - 1 <= set_size(neg_n);
  for (int p = 0; p < pairs; ++p)
    for (int j = 0; j < top_k; ++j) {
      // kpos == NULL: emit the raw list positions; the caller maps them (pos >= kp ? pos+1 : pos) on the
      // device, so the draw — which does not depend on kp — can run while the GPU is still computing kp
      const int64_t kp = kpos ? kpos[(size_t)p * top_k + j] : (int64_t)hw;
      sample_positions(g, hw - 1, neg_n, pool_path, pool, stamp, tick, tmp);   // list(range(hw)) with kp removed
      for (int e = 0; e < neg_n; ++e) out[((size_t)p * top_k + j) * neg_n + e] = tmp[e] < kp ? tmp[e] : tmp[e] + 1;
    }
  return DCN_OK;
}

// out: [n][rows][neg_n] = the draw for index == n-1 of every (ii, jj); all n draws advance the stream.
extern "C" int dcn_mt_sample_crossmodal(uint32_t* state, int n, int rows, int neg_n, int64_t* out) {
  DCN_CHECK_ARG(state && out && n > 0 && rows > 1 && neg_n > 0 && neg_n <= rows - 1 && neg_n <= 64,
                "mt_sample_crossmodal: bad argument");
  const int setsz = set_size(neg_n);
  const int bits = bit_length((uint32_t)rows);
  if (rows - 1 > setsz && bits == bit_length((uint32_t)(rows - 1)) && neg_n <= 8 && bits <= 16) {     // (survivors are kept as uint16_t)
    // both populations (rows, rows - 1 entries) take the rejection branch with the same bit count: the block form.
    // A draw is compared with the <= 7 earlier ones of its sample instead of a set; only the n-th sample of a
    // (ii, jj) is written, the others just advance the stream.
    Accepted a; a.mt = state; a.pos = state + 624; a.shift = 32 - bits; a.limit = (uint32_t)rows;
    a.begin();
    for (int ii = 0; ii < n; ++ii)
      for (int jj = 0; jj < rows; ++jj) {
        uint32_t sel[8];
        for (int index = 0; index < n; ++index) {
          const uint32_t lim = (uint32_t)(index == ii ? rows - 1 : rows);
          if (neg_n == 5 && a.ci + 5 <= a.cn) {
            // the common case (94 % at 169 rows): the next five survivors are distinct and inside the population
            const uint16_t* v = a.val + a.ci;
            const uint32_t v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3], v4 = v[4];
            const bool bad = (v0 == v1) | (v0 == v2) | (v0 == v3) | (v0 == v4) | (v1 == v2) | (v1 == v3) | (v1 == v4) |
                             (v2 == v3) | (v2 == v4) | (v3 == v4) |
                             (v0 >= lim) | (v1 >= lim) | (v2 >= lim) | (v3 >= lim) | (v4 >= lim);
            if (!bad) {
              a.ci += 5;
              if (index == n - 1) { sel[0] = v0; sel[1] = v1; sel[2] = v2; sel[3] = v3; sel[4] = v4; }
              continue;
            }
          }
          for (int i = 0; i < neg_n; ++i) {
            uint32_t r; bool again;
            do {
              r = a.next();
              again = r >= lim;
              for (int e = 0; e < i; ++e) again |= (sel[e] == r);
            } while (again);
            sel[i] = r;
          }
        }
        const bool removed = (ii == n - 1);
        for (int e = 0; e < neg_n; ++e)
          out[((size_t)ii * rows + jj) * neg_n + e] = (removed && sel[e] >= (uint32_t)jj) ? (int64_t)sel[e] + 1 : (int64_t)sel[e];
      }
    a.end();
    return DCN_OK;
  }
  MT g{state, state + 624};
  std::vector<int> pool; std::vector<uint32_t> stamp((size_t)rows, 0u); uint32_t tick = 0; int tmp[64];
  for (int ii = 0; ii < n; ++ii)
    for (int jj = 0; jj < rows; ++jj) {
      for (int index = 0; index < n; ++index) {
        const int pop = index == ii ? rows - 1 : rows;
        sample_positions(g, pop, neg_n, pop <= setsz, pool, stamp, tick, tmp);
      }
      const bool removed = (ii == n - 1);
      for (int e = 0; e < neg_n; ++e)
        out[((size_t)ii * rows + jj) * neg_n + e] = (removed && tmp[e] >= jj) ? tmp[e] + 1 : tmp[e];
    }
  return DCN_OK;
}

// Counting sort of the (ii, jj, m) -> position table by position: csr_src lists, for every position of the last image,
// the flat indices (ii*rows + jj)*neg_n + m that drew it, in ascending order (a deterministic backward of the gather).
extern "C" int dcn_mt_sample_crossmodal_csr(const int64_t* out, int n, int rows, int neg_n, int32_t* csr_off, int32_t* csr_src) {
  DCN_CHECK_ARG(out && csr_off && csr_src && n > 0 && rows > 0 && neg_n > 0, "mt_sample_crossmodal_csr: bad argument");
  const int64_t total = (int64_t)n * rows * neg_n;
  DCN_CHECK_ARG(total < (1LL << 31), "mt_sample_crossmodal_csr: %lld entries exceed 31 bits", (long long)total);
  for (int p = 0; p <= rows; ++p) csr_off[p] = 0;
  for (int64_t i = 0; i < total; ++i) {
    const int64_t p = out[i];
    DCN_CHECK_ARG(p >= 0 && p < rows, "mt_sample_crossmodal_csr: position out of range");
    ++csr_off[p + 1];
  }
  for (int p = 0; p < rows; ++p) csr_off[p + 1] += csr_off[p];
  std::vector<int32_t> cur(csr_off, csr_off + rows);
  for (int64_t i = 0; i < total; ++i) csr_src[cur[out[i]]++] = (int32_t)i;
  return DCN_OK;
}

// The three calls of one training forward as ONE call, timed inside (round 6).  The worker thread of the Python side needs the
// interpreter lock between two ctypes calls and to read its clock; while the main thread spins in Python that is up to one switch
// interval (5 ms) per acquisition — the "sampler_thread" figure of the round-5 bench line (40.6 ms against 10-16 ms of draws) was mostly
// that wait.  *seconds = what the draws cost this thread (CLOCK_THREAD_CPUTIME_ID: time it ran, not time it waited).
#include <time.h>
extern "C" int dcn_mt_sample_step(uint32_t* state, int n, int top_k, int hw, int neg_n, int neg_c, int64_t* k9, int64_t* k14,
                                  int32_t* csr_off, int32_t* csr_src, double* seconds) {
  timespec t0, t1;
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t0);
  int rc = dcn_mt_sample_interframe(state, nullptr, n / 2, top_k, hw, neg_n, k9);
  if (rc == DCN_OK) rc = dcn_mt_sample_crossmodal(state, n, hw, neg_c, k14);
  if (rc == DCN_OK) rc = dcn_mt_sample_crossmodal_csr(k14, n, hw, neg_c, csr_off, csr_src);
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t1);
  if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  return rc;
}
