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
#include <vector>
#include "common.h"

namespace {

struct MT {
  uint32_t* mt;   // 624 words
  uint32_t* pos;  // index word
  uint32_t next() {
    static const uint32_t mag01[2] = {0x0U, 0x9908b0dfU};
    if (*pos >= 624) {
      int kk;
      uint32_t y;
      for (kk = 0; kk < 624 - 397; kk++) { y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU); mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1U]; }
      for (; kk < 623; kk++) { y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU); mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1U]; }
      y = (mt[623] & 0x80000000U) | (mt[0] & 0x7fffffffU);
      mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 1U];
      *pos = 0;
    }
    uint32_t y = mt[(*pos)++];
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680U; y ^= (y << 15) & 0xefc60000U; y ^= (y >> 18);
    return y;
  }
  uint32_t randbelow(uint32_t n) {
    int bits = 0;
    for (uint32_t v = n; v; v >>= 1) ++bits;
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

// indices (positions in the population list) of random.sample(population of length n, k)
void sample_positions(MT& g, int n, int k, std::vector<int>& pool, std::vector<uint8_t>& seen, int* out) {
  if (n <= set_size(k)) {
    pool.resize(n);
    for (int i = 0; i < n; ++i) pool[i] = i;
    for (int i = 0; i < k; ++i) {
      const uint32_t j = g.randbelow((uint32_t)(n - i));
      out[i] = pool[j];
      pool[j] = pool[n - i - 1];
    }
  } else {
    seen.assign(n, 0);
    for (int i = 0; i < k; ++i) {
      uint32_t j = g.randbelow((uint32_t)n);
      while (seen[j]) j = g.randbelow((uint32_t)n);
      seen[j] = 1;
      out[i] = (int)j;
    }
  }
}

}  // namespace

// state: uint32[625].  kpos: [pairs][top_k] matched frame-2 positions.  out: [pairs][top_k][neg_n].
extern "C" int dcn_mt_sample_interframe(uint32_t* state, const int64_t* kpos, int pairs, int top_k, int hw, int neg_n,
                                        int64_t* out) {
  DCN_CHECK_ARG(state && out && pairs > 0 && top_k > 0 && hw > 1 && neg_n > 0 && neg_n <= hw - 1 && neg_n <= 64,
                "mt_sample_interframe: bad argument");
  MT g{state, state + 624};
  std::vector<int> pool; std::vector<uint8_t> seen; int tmp[64];
  for (int p = 0; p < pairs; ++p)
    for (int j = 0; j < top_k; ++j) {
      // kpos == NULL: emit the raw list positions; the caller maps them (pos >= kp ? pos+1 : pos) on the
      // device, so the draw — which does not depend on kp — can run while the GPU is still computing kp
      const int64_t kp = kpos ? kpos[(size_t)p * top_k + j] : (int64_t)hw;
      sample_positions(g, hw - 1, neg_n, pool, seen, tmp);       // list(range(hw)) with kp removed
      for (int e = 0; e < neg_n; ++e) out[((size_t)p * top_k + j) * neg_n + e] = tmp[e] < kp ? tmp[e] : tmp[e] + 1;
    }
  return DCN_OK;
}

// out: [n][rows][neg_n] = the draw for index == n-1 of every (ii, jj); all n draws advance the stream.
extern "C" int dcn_mt_sample_crossmodal(uint32_t* state, int n, int rows, int neg_n, int64_t* out) {
  DCN_CHECK_ARG(state && out && n > 0 && rows > 1 && neg_n > 0 && neg_n <= rows - 1 && neg_n <= 64,
                "mt_sample_crossmodal: bad argument");
  MT g{state, state + 624};
  std::vector<int> pool; std::vector<uint8_t> seen; int tmp[64];
  for (int ii = 0; ii < n; ++ii)
    for (int jj = 0; jj < rows; ++jj) {
      for (int index = 0; index < n; ++index)
        sample_positions(g, index == ii ? rows - 1 : rows, neg_n, pool, seen, tmp);
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
