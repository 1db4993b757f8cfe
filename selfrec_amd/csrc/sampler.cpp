// (a-1) Pairwise BPR sampler: a bit-exact replay of what the reference does with CPython's
// global `random` stream in util/sampler.py:5-28, and of random.sample as used by
// data/augmentor.py:15-16,35.
//
// The algorithm is fixed by CPython (Lib/random.py + Modules/_randommodule.c, identical in
// 3.9-3.12 for the calls used here):
//   genrand_uint32        MT19937 (Matsumoto & Nishimura 2002 reference implementation)
//   getrandbits(k<=32)    genrand_uint32() >> (32-k)
//   _randbelow(n)         k = n.bit_length(); r = getrandbits(k); while r >= n: redraw
//   shuffle(x)            for i in reversed(range(1,len(x))): j=_randbelow(i+1); swap x[i],x[j]
//   choice(seq)           seq[_randbelow(len(seq))]
//   sample(range(n),k)    pool-based partial shuffle when n <= 21 + 4**ceil(log(3k,4)) (k>5),
//                         else set-based rejection
// The data-dependent rejection loop makes the stream inherently sequential, so this runs on
// a host core (tens of ns per draw) and is overlapped with the device step by the caller.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <unordered_set>
#include <vector>

#include "common.h"

namespace {

struct MT19937 {
  static constexpr int N = 624, M = 397;
  uint32_t mt[N];
  int pos = N + 1;

  void init_genrand(uint32_t s) {
    mt[0] = s;
    for (int i = 1; i < N; ++i) mt[i] = 1812433253U * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    pos = N;
  }
  void init_by_array(const uint32_t* key, size_t len) {
    init_genrand(19650218U);
    size_t i = 1, j = 0;
    size_t k = (N > len ? (size_t)N : len);
    for (; k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525U)) + key[j] + (uint32_t)j;
      ++i; ++j;
      if (i >= (size_t)N) { mt[0] = mt[N - 1]; i = 1; }
      if (j >= len) j = 0;
    }
    for (k = N - 1; k; --k) {
      mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941U)) - (uint32_t)i;
      ++i;
      if (i >= (size_t)N) { mt[0] = mt[N - 1]; i = 1; }
    }
    mt[0] = 0x80000000U;
  }
  void refill() {
    static const uint32_t mag01[2] = {0x0U, 0x9908b0dfU};
    int kk;
    uint32_t y;
    // Word kk reads words kk, kk + 1 (not yet rewritten) and kk + M resp. kk + M - N (rewritten >= N - M = 227 words
    // earlier): any vector width below 227 keeps the scalar order's values, which the compiler cannot see by itself.
#pragma clang loop vectorize(assume_safety)
    for (kk = 0; kk < N - M; ++kk) {
      y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
      mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((0U - (y & 1U)) & 0x9908b0dfU);
    }
#pragma clang loop vectorize(assume_safety)
    for (kk = N - M; kk < N - 1; ++kk) {
      y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
      mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((0U - (y & 1U)) & 0x9908b0dfU);
    }
    y = (mt[N - 1] & 0x80000000U) | (mt[0] & 0x7fffffffU);
    mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ mag01[y & 1U];
    pos = 0;
  }
  inline uint32_t next_u32() {
    if (pos >= N) refill();
    uint32_t y = mt[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
  }
  // the k-th output AFTER the next one, without drawing it (k < N - pos: outputs of the current block only)
  inline uint32_t peek_u32(int at) const {
    uint32_t y = mt[at];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
  }
  // n in [1, 2^32)
  inline uint32_t randbelow(uint32_t n) {
    const int shift = __builtin_clz(n);  // 32 - bit_length(n)
    uint32_t r = next_u32() >> shift;
    while (r >= n) r = next_u32() >> shift;
    return r;
  }
};

}  // namespace

struct srh_sampler {
  int64_t n_users = 0, n_items = 0, n_edges = 0;
  struct Edge { int32_t u, i; };
  std::vector<Edge> edges;                  // as given at create time (one cache line per edge, not one per array)
  std::vector<uint32_t> order;              // persistent order of training_data (edge numbers)
  std::vector<int64_t> row_ptr;             // user -> sorted positive items
  std::vector<int32_t> row_items;
  // Membership signature per user: a bitmap of >= 16 bits per positive item (256 bits minimum, a
  // few MB in total, cache resident).  A clear bit proves "not rated" without touching the user's
  // item list -- the common case by far; once the bitmap is as large as the catalogue it is exact.
  std::vector<uint64_t> sig;
  // per user, in ONE word (the sampler's inner loop misses the cache on every array it touches per pair: 5 -> 3 lines):
  //   info >> 8  = first 64-bit word of user u's bitmap;  info & 255 = log2(bits) of it, 255 = exact item bitmap
  std::vector<uint64_t> sig_info;
  std::vector<uint64_t> seen_u, seen_i;     // scratch bitmaps for the per-batch sorted unique ids
  // srh_sampler_epoch_segments: one scratch set per worker thread, kept between epochs (280 KB each at the Yelp2018 shape:
  // freed and re-mapped every epoch they would be munmap traffic beside the thread that feeds the GPU)
  struct SegScratch {
    std::vector<int32_t> pos_u, pos_i, fill, neg_only, un;
    std::vector<uint64_t> seen;
  };
  std::vector<SegScratch> seg_scratch;
  MT19937 rng;
  bool seeded = false;

  static inline uint32_t sig_bit_of(uint64_t info, int32_t item) {
    const uint32_t lg = (uint32_t)(info & 255);
    return lg == 255 ? (uint32_t)item : ((uint32_t)item * 0x9E3779B1U) >> (32 - lg);
  }
  inline uint32_t sig_bit(int32_t u, int32_t item) const { return sig_bit_of(sig_info[u], item); }
  inline bool rated(int32_t u, int32_t item) const {
    const uint64_t info = sig_info[u];
    const uint32_t b = sig_bit_of(info, item);
    if (!((sig[(size_t)(info >> 8) + (b >> 6)] >> (b & 63)) & 1ULL)) return false;
    if ((info & 255) == 255) return true;
    const int32_t* lo = row_items.data() + row_ptr[u];
    const int32_t* hi = row_items.data() + row_ptr[u + 1];
    return std::binary_search(lo, hi, item);
  }
};

extern "C" {

srh_status_t srh_sampler_create(srh_sampler_t** out, int64_t n_users, int64_t n_items,
                                int64_t n_edges, const int32_t* h_edge_user,
                                const int32_t* h_edge_item) {
  SRH_REQUIRE(out && ((h_edge_user && h_edge_item) || n_edges == 0), "sampler_create: null argument");
  SRH_REQUIRE(n_users > 0 && n_items > 0 && n_edges >= 0, "sampler_create: bad sizes");
  SRH_REQUIRE(n_items < (int64_t(1) << 31) && n_users < (int64_t(1) << 31) && n_edges < (int64_t(1) << 32),
              "sampler_create: sizes beyond 32-bit draws are not supported");
  srh_sampler* s = new (std::nothrow) srh_sampler();
  if (!s) { srh::set_error("sampler_create: out of memory"); return SRH_ERR_NOMEM; }
  s->n_users = n_users; s->n_items = n_items; s->n_edges = n_edges;
  s->edges.resize((size_t)n_edges);
  for (int64_t e = 0; e < n_edges; ++e) s->edges[(size_t)e] = {h_edge_user[e], h_edge_item[e]};
  s->order.resize(n_edges);
  for (int64_t e = 0; e < n_edges; ++e) s->order[e] = (uint32_t)e;
  s->row_ptr.assign(n_users + 1, 0);
  for (int64_t e = 0; e < n_edges; ++e) {
    int32_t u = s->edges[e].u, it = s->edges[e].i;
    if (u < 0 || u >= n_users || it < 0 || it >= n_items) {
      delete s;
      srh::set_error("sampler_create: edge %lld (%d,%d) out of range", (long long)e, u, it);
      return SRH_ERR_INVALID_ARG;
    }
    s->row_ptr[u + 1]++;
  }
  for (int64_t u = 0; u < n_users; ++u) s->row_ptr[u + 1] += s->row_ptr[u];
  s->row_items.resize(n_edges);
  std::vector<int64_t> fill(s->row_ptr.begin(), s->row_ptr.end() - 1);
  for (int64_t e = 0; e < n_edges; ++e) s->row_items[fill[s->edges[e].u]++] = s->edges[e].i;
  // the per-user work below (sorting a user's items, filling a user's bitmap) touches disjoint memory: user ranges of about
  // equal numbers of edges, one host thread each
  int n_thr = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())),
                                                         n_edges / (1 << 20)));
  if (const char* forced = std::getenv("SRH_SAMPLER_THREADS")) n_thr = std::max(1, std::min(64, std::atoi(forced)));   // (tests)
  std::vector<int64_t> cut((size_t)n_thr + 1, n_users);
  cut[0] = 0;
  for (int t = 1; t < n_thr; ++t)
    cut[(size_t)t] = std::lower_bound(s->row_ptr.begin(), s->row_ptr.end(), n_edges * t / n_thr) - s->row_ptr.begin();
  auto over_users = [&](auto&& body) {
    if (n_thr == 1) { body((int64_t)0, n_users); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < n_thr; ++t) pool.emplace_back([&, t] { body(cut[(size_t)t], std::min<int64_t>(cut[(size_t)t + 1], n_users)); });
    for (auto& th : pool) th.join();
  };
  over_users([&](int64_t u0, int64_t u1) {
    for (int64_t u = u0; u < u1; ++u)
      std::sort(s->row_items.begin() + s->row_ptr[u], s->row_items.begin() + s->row_ptr[u + 1]);
  });
  s->sig_info.resize(n_users);
  size_t words = 0;
  for (int64_t u = 0; u < n_users; ++u) {
    const int64_t deg = s->row_ptr[u + 1] - s->row_ptr[u];
    int lg = 8;
    while ((int64_t(1) << lg) < 16 * deg) ++lg;
    if ((int64_t(1) << lg) >= n_items) {                  // as large as the catalogue: make it exact
      s->sig_info[u] = ((uint64_t)words << 8) | 255u;
      words += (size_t)(n_items + 63) / 64;
    } else {
      s->sig_info[u] = ((uint64_t)words << 8) | (uint64_t)lg;
      words += (size_t)1 << (lg - 6);
    }
  }
  s->sig.assign(words, 0);
  over_users([&](int64_t u0, int64_t u1) {            // (by user, from the grouped item lists: a user's bitmap stays in cache)
    for (int64_t u = u0; u < u1; ++u) {
      const uint64_t info = s->sig_info[u];
      uint64_t* bits = s->sig.data() + (size_t)(info >> 8);
      for (int64_t p = s->row_ptr[u]; p < s->row_ptr[u + 1]; ++p) {
        const uint32_t b = srh_sampler::sig_bit_of(info, s->row_items[p]);
        bits[b >> 6] |= 1ULL << (b & 63);
      }
    }
  });
  // (two levels each: one bit per id, then one bit per 64-bit word of those -- sorted_unique)
  s->seen_u.assign((size_t)(n_users + 63) / 64 + ((size_t)(n_users + 63) / 64 + 63) / 64, 0);
  s->seen_i.assign((size_t)(n_items + 63) / 64 + ((size_t)(n_items + 63) / 64 + 63) / 64, 0);
  *out = s;
  return SRH_OK;
}

void srh_sampler_destroy(srh_sampler_t* s) { delete s; }

srh_status_t srh_sampler_set_state(srh_sampler_t* s, const uint32_t* h_mt624, int32_t pos) {
  SRH_REQUIRE(s && h_mt624, "sampler_set_state: null argument");
  SRH_REQUIRE(pos >= 0 && pos <= 624, "sampler_set_state: position %d out of [0,624]", pos);
  std::memcpy(s->rng.mt, h_mt624, sizeof(uint32_t) * 624);
  s->rng.pos = pos;
  s->seeded = true;
  return SRH_OK;
}

srh_status_t srh_sampler_get_state(const srh_sampler_t* s, uint32_t* h_mt624, int32_t* pos) {
  SRH_REQUIRE(s && h_mt624 && pos, "sampler_get_state: null argument");
  if (!s->seeded) { srh::set_error("sampler_get_state: generator was never seeded"); return SRH_ERR_STATE; }
  std::memcpy(h_mt624, s->rng.mt, sizeof(uint32_t) * 624);
  *pos = s->rng.pos;
  return SRH_OK;
}

srh_status_t srh_sampler_seed(srh_sampler_t* s, uint64_t seed) {
  SRH_REQUIRE(s, "sampler_seed: null handle");
  uint32_t key[2] = {(uint32_t)(seed & 0xffffffffU), (uint32_t)(seed >> 32)};
  s->rng.init_by_array(key, key[1] ? 2 : 1);
  s->seeded = true;
  return SRH_OK;
}

static srh_status_t require_seeded(const srh_sampler_t* s, const char* who) {
  if (!s) { srh::set_error("%s: null handle", who); return SRH_ERR_INVALID_ARG; }
  if (!s->seeded) {
    srh::set_error("%s: generator state not set (call srh_sampler_set_state or srh_sampler_seed)", who);
    return SRH_ERR_STATE;
  }
  return SRH_OK;
}

srh_status_t srh_sampler_shuffle(srh_sampler_t* s) {
  srh_status_t st = require_seeded(s, "sampler_shuffle");
  if (st) return st;
  // Fisher-Yates.  The draws do not depend on the data, so they run a few iterations ahead of
  // the swaps and their targets are prefetched: the random access into the (5 MB at Yelp
  // shape) order array is what costs, not the generator.
  uint32_t* x = s->order.data();
  constexpr int kAhead = 16;
  uint32_t js[kAhead];
  int64_t i = s->n_edges - 1;
  while (i >= 1) {
    const int n = (int)std::min<int64_t>(kAhead, i);
    for (int k = 0; k < n; ++k) {
      js[k] = s->rng.randbelow((uint32_t)(i - k + 1));
      __builtin_prefetch(x + js[k], 1);
    }
    for (int k = 0; k < n; ++k) std::swap(x[i - k], x[js[k]]);
    i -= n;
  }
  return SRH_OK;
}

srh_status_t srh_sampler_get_order(const srh_sampler_t* s, int64_t* h_perm) {
  SRH_REQUIRE(s && h_perm, "sampler_get_order: null argument");
  for (int64_t e = 0; e < s->n_edges; ++e) h_perm[e] = (int64_t)s->order[e];
  return SRH_OK;
}

static inline int64_t batch_into(srh_sampler_t* s, int64_t ptr, int64_t batch_size, int32_t n_negs,
                                 int32_t* u, int32_t* it, int32_t* neg) {
  const int64_t end = (ptr + batch_size < s->n_edges) ? ptr + batch_size : s->n_edges;
  const uint32_t n_items = (uint32_t)s->n_items;
  int64_t w = 0;
  constexpr int64_t kAhead = 32;
  // The membership test of a candidate reads one word of the user's signature: cache resident at the Yelp2018 shape, a miss
  // per draw once the signatures are tens of MB (1 M users).  The candidate of pair p + kSpec is known before it is drawn:
  // the generator's coming outputs can be LOOKED AT (peek_u32: the current block of 624 words, nothing is consumed), and unless
  // an earlier candidate is rejected as rated -- rare -- the pair will draw exactly those.  So a cursor runs kSpec pairs ahead,
  // replays randbelow on the peeked outputs and prefetches the word the real test will read.  Purely a hint: the draws
  // themselves are the loop below, unchanged; a rejection just restarts the cursor at the real position.
  constexpr int64_t kSpec = 8;
  const int shift = __builtin_clz(n_items);
  int64_t spec_p = ptr;              // next pair the cursor will look at
  int spec_pos = s->rng.pos;         // generator position that pair is expected to start at
  bool spec_ok = true;               // (false: ran into the end of the block; resumes after the refill)
  int prev_pos = s->rng.pos;
  auto speculate = [&](int64_t upto) {
    while (spec_ok && spec_p < upto && spec_p < end) {
      const int32_t un = s->edges[s->order[spec_p]].u;
      const uint64_t info = s->sig_info[un];
      int q = spec_pos;
      for (int32_t m = 0; m < n_negs; ++m) {
        uint32_t r;
        do {
          if (q >= MT19937::N) { spec_ok = false; return; }
          r = s->rng.peek_u32(q++) >> shift;
        } while (r >= n_items);
        const uint32_t b = srh_sampler::sig_bit_of(info, (int32_t)r);
        __builtin_prefetch(s->sig.data() + (size_t)(info >> 8) + (b >> 6));
      }
      spec_pos = q;
      ++spec_p;
    }
  };
  for (int64_t p = ptr; p < end; ++p) {
    if (p + kAhead < end) {                       // hide the two dependent random accesses
      const uint32_t ea = s->order[p + kAhead];
      __builtin_prefetch(s->edges.data() + ea);
    }
    if (p + kAhead / 2 < end) {
      const int32_t un = s->edges[s->order[p + kAhead / 2]].u;
      __builtin_prefetch(s->sig_info.data() + un);
    }
    if (s->rng.pos < prev_pos) spec_ok = true, spec_p = p, spec_pos = s->rng.pos;    // (a new block of outputs: look ahead again)
    prev_pos = s->rng.pos;
    if (spec_p <= p) spec_p = p, spec_pos = s->rng.pos;                              // (never behind the real position)
    speculate(p + kSpec);
    const int64_t e = s->order[p];
    const srh_sampler::Edge ed = s->edges[e];
    const int32_t uu = ed.u;
    u[p - ptr] = uu;
    it[p - ptr] = ed.i;
    bool rejected = false;
    for (int32_t m = 0; m < n_negs; ++m) {
      int32_t cand = (int32_t)s->rng.randbelow(n_items);
      while (s->rated(uu, cand)) { cand = (int32_t)s->rng.randbelow(n_items); rejected = true; }
      neg[w++] = cand;
    }
    if (rejected) spec_p = p + 1, spec_pos = s->rng.pos;       // the outputs moved on: the cursor's guesses are stale
  }
  return end - ptr;
}

srh_status_t srh_sampler_next_batch(srh_sampler_t* s, int64_t ptr, int64_t batch_size,
                                    int32_t n_negs, int32_t* h_u, int32_t* h_i, int32_t* h_j,
                                    int64_t* out_count) {
  srh_status_t st = require_seeded(s, "sampler_next_batch");
  if (st) return st;
  SRH_REQUIRE(h_u && h_i && h_j && out_count, "sampler_next_batch: null argument");
  SRH_REQUIRE(ptr >= 0 && ptr <= s->n_edges && batch_size > 0 && n_negs >= 1,
              "sampler_next_batch: bad ptr/batch_size/n_negs");
  *out_count = batch_into(s, ptr, batch_size, n_negs, h_u, h_i, h_j);
  return SRH_OK;
}

// sorted unique ids of one batch: mark a bitmap, then visit its non-empty 64-bit words in order.  The words are found
// through a SECOND bitmap with one bit per word of the first: a batch of 2048 ids touches at most 2048 of the 15.6 k words a
// million users span, and sweeping all of them for every batch was a third of an epoch's sampling at the 1 M x 500 k shape
// (2.7 of 7.6 s on the build host; with the summary 0.3).  `seen` holds both levels: [words of ids | words of words].
static int32_t sorted_unique(const int32_t* src, int64_t n, int32_t* dst, std::vector<uint64_t>& seen, size_t n_words) {
  uint64_t* l0 = seen.data();
  uint64_t* l1 = seen.data() + n_words;
  int32_t lo = INT32_MAX, hi = -1;                 // range of touched level-1 words
  for (int64_t k = 0; k < n; ++k) {
    const int32_t v = src[k];
    const int32_t w = v >> 6;
    l0[w] |= 1ULL << (v & 63);
    l1[w >> 6] |= 1ULL << (w & 63);
    lo = std::min(lo, w >> 6);
    hi = std::max(hi, w >> 6);
  }
  int32_t m = 0;
  for (int32_t w1 = lo; w1 <= hi; ++w1) {
    uint64_t words = l1[w1];
    l1[w1] = 0;
    while (words) {
      const int32_t wd = (w1 << 6) + __builtin_ctzll(words);
      words &= words - 1;
      uint64_t bits = l0[wd];
      l0[wd] = 0;
      while (bits) {
        dst[m++] = (wd << 6) + __builtin_ctzll(bits);
        bits &= bits - 1;
      }
    }
  }
  return m;
}

srh_status_t srh_sampler_epoch(srh_sampler_t* s, int64_t batch_size, int32_t n_negs,
                               int32_t* h_u, int32_t* h_i, int32_t* h_j,
                               int32_t* h_uniq_u, int32_t* h_n_uniq_u,
                               int32_t* h_uniq_i, int32_t* h_n_uniq_i) {
  srh_status_t st = require_seeded(s, "sampler_epoch");
  if (st) return st;
  SRH_REQUIRE(h_u && h_i && h_j, "sampler_epoch: null output");
  SRH_REQUIRE(batch_size > 0 && n_negs >= 1, "sampler_epoch: bad batch_size/n_negs");
  const bool want_uniq = h_uniq_u || h_uniq_i;
  SRH_REQUIRE(!want_uniq || (h_uniq_u && h_uniq_i && h_n_uniq_u && h_n_uniq_i),
              "sampler_epoch: unique outputs must be given together");
  st = srh_sampler_shuffle(s);
  if (st) return st;
  int64_t b = 0;
  for (int64_t ptr = 0; ptr < s->n_edges; ++b) {
    int64_t cnt = batch_into(s, ptr, batch_size, n_negs, h_u + ptr, h_i + ptr, h_j + ptr * n_negs);
    if (want_uniq) {
      h_n_uniq_u[b] = sorted_unique(h_u + ptr, cnt, h_uniq_u + b * batch_size, s->seen_u, (size_t)(s->n_users + 63) / 64);
      h_n_uniq_i[b] = sorted_unique(h_i + ptr, cnt, h_uniq_i + b * batch_size, s->seen_i, (size_t)(s->n_items + 63) / 64);
    }
    ptr += cnt;
  }
  return SRH_OK;
}

// The fixed-order row -> slot lists of every batch of an epoch (see include/selfrec_hip.h).  Pure host arithmetic on the
// arrays srh_sampler_epoch filled: no draw from the generator.
srh_status_t srh_sampler_epoch_segments(srh_sampler_t* s, int64_t batch_size, const int32_t* h_u, const int32_t* h_i,
                                        const int32_t* h_j, const int32_t* h_uniq_u, const int32_t* h_n_uniq_u,
                                        const int32_t* h_uniq_i, const int32_t* h_n_uniq_i, int32_t user_row0,
                                        int32_t item_row0, int32_t* h_n_uniq_n, int32_t* h_seg_rows, int32_t* h_seg_end,
                                        int32_t* h_seg, int32_t* h_seg_a, int32_t* h_seg_b) {
  SRH_REQUIRE(s && h_u && h_i && h_j && h_uniq_u && h_n_uniq_u && h_uniq_i && h_n_uniq_i, "sampler_epoch_segments: null input");
  SRH_REQUIRE(h_n_uniq_n && h_seg_rows && h_seg_end && h_seg && h_seg_a && h_seg_b, "sampler_epoch_segments: null output");
  SRH_REQUIRE(batch_size > 0 && batch_size < (int64_t(1) << 28), "sampler_epoch_segments: bad batch_size");
  // batches are independent: a few host threads, each with its own id -> row-group scratch (the sampler's draws are one
  // sequential stream; this is not)
  const int64_t n_batches = (s->n_edges + batch_size - 1) / batch_size;
  int n_thr = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency() / 2)),
                                                         n_batches / 16));
  if (const char* forced = std::getenv("SRH_SAMPLER_THREADS")) n_thr = std::max(1, std::min(64, std::atoi(forced)));
  const size_t item_words = (size_t)(s->n_items + 63) / 64;
  if ((int)s->seg_scratch.size() < n_thr) s->seg_scratch.resize((size_t)n_thr);
  for (int t = 0; t < n_thr; ++t) {
    srh_sampler::SegScratch& sc = s->seg_scratch[(size_t)t];
    if (sc.pos_u.empty()) {
      sc.pos_u.assign((size_t)s->n_users, -1);
      sc.pos_i.assign((size_t)s->n_items, -1);
      sc.seen.assign(item_words + (item_words + 63) / 64, 0);
    }
    if (sc.fill.size() < (size_t)(3 * batch_size)) {
      sc.fill.resize((size_t)(3 * batch_size));
      sc.neg_only.resize((size_t)batch_size);
      sc.un.resize((size_t)batch_size);
    }
  }
  auto work = [&](int t) {
    srh_sampler::SegScratch& sc = s->seg_scratch[(size_t)t];
    std::vector<int32_t>&neg_only = sc.neg_only, &un = sc.un;
    std::vector<uint64_t>& seen = sc.seen;
    int32_t* pos_u = sc.pos_u.data();
    int32_t* pos_i = sc.pos_i.data();
    int32_t* fill = sc.fill.data();
    for (int64_t b = t; b < n_batches; b += n_thr) {
      const int64_t ptr = b * batch_size;
      const int64_t cnt = std::min<int64_t>(batch_size, s->n_edges - ptr);
      const int32_t *u = h_u + ptr, *it = h_i + ptr, *jt = h_j + ptr;
      const int32_t *uu = h_uniq_u + b * batch_size, *ui = h_uniq_i + b * batch_size;
      const int32_t nuu = h_n_uniq_u[b], nui = h_n_uniq_i[b];
      int32_t* rows = h_seg_rows + b * 3 * batch_size;
      int32_t* end = h_seg_end + b * 3 * batch_size;
      int32_t* seg = h_seg + b * 3 * batch_size;
      int32_t* sa = h_seg_a + b * 3 * batch_size;
      int32_t* sb = h_seg_b + b * batch_size;
      for (int32_t k = 0; k < nuu; ++k) pos_u[uu[k]] = k;
      for (int32_t k = 0; k < nui; ++k) pos_i[ui[k]] = nuu + k;
      // sorted unique negatives that are nobody's positive in this batch
      int64_t m = 0;
      for (int64_t r = 0; r < cnt; ++r)
        if (pos_i[jt[r]] < 0) neg_only[(size_t)m++] = jt[r];
      const int32_t nun = sorted_unique(neg_only.data(), m, un.data(), seen, item_words);
      for (int32_t k = 0; k < nun; ++k) pos_i[un[(size_t)k]] = nuu + nui + k;
      const int32_t groups = nuu + nui + nun;
      for (int32_t k = 0; k < nuu; ++k) rows[k] = uu[k] + user_row0;
      for (int32_t k = 0; k < nui; ++k) rows[nuu + k] = ui[k] + item_row0;
      for (int32_t k = 0; k < nun; ++k) rows[nuu + nui + k] = un[(size_t)k] + item_row0;
      for (int64_t g = groups; g < 3 * batch_size; ++g) rows[g] = -1;          // (no row: the group has nothing to do)
      // counting sort of the 3 cnt (slot, role) entries by row group; inside a group: slot ascending, positive before negative
      for (int32_t g = 0; g < groups; ++g) end[g] = 0;
      for (int64_t r = 0; r < cnt; ++r) { ++end[pos_u[u[r]]]; ++end[pos_i[it[r]]]; ++end[pos_i[jt[r]]]; }
      int32_t run = 0;
      for (int32_t g = 0; g < groups; ++g) { fill[g] = run; run += end[g]; end[g] = run; }
      for (int64_t r = 0; r < cnt; ++r) {
        const int32_t eu = fill[pos_u[u[r]]]++;          // (user groups come first: eu < cnt)
        seg[eu] = (int32_t)(r * 4 + 0); sa[eu] = it[r] + item_row0; sb[eu] = jt[r] + item_row0;
        const int32_t ep = fill[pos_i[it[r]]]++;
        seg[ep] = (int32_t)(r * 4 + 1); sa[ep] = u[r] + user_row0;
        const int32_t en = fill[pos_i[jt[r]]]++;
        seg[en] = (int32_t)(r * 4 + 2); sa[en] = u[r] + user_row0;
      }
      for (int32_t k = 0; k < nuu; ++k) pos_u[uu[k]] = -1;
      for (int32_t k = 0; k < nui; ++k) pos_i[ui[k]] = -1;
      for (int32_t k = 0; k < nun; ++k) pos_i[un[(size_t)k]] = -1;
      h_n_uniq_n[b] = nun;
    }
  };
  if (n_thr == 1) work(0);
  else {
    std::vector<std::thread> pool;
    for (int t = 0; t < n_thr; ++t) pool.emplace_back(work, t);
    for (auto& th : pool) th.join();
  }
  return SRH_OK;
}

srh_status_t srh_sampler_sample_range(srh_sampler_t* s, int64_t n, int64_t k, int64_t* h_out) {
  srh_status_t st = require_seeded(s, "sampler_sample_range");
  if (st) return st;
  SRH_REQUIRE(h_out || k == 0, "sampler_sample_range: null output");
  SRH_REQUIRE(n >= 0 && n < (int64_t(1) << 32), "sampler_sample_range: n out of range");
  SRH_REQUIRE(k >= 0 && k <= n, "sampler_sample_range: Sample larger than population or is negative");
  // Lib/random.py sample(): setsize = 21; if k > 5: setsize += 4 ** _ceil(_log(k * 3, 4))
  double setsize = 21.0;
  if (k > 5) setsize += std::pow(4.0, std::ceil(std::log((double)(k * 3)) / std::log(4.0)));
  if ((double)n <= setsize) {
    std::vector<int64_t> pool((size_t)n);
    for (int64_t i = 0; i < n; ++i) pool[i] = i;
    for (int64_t i = 0; i < k; ++i) {
      uint32_t j = s->rng.randbelow((uint32_t)(n - i));
      h_out[i] = pool[j];
      pool[j] = pool[n - i - 1];
    }
  } else {
    std::unordered_set<int64_t> selected;
    selected.reserve((size_t)k * 2);
    for (int64_t i = 0; i < k; ++i) {
      int64_t j = s->rng.randbelow((uint32_t)n);
      while (selected.count(j)) j = s->rng.randbelow((uint32_t)n);
      selected.insert(j);
      h_out[i] = j;
    }
  }
  return SRH_OK;
}

srh_status_t srh_sampler_next_u32(srh_sampler_t* s, uint32_t* out) {
  srh_status_t st = require_seeded(s, "sampler_next_u32");
  if (st) return st;
  SRH_REQUIRE(out, "sampler_next_u32: null output");
  *out = s->rng.next_u32();
  return SRH_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// find_k_largest (reference util/algorithm.py:144-156) on the host, tie order included.  The reference keeps a min-heap of
// (score, position) pairs -- python's heapq on tuples -- seeded with the first K candidates, replaces the root whenever a
// later score is STRICTLY larger, and finally sorts the heap by score, descending and stable.  Which of several equal scores
// survive, and in what order they come out, is therefore a property of heapq's sift routines; they are restated here step
// for step (CPython Lib/heapq.py: heapify, _siftup, _siftdown, heapreplace), so that the few rows of a ranking whose best
// scores tie come out as the reference's would without walking 38 k python floats (6 ms per row).
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct HeapItem {
  float score;
  int64_t pos;
};
inline bool heap_less(const HeapItem& a, const HeapItem& b) {          // tuple comparison: score, then position
  return a.score < b.score || (a.score == b.score && a.pos < b.pos);
}
void heap_siftdown(HeapItem* heap, int64_t startpos, int64_t pos) {
  const HeapItem item = heap[pos];
  while (pos > startpos) {
    const int64_t parent = (pos - 1) >> 1;
    if (!heap_less(item, heap[parent])) break;
    heap[pos] = heap[parent];
    pos = parent;
  }
  heap[pos] = item;
}
void heap_siftup(HeapItem* heap, int64_t n, int64_t pos) {
  const int64_t startpos = pos;
  const HeapItem item = heap[pos];
  int64_t child = 2 * pos + 1;
  while (child < n) {                                                   // bubble the smaller child up until a leaf ...
    const int64_t right = child + 1;
    if (right < n && !heap_less(heap[child], heap[right])) child = right;
    heap[pos] = heap[child];
    pos = child;
    child = 2 * pos + 1;
  }
  heap[pos] = item;                                                     // ... then sift the item down from there
  heap_siftdown(heap, startpos, pos);
}
}  // namespace

srh_status_t srh_find_k_largest_host(int64_t k, const float* h_candidates, int64_t n, int64_t* h_out_ids, float* h_out_scores,
                                     int64_t* out_count) {
  SRH_REQUIRE(h_candidates && h_out_ids && h_out_scores && out_count, "find_k_largest_host: null argument");
  SRH_REQUIRE(k >= 1 && n >= 0, "find_k_largest_host: bad K or length");
  const int64_t m = std::min(k, n);                                     // (candidates[:K] of a shorter list: all of it)
  std::vector<HeapItem> heap((size_t)m);
  for (int64_t i = 0; i < m; ++i) heap[(size_t)i] = {h_candidates[i], i};
  for (int64_t i = m / 2 - 1; i >= 0; --i) heap_siftup(heap.data(), m, i);            // heapq.heapify
  HeapItem* hp = heap.data();
  float root = m > 0 ? hp[0].score : 0.f;
  for (int64_t i = k; i < n; ++i) {
    const float c = h_candidates[i];
    if (c > root) {                                                     // heapq.heapreplace
      hp[0] = {c, i};
      heap_siftup(hp, m, 0);
      root = hp[0].score;
    }
  }
  // heap.sort(key=score, reverse=True): stable, equal scores keep their order in the heap array
  std::stable_sort(heap.begin(), heap.end(), [](const HeapItem& a, const HeapItem& b) { return a.score > b.score; });
  for (int64_t i = 0; i < m; ++i) {
    h_out_ids[i] = heap[(size_t)i].pos;
    h_out_scores[i] = heap[(size_t)i].score;
  }
  *out_count = m;
  return SRH_OK;
}

srh_status_t srh_mt19937_uniform_f32(uint32_t* h_mt624, int32_t* pos, int64_t n, float* h_out,
                                     float keep_addend, uint8_t* h_keep) {
  SRH_REQUIRE(h_mt624 && pos, "mt19937_uniform_f32: null state");
  SRH_REQUIRE(*pos >= 0 && *pos <= MT19937::N, "mt19937_uniform_f32: position outside [0, 624]");
  SRH_REQUIRE(n >= 0, "mt19937_uniform_f32: negative count");
  MT19937 g;
  std::memcpy(g.mt, h_mt624, sizeof(g.mt));
  g.pos = *pos;
  int64_t done = 0;
  while (done < n) {
    if (g.pos >= MT19937::N) g.refill();
    const int take = (int)std::min<int64_t>(MT19937::N - g.pos, n - done);
    const uint32_t* w = g.mt + g.pos;
    float u[MT19937::N];
    for (int k = 0; k < take; ++k) {            // (no branches, no carried dependence: vectorised by the host compiler)
      uint32_t y = w[k];
      y ^= (y >> 11);
      y ^= (y << 7) & 0x9d2c5680U;
      y ^= (y << 15) & 0xefc60000U;
      y ^= (y >> 18);
      u[k] = (float)(y & 0xFFFFFFU) * 0x1p-24f;
    }
    if (h_out) std::memcpy(h_out + done, u, (size_t)take * sizeof(float));
    if (h_keep) {
      // floorf(x) != 0  <=>  not (0 <= x < 1), NaN and -0.0 included -- a compare the compiler vectorises
      uint8_t* kp = h_keep + done;
      for (int k = 0; k < take; ++k) {
        const float x = keep_addend + u[k];
        kp[k] = (uint8_t)!(x >= 0.0f && x < 1.0f);
      }
    }
    g.pos += take;
    done += take;
  }
  std::memcpy(h_mt624, g.mt, sizeof(g.mt));
  *pos = g.pos;
  return SRH_OK;
}

}  // extern "C"
