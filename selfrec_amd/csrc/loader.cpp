// (f-1) Dataset text files -> id arrays, natively.  Replaces the python loops of reference
// data/loader.py:22-33 (FileIO.load_data_set, "user item weight" lines) and data/ui_graph.py:29-45
// (first-appearance id maps; test pairs kept only when both ends were seen in training), which cost
// ~3 s at Yelp2018 shape and minutes at 50 M interactions.  Pure host code.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "common.h"

struct srh_dataset {
  std::vector<int32_t> train_u, train_i, test_u, test_i;
  std::vector<float> train_w;
  std::vector<std::string> user_names, item_names;
  int64_t test_lines = 0;
};

namespace {

struct Line {
  const char *u, *i, *w;
  size_t ul, il, wl;
};

// python: items = split(' ', line.strip()); user, item, weight = items[0], items[1], items[2]
bool split_line(const char* b, const char* e, Line& out) {
  auto is_ws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f' || c == '\v'; };
  while (b < e && is_ws(*b)) ++b;
  while (e > b && is_ws(e[-1])) --e;
  const char* p1 = (const char*)memchr(b, ' ', e - b);
  if (!p1) return false;
  const char* p2 = (const char*)memchr(p1 + 1, ' ', e - (p1 + 1));
  if (!p2) return false;
  const char* p3 = (const char*)memchr(p2 + 1, ' ', e - (p2 + 1));
  if (!p3) p3 = e;
  out = {b, p1 + 1, p2 + 1, (size_t)(p1 - b), (size_t)(p2 - p1 - 1), (size_t)(p3 - p2 - 1)};
  return true;
}

bool read_file(const char* path, std::vector<char>& buf) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  buf.resize((size_t)std::max(0L, n));
  const size_t got = n > 0 ? fread(buf.data(), 1, (size_t)n, f) : 0;
  fclose(f);
  return got == (size_t)std::max(0L, n);
}

// name (bytes inside the file buffer) -> dense id, open addressing, no per-name allocation.  40 M lines spend their time in
// this table: std::unordered_map<std::string, ...> built two heap strings per line.
struct NameTable {
  struct Slot { const char* p; uint32_t len; int32_t id; uint64_t h; };
  std::vector<Slot> slots;
  struct Name { const char* first; uint32_t second; uint64_t h; };
  std::vector<Name> names;                                   // id -> name (+ its hash), in first-appearance order
  size_t mask = 0;
  explicit NameTable(size_t cap_log2 = 12) { slots.assign((size_t)1 << cap_log2, Slot{nullptr, 0, -1, 0}); mask = slots.size() - 1; }
  static uint64_t hash(const char* p, size_t n) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xff51afd7ed558ccdull);
    while (n >= 8) { uint64_t w; memcpy(&w, p, 8); h = (h ^ w) * 0xc4ceb9fe1a85ec53ull; h ^= h >> 29; p += 8; n -= 8; }
    uint64_t w = 0;
    memcpy(&w, p, n);
    h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
    return h ^ (h >> 32);
  }
  void grow() {
    std::vector<Slot> old;
    old.swap(slots);
    slots.assign(old.size() * 2, Slot{nullptr, 0, -1, 0});
    mask = slots.size() - 1;
    for (const Slot& s : old)
      if (s.p) { size_t k = s.h & mask; while (slots[k].p) k = (k + 1) & mask; slots[k] = s; }
  }
  int32_t find(const char* p, size_t n, uint64_t h) const {
    for (size_t k = h & mask;; k = (k + 1) & mask) {
      const Slot& s = slots[k];
      if (!s.p) return -1;
      if (s.h == h && s.len == n && memcmp(s.p, p, n) == 0) return s.id;
    }
  }
  int32_t find(const char* p, size_t n) const { return find(p, n, hash(p, n)); }
  int32_t find_or_add(const char* p, size_t n) {
    const uint64_t h = hash(p, n);
    for (size_t k = h & mask;; k = (k + 1) & mask) {
      Slot& s = slots[k];
      if (!s.p) {
        const int32_t id = (int32_t)names.size();
        s = Slot{p, (uint32_t)n, id, h};
        names.push_back(Name{p, (uint32_t)n, h});
        if (names.size() * 10 > slots.size() * 6) grow();
        return id;
      }
      if (s.h == h && s.len == n && memcmp(s.p, p, n) == 0) return s.id;
    }
  }
};

float parse_weight(const char* p, size_t n) {
  if (n == 1 && *p == '1') return 1.0f;                    // (the common case of implicit-feedback files)
  char tmp[64];
  if (n < sizeof(tmp)) { memcpy(tmp, p, n); tmp[n] = 0; return strtof(tmp, nullptr); }
  return strtof(std::string(p, n).c_str(), nullptr);
}

// [b, e) cut into `parts` pieces that end at line ends
std::vector<const char*> cut_at_lines(const char* b, const char* e, int parts) {
  std::vector<const char*> cuts{b};
  for (int k = 1; k < parts; ++k) {
    const char* p = b + (size_t)(e - b) * k / parts;
    if (p <= cuts.back()) continue;
    const char* nl = (const char*)memchr(p, '\n', e - p);
    if (!nl) break;
    if (nl + 1 > cuts.back() && nl + 1 < e) cuts.push_back(nl + 1);
  }
  cuts.push_back(e);
  return cuts;
}

int worker_count(size_t bytes) {
  const char* env = getenv("SRH_LOADER_THREADS");          // (tests force a count; otherwise one worker per 2 MB, <= 32)
  if (env) return std::max(1, std::min(atoi(env), 256));
  const int hw = (int)std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency()));
  return std::max(1, std::min(hw, (int)(bytes / (2u << 20)) + 1));
}

}  // namespace

extern "C" {

// First-appearance ids over the WHOLE file, built in parallel: every worker maps the names of its piece of the file to
// piece-local ids in local first-appearance order; the pieces' name lists are then merged in file order (distinct names
// only: cheap and sequential, which is what makes the global order the python loop's), and the workers translate their
// lines.  Same ids, same name order, same kept test pairs as the single-threaded loop at any thread count.
srh_status_t srh_dataset_load(srh_dataset_t** out, const char* train_path, const char* test_path) {
  SRH_REQUIRE(out && train_path, "dataset_load: null argument");
  std::vector<char> buf;
  if (!read_file(train_path, buf)) { srh::set_error("dataset_load: cannot read %s", train_path); return SRH_ERR_INVALID_ARG; }
  srh_dataset* ds = new (std::nothrow) srh_dataset();
  if (!ds) { srh::set_error("dataset_load: out of memory"); return SRH_ERR_NOMEM; }
  struct Piece {
    NameTable users{10}, items{10};
    std::vector<int32_t> lu, li;
    std::vector<float> w;
    int64_t lines = 0, bad_line = -1;
    std::vector<int32_t> tr_u, tr_i;                      // local id -> global id
  };
  const char *b = buf.data(), *e = buf.data() + buf.size();
  const std::vector<const char*> cuts = cut_at_lines(b, e, worker_count(buf.size()));
  const int T = (int)cuts.size() - 1;
  std::vector<Piece> pieces((size_t)std::max(T, 0));
  auto scan = [&](int t) {
    Piece& pc = pieces[t];
    const char *p = cuts[t], *end = cuts[t + 1];
    const size_t guess = (size_t)(end - p) / 12 + 16;
    pc.lu.reserve(guess); pc.li.reserve(guess); pc.w.reserve(guess);
    while (p < end) {
      const char* nl = (const char*)memchr(p, '\n', end - p);
      const char* le = nl ? nl : end;
      ++pc.lines;
      Line ln;
      if (!split_line(p, le, ln)) { pc.bad_line = pc.lines; return; }
      pc.lu.push_back(pc.users.find_or_add(ln.u, ln.ul));
      pc.li.push_back(pc.items.find_or_add(ln.i, ln.il));
      pc.w.push_back(parse_weight(ln.w, ln.wl));
      p = nl ? nl + 1 : end;
    }
  };
  auto run_all = [&](auto&& fn, int n) {
    if (n <= 1) { for (int t = 0; t < n; ++t) fn(t); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < n; ++t) th.emplace_back(fn, t);
    for (auto& x : th) x.join();
  };
  run_all(scan, T);
  int64_t lines_before = 0;
  for (int t = 0; t < T; ++t) {
    if (pieces[t].bad_line >= 0) {
      const long long at = (long long)(lines_before + pieces[t].bad_line);
      delete ds;
      srh::set_error("dataset_load: %s line %lld does not have the form 'user item weight'", train_path, at);
      return SRH_ERR_INVALID_ARG;
    }
    lines_before += pieces[t].lines;
  }
  // ---- merge, in parallel by NAME: partition q owns the names whose hash falls to it, walks every piece's list of ITS
  // names in file order and records, for each, the (piece, local id) of its first appearance.  (Merging the pieces' lists
  // one after the other into one table is sequential and, with mostly-distinct names in every piece, costs T x names
  // look-ups: it was 60 % of the loader's time at 8 threads.)
  const int Q = std::max(T, 1);
  auto part_of = [Q](uint64_t h) { return (int)((h >> 33) % (uint64_t)Q); };
  struct Owner { int32_t piece, local; };
  struct Part { NameTable users{12}, items{12}; std::vector<Owner> own_u, own_i; };
  std::vector<Part> parts((size_t)Q);
  std::vector<std::vector<Owner>> owner_u((size_t)T), owner_i((size_t)T);       // per piece, per local id: where it first appeared
  std::vector<std::vector<uint8_t>> new_u((size_t)T), new_i((size_t)T);
  for (int t = 0; t < T; ++t) {
    owner_u[t].resize(pieces[t].users.names.size()); new_u[t].assign(pieces[t].users.names.size(), 0);
    owner_i[t].resize(pieces[t].items.names.size()); new_i[t].assign(pieces[t].items.names.size(), 0);
  }
  run_all([&](int q) {
    Part& pt = parts[q];
    auto walk = [&](bool is_user) {
      NameTable& tab = is_user ? pt.users : pt.items;
      std::vector<Owner>& own = is_user ? pt.own_u : pt.own_i;
      for (int t = 0; t < T; ++t) {
        const auto& names = is_user ? pieces[t].users.names : pieces[t].items.names;
        auto& owner = is_user ? owner_u[t] : owner_i[t];
        auto& fresh = is_user ? new_u[t] : new_i[t];
        for (size_t k = 0; k < names.size(); ++k) {
          if (part_of(names[k].h) != q) continue;
          const int32_t id = tab.find_or_add(names[k].first, names[k].second);
          if ((size_t)id == own.size()) { own.push_back(Owner{t, (int32_t)k}); fresh[k] = 1; }
          owner[k] = own[id];
        }
      }
    };
    walk(true);
    walk(false);
  }, Q);
  // global id of a name = how many names appeared for the first time before it: pieces in file order, local ids in order
  std::vector<std::vector<int32_t>> rank_u((size_t)T), rank_i((size_t)T);
  int32_t n_users = 0, n_items = 0;
  for (int t = 0; t < T; ++t) {
    rank_u[t].resize(new_u[t].size()); rank_i[t].resize(new_i[t].size());
    for (size_t k = 0; k < new_u[t].size(); ++k) { rank_u[t][k] = n_users; n_users += new_u[t][k]; }
    for (size_t k = 0; k < new_i[t].size(); ++k) { rank_i[t][k] = n_items; n_items += new_i[t][k]; }
  }
  std::vector<int64_t> offset((size_t)T + 1, 0);
  for (int t = 0; t < T; ++t) offset[t + 1] = offset[t] + (int64_t)pieces[t].lu.size();
  ds->user_names.resize((size_t)n_users);
  ds->item_names.resize((size_t)n_items);
  run_all([&](int t) {
    Piece& pc = pieces[t];
    pc.tr_u.resize(owner_u[t].size());
    pc.tr_i.resize(owner_i[t].size());
    for (size_t k = 0; k < owner_u[t].size(); ++k) {
      pc.tr_u[k] = rank_u[owner_u[t][k].piece][owner_u[t][k].local];
      if (new_u[t][k]) ds->user_names[(size_t)pc.tr_u[k]].assign(pc.users.names[k].first, pc.users.names[k].second);
    }
    for (size_t k = 0; k < owner_i[t].size(); ++k) {
      pc.tr_i[k] = rank_i[owner_i[t][k].piece][owner_i[t][k].local];
      if (new_i[t][k]) ds->item_names[(size_t)pc.tr_i[k]].assign(pc.items.names[k].first, pc.items.names[k].second);
    }
  }, T);
  // name -> global id for the test file: through the partition that owns the name
  auto global_id = [&](bool is_user, const char* p, size_t n) -> int32_t {
    const uint64_t h = NameTable::hash(p, n);
    const Part& pt = parts[part_of(h)];
    const int32_t id = (is_user ? pt.users : pt.items).find(p, n, h);
    if (id < 0) return -1;
    const Owner o = (is_user ? pt.own_u : pt.own_i)[(size_t)id];
    return (is_user ? rank_u : rank_i)[o.piece][o.local];
  };
  ds->train_u.resize((size_t)offset[T]); ds->train_i.resize((size_t)offset[T]); ds->train_w.resize((size_t)offset[T]);
  run_all([&](int t) {
    const Piece& pc = pieces[t];
    int32_t* du = ds->train_u.data() + offset[t];
    int32_t* di = ds->train_i.data() + offset[t];
    for (size_t k = 0; k < pc.lu.size(); ++k) { du[k] = pc.tr_u[pc.lu[k]]; di[k] = pc.tr_i[pc.li[k]]; }
    if (!pc.w.empty()) memcpy(ds->train_w.data() + offset[t], pc.w.data(), sizeof(float) * pc.w.size());
  }, T);
  // (the partitions' tables point into `buf` and into the pieces' name lists: both stay alive until the test file is done)
  if (test_path) {
    std::vector<char> tbuf;                                // (the name tables point into `buf`: keep it)
    if (!read_file(test_path, tbuf)) { delete ds; srh::set_error("dataset_load: cannot read %s", test_path); return SRH_ERR_INVALID_ARG; }
    const char *tb = tbuf.data(), *te = tbuf.data() + tbuf.size();
    const std::vector<const char*> tcuts = cut_at_lines(tb, te, worker_count(tbuf.size()));
    const int TT = (int)tcuts.size() - 1;
    struct TestPiece { std::vector<int32_t> u, i; int64_t lines = 0; };
    std::vector<TestPiece> tp((size_t)std::max(TT, 0));
    run_all([&](int t) {
      const char *p = tcuts[t], *end = tcuts[t + 1];
      while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', end - p);
        const char* le = nl ? nl : end;
        Line ln;
        if (split_line(p, le, ln)) {
          ++tp[t].lines;
          const int32_t u = global_id(true, ln.u, ln.ul), i = global_id(false, ln.i, ln.il);
          if (u >= 0 && i >= 0) { tp[t].u.push_back(u); tp[t].i.push_back(i); }
        }
        p = nl ? nl + 1 : end;
      }
    }, TT);
    for (const TestPiece& q : tp) {
      ds->test_lines += q.lines;
      ds->test_u.insert(ds->test_u.end(), q.u.begin(), q.u.end());
      ds->test_i.insert(ds->test_i.end(), q.i.begin(), q.i.end());
    }
  }
  *out = ds;
  return SRH_OK;
}

void srh_dataset_destroy(srh_dataset_t* ds) { delete ds; }

srh_status_t srh_dataset_sizes(const srh_dataset_t* ds, int64_t* h_sizes5) {
  SRH_REQUIRE(ds && h_sizes5, "dataset_sizes: null argument");
  h_sizes5[0] = (int64_t)ds->user_names.size();
  h_sizes5[1] = (int64_t)ds->item_names.size();
  h_sizes5[2] = (int64_t)ds->train_u.size();
  h_sizes5[3] = (int64_t)ds->test_u.size();
  h_sizes5[4] = ds->test_lines;
  return SRH_OK;
}

srh_status_t srh_dataset_copy_ids(const srh_dataset_t* ds, int32_t* h_train_u, int32_t* h_train_i, float* h_train_w,
                                  int32_t* h_test_u, int32_t* h_test_i) {
  SRH_REQUIRE(ds, "dataset_copy_ids: null handle");
  if (h_train_u) memcpy(h_train_u, ds->train_u.data(), sizeof(int32_t) * ds->train_u.size());
  if (h_train_i) memcpy(h_train_i, ds->train_i.data(), sizeof(int32_t) * ds->train_i.size());
  if (h_train_w) memcpy(h_train_w, ds->train_w.data(), sizeof(float) * ds->train_w.size());
  if (h_test_u) memcpy(h_test_u, ds->test_u.data(), sizeof(int32_t) * ds->test_u.size());
  if (h_test_i) memcpy(h_test_i, ds->test_i.data(), sizeof(int32_t) * ds->test_i.size());
  return SRH_OK;
}

int64_t srh_dataset_names_bytes(const srh_dataset_t* ds, int32_t which) {
  if (!ds) return 0;
  const auto& v = which ? ds->item_names : ds->user_names;
  int64_t n = 0;
  for (const auto& s : v) n += (int64_t)s.size();
  return n;
}

srh_status_t srh_dataset_copy_names(const srh_dataset_t* ds, int32_t which, char* h_buf, int64_t* h_offsets) {
  SRH_REQUIRE(ds && h_buf, "dataset_copy_names: null argument");
  const auto& v = which ? ds->item_names : ds->user_names;
  int64_t at = 0;
  if (!h_offsets) {                       // joined form: name '\n' name '\n' ... (names_bytes + count bytes; names hold no '\n')
    for (size_t k = 0; k < v.size(); ++k) {
      memcpy(h_buf + at, v[k].data(), v[k].size());
      at += (int64_t)v[k].size();
      h_buf[at++] = '\n';
    }
    return SRH_OK;
  }
  for (size_t k = 0; k < v.size(); ++k) {
    h_offsets[k] = at;
    memcpy(h_buf + at, v[k].data(), v[k].size());
    at += (int64_t)v[k].size();
  }
  h_offsets[v.size()] = at;
  return SRH_OK;
}

}  // extern "C"
