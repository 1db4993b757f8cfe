// (f-1) Dataset text files -> id arrays, natively.  Replaces the python loops of reference
// data/loader.py:22-33 (FileIO.load_data_set, "user item weight" lines) and data/ui_graph.py:29-45
// (first-appearance id maps; test pairs kept only when both ends were seen in training), which cost
// ~3 s at Yelp2018 shape and minutes at 50 M interactions.  Pure host code.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
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

}  // namespace

extern "C" {

srh_status_t srh_dataset_load(srh_dataset_t** out, const char* train_path, const char* test_path) {
  SRH_REQUIRE(out && train_path, "dataset_load: null argument");
  std::vector<char> buf;
  if (!read_file(train_path, buf)) { srh::set_error("dataset_load: cannot read %s", train_path); return SRH_ERR_INVALID_ARG; }
  srh_dataset* ds = new (std::nothrow) srh_dataset();
  if (!ds) { srh::set_error("dataset_load: out of memory"); return SRH_ERR_NOMEM; }
  std::unordered_map<std::string, int32_t> users, items;
  users.reserve(1 << 16); items.reserve(1 << 16);
  int64_t line_no = 0;
  const char *p = buf.data(), *end = buf.data() + buf.size();
  while (p < end) {
    const char* nl = (const char*)memchr(p, '\n', end - p);
    const char* le = nl ? nl : end;
    ++line_no;
    Line ln;
    if (!split_line(p, le, ln)) {
      delete ds;
      srh::set_error("dataset_load: %s line %lld does not have the form 'user item weight'", train_path, (long long)line_no);
      return SRH_ERR_INVALID_ARG;
    }
    std::string us(ln.u, ln.ul), is(ln.i, ln.il);
    auto iu = users.find(us);
    if (iu == users.end()) { iu = users.emplace(us, (int32_t)users.size()).first; ds->user_names.push_back(us); }
    auto ii = items.find(is);
    if (ii == items.end()) { ii = items.emplace(is, (int32_t)items.size()).first; ds->item_names.push_back(is); }
    ds->train_u.push_back(iu->second);
    ds->train_i.push_back(ii->second);
    ds->train_w.push_back(strtof(std::string(ln.w, ln.wl).c_str(), nullptr));
    p = nl ? nl + 1 : end;
  }
  if (test_path) {
    if (!read_file(test_path, buf)) { delete ds; srh::set_error("dataset_load: cannot read %s", test_path); return SRH_ERR_INVALID_ARG; }
    p = buf.data(); end = buf.data() + buf.size();
    while (p < end) {
      const char* nl = (const char*)memchr(p, '\n', end - p);
      const char* le = nl ? nl : end;
      Line ln;
      if (split_line(p, le, ln)) {
        ++ds->test_lines;
        auto iu = users.find(std::string(ln.u, ln.ul));
        auto ii = items.find(std::string(ln.i, ln.il));
        if (iu != users.end() && ii != items.end()) { ds->test_u.push_back(iu->second); ds->test_i.push_back(ii->second); }
      }
      p = nl ? nl + 1 : end;
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
  SRH_REQUIRE(ds && h_buf && h_offsets, "dataset_copy_names: null argument");
  const auto& v = which ? ds->item_names : ds->user_names;
  int64_t at = 0;
  for (size_t k = 0; k < v.size(); ++k) {
    h_offsets[k] = at;
    memcpy(h_buf + at, v[k].data(), v[k].size());
    at += (int64_t)v[k].size();
  }
  h_offsets[v.size()] = at;
  return SRH_OK;
}

}  // extern "C"
