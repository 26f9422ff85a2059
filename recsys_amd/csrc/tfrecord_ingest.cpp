// Host-side TFRecord / tf.train.Example ingest of librsx.so (no device code).
// Replaces tf.data.TFRecordDataset + tf.parse_single_example of fm/fm.py:100-112 (deepfm/deepfm.py:54-70,
// xdeepfm/xdeepfm.py:95-118, dcn/dcn.py:100-112, din/din.py:52-80) and produces what the reference's
// feature columns produce next: table-local ids (FarmHash % bucket, bucketize(log(x+shift))), the label
// and the log-normalised numerics.  Format per SURVEY.md Appendix A-14 (what xdeepfm/gen_tfrecords.py:31-40
// writes through spark-tensorflow-connector).  Multi-threaded over records; a writer for synthetic shards
// is included because the reference's only sample shard is a missing blob.
#include <atomic>
#include <cmath>
#include <cstring>
#include <string>
#include <pthread.h>
#include <sched.h>
#include <cstdlib>
#include <thread>
#include <vector>

#include "rsx.h"

extern "C" uint64_t rsx_fingerprint64_h(const uint8_t* s, size_t n);

namespace {
struct Span { const uint8_t* p; size_t n; };

inline bool rd_varint(const uint8_t*& p, const uint8_t* e, uint64_t& v) {
  v = 0;
  for (int s = 0; s < 64 && p < e; s += 7) {
    const uint8_t b = *p++;
    v |= (uint64_t)(b & 0x7f) << s;
    if (!(b & 0x80)) return true;
  }
  return false;
}
// next field of a message: returns false at end or on malformed input (ok=false)
inline bool next_field(const uint8_t*& p, const uint8_t* e, uint32_t& fno, uint32_t& wt, Span& val, uint64_t& num, bool& ok) {
  if (p >= e) return false;
  uint64_t tag;
  if (!rd_varint(p, e, tag)) { ok = false; return false; }
  fno = (uint32_t)(tag >> 3);
  wt = (uint32_t)(tag & 7);
  if (wt == 2) {
    uint64_t n;
    if (!rd_varint(p, e, n) || n > (uint64_t)(e - p)) { ok = false; return false; }
    val = {p, (size_t)n};
    p += n;
  } else if (wt == 0) {
    if (!rd_varint(p, e, num)) { ok = false; return false; }
  } else if (wt == 5) {
    if (e - p < 4) { ok = false; return false; }
    val = {p, 4};
    p += 4;
  } else if (wt == 1) {
    if (e - p < 8) { ok = false; return false; }
    val = {p, 8};
    p += 8;
  } else {
    ok = false;
    return false;
  }
  return true;
}

// Visits every (key, Feature payload) of one serialized Example.
template <class F>
bool for_each_feature(const uint8_t* rec, size_t n, F f) {
  bool ok = true;
  const uint8_t *p = rec, *e = rec + n;
  uint32_t fno, wt; Span v; uint64_t num;
  while (next_field(p, e, fno, wt, v, num, ok)) {
    if (fno != 1 || wt != 2) continue;                 // Example.features
    const uint8_t *q = v.p, *qe = v.p + v.n;
    Span ent;
    while (next_field(q, qe, fno, wt, ent, num, ok)) {
      if (fno != 1 || wt != 2) continue;               // Features.feature (map entry)
      const uint8_t *r = ent.p, *re = ent.p + ent.n;
      Span key{nullptr, 0}, feat{nullptr, 0}, t;
      while (next_field(r, re, fno, wt, t, num, ok)) {
        if (fno == 1 && wt == 2) key = t;
        else if (fno == 2 && wt == 2) feat = t;
      }
      if (!ok) return false;
      if (key.p) f(key, feat);
    }
  }
  return ok;
}

// Feature -> first float of float_list (kind 2)
inline bool feat_first_float(Span feat, float& out) {
  bool ok = true;
  const uint8_t *p = feat.p, *e = feat.p + feat.n;
  uint32_t fno, wt; Span v; uint64_t num;
  while (next_field(p, e, fno, wt, v, num, ok)) {
    if (fno != 2 || wt != 2) continue;
    const uint8_t *q = v.p, *qe = v.p + v.n;
    Span x;
    while (next_field(q, qe, fno, wt, x, num, ok)) {
      if (fno != 1) continue;
      if ((wt == 2 && x.n >= 4) || wt == 5) { std::memcpy(&out, x.p, 4); return true; }
    }
  }
  return false;
}
// Feature -> first bytes value of bytes_list (kind 1)
inline bool feat_first_bytes(Span feat, Span& out) {
  bool ok = true;
  const uint8_t *p = feat.p, *e = feat.p + feat.n;
  uint32_t fno, wt; Span v; uint64_t num;
  while (next_field(p, e, fno, wt, v, num, ok)) {
    if (fno != 1 || wt != 2) continue;
    const uint8_t *q = v.p, *qe = v.p + v.n;
    Span x;
    while (next_field(q, qe, fno, wt, x, num, ok))
      if (fno == 1 && wt == 2) { out = x; return true; }
  }
  return false;
}
// Feature -> all int64 of int64_list (kind 3); returns count (writes at most cap)
inline int64_t feat_int64s(Span feat, int64_t* out, int64_t cap) {
  bool ok = true;
  int64_t n = 0;
  const uint8_t *p = feat.p, *e = feat.p + feat.n;
  uint32_t fno, wt; Span v; uint64_t num;
  while (next_field(p, e, fno, wt, v, num, ok)) {
    if (fno != 3 || wt != 2) continue;
    const uint8_t *q = v.p, *qe = v.p + v.n;
    Span x;
    while (next_field(q, qe, fno, wt, x, num, ok)) {
      if (fno != 1) continue;
      if (wt == 0) { if (n < cap) out[n] = (int64_t)num; ++n; }
      else if (wt == 2) {
        const uint8_t *r = x.p, *re = x.p + x.n;
        uint64_t y;
        while (r < re && rd_varint(r, re, y)) { if (n < cap) out[n] = (int64_t)y; ++n; }
      }
    }
  }
  return n;
}

inline int key_cN(Span key) {   // "_c0".."_c39" -> 0..39, else -1
  if (key.n < 3 || key.n > 4 || key.p[0] != '_' || key.p[1] != 'c') return -1;
  int v = 0;
  for (size_t i = 2; i < key.n; ++i) {
    if (key.p[i] < '0' || key.p[i] > '9') return -1;
    v = v * 10 + (key.p[i] - '0');
  }
  return v <= 39 ? v : -1;
}

// The CPUs this process may run on (its affinity mask at first use).
const std::vector<int>& allowed_cpus() {
  static const std::vector<int> cpus = [] {
    std::vector<int> v;
    cpu_set_t m;
    if (sched_getaffinity(0, sizeof(m), &m) == 0)
      for (int c = 0; c < CPU_SETSIZE; ++c)
        if (CPU_ISSET(c, &m)) v.push_back(c);
    return v;
  }();
  return cpus;
}

// Splits [0, n) over at most `threads` worker threads -- never more than the process may run on: oversubscribed workers only
// time-slice one another -- each PINNED to one allowed CPU.  (Unpinned, the short-lived workers of a call were seen staying
// on the caller's CPU for their whole life on the 8-vCPU build container: 1, 2, 4 and 8 threads all parsed 0.53 M records/s.
// RSX_HOST_PIN=0 leaves placement to the scheduler.)
template <class F>
void parallel_for(int64_t n, int threads, F f) {
  const std::vector<int>& cpus = allowed_cpus();
  if (!cpus.empty() && threads > (int)cpus.size()) threads = (int)cpus.size();
  if (threads <= 1 || n < 64) { f(0, n); return; }
  static const bool pin = [] { const char* e = std::getenv("RSX_HOST_PIN"); return !(e && e[0] == '0'); }();
  // (one process per GPU on a shared host: rank r starts at CPU r * threads, so that the ranks' workers do not pile up on CPU 0..)
  static const int rank = [] { const char* e = std::getenv("LOCAL_RANK"); return e ? std::atoi(e) : 0; }();
  const size_t base = (size_t)(rank < 0 ? 0 : rank) * (size_t)threads;
  std::vector<std::thread> th;
  const int64_t chunk = (n + threads - 1) / threads;
  for (int t = 0; t < threads; ++t) {
    const int64_t a = t * chunk, b = a + chunk < n ? a + chunk : n;
    if (a >= b) break;
    th.emplace_back([=, &cpus] {
      if (pin && !cpus.empty()) {
        cpu_set_t m;
        CPU_ZERO(&m);
        CPU_SET(cpus[(base + t) % cpus.size()], &m);
        pthread_setaffinity_np(pthread_self(), sizeof(m), &m);
      }
      f(a, b);
    });
  }
  for (auto& x : th) x.join();
}

inline void put_varint(std::string& s, uint64_t v) {
  while (v >= 0x80) { s.push_back((char)(v | 0x80)); v >>= 7; }
  s.push_back((char)v);
}
inline void put_ld(std::string& s, uint32_t field, const std::string& payload) {
  put_varint(s, (field << 3) | 2);
  put_varint(s, payload.size());
  s += payload;
}
}  // namespace

// Scans the framing `u64 len | u32 masked_crc(len) | data | u32 masked_crc(data)` of a whole shard image.
// Returns the number of records (offsets/lengths of the payloads are written up to max_records), or a negative
// rsx_status: RSX_EDATA on a truncated record or a crc mismatch (TF raises DataLossError there).
extern "C" int64_t rsx_tfrecord_index_h(const uint8_t* buf_h, size_t n, int64_t* offsets_h, int64_t* lengths_h,
                                        int64_t max_records, int verify_crc) {
  if (n > 0 && !buf_h) return RSX_EINVAL;
  size_t p = 0;
  int64_t cnt = 0;
  while (p < n) {
    if (n - p < 12) return RSX_EDATA;
    uint64_t len;
    std::memcpy(&len, buf_h + p, 8);
    uint32_t c;
    std::memcpy(&c, buf_h + p + 8, 4);
    if (verify_crc && c != rsx_masked_crc32c_h(buf_h + p, 8)) return RSX_EDATA;
    if (len > n - p - 12 || n - p - 12 - len < 4) return RSX_EDATA;
    if (verify_crc) {
      std::memcpy(&c, buf_h + p + 12 + len, 4);
      if (c != rsx_masked_crc32c_h(buf_h + p + 12, (size_t)len)) return RSX_EDATA;
    }
    if (cnt < max_records && offsets_h && lengths_h) {
      offsets_h[cnt] = (int64_t)(p + 12);
      lengths_h[cnt] = (int64_t)len;
    }
    ++cnt;
    p += 16 + len;
  }
  return cnt;
}

// One Criteo record -> one row: label, cont_log[13] = log(x + shift_j) (nullable), ids[F] in slot order.
//   slot_src[F]   source feature index j of each slot (1..13 numeric, 14..39 categorical)
//   slot_rows[F]  bucket count of the slot; bnd / bnd_off[F+1]: boundaries of the numeric slots
//   shift[13]     log shift of _c1.._c13 (1, except 4 for _c2: fm/fm.py:77-78)
// Absent categorical feature -> the 'NULL' default (fm/fm.py:44); absent numeric or label -> RSX_EDATA (TF:
// FixedLenFeature without default).  Shared by the batch entry point below and the streaming reader.
int rsx_criteo_parse_row(const uint8_t* rec, size_t n, const int32_t* slot_src, const int32_t* slot_rows, const float* bnd,
                         const int32_t* bnd_off, const float* shift, int F, uint64_t null_hash, float* label,
                         float* cont_log, int32_t* ids, bool label_optional) {
  float fv[14];
  fv[0] = 0.f;
  bool have_f[14] = {false};
  uint64_t hv[40];
  bool have_h[40] = {false};
  const bool ok = for_each_feature(rec, n, [&](Span key, Span feat) {
    const int j = key_cN(key);
    if (j < 0) return;
    if (j <= 13) {
      float x;
      if (feat_first_float(feat, x)) { fv[j] = x; have_f[j] = true; }
    } else {
      Span s;
      if (feat_first_bytes(feat, s)) { hv[j] = rsx_fingerprint64_h(s.p, s.n); have_h[j] = true; }
    }
  });
  bool good = ok;
  for (int j = label_optional ? 1 : 0; j <= 13; ++j) good = good && have_f[j];
  if (!good) return RSX_EDATA;
  *label = fv[0];
  float lg[14];
  for (int j = 1; j <= 13; ++j) {
    lg[j] = logf(fv[j] + shift[j - 1]);
    if (cont_log) cont_log[j - 1] = lg[j];
  }
  for (int s = 0; s < F; ++s) {
    const int j = slot_src[s];
    int32_t id;
    if (j <= 13) {
      const float v = lg[j];
      const float* bd = bnd + bnd_off[s];
      const int nb = bnd_off[s + 1] - bnd_off[s];
      if (v != v) id = nb;
      else {
        int lo = 0, hi = nb;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (bd[mid] <= v) lo = mid + 1; else hi = mid; }
        id = lo;
      }
    } else {
      id = (int32_t)((have_h[j] ? hv[j] : null_hash) % (uint64_t)slot_rows[s]);
    }
    ids[s] = id;
  }
  return RSX_OK;
}

// Criteo records -> label[n], cont_log[n,13], ids[n,F]: the row parser over a list of records, multi-threaded.
extern "C" int rsx_criteo_parse_h(const uint8_t* buf_h, const int64_t* offsets_h, const int64_t* lengths_h, int64_t n,
                                  const int32_t* slot_src_h, const int32_t* slot_rows_h, const float* bnd_h,
                                  const int32_t* bnd_off_h, const float* shift_h, int F, float* label_h,
                                  float* cont_log_h, int32_t* ids_h, int threads) {
  if (n < 0 || F <= 0 || F > 64) return RSX_EINVAL;
  if (n == 0) return RSX_OK;
  if (!buf_h || !offsets_h || !lengths_h || !slot_src_h || !slot_rows_h || !bnd_off_h || !shift_h || !label_h || !ids_h)
    return RSX_EINVAL;
  const uint64_t null_hash = rsx_fingerprint64_h(reinterpret_cast<const uint8_t*>("NULL"), 4);
  const bool label_optional = (threads & 0x10000) != 0;      // serving requests carry no label (see rsx.h)
  threads &= 0xFFFF;
  std::atomic<int> status{RSX_OK};
  parallel_for(n, threads, [&](int64_t a, int64_t b) {
    for (int64_t r = a; r < b; ++r) {
      const int st = rsx_criteo_parse_row(buf_h + offsets_h[r], (size_t)lengths_h[r], slot_src_h, slot_rows_h, bnd_h,
                                          bnd_off_h, shift_h, F, null_hash, label_h + r,
                                          cont_log_h ? cont_log_h + r * 13 : nullptr, ids_h + r * F, label_optional);
      if (st != RSX_OK) status.store(st);
    }
  });
  return status.load();
}

// One DIN record (din/din.py:44-57): label, i_id, i_cate scalars (int64) and the VarLen histories, densified and zero
// padded / truncated to P like sparse_tensor_to_dense + batch.
int rsx_din_parse_row(const uint8_t* rec, size_t n, int P, int64_t* label, int64_t* i_id, int64_t* i_cate, int64_t* hist_i,
                      int64_t* hist_c) {
  bool got[3] = {false, false, false};
  std::memset(hist_i, 0, sizeof(int64_t) * P);
  std::memset(hist_c, 0, sizeof(int64_t) * P);
  auto is = [](Span k, const char* lit, size_t len) { return k.n == len && std::memcmp(k.p, lit, len) == 0; };
  const bool ok = for_each_feature(rec, n, [&](Span key, Span feat) {
    if (is(key, "label", 5)) got[0] = feat_int64s(feat, label, 1) >= 1;
    else if (is(key, "i_id", 4)) got[1] = feat_int64s(feat, i_id, 1) >= 1;
    else if (is(key, "i_cate", 6)) got[2] = feat_int64s(feat, i_cate, 1) >= 1;
    else if (is(key, "u_iid_seq", 9)) feat_int64s(feat, hist_i, P);
    else if (is(key, "u_icat_seq", 10)) feat_int64s(feat, hist_c, P);
  });
  return (ok && got[0] && got[1] && got[2]) ? RSX_OK : RSX_EDATA;
}

extern "C" int rsx_din_parse_h(const uint8_t* buf_h, const int64_t* offsets_h, const int64_t* lengths_h, int64_t n, int P,
                               int64_t* label_h, int64_t* i_id_h, int64_t* i_cate_h, int64_t* hist_i_h,
                               int64_t* hist_c_h, int threads) {
  if (n < 0 || P <= 0) return RSX_EINVAL;
  if (n == 0) return RSX_OK;
  if (!buf_h || !offsets_h || !lengths_h || !label_h || !i_id_h || !i_cate_h || !hist_i_h || !hist_c_h) return RSX_EINVAL;
  std::atomic<int> status{RSX_OK};
  parallel_for(n, threads, [&](int64_t a, int64_t b) {
    for (int64_t r = a; r < b; ++r) {
      const int st = rsx_din_parse_row(buf_h + offsets_h[r], (size_t)lengths_h[r], P, label_h + r, i_id_h + r, i_cate_h + r,
                                       hist_i_h + r * P, hist_c_h + r * P);
      if (st != RSX_OK) status.store(st);
    }
  });
  return status.load();
}

// deepfm/deepfm.py:28-33,53-56 (the script as committed): scalar int64 features by name.  out_h[k*n + r] = first value of
// feature names_h[k] in record r; a missing feature is RSX_EDATA (FixedLenFeature without default).
extern "C" int rsx_int64_features_parse_h(const uint8_t* buf_h, const int64_t* offsets_h, const int64_t* lengths_h, int64_t n,
                                          const char* const* names_h, int k, int64_t* out_h, int threads) {
  if (n < 0 || k <= 0 || k > 16) return RSX_EINVAL;
  if (n == 0) return RSX_OK;
  if (!buf_h || !offsets_h || !lengths_h || !names_h || !out_h) return RSX_EINVAL;
  size_t nlen[16];
  for (int i = 0; i < k; ++i) {
    if (!names_h[i]) return RSX_EINVAL;
    nlen[i] = std::strlen(names_h[i]);
  }
  std::atomic<int> status{RSX_OK};
  parallel_for(n, threads, [&](int64_t a, int64_t b) {
    for (int64_t r = a; r < b; ++r) {
      bool got[16] = {false};
      const bool ok = for_each_feature(buf_h + offsets_h[r], (size_t)lengths_h[r], [&](Span key, Span feat) {
        for (int i = 0; i < k; ++i)
          if (key.n == nlen[i] && std::memcmp(key.p, names_h[i], nlen[i]) == 0) got[i] = feat_int64s(feat, out_h + (size_t)i * n + r, 1) >= 1;
      });
      bool all = ok;
      for (int i = 0; i < k; ++i) all = all && got[i];
      if (!all) status.store(RSX_EDATA);
    }
  });
  return status.load();
}

// categorical_column_with_hash_bucket(key, buckets, dtype=int64) (deepfm/deepfm.py:41,46): TF formats the integer as a
// decimal string (as_string) and hashes that: id = Fingerprint64(str(key)) % buckets  (SURVEY.md Appendix A-2).
extern "C" int rsx_hash_int64_keys_h(const int64_t* keys_h, int64_t n, uint64_t buckets, int32_t* out_h) {
  if (n < 0 || buckets == 0 || buckets > 0x7fffffffull || (n > 0 && (!keys_h || !out_h))) return RSX_EINVAL;
  for (int64_t i = 0; i < n; ++i) {
    char tmp[24];
    int64_t v = keys_h[i];
    uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    int p = 24;
    do { tmp[--p] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) tmp[--p] = '-';
    out_h[i] = (int32_t)(rsx_fingerprint64_h(reinterpret_cast<const uint8_t*>(tmp + p), (size_t)(24 - p)) % buckets);
  }
  return RSX_OK;
}

static void frame_into(std::string& out, const std::string& ex) {
  uint64_t len = ex.size();
  char hdr[8];
  std::memcpy(hdr, &len, 8);
  uint32_t c = rsx_masked_crc32c_h(reinterpret_cast<const uint8_t*>(hdr), 8);
  out.append(hdr, 8);
  out.append(reinterpret_cast<const char*>(&c), 4);
  out += ex;
  c = rsx_masked_crc32c_h(reinterpret_cast<const uint8_t*>(ex.data()), ex.size());
  out.append(reinterpret_cast<const char*>(&c), 4);
}
static void add_feature(std::string& feats, const std::string& key, uint32_t kind, const std::string& list_payload) {
  std::string lst, feat, entry;
  put_ld(lst, 1, list_payload);       // value = 1 (packed for float/int64)
  put_ld(feat, kind, lst);
  put_ld(entry, 1, key);
  put_ld(entry, 2, feat);
  put_ld(feats, 1, entry);
}

// Writer for synthetic Criteo shards with the schema of fm/fm.py:39-44: floats _c0.._c13 as float_list,
// strings _c14.._c39 as bytes_list, values equal to "NULL" omitted like the Spark connector omits nulls.
// Returns bytes written, or -needed when cap is too small, or a negative rsx_status.
extern "C" int64_t rsx_criteo_encode_h(const float* label_h, const float* cont_h, const uint8_t* cat_bytes_h,
                                       const int64_t* cat_offs_h, int64_t n, uint8_t* out_h, int64_t cap) {
  if (n < 0 || (n > 0 && (!label_h || !cont_h || !cat_offs_h))) return RSX_EINVAL;
  std::string out;
  for (int64_t r = 0; r < n; ++r) {
    std::string feats, ex;
    for (int j = 0; j <= 13; ++j) {
      const float v = j == 0 ? label_h[r] : cont_h[r * 13 + j - 1];
      add_feature(feats, "_c" + std::to_string(j), 2, std::string(reinterpret_cast<const char*>(&v), 4));
    }
    for (int j = 14; j <= 39; ++j) {
      const int64_t a = cat_offs_h[r * 26 + j - 14], b = cat_offs_h[r * 26 + j - 14 + 1];
      if (b - a == 4 && std::memcmp(cat_bytes_h + a, "NULL", 4) == 0) continue;
      std::string lst, feat, entry;
      put_ld(lst, 1, std::string(reinterpret_cast<const char*>(cat_bytes_h + a), (size_t)(b - a)));
      put_ld(feat, 1, lst);
      put_ld(entry, 1, "_c" + std::to_string(j));
      put_ld(entry, 2, feat);
      put_ld(feats, 1, entry);
    }
    put_ld(ex, 1, feats);
    frame_into(out, ex);
  }
  if ((int64_t)out.size() > cap || !out_h) return -(int64_t)out.size() - 16;
  std::memcpy(out_h, out.data(), out.size());
  return (int64_t)out.size();
}

// Writer for synthetic DIN shards (din/din.py:44-50): int64 label, i_id, i_cate and the two VarLen histories
// (trailing zero padding is dropped, so the sequences are really variable-length on disk).
extern "C" int64_t rsx_din_encode_h(const int64_t* label_h, const int64_t* i_id_h, const int64_t* i_cate_h,
                                    const int64_t* hist_i_h, const int64_t* hist_c_h, int64_t n, int P, int keep_padding,
                                    uint8_t* out_h, int64_t cap) {
  if (n < 0 || P <= 0) return RSX_EINVAL;
  std::string out;
  auto ints = [](const int64_t* v, int64_t cnt) {
    std::string s;
    for (int64_t i = 0; i < cnt; ++i) put_varint(s, (uint64_t)v[i]);
    return s;
  };
  for (int64_t r = 0; r < n; ++r) {
    std::string feats, ex;
    add_feature(feats, "label", 3, ints(label_h + r, 1));
    add_feature(feats, "i_id", 3, ints(i_id_h + r, 1));
    add_feature(feats, "i_cate", 3, ints(i_cate_h + r, 1));
    int64_t len = P;
    if (!keep_padding) while (len > 0 && hist_i_h[r * P + len - 1] == 0) --len;
    add_feature(feats, "u_iid_seq", 3, ints(hist_i_h + r * P, len));
    add_feature(feats, "u_icat_seq", 3, ints(hist_c_h + r * P, len));
    put_ld(ex, 1, feats);
    frame_into(out, ex);
  }
  if ((int64_t)out.size() > cap || !out_h) return -(int64_t)out.size() - 16;
  std::memcpy(out_h, out.data(), out.size());
  return (int64_t)out.size();
}
