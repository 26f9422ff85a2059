// Host-side ingest helpers of librsx.so (no device code): FarmHash Fingerprint64, log-bucketize,
// CRC-32C.  Reference call sites: categorical_column_with_hash_bucket fm/fm.py:89,
// bucketized_column fm/fm.py:76-79, TFRecordDataset fm/fm.py:107.
// The hash is the published farmhashna::Hash64 (what TF's Fingerprint64 wraps).
#include <cmath>
#include <cstring>

#include "rsx.h"

namespace {
typedef uint64_t u64;
constexpr u64 k0 = 0xc3a5c85c97cb3127ULL, k1 = 0xb492b66fbe98f273ULL, k2 = 0x9ae16a3b2f90404fULL;

inline u64 ld64(const uint8_t* p) { u64 v; std::memcpy(&v, p, 8); return v; }
inline uint32_t ld32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
inline u64 rot(u64 v, int s) { return s == 0 ? v : (v >> s) | (v << (64 - s)); }
inline u64 smix(u64 v) { return v ^ (v >> 47); }
inline u64 h16(u64 u, u64 v, u64 mul) {
  u64 a = (u ^ v) * mul; a ^= a >> 47;
  u64 b = (v ^ a) * mul; b ^= b >> 47;
  return b * mul;
}
struct P2 { u64 a, b; };
inline P2 weak32(const uint8_t* s, u64 a, u64 b) {
  const u64 w = ld64(s), x = ld64(s + 8), y = ld64(s + 16), z = ld64(s + 24);
  a += w; b = rot(b + a + z, 21);
  const u64 c = a;
  a += x; a += y; b += rot(a, 44);
  return {a + z, b + c};
}
u64 fp64(const uint8_t* s, size_t n) {
  if (n <= 16) {
    if (n >= 8) {
      const u64 mul = k2 + n * 2, a = ld64(s) + k2, b = ld64(s + n - 8);
      return h16(rot(b, 37) * mul + a, (rot(a, 25) + b) * mul, mul);
    }
    if (n >= 4) {
      const u64 mul = k2 + n * 2, a = ld32(s);
      return h16(n + (a << 3), ld32(s + n - 4), mul);
    }
    if (n > 0) {
      const uint32_t y = (uint32_t)s[0] + ((uint32_t)s[n >> 1] << 8), z = (uint32_t)n + ((uint32_t)s[n - 1] << 2);
      return smix(y * k2 ^ z * k0) * k2;
    }
    return k2;
  }
  if (n <= 32) {
    const u64 mul = k2 + n * 2, a = ld64(s) * k1, b = ld64(s + 8), c = ld64(s + n - 8) * mul, d = ld64(s + n - 16) * k2;
    return h16(rot(a + b, 43) + rot(c, 30) + d, a + rot(b + k2, 18) + c, mul);
  }
  if (n <= 64) {
    const u64 mul = k2 + n * 2, a = ld64(s) * k2, b = ld64(s + 8), c = ld64(s + n - 8) * mul, d = ld64(s + n - 16) * k2;
    const u64 y = rot(a + b, 43) + rot(c, 30) + d, z = h16(y, a + rot(b + k2, 18) + c, mul);
    const u64 e = ld64(s + 16) * mul, f = ld64(s + 24), g = (y + ld64(s + n - 32)) * mul, h = (z + ld64(s + n - 24)) * mul;
    return h16(rot(e + f, 43) + rot(g, 30) + h, e + rot(f + a, 18) + g, mul);
  }
  const u64 seed = 81;
  u64 x = seed, y = seed * k1 + 113, z = smix(y * k2 + 113) * k2;
  P2 v{0, 0}, w{0, 0};
  x = x * k2 + ld64(s);
  const uint8_t* end = s + ((n - 1) / 64) * 64;
  const uint8_t* last64 = end + ((n - 1) & 63) - 63;
  do {
    x = rot(x + y + v.a + ld64(s + 8), 37) * k1;
    y = rot(y + v.b + ld64(s + 48), 42) * k1;
    x ^= w.b;
    y += v.a + ld64(s + 40);
    z = rot(z + w.a, 33) * k1;
    v = weak32(s, v.b * k1, x + w.a);
    w = weak32(s + 32, z + w.b, y + ld64(s + 16));
    const u64 t = z; z = x; x = t;
    s += 64;
  } while (s != end);
  const u64 mul = k1 + ((z & 0xff) << 1);
  s = last64;
  w.a += (n - 1) & 63;
  v.a += w.a;
  w.a += v.a;
  x = rot(x + y + v.a + ld64(s + 8), 37) * mul;
  y = rot(y + v.b + ld64(s + 48), 42) * mul;
  x ^= w.b * 9;
  y += v.a * 9 + ld64(s + 40);
  z = rot(z + w.a, 33) * mul;
  v = weak32(s, v.b * mul, x + w.a);
  w = weak32(s + 32, z + w.b, y + ld64(s + 16));
  const u64 t = z; z = x; x = t;
  return h16(h16(v.a, w.a, mul) + smix(y) * k0 + z, h16(v.b, w.b, mul) + x, mul);
}

uint32_t crc_tab[8][256];
bool crc_ready = [] {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
    crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) crc_tab[t][i] = (crc_tab[t - 1][i] >> 8) ^ crc_tab[0][crc_tab[t - 1][i] & 0xff];
  return true;
}();
}  // namespace

extern "C" uint64_t rsx_fingerprint64_h(const uint8_t* s, size_t n) { return fp64(s, n); }

extern "C" int rsx_hash_fp64_h(const uint8_t* bytes_h, const int64_t* offs_h, int64_t n, uint64_t* out_h) {
  if (n < 0 || (n > 0 && (!offs_h || !out_h))) return RSX_EINVAL;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t a = offs_h[i], b = offs_h[i + 1];
    if (b < a) return RSX_EINVAL;
    out_h[i] = fp64(bytes_h + a, (size_t)(b - a));
  }
  return RSX_OK;
}

extern "C" int rsx_bucketize_log_h(const float* x_h, int64_t n, const float* boundaries_h, int nb, float shift,
                                   int32_t* out_h) {
  if (n < 0 || nb < 0 || (n > 0 && (!x_h || !out_h)) || (nb > 0 && !boundaries_h)) return RSX_EINVAL;
  for (int64_t i = 0; i < n; ++i) {
    const float v = logf(x_h[i] + shift);
    int idx = 0;  // std::upper_bound: first boundary > v; NaN compares false everywhere -> nb
    if (v != v) {
      idx = nb;
    } else {
      int lo = 0, hi = nb;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (boundaries_h[mid] <= v) lo = mid + 1; else hi = mid;
      }
      idx = lo;
    }
    out_h[i] = idx;
  }
  return RSX_OK;
}

static uint32_t crc32c_table(const uint8_t* p, size_t n) {
  (void)crc_ready;
  uint32_t c = 0xFFFFFFFFu;
  while (n >= 8) {  // slicing-by-8
    uint64_t w;
    std::memcpy(&w, p, 8);
    w ^= c;
    c = crc_tab[7][w & 0xff] ^ crc_tab[6][(w >> 8) & 0xff] ^ crc_tab[5][(w >> 16) & 0xff] ^ crc_tab[4][(w >> 24) & 0xff] ^
        crc_tab[3][(w >> 32) & 0xff] ^ crc_tab[2][(w >> 40) & 0xff] ^ crc_tab[1][(w >> 48) & 0xff] ^ crc_tab[0][w >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) c = crc_tab[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

// CRC-32C is the polynomial of the x86 `crc32` instruction (SSE4.2): 8 bytes per instruction, 3-cycle latency.  Three
// independent streams would hide the latency; records are ~800 B and are parsed right after, so the plain chain
// (~2.7 GB/s per core, 3x the table version) already leaves the CRC well below the parse cost.
__attribute__((target("sse4.2"))) static uint32_t crc32c_hw(const uint8_t* p, size_t n) {
  uint64_t c = 0xFFFFFFFFu;
  while (n >= 8) {
    uint64_t w;
    std::memcpy(&w, p, 8);
    c = __builtin_ia32_crc32di(c, w);
    p += 8;
    n -= 8;
  }
  uint32_t c32 = (uint32_t)c;
  while (n--) c32 = __builtin_ia32_crc32qi(c32, *p++);
  return c32 ^ 0xFFFFFFFFu;
}

extern "C" uint32_t rsx_crc32c_h(const uint8_t* p, size_t n) {
  static const bool hw = __builtin_cpu_supports("sse4.2");
  return hw ? crc32c_hw(p, n) : crc32c_table(p, n);
}

// test hook: the table implementation regardless of the CPU (the two must agree bit for bit)
extern "C" uint32_t rsx_crc32c_table_h(const uint8_t* p, size_t n) { return crc32c_table(p, n); }

extern "C" uint32_t rsx_masked_crc32c_h(const uint8_t* p, size_t n) {
  const uint32_t c = rsx_crc32c_h(p, n);
  return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}
