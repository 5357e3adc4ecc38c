// Host-side entropy coding for the cra5_amd C ABI (include/cra5_amd.h).
//
// rANS64 coder with the wire semantics of the reference's `compressai.ans`
// (rans_interface.cpp:108-284 in taohan10200/CRA5: 16-bit probabilities, escape bin +
// 4-bit bypass nibbles, symbols coded in reverse, 32-bit little-endian word stream) and
// the pmf -> quantised-CDF routine of `compressai._CXX` (ops.cpp:40-108).
//
// Design differences from the reference (same bytes out):
//   * flat int32 arrays in, no Python-list marshalling, no GIL, no global state;
//   * the encoder never materialises the (start, range, bypass) symbol vector: one
//     counting pass sizes the word buffer, then one backward pass codes straight from
//     the symbol array (bypass nibbles are regenerated in reverse order);
//   * the decoder finds the bin with a binary search instead of the reference's linear
//     `find_if` scan (identical result on a strictly increasing CDF row);
//   * range-checked: bad table indexes / truncated streams return an error code.
#include "../../include/cra5_amd.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

namespace {

constexpr uint64_t kRansL = 1ull << 31;  // rans64.h RANS64_L
constexpr uint32_t kProbBits = 16;       // rans_interface.cpp:49
constexpr uint32_t kBypassBits = 4;      // rans_interface.cpp:51
constexpr uint32_t kBypassMax = (1u << kBypassBits) - 1;

struct Tables {
  const int32_t *cdfs;
  int n_cdfs;
  int stride;
  const int32_t *sizes;
  const int32_t *offsets;
};

// One coded symbol resolved against its table: bin + escape payload.
struct Resolved {
  uint32_t start, range;
  bool escape;
  uint32_t raw;
  int n_nibbles;
};

inline int nibbles_of(uint32_t raw) {
  int n = 0;
  while (n < 8 && (raw >> (n * kBypassBits)) != 0) ++n;
  return n;
}

inline Resolved resolve(const Tables &t, int32_t sym, int32_t ci) {
  const int32_t *cdf = t.cdfs + static_cast<size_t>(ci) * t.stride;
  const int32_t max_value = t.sizes[ci] - 2;
  int32_t value = sym - t.offsets[ci];
  Resolved r{0, 0, false, 0, 0};
  if (value < 0) {
    r.raw = static_cast<uint32_t>(-2 * value - 1);
    value = max_value;
  } else if (value >= max_value) {
    r.raw = static_cast<uint32_t>(2 * (value - max_value));
    value = max_value;
  }
  r.start = static_cast<uint32_t>(cdf[value]) & 0xFFFFu;
  r.range = static_cast<uint32_t>(cdf[value + 1] - cdf[value]) & 0xFFFFu;
  if (value == max_value) {
    r.escape = true;
    r.n_nibbles = nibbles_of(r.raw);
  }
  return r;
}

// Division by a bin's frequency as a multiplication (round 6).  Rans64EncPut computes x / freq and x % freq with a 64-bit
// divide on the coder's serial dependency chain (~20 cycles of the ~28 a symbol takes); the published rans64 encoder
// (Rans64EncSymbolInit / Rans64EncPutSymbol) replaces it by a multiplication with a pre-computed reciprocal - Alverson,
// "Integer division using reciprocals": q = mulhi(x, rcp) >> shift is EXACTLY floor(x / freq) for every state the coder
// can be in (x < 2^47 freq) - and folds the remainder into x + bias + q (2^16 - freq).  The resolved encoders get
// (start, freq) per symbol from the device, so the reciprocals live in one table over all 65 536 frequencies (1 MB, built
// once, thread-safe; a frame touches a few hundred of its lines).  Same states, same words, same streams.
struct Rcp {
  uint64_t rcp;
  uint32_t shift;
  uint32_t pad;
};
const Rcp *rcp_table() {
  static const std::vector<Rcp> tab = [] {
    std::vector<Rcp> t(65537);
    t[0] = Rcp{0, 0, 0};
    t[1] = Rcp{~0ull, 0, 0};                      // q = mulhi(x, 2^64 - 1) = x - 1; the bias below makes up for it
    for (uint32_t freq = 2; freq <= 65536; ++freq) {
      uint32_t shift = 0;
      while (freq > (1u << shift)) ++shift;
      // ceil(2^(shift + 63) / freq) by long division in two 32-bit steps
      uint64_t x0 = freq - 1;
      const uint64_t x1 = 1ull << (shift + 31);
      const uint64_t t1 = x1 / freq;
      x0 += (x1 % freq) << 32;
      const uint64_t t0 = x0 / freq;
      t[freq] = Rcp{t0 + (t1 << 32), shift - 1, 0};
    }
    return t;
  }();
  return tab.data();
}
inline uint64_t mulhi64(uint64_t a, uint64_t b) {
  return static_cast<uint64_t>((static_cast<unsigned __int128>(a) * b) >> 64);
}

struct Encoder {
  uint64_t x = kRansL;
  uint32_t *ptr;
  const Rcp *rcp = rcp_table();
  inline void put(uint32_t start, uint32_t freq) {
    // Rans64EncPutSymbol.  The renormalisation is taken for ~40 % of the symbols of a 13-bit/symbol stream:
    const uint64_t x_max = ((kRansL >> kProbBits) << 32) * freq;
    {
      // by select, not by branch: the word is written below the cursor either way (the buffers carry one spare word) and
      // cursor / state move when the state is due - a coin flip the predictor loses (default stream 13.7 -> 11.4 ms)
      const bool rn = x >= x_max;
      ptr[-1] = static_cast<uint32_t>(x);
      ptr -= rn ? 1 : 0;
      x = rn ? (x >> 32) : x;
    }
    const Rcp r = rcp[freq];
    const uint64_t q = mulhi64(x, r.rcp) >> r.shift;
    // x / freq * 2^16 + x % freq + start  ==  x + start + q (2^16 - freq); freq = 1: q = x - 1 -> + (2^16 - 1)
    x = x + start + (freq < 2 ? ((1u << kProbBits) - 1) : 0u) + q * ((1u << kProbBits) - freq);
  }
  // `total` (1..8) bypass nibbles at once - bits = nibble_first << 4 (total - 1) | ... | nibble_last, pushed first to
  // last: what `total` calls of put_bits() do, with the ONE renormalisation they can take (x >= 2^59 before a nibble;
  // after it the state is below 2^31 and seven more nibbles fit) placed where they would take it.
  inline void put_nibbles(uint32_t bits, int total) {
    const int msb = 63 - __builtin_clzll(x);
    int first = (62 - msb) >> 2;                  // nibbles that go in before x reaches 2^59
    if (first > total) first = total;
    const int rest = total - first;
    x = (x << (4 * first)) | (bits >> (4 * rest));
    {
      const bool rn = rest > 0;                   // (by select, like put(): default stream 11.7 -> 10.7 ms)
      ptr[-1] = static_cast<uint32_t>(x);
      ptr -= rn ? 1 : 0;
      x = rn ? (x >> 32) : x;
      x = (x << (4 * rest)) | (bits & ((1u << (4 * rest)) - 1));
    }
  }
  inline void put_bits(uint32_t val) {  // 4-bit bypass symbol
    constexpr uint32_t freq = 1u << (16 - kBypassBits);
    constexpr uint64_t x_max = ((kRansL >> 16) << 32) * freq;
    if (x >= x_max) {
      *--ptr = static_cast<uint32_t>(x);
      x >>= 32;
    }
    x = (x << kBypassBits) | val;
  }
};

int encode_impl(const int32_t *symbols, const int32_t *indexes, size_t n, const Tables &t,
                uint8_t **out, size_t *out_len) {
  if (!out || !out_len || (n && (!symbols || !indexes))) return CRA5_ERR_ARG;
  // One pass, last symbol first.  Every coded sub-symbol emits at most one 32-bit word and a symbol
  // has at most 1 bin + 1 count nibble + 8 payload nibbles (a uint32 payload), so 10 n + 2 words always
  // suffice; the words are written from the END of the buffer, only the pages actually reached are
  // ever touched (a frame: 106 MB reserved, ~4 MB used).  If that reservation fails, count first.
  size_t cap = 10 * n + 3;   // (+ 1: put() writes the word below the cursor before it knows whether it keeps it)
  uint32_t *buf = static_cast<uint32_t *>(std::malloc(cap * sizeof(uint32_t)));
  if (!buf) {
    size_t n_sub = 0;
    for (size_t i = 0; i < n; ++i) {
      const int32_t ci = indexes[i];
      if (ci < 0 || ci >= t.n_cdfs || t.sizes[ci] < 2 || t.sizes[ci] > t.stride) return CRA5_ERR_INDEX;
      const Resolved r = resolve(t, symbols[i], ci);
      n_sub += 1;
      if (r.escape) n_sub += static_cast<size_t>(r.n_nibbles) / kBypassMax + 1 + r.n_nibbles;
    }
    cap = n_sub + 3;
    buf = static_cast<uint32_t *>(std::malloc(cap * sizeof(uint32_t)));
    if (!buf) return CRA5_ERR_ALLOC;
  }
  Encoder e;
  e.ptr = buf + cap;
  // inside a symbol the reference pushes [bin, count nibbles (15,15,..,rem), payload nibbles
  // lsb-first] and pops in reverse.
  for (size_t i = n; i-- > 0;) {
    const int32_t ci = indexes[i];
    if (ci < 0 || ci >= t.n_cdfs || t.sizes[ci] < 2 || t.sizes[ci] > t.stride) {
      std::free(buf);
      return CRA5_ERR_INDEX;
    }
    const Resolved r = resolve(t, symbols[i], ci);
    if (r.range == 0) {   // malformed table (zero-width bin): would divide by zero in put()
      std::free(buf);
      return CRA5_ERR_INDEX;
    }
    if (r.escape) {
      for (int j = r.n_nibbles - 1; j >= 0; --j) e.put_bits((r.raw >> (j * kBypassBits)) & kBypassMax);
      const uint32_t full = static_cast<uint32_t>(r.n_nibbles) / kBypassMax;
      e.put_bits(static_cast<uint32_t>(r.n_nibbles) - full * kBypassMax);
      for (uint32_t k = 0; k < full; ++k) e.put_bits(kBypassMax);
    }
    e.put(r.start, r.range);
  }
  e.ptr -= 2;  // Rans64EncFlush
  e.ptr[0] = static_cast<uint32_t>(e.x);
  e.ptr[1] = static_cast<uint32_t>(e.x >> 32);
  const size_t nbytes = static_cast<size_t>((buf + cap) - e.ptr) * sizeof(uint32_t);
  uint8_t *res = static_cast<uint8_t *>(std::malloc(nbytes));
  if (!res) {
    std::free(buf);
    return CRA5_ERR_ALLOC;
  }
  std::memcpy(res, e.ptr, nbytes);
  std::free(buf);
  *out = res;
  *out_len = nbytes;
  return CRA5_OK;
}

struct Decoder {
  uint64_t x;
  const uint8_t *p, *end;
  bool ok = true;
  inline uint32_t word() {
    if (p + 4 > end) {
      ok = false;
      return 0;
    }
    uint32_t w;
    std::memcpy(&w, p, 4);  // no alignment assumption (the reference casts in place)
    p += 4;
    return w;
  }
  inline uint32_t get_bits() {
    const uint32_t val = static_cast<uint32_t>(x & kBypassMax);
    x >>= kBypassBits;
    if (x < kRansL) x = (x << 32) | word();
    return val;
  }
};

// Bin lookup for the decoder.  The reference scans the CDF row linearly (rans_interface.cpp:246-250), a binary search is
// ~6 unpredictable branches per symbol.  Every row gets, on first use, a table from the top bits of the cumulative
// frequency to a PACKED entry (round 6; rounds 2-5: the bin index alone, followed by loads of cdf[s + 1], cdf[s] on the
// coder's serial chain): bin | start << 16 | freq << 32, flagged EXACT when the whole bucket lies inside that bin - then
// the one 8-byte load is all the state update needs (a Gaussian row's mass sits in a few wide bins: 95-99 % of the
// symbols); otherwise the entry names the bin of the bucket's first value and the scan from there is 0-2 steps.  The
// table is a pure function of the row: the decoded symbols are the linear scan's.
// Measured on a real frame of the synthetic-weight model and on an entropy-matched stream (EPYC-class host, one core):
// see profiles/EXPERIMENTS.md, round 6.
constexpr uint64_t kExact = 1ull << 63;
constexpr int kLutBits = 10;               // 1024 buckets of 64 cumulative values: 8 KB per row in use (12 bits: the hot rows
                                           // of a frame fall out of L1 / L2 - measured slower on both test streams)
struct RowDec {
  const uint64_t *tab = nullptr;           // nullptr: malformed row -> generic search
  const int32_t *cdf = nullptr;
  int32_t csz = 0, max_value = 0, offset = 0;
  bool built = false;
};

static void build_row_dec(RowDec &r, std::vector<uint64_t> &store, const int32_t *cdf, int32_t csz, int32_t offset) {
  r.cdf = cdf;
  r.csz = csz;
  r.max_value = csz - 2;
  r.offset = offset;
  r.built = true;
  // a well-formed row is non-decreasing, starts at 0, ends within 2^16 and fits uint16 bins; anything else goes through
  // the generic search
  bool ok = csz <= 65535 && cdf[0] == 0 && cdf[csz - 1] <= (1 << kProbBits);
  for (int32_t k = 1; ok && k < csz; ++k) ok = cdf[k] >= cdf[k - 1];
  if (!ok) return;
  constexpr int shift = kProbBits - kLutBits;
  store.assign(static_cast<size_t>(1) << kLutBits, 0);
  int32_t s = 0;
  for (int32_t b = 0; b < (1 << kLutBits); ++b) {
    const int32_t lo = b << shift, hi = lo + (1 << shift) - 1;
    while (s + 1 < csz && cdf[s + 1] <= lo) ++s;   // largest s with cdf[s] <= lo (cdf[0] = 0)
    uint64_t e = static_cast<uint64_t>(s);
    if (s <= r.max_value && cdf[s + 1] > hi)       // the bucket lies inside bin s (a real bin: s + 1 < csz)
      e |= kExact | (static_cast<uint64_t>(cdf[s]) << 16) | (static_cast<uint64_t>(cdf[s + 1] - cdf[s]) << 32);
    store[static_cast<size_t>(b)] = e;
  }
  r.tab = store.data();
}

// IdxT / OutT: int32_t / int32_t is the reference's interface; uint8_t / int16_t the compact records of the frame path
// (a CDF index is < 256, a symbol almost always fits 16 bits: CRA5_ERR_RANGE when one does not - the caller then takes
// the 32-bit route)
template <class IdxT, class OutT>
int decode_symbols(Decoder &d, const IdxT *indexes, size_t n, const Tables &t, OutT *out) {
  constexpr uint64_t mask = (1ull << kProbBits) - 1;
  constexpr int shift = kProbBits - kLutBits;
  const size_t n_rows = static_cast<size_t>(t.n_cdfs > 0 ? t.n_cdfs : 0);
  std::vector<RowDec> rows(n_rows);
  std::vector<std::vector<uint64_t>> store(n_rows);
  for (size_t i = 0; i < n; ++i) {
    const int32_t ci = static_cast<int32_t>(indexes[i]);
    if (ci < 0 || ci >= t.n_cdfs) return CRA5_ERR_INDEX;
    RowDec &r = rows[static_cast<size_t>(ci)];
    if (!r.built) {
      if (t.sizes[ci] < 2 || t.sizes[ci] > t.stride) return CRA5_ERR_INDEX;
      build_row_dec(r, store[static_cast<size_t>(ci)], t.cdfs + static_cast<size_t>(ci) * t.stride, t.sizes[ci], t.offsets[ci]);
    }
    const int32_t cum = static_cast<int32_t>(d.x & mask);
    int32_t s;
    uint32_t start, freq;
    const uint64_t e = r.tab ? r.tab[cum >> shift] : 0;
    if (e & kExact) {
      s = static_cast<int32_t>(e & 0xFFFFu);
      start = static_cast<uint32_t>(e >> 16) & 0xFFFFu;
      freq = static_cast<uint32_t>(e >> 32) & 0x1FFFFu;
    } else {
      const int32_t *cdf = r.cdf;
      const int32_t csz = r.csz;
      if (r.tab) {
        s = static_cast<int32_t>(e & 0xFFFFu);
        // first entry > cum, minus one: 0-2 steps from the bucket's bin
        while (s + 1 < csz && cdf[s + 1] <= cum) ++s;
        if (cdf[csz - 1] <= cum) s = csz - 1;            // cum beyond the row's total: rejected below
      } else {
        s = static_cast<int32_t>(std::upper_bound(cdf, cdf + csz, cum) - cdf) - 1;
      }
      if (s < 0 || s > r.max_value) return CRA5_ERR_STREAM;
      start = static_cast<uint32_t>(cdf[s]);
      freq = static_cast<uint32_t>(cdf[s + 1] - cdf[s]);
    }
    d.x = freq * (d.x >> kProbBits) + static_cast<uint64_t>(cum) - start;
    {
      // renormalisation by select: the word below the cursor is read either way (from the last word of the stream when
      // the cursor is at its end) and taken when the state fell below 2^31 - ~40 % of the symbols, a coin flip as a branch
      const bool need = d.x < kRansL;
      const bool room = d.p + 4 <= d.end;
      uint32_t w;
      std::memcpy(&w, room ? d.p : d.end - 4, 4);
      d.ok = d.ok && (room || !need);
      d.x = need ? ((d.x << 32) | w) : d.x;
      d.p += need ? 4 : 0;
    }
    int32_t value = s;
    if (value == r.max_value) {
      // An escape = count nibble + n payload nibbles, each of which get_bits() follows with a renormalisation when the
      // state dropped below 2^31 - at most ONE of them can (the renormalised state is above 2^59: seven more nibbles
      // fit).  Where it falls is arithmetic: j0 = nibbles until the state is below 2^31.  So: the first min(j0, T) nibbles
      // as one mask / shift, the renormalisation by select, the rest as another mask / shift - straight-line code instead
      // of a loop whose trip count and per-nibble branches mispredict (default stream 28.6 -> 27.7 ms).
      uint32_t raw;
      int32_t n_bypass = static_cast<int32_t>(d.x & kBypassMax);
      const int T = n_bypass + 1;                               // count nibble + payload nibbles
      const int msb = 63 - __builtin_clzll(d.x | 1);
      const int j0 = ((msb - 31) >> 2) + 1;                     // (>= 1 for a state in range; a broken stream: slow path)
      if (n_bypass <= 8 && j0 >= 1 && T - j0 < 7) {
        const int t1 = j0 < T ? j0 : T;
        const int r2 = T - t1;
        uint64_t v = d.x & ((1ull << (4 * t1)) - 1);
        d.x >>= 4 * t1;
        {
          const bool need = d.x < kRansL;
          const bool room = d.p + 4 <= d.end;
          uint32_t w;
          std::memcpy(&w, room ? d.p : d.end - 4, 4);
          d.ok = d.ok && (room || !need);
          d.x = need ? ((d.x << 32) | w) : d.x;
          d.p += need ? 4 : 0;
        }
        v |= (d.x & ((1ull << (4 * r2)) - 1)) << (4 * t1);
        d.x >>= 4 * r2;
        raw = static_cast<uint32_t>(v >> kBypassBits);
      } else {
        uint32_t val = d.get_bits();
        n_bypass = static_cast<int32_t>(val);
        while (val == kBypassMax && d.ok) {
          val = d.get_bits();
          n_bypass += static_cast<int32_t>(val);
        }
        if (n_bypass > 8) return CRA5_ERR_STREAM;  // a uint32 payload has at most 8 nibbles
        raw = 0;
        for (int j = 0; j < n_bypass; ++j) raw |= d.get_bits() << (j * kBypassBits);
      }
      // raw odd: -(raw >> 1) - 1, even: (raw >> 1) + max_value - by mask, the sign bit of a payload is a coin flip
      const int32_t neg = -static_cast<int32_t>(raw & 1u);
      value = (static_cast<int32_t>(raw >> 1) ^ neg) + (r.max_value & ~neg);
    }
    if (!d.ok) return CRA5_ERR_STREAM;
    const int32_t sym = value + r.offset;
    if (sizeof(OutT) < sizeof(int32_t) && (sym < -32768 || sym > 32767)) return CRA5_ERR_RANGE;
    out[i] = static_cast<OutT>(sym);
  }
  return CRA5_OK;
}

int decoder_init(Decoder &d, const uint8_t *enc, size_t len) {
  if (!enc) return CRA5_ERR_ARG;
  if (len < 8) return CRA5_ERR_STREAM;
  d.p = enc;
  d.end = enc + len;
  d.ok = true;
  const uint64_t lo = d.word();
  const uint64_t hi = d.word();
  d.x = lo | (hi << 32);  // Rans64DecInit
  return CRA5_OK;
}

template <class IdxT, class OutT>
int decode_impl(const uint8_t *enc, size_t len, const IdxT *indexes, size_t n, const Tables &t, OutT *out) {
  if (n && (!indexes || !out)) return CRA5_ERR_ARG;
  Decoder d;
  const int rc = decoder_init(d, enc, len);
  if (rc) return rc;
  const int rs = decode_symbols(d, indexes, n, t, out);
  if (rs) return rs;
  // End-state check (the reference has none: rans_interface.cpp:215-284 hands back whatever it decoded).  rANS is a
  // bijection: the encoder started from RANS64_L (Rans64EncInit) and wrote exactly the words the decoder needs, so
  // after the last symbol of a stream decoded with the ENCODER's tables and indexes the state is RANS64_L again and
  // no word is left.  Anything else means the symbols handed back are not the ones that were coded: a CDF index that
  // differs from the encoder's (h_s evaluated on another platform), the wrong tables, or a damaged stream.
  if (d.x != kRansL || d.p != d.end) return CRA5_ERR_DESYNC;
  return CRA5_OK;
}

// ---- stateful objects: BufferedRansEncoder / RansDecoder.set_stream + decode_stream --------
struct BufferedSym {
  uint16_t start, range;
  bool bypass;
};
struct BufferedEncoder {
  std::vector<BufferedSym> syms;  // forward order, like the reference's _syms (rans_interface.cpp:142-170)
};
struct StreamDecoder {
  std::vector<uint8_t> stream;
  Decoder d;
  bool ready = false;
};

template <class F>
void parallel_for(int n, int n_threads, F &&f) {
  if (n_threads <= 1 || n <= 1) {
    for (int i = 0; i < n; ++i) f(i);
    return;
  }
  std::atomic<int> next{0};
  std::vector<std::thread> pool;
  const int nt = std::min(n_threads, n);
  pool.reserve(nt);
  for (int t = 0; t < nt; ++t)
    pool.emplace_back([&] {
      for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i);
    });
  for (auto &th : pool) th.join();
}

}  // namespace

extern "C" {

int cra5_abi_version(void) { return 1; }

int cra5_rans_encode_with_indexes(const int32_t *symbols, const int32_t *indexes, size_t n,
                                  const int32_t *cdfs, int n_cdfs, int cdf_stride,
                                  const int32_t *cdf_sizes, const int32_t *offsets, uint8_t **out,
                                  size_t *out_len) {
  if (!cdfs || !cdf_sizes || !offsets || n_cdfs <= 0 || cdf_stride < 2) return CRA5_ERR_ARG;
  return encode_impl(symbols, indexes, n, Tables{cdfs, n_cdfs, cdf_stride, cdf_sizes, offsets}, out, out_len);
}

int cra5_rans_encode_resolved_compact(const uint32_t *start_range, const uint16_t *rec16, size_t n, uint8_t **out,
                                      size_t *out_len) {
  if (!out || !out_len || (n && (!start_range || !rec16))) return CRA5_ERR_ARG;
  const size_t cap = 10 * n + 3;   // see encode_impl
  uint32_t *buf = static_cast<uint32_t *>(std::malloc(cap * sizeof(uint32_t)));
  if (!buf) return CRA5_ERR_ALLOC;
  Encoder e;
  e.ptr = buf + cap;
  for (size_t i = n; i-- > 0;) {
    const uint32_t sr = start_range[i];
    const uint32_t rec = rec16[i];
    if (rec) {
      if (rec == 0xFFFFu) {   // payload beyond 12 bits / invalid index: the caller uses the 32-bit records
        std::free(buf);
        return CRA5_ERR_RANGE;
      }
      const int n_nibbles = static_cast<int>(rec >> 12) - 1;
      const uint32_t r = rec & 0xFFFu;
      if (n_nibbles < 0 || n_nibbles > 3) {
        std::free(buf);
        return CRA5_ERR_INDEX;
      }
      // payload nibbles most significant first, then the count nibble: one shifted word (<= 4 nibbles)
      e.put_nibbles((r << kBypassBits) | static_cast<uint32_t>(n_nibbles), n_nibbles + 1);
    }
    const uint32_t freq = sr >> 16;
    if (!freq) {
      std::free(buf);
      return CRA5_ERR_INDEX;
    }
    e.put(sr & 0xFFFFu, freq);
  }
  e.ptr -= 2;  // Rans64EncFlush
  e.ptr[0] = static_cast<uint32_t>(e.x);
  e.ptr[1] = static_cast<uint32_t>(e.x >> 32);
  const size_t nbytes = static_cast<size_t>((buf + cap) - e.ptr) * sizeof(uint32_t);
  uint8_t *res = static_cast<uint8_t *>(std::malloc(nbytes));
  if (!res) {
    std::free(buf);
    return CRA5_ERR_ALLOC;
  }
  std::memcpy(res, e.ptr, nbytes);
  std::free(buf);
  *out = res;
  *out_len = nbytes;
  return CRA5_OK;
}

int cra5_rans_encode_resolved(const uint32_t *start_range, const uint32_t *raw, const uint8_t *esc, size_t n,
                              uint8_t **out, size_t *out_len) {
  if (!out || !out_len || (n && (!start_range || !raw || !esc))) return CRA5_ERR_ARG;
  const size_t cap = 10 * n + 3;   // see encode_impl
  uint32_t *buf = static_cast<uint32_t *>(std::malloc(cap * sizeof(uint32_t)));
  if (!buf) return CRA5_ERR_ALLOC;
  Encoder e;
  e.ptr = buf + cap;
  for (size_t i = n; i-- > 0;) {
    const uint32_t sr = start_range[i];
    const uint32_t ec = esc[i];
    if (ec) {
      if (ec > 9) {
        std::free(buf);
        return CRA5_ERR_INDEX;
      }
      const int n_nibbles = static_cast<int>(ec) - 1;
      const uint32_t r = raw[i];
      for (int j = n_nibbles - 1; j >= 0; --j) e.put_bits((r >> (j * kBypassBits)) & kBypassMax);
      e.put_bits(static_cast<uint32_t>(n_nibbles));   // n_nibbles <= 8 < 15: one count nibble
    }
    const uint32_t freq = sr >> 16;
    if (!freq) {   // a zero-width bin cannot be coded (malformed table)
      std::free(buf);
      return CRA5_ERR_INDEX;
    }
    e.put(sr & 0xFFFFu, freq);
  }
  e.ptr -= 2;  // Rans64EncFlush
  e.ptr[0] = static_cast<uint32_t>(e.x);
  e.ptr[1] = static_cast<uint32_t>(e.x >> 32);
  const size_t nbytes = static_cast<size_t>((buf + cap) - e.ptr) * sizeof(uint32_t);
  uint8_t *res = static_cast<uint8_t *>(std::malloc(nbytes));
  if (!res) {
    std::free(buf);
    return CRA5_ERR_ALLOC;
  }
  std::memcpy(res, e.ptr, nbytes);
  std::free(buf);
  *out = res;
  *out_len = nbytes;
  return CRA5_OK;
}

int cra5_rans_decode_with_indexes(const uint8_t *encoded, size_t len, const int32_t *indexes, size_t n,
                                  const int32_t *cdfs, int n_cdfs, int cdf_stride,
                                  const int32_t *cdf_sizes, const int32_t *offsets, int32_t *out) {
  if (!cdfs || !cdf_sizes || !offsets || n_cdfs <= 0 || cdf_stride < 2) return CRA5_ERR_ARG;
  return decode_impl(encoded, len, indexes, n, Tables{cdfs, n_cdfs, cdf_stride, cdf_sizes, offsets}, out);
}

int cra5_rans_decode_with_indexes_u8_i16(const uint8_t *encoded, size_t len, const uint8_t *indexes, size_t n,
                                         const int32_t *cdfs, int n_cdfs, int cdf_stride, const int32_t *cdf_sizes,
                                         const int32_t *offsets, int16_t *out) {
  if (!cdfs || !cdf_sizes || !offsets || n_cdfs <= 0 || cdf_stride < 2) return CRA5_ERR_ARG;
  return decode_impl(encoded, len, indexes, n, Tables{cdfs, n_cdfs, cdf_stride, cdf_sizes, offsets}, out);
}

int cra5_rans_encode_batch(int n_streams, const int32_t *const *symbols, const int32_t *const *indexes,
                           const size_t *n, const int32_t *const *cdfs, const int *n_cdfs,
                           const int *cdf_stride, const int32_t *const *cdf_sizes,
                           const int32_t *const *offsets, uint8_t **out, size_t *out_len, int *rc,
                           int n_threads) {
  if (n_streams < 0 || !rc) return CRA5_ERR_ARG;
  parallel_for(n_streams, n_threads, [&](int i) {
    rc[i] = cra5_rans_encode_with_indexes(symbols[i], indexes[i], n[i], cdfs[i], n_cdfs[i], cdf_stride[i],
                                          cdf_sizes[i], offsets[i], &out[i], &out_len[i]);
  });
  for (int i = 0; i < n_streams; ++i)
    if (rc[i]) return rc[i];
  return CRA5_OK;
}

int cra5_rans_decode_batch(int n_streams, const uint8_t *const *encoded, const size_t *len,
                           const int32_t *const *indexes, const size_t *n, const int32_t *const *cdfs,
                           const int *n_cdfs, const int *cdf_stride, const int32_t *const *cdf_sizes,
                           const int32_t *const *offsets, int32_t *const *out, int *rc, int n_threads) {
  if (n_streams < 0 || !rc) return CRA5_ERR_ARG;
  parallel_for(n_streams, n_threads, [&](int i) {
    rc[i] = cra5_rans_decode_with_indexes(encoded[i], len[i], indexes[i], n[i], cdfs[i], n_cdfs[i],
                                          cdf_stride[i], cdf_sizes[i], offsets[i], out[i]);
  });
  for (int i = 0; i < n_streams; ++i)
    if (rc[i]) return rc[i];
  return CRA5_OK;
}

void cra5_free(void *p) { std::free(p); }

/* BufferedRansEncoder (rans_interface.cpp:108-200): symbols of several calls (possibly with
 * different tables) are buffered, flush() codes them all in reverse into one stream. */
void *cra5_rans_encoder_new(void) { return new (std::nothrow) BufferedEncoder(); }
void cra5_rans_encoder_free(void *enc) { delete static_cast<BufferedEncoder *>(enc); }

int cra5_rans_encoder_push(void *enc, const int32_t *symbols, const int32_t *indexes, size_t n,
                           const int32_t *cdfs, int n_cdfs, int cdf_stride, const int32_t *cdf_sizes,
                           const int32_t *offsets) {
  if (!enc || !cdfs || !cdf_sizes || !offsets || n_cdfs <= 0 || cdf_stride < 2 || (n && (!symbols || !indexes)))
    return CRA5_ERR_ARG;
  auto *e = static_cast<BufferedEncoder *>(enc);
  const Tables t{cdfs, n_cdfs, cdf_stride, cdf_sizes, offsets};
  for (size_t i = 0; i < n; ++i) {
    const int32_t ci = indexes[i];
    if (ci < 0 || ci >= n_cdfs || cdf_sizes[ci] < 2 || cdf_sizes[ci] > cdf_stride) return CRA5_ERR_INDEX;
    if (resolve(t, symbols[i], ci).range == 0) return CRA5_ERR_INDEX;   // zero-width bin: nothing is buffered
  }
  for (size_t i = 0; i < n; ++i) {
    const Resolved r = resolve(t, symbols[i], indexes[i]);
    e->syms.push_back({static_cast<uint16_t>(r.start), static_cast<uint16_t>(r.range), false});
    if (r.escape) {
      uint32_t val = static_cast<uint32_t>(r.n_nibbles);
      while (val >= kBypassMax) {
        e->syms.push_back({static_cast<uint16_t>(kBypassMax), static_cast<uint16_t>(kBypassMax + 1), true});
        val -= kBypassMax;
      }
      e->syms.push_back({static_cast<uint16_t>(val), static_cast<uint16_t>(val + 1), true});
      for (int j = 0; j < r.n_nibbles; ++j) {
        const uint32_t v = (r.raw >> (j * kBypassBits)) & kBypassMax;
        e->syms.push_back({static_cast<uint16_t>(v), static_cast<uint16_t>(v + 1), true});
      }
    }
  }
  return CRA5_OK;
}

int cra5_rans_encoder_flush(void *enc, uint8_t **out, size_t *out_len) {
  if (!enc || !out || !out_len) return CRA5_ERR_ARG;
  auto *e = static_cast<BufferedEncoder *>(enc);
  const size_t cap = e->syms.size() + 3;   // (+ 1: put() writes one word below the cursor speculatively)
  uint32_t *buf = static_cast<uint32_t *>(std::malloc(cap * sizeof(uint32_t)));
  if (!buf) return CRA5_ERR_ALLOC;
  Encoder c;
  c.ptr = buf + cap;
  for (size_t k = e->syms.size(); k-- > 0;) {
    const BufferedSym &s = e->syms[k];
    if (!s.bypass) c.put(s.start, s.range);
    else c.put_bits(s.start);
  }
  c.ptr -= 2;
  c.ptr[0] = static_cast<uint32_t>(c.x);
  c.ptr[1] = static_cast<uint32_t>(c.x >> 32);
  const size_t nbytes = static_cast<size_t>((buf + cap) - c.ptr) * sizeof(uint32_t);
  uint8_t *res = static_cast<uint8_t *>(std::malloc(nbytes));
  if (!res) {
    std::free(buf);
    return CRA5_ERR_ALLOC;
  }
  std::memcpy(res, c.ptr, nbytes);
  std::free(buf);
  e->syms.clear();
  *out = res;
  *out_len = nbytes;
  return CRA5_OK;
}

/* RansDecoder.set_stream / decode_stream (rans_interface.cpp:286-359): the decoder keeps its
 * state between calls, so one stream can be decoded in several pieces with different tables. */
void *cra5_rans_decoder_new(void) { return new (std::nothrow) StreamDecoder(); }
void cra5_rans_decoder_free(void *dec) { delete static_cast<StreamDecoder *>(dec); }

int cra5_rans_decoder_set_stream(void *dec, const uint8_t *encoded, size_t len) {
  if (!dec || !encoded) return CRA5_ERR_ARG;
  auto *d = static_cast<StreamDecoder *>(dec);
  d->stream.assign(encoded, encoded + len);
  const int rc = decoder_init(d->d, d->stream.data(), d->stream.size());
  d->ready = (rc == CRA5_OK);
  return rc;
}

int cra5_rans_decoder_decode_stream(void *dec, const int32_t *indexes, size_t n, const int32_t *cdfs, int n_cdfs,
                                    int cdf_stride, const int32_t *cdf_sizes, const int32_t *offsets, int32_t *out) {
  if (!dec || !cdfs || !cdf_sizes || !offsets || n_cdfs <= 0 || cdf_stride < 2 || (n && (!indexes || !out)))
    return CRA5_ERR_ARG;
  auto *d = static_cast<StreamDecoder *>(dec);
  if (!d->ready) return CRA5_ERR_STREAM;
  return decode_symbols(d->d, indexes, n, Tables{cdfs, n_cdfs, cdf_stride, cdf_sizes, offsets}, out);
}

int cra5_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf) {
  if (!pmf || !cdf || n <= 0 || precision < 1 || precision > 16) return CRA5_ERR_ARG;
  for (int i = 0; i < n; ++i)
    if (!(pmf[i] >= 0.0f) || !std::isfinite(pmf[i])) return CRA5_ERR_PMF_DOMAIN;
  const float one = static_cast<float>(1 << precision);
  // frequencies first (kept separately so the fix-up below can work on widths)
  std::vector<uint32_t> freq(static_cast<size_t>(n));
  int total_i = 0;  // the reference accumulates into an `int` (std::accumulate(.., 0))
  for (int i = 0; i < n; ++i) {
    freq[i] = static_cast<uint32_t>(std::round(pmf[i] * one));
    total_i = static_cast<int>(static_cast<unsigned>(total_i) + freq[i]);
  }
  const uint32_t total = static_cast<uint32_t>(total_i);
  if (total == 0) return CRA5_ERR_PMF_ZERO;
  cdf[0] = 0;
  uint32_t run = 0;
  for (int i = 0; i < n; ++i) {
    run += static_cast<uint32_t>((static_cast<uint64_t>(1u << precision) * freq[i]) / total);
    cdf[i + 1] = run;
  }
  cdf[n] = 1u << precision;
  // zero-width bins steal one count from the narrowest bin that is wider than 1
  // (first such bin on ties), shifting the boundaries in between (ops.cpp:74-100)
  for (int i = 0; i < n; ++i) {
    if (cdf[i] != cdf[i + 1]) continue;
    uint32_t best = ~0u;
    int donor = -1;
    for (int j = 0; j < n; ++j) {
      const uint32_t w = cdf[j + 1] - cdf[j];
      if (w > 1 && w < best) {
        best = w;
        donor = j;
      }
    }
    if (donor < 0) return CRA5_ERR_PMF_STEAL;
    if (donor < i) {
      for (int j = donor + 1; j <= i; ++j) --cdf[j];
    } else {
      for (int j = i + 1; j <= donor; ++j) ++cdf[j];
    }
  }
  return CRA5_OK;
}

}  // extern "C"
