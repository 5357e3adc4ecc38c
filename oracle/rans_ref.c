/* TEST INFRASTRUCTURE ONLY (oracle) - never linked into the product library.
 *
 * Plain-C restatement of the reference's rANS64 entropy coder
 *   /root/reference/cra5/models/compressai/cpp_exts/rans/rans_interface.cpp
 *     :69-105  bypass put/get bits      :108-173 symbol buffering (escape coding)
 *     :175-200 flush (reverse order)    :215-284 decode_with_indexes
 * plus the six primitives of ryg_rans `rans64.h` (F. Giesen, public domain) that
 * the reference includes from an UN-VENDORED submodule (rans_interface.hpp:36 ->
 * third_party/ryg_rans, setup.py:68; no version pin in the reference tree).  Their
 * published definitions are restated inline below.
 *
 * PARITY UNPINNED at the byte level: the reference extension cannot be built here
 * (missing header) and the reference ships no known-answer stream.  This file is
 * cross-checked against an independent pure-Python restatement (oracle/rans_py.py)
 * and by encode->decode round trips on adversarial inputs (tests/test_rans.py).
 *
 * Faithful quirks kept: symbols buffered forward / coded in reverse; output is the
 * tail of a uint32 buffer; decoder does the LINEAR cdf scan of :246-250.
 * Guarded quirk: the reference sizes the word buffer as #buffered symbols, which
 * under-runs for < 2 symbols (flush writes 2 words); we allocate n_syms + 2.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RANS64_L (1ull << 31) /* rans64.h */
enum { PRECISION = 16, BYPASS_PRECISION = 4, MAX_BYPASS_VAL = 15 };

typedef struct { uint16_t start, range; uint8_t bypass; } sym_t;

/* rans64.h Rans64EncPut */
static inline void enc_put(uint64_t *r, uint32_t **pp, uint32_t start, uint32_t freq, uint32_t bits) {
  uint64_t x = *r;
  uint64_t x_max = ((RANS64_L >> bits) << 32) * freq;
  if (x >= x_max) { *pp -= 1; **pp = (uint32_t)x; x >>= 32; }
  *r = ((x / freq) << bits) + (x % freq) + start;
}
/* rans_interface.cpp:69-87 */
static inline void enc_put_bits(uint64_t *r, uint32_t **pp, uint32_t val, uint32_t nbits) {
  uint64_t x = *r;
  uint32_t freq = 1u << (16 - nbits);
  uint64_t x_max = ((RANS64_L >> 16) << 32) * freq;
  if (x >= x_max) { *pp -= 1; **pp = (uint32_t)x; x >>= 32; }
  *r = (x << nbits) | val;
}
/* rans_interface.cpp:89-105 */
static inline uint32_t dec_get_bits(uint64_t *r, const uint32_t **pp, uint32_t nbits) {
  uint64_t x = *r;
  uint32_t val = (uint32_t)(x & ((1u << nbits) - 1));
  x >>= nbits;
  if (x < RANS64_L) { x = (x << 32) | **pp; *pp += 1; }
  *r = x;
  return val;
}

/* Returns malloc'ed stream in *out (caller frees with oracle_free), 0 on success. */
int oracle_rans_encode(const int32_t *symbols, const int32_t *indexes, size_t n,
                       const int32_t *cdfs, int n_cdfs, int cdf_stride,
                       const int32_t *cdf_sizes, const int32_t *offsets,
                       uint8_t **out, size_t *out_len) {
  size_t cap = n + 16, ns = 0;
  sym_t *syms = (sym_t *)malloc(cap * sizeof(sym_t));
  if (!syms) return -1;
#define PUSH(a, b, c)                                                        \
  do {                                                                       \
    if (ns == cap) { cap *= 2; syms = (sym_t *)realloc(syms, cap * sizeof(sym_t)); } \
    syms[ns].start = (uint16_t)(a); syms[ns].range = (uint16_t)(b); syms[ns].bypass = (c); ns++; \
  } while (0)
  for (size_t i = 0; i < n; ++i) { /* rans_interface.cpp:117-172 */
    int32_t ci = indexes[i];
    if (ci < 0 || ci >= n_cdfs) { free(syms); return -2; }
    const int32_t *cdf = cdfs + (size_t)ci * cdf_stride;
    int32_t max_value = cdf_sizes[ci] - 2;
    int32_t value = symbols[i] - offsets[ci];
    uint32_t raw_val = 0;
    if (value < 0) { raw_val = (uint32_t)(-2 * value - 1); value = max_value; }
    else if (value >= max_value) { raw_val = (uint32_t)(2 * (value - max_value)); value = max_value; }
    PUSH(cdf[value], cdf[value + 1] - cdf[value], 0);
    if (value == max_value) {
      /* raw_val >= 2^28 needs 8 nibbles: the reference's loop below then evaluates `raw_val >> 32`
       * on a uint32_t (undefined; x86 masks the count to 0 and the loop never ends,
       * rans_interface.cpp:152-154).  Outside the reference's defined domain: refuse instead of
       * reproducing the hang (|symbol - offset| < 2^27 is always safe). */
      if (raw_val >> 28) { free(syms); return -3; }
      int32_t n_bypass = 0;
      while ((raw_val >> (n_bypass * BYPASS_PRECISION)) != 0) ++n_bypass;
      int32_t val = n_bypass;
      while (val >= MAX_BYPASS_VAL) { PUSH(MAX_BYPASS_VAL, MAX_BYPASS_VAL + 1, 1); val -= MAX_BYPASS_VAL; }
      PUSH(val, val + 1, 1);
      for (int32_t j = 0; j < n_bypass; ++j) {
        int32_t v = (raw_val >> (j * BYPASS_PRECISION)) & MAX_BYPASS_VAL;
        PUSH(v, v + 1, 1);
      }
    }
  }
#undef PUSH
  /* flush: rans_interface.cpp:175-200 */
  size_t nwords = ns + 2;
  uint32_t *buf = (uint32_t *)malloc(nwords * sizeof(uint32_t));
  if (!buf) { free(syms); return -1; }
  uint32_t *ptr = buf + nwords;
  uint64_t rans = RANS64_L; /* Rans64EncInit */
  for (size_t k = ns; k-- > 0;) {
    if (!syms[k].bypass) enc_put(&rans, &ptr, syms[k].start, syms[k].range, PRECISION);
    else enc_put_bits(&rans, &ptr, syms[k].start, BYPASS_PRECISION);
  }
  ptr -= 2; ptr[0] = (uint32_t)rans; ptr[1] = (uint32_t)(rans >> 32); /* Rans64EncFlush */
  size_t nbytes = (size_t)((buf + nwords) - ptr) * sizeof(uint32_t);
  *out = (uint8_t *)malloc(nbytes ? nbytes : 1);
  memcpy(*out, ptr, nbytes);
  *out_len = nbytes;
  free(buf); free(syms);
  return 0;
}

void oracle_free(void *p) { free(p); }

int oracle_rans_decode(const uint8_t *encoded, size_t len, const int32_t *indexes, size_t n,
                       const int32_t *cdfs, int n_cdfs, int cdf_stride,
                       const int32_t *cdf_sizes, const int32_t *offsets, int32_t *out) {
  (void)len;
  uint32_t *words = (uint32_t *)malloc(len + 8); /* aligned copy (reference casts in place, :227) */
  memcpy(words, encoded, len);
  const uint32_t *ptr = words;
  uint64_t rans = (uint64_t)ptr[0] | ((uint64_t)ptr[1] << 32); ptr += 2; /* Rans64DecInit */
  for (size_t i = 0; i < n; ++i) {
    int32_t ci = indexes[i];
    if (ci < 0 || ci >= n_cdfs) { free(words); return -2; }
    const int32_t *cdf = cdfs + (size_t)ci * cdf_stride;
    int32_t csz = cdf_sizes[ci];
    int32_t max_value = csz - 2;
    uint32_t cum_freq = (uint32_t)(rans & ((1u << PRECISION) - 1)); /* Rans64DecGet */
    int32_t s = 0;
    while (s < csz && !((uint32_t)cdf[s] > cum_freq)) ++s; /* std::find_if, :246-250 */
    s -= 1;
    { /* Rans64DecAdvance */
      uint64_t mask = (1ull << PRECISION) - 1;
      uint32_t start = (uint32_t)cdf[s], freq = (uint32_t)(cdf[s + 1] - cdf[s]);
      uint64_t x = rans;
      x = freq * (x >> PRECISION) + (x & mask) - start;
      if (x < RANS64_L) { x = (x << 32) | *ptr; ptr += 1; }
      rans = x;
    }
    int32_t value = s;
    if (value == max_value) {
      int32_t val = (int32_t)dec_get_bits(&rans, &ptr, BYPASS_PRECISION);
      int32_t n_bypass = val;
      while (val == MAX_BYPASS_VAL) { val = (int32_t)dec_get_bits(&rans, &ptr, BYPASS_PRECISION); n_bypass += val; }
      int32_t raw_val = 0;
      for (int j = 0; j < n_bypass; ++j) { val = (int32_t)dec_get_bits(&rans, &ptr, BYPASS_PRECISION); raw_val |= val << (j * BYPASS_PRECISION); }
      value = raw_val >> 1;
      if (raw_val & 1) value = -value - 1; else value += max_value;
    }
    out[i] = value + offsets[ci];
  }
  free(words);
  return 0;
}
