/* TEST INFRASTRUCTURE ONLY (oracle) - never linked into the product library.
 *
 * Plain-C restatement of the reference's pmf -> 16-bit quantised CDF routine,
 *   /root/reference/cra5/models/compressai/cpp_exts/ops/ops.cpp:40-108
 * Pinned against the reference itself: oracle/_ref/_CXX*.so is that very file
 * compiled where it lies (oracle/Makefile `ref`), and tests/golden/pmf_cdf.json
 * holds its outputs (tests/golden/make_golden.py).
 *
 * Returns 0 on success, -1 for a negative / non-finite entry (ops.cpp:46-52),
 * -2 if every entry rounds to zero (ops.cpp:60-64), -3 if no bin can donate.
 */
#include <math.h>
#include <stdint.h>

int oracle_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf /* n+1 */) {
  for (int i = 0; i < n; ++i)
    if (pmf[i] < 0 || !isfinite(pmf[i])) return -1;

  /* ops.cpp:54-58: cdf[0]=0; cdf[i+1] = round(p * 2^precision) (float arithmetic) */
  cdf[0] = 0;
  for (int i = 0; i < n; ++i) cdf[i + 1] = (uint32_t)roundf(pmf[i] * (float)(1 << precision));

  /* ops.cpp:60: std::accumulate(..., 0) -> the running sum is an `int` */
  int acc = 0;
  for (int i = 0; i <= n; ++i) acc = (int)((unsigned)acc + cdf[i]);
  const uint32_t total = (uint32_t)acc;
  if (total == 0) return -2;

  /* ops.cpp:66-69: renormalise with 64-bit integer division */
  for (int i = 0; i <= n; ++i) cdf[i] = (uint32_t)((((uint64_t)(1 << precision)) * cdf[i]) / total);

  /* ops.cpp:71-72: inclusive prefix sum, force the last entry */
  for (int i = 1; i <= n; ++i) cdf[i] += cdf[i - 1];
  cdf[n] = 1u << precision;

  /* ops.cpp:74-100: no zero-width bin; steal from the smallest bin with freq > 1
   * (first such bin on ties) */
  for (int i = 0; i < n; ++i) {
    if (cdf[i] == cdf[i + 1]) {
      uint32_t best_freq = ~0u;
      int best_steal = -1;
      for (int j = 0; j < n; ++j) {
        uint32_t freq = cdf[j + 1] - cdf[j];
        if (freq > 1 && freq < best_freq) {
          best_freq = freq;
          best_steal = j;
        }
      }
      if (best_steal < 0) return -3;
      if (best_steal < i) {
        for (int j = best_steal + 1; j <= i; ++j) cdf[j]--;
      } else {
        for (int j = i + 1; j <= best_steal; ++j) cdf[j]++;
      }
    }
  }
  return 0;
}
