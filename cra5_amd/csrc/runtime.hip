// Host <-> device frame copies for the single-frame API (cra5_api.encode_era5_as_bin / decode_from_bin on HOST
// arrays, cra5_api.py:81-125,153-192 in the reference, where `.to(device)` / `.cpu()` do this job).
//
// A 1.11 GB ERA5 frame in pageable memory moves at 8-20 GB/s through the runtime's own pageable path (and a fresh
// destination array adds 270 k page faults on one thread).  Here the copy is cut into chunks that flow through a
// PINNED staging buffer: a small team of host threads memcpy's chunk c + 1 while the DMA engine moves chunk c, so the
// whole transfer takes max(host memcpy, PCIe) instead of their sum.  Plain C ABI: raw pointers, sizes, a hipStream_t.
#include <hip/hip_runtime.h>

#include <sched.h>

#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/cra5_amd.h"

namespace {

inline void cpu_relax() {
#if defined(__x86_64__)
  __builtin_ia32_pause();
#endif
}

// wait loops: a few pauses, then give the core away (the waiters used to spin for the whole transfer - 7 busy cores per
// copy, and under an N-rank job's per-rank core share the spinners could starve the thread they wait for)
struct Backoff {
  int n = 0;
  void wait() {
    if (++n < 64) cpu_relax();
    else std::this_thread::yield();
  }
};

// the team never exceeds the CPUs this process may run on (a rank of an N-rank job is pinned to its NUMA share)
inline int cap_threads(int n_threads) {
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) {
    const int allowed = CPU_COUNT(&set);
    if (allowed >= 1 && n_threads > allowed) n_threads = allowed;
  }
  return n_threads < 1 ? 1 : n_threads;
}

// The chunks of one staged copy.  `ramp`: the first chunk is 1 MB and the sizes double up to `chunk` - the DMA engine
// starts after 10 us of host memcpy instead of after a whole 32 MB chunk's (host -> device: 20.4 -> 19.8 ms of a 19.3 ms
// transfer; the device -> host direction has nothing to wait for and keeps uniform chunks).
struct Slices {
  size_t bytes, chunk;
  int threads;
  std::vector<size_t> start;   // n_chunks + 1 offsets
  Slices(size_t bytes_, size_t chunk_, int threads_, bool ramp) : bytes(bytes_), chunk(chunk_), threads(threads_) {
    size_t c = ramp ? (size_t(1) << 20 < chunk_ ? size_t(1) << 20 : chunk_) : chunk_;
    for (size_t o = 0; o < bytes_;) {
      start.push_back(o);
      o += c;
      if (c < chunk_) c = (2 * c < chunk_) ? 2 * c : chunk_;
    }
    start.push_back(bytes_);
  }
  size_t n_chunks() const { return start.size() - 1; }
  size_t offset(size_t c) const { return start[c]; }
  size_t length(size_t c) const { return start[c + 1] - start[c]; }
  // thread t's byte range inside chunk c (64-byte aligned cuts)
  void range(size_t c, int t, size_t &lo, size_t &hi) const {
    const size_t c0 = start[c], len = start[c + 1] - c0;
    const size_t per = ((len + threads - 1) / threads + 63) & ~size_t(63);
    lo = c0 + (size_t(t) * per < len ? size_t(t) * per : len);
    hi = c0 + (size_t(t + 1) * per < len ? size_t(t + 1) * per : len);
  }
};

}  // namespace

extern "C" int cra5_copy_h2d_staged(void *dst_dev, const void *src_host, void *pinned, size_t bytes, size_t chunk_bytes,
                                    int n_threads, void *stream) {
  if (!dst_dev || !src_host || !pinned || bytes == 0 || n_threads < 1 || n_threads > 64) return CRA5_ERR_ARG;
  if (chunk_bytes < (1u << 16)) chunk_bytes = 1u << 16;
  n_threads = cap_threads(n_threads);
  const Slices S(bytes, chunk_bytes, n_threads, true);
  const size_t nc = S.n_chunks();
  std::vector<std::atomic<int>> done(nc);
  for (auto &d : done) d.store(0, std::memory_order_relaxed);
  auto work = [&](int t) {
    for (size_t c = 0; c < nc; ++c) {
      size_t lo, hi;
      S.range(c, t, lo, hi);
      if (hi > lo) std::memcpy(static_cast<char *>(pinned) + lo, static_cast<const char *>(src_host) + lo, hi - lo);
      done[c].fetch_add(1, std::memory_order_release);
    }
  };
  std::vector<std::thread> team;
  for (int t = 1; t < n_threads; ++t) team.emplace_back(work, t);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = 0;
  // the calling thread is team member 0 for chunk c, then hands chunk c to the DMA engine
  for (size_t c = 0; c < nc; ++c) {
    size_t lo, hi;
    S.range(c, 0, lo, hi);
    if (hi > lo) std::memcpy(static_cast<char *>(pinned) + lo, static_cast<const char *>(src_host) + lo, hi - lo);
    done[c].fetch_add(1, std::memory_order_release);
    for (Backoff b; done[c].load(std::memory_order_acquire) < n_threads;) b.wait();
    const size_t c0 = S.offset(c), len = S.length(c);
    if (!rc)
      rc = (int)hipMemcpyAsync(static_cast<char *>(dst_dev) + c0, static_cast<const char *>(pinned) + c0, len,
                               hipMemcpyHostToDevice, st);
  }
  for (auto &th : team) th.join();
  // Return only when the DMA engine has READ the last chunk out of `pinned` (earlier chunks are done by stream order):
  // the caller re-uses the same staging buffer for its next frame, and overwriting memory a DMA still reads would
  // corrupt the frame on the device silently.  Costs the tail of one chunk (~0.5 ms of a ~18 ms copy).
  if (!rc) {
    hipEvent_t ev = nullptr;
    rc = (int)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (!rc) rc = (int)hipEventRecord(ev, st);
    if (!rc) rc = (int)hipEventSynchronize(ev);
    if (ev) (void)hipEventDestroy(ev);
  }
  return rc;   // 0 or the hipError_t, like every device launcher
}

extern "C" int cra5_copy_d2h_staged(void *dst_host, const void *src_dev, void *pinned, size_t bytes, size_t chunk_bytes,
                                    int n_threads, void *stream) {
  if (!dst_host || !src_dev || !pinned || bytes == 0 || n_threads < 1 || n_threads > 64) return CRA5_ERR_ARG;
  if (chunk_bytes < (1u << 16)) chunk_bytes = 1u << 16;
  n_threads = cap_threads(n_threads);
  const Slices S(bytes, chunk_bytes, n_threads, false);
  const size_t nc = S.n_chunks();
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::vector<hipEvent_t> ev(nc, nullptr);
  int rc = 0;
  for (size_t c = 0; c < nc && !rc; ++c) {
    const size_t c0 = S.offset(c), len = S.length(c);
    rc = (int)hipEventCreateWithFlags(&ev[c], hipEventDisableTiming);
    if (!rc)
      rc = (int)hipMemcpyAsync(static_cast<char *>(pinned) + c0, static_cast<const char *>(src_dev) + c0, len,
                               hipMemcpyDeviceToHost, st);
    if (!rc) rc = (int)hipEventRecord(ev[c], st);
  }
  std::vector<std::atomic<int>> ready(nc);
  for (auto &r : ready) r.store(0, std::memory_order_relaxed);
  std::atomic<int> failed(rc);
  auto work = [&](int t) {
    for (size_t c = 0; c < nc; ++c) {
      for (Backoff b; !ready[c].load(std::memory_order_acquire);) {
        if (failed.load(std::memory_order_relaxed)) return;
        b.wait();
      }
      size_t lo, hi;
      S.range(c, t, lo, hi);
      if (hi > lo) std::memcpy(static_cast<char *>(dst_host) + lo, static_cast<const char *>(pinned) + lo, hi - lo);
    }
  };
  std::vector<std::thread> team;
  if (!rc)
    for (int t = 1; t < n_threads; ++t) team.emplace_back(work, t);
  for (size_t c = 0; c < nc && !rc; ++c) {
    rc = (int)hipEventSynchronize(ev[c]);
    if (rc) {
      failed.store(rc);
      break;
    }
    ready[c].store(1, std::memory_order_release);
    size_t lo, hi;
    S.range(c, 0, lo, hi);
    if (hi > lo) std::memcpy(static_cast<char *>(dst_host) + lo, static_cast<const char *>(pinned) + lo, hi - lo);
  }
  if (rc) failed.store(rc);
  for (auto &th : team) th.join();
  for (auto e : ev)
    if (e) (void)hipEventDestroy(e);
  return rc;   // 0 or the hipError_t, like every device launcher
}

// ---- shader-clock telemetry (bench.py: the sustained clock of the timed region travels with the JSON line) ----------
// SHORT probes, never a resident sampler wave: HIP multiplexes its streams onto a handful of hardware queues and a
// long-running kernel blocks every stream that shares its queue (a 4-second sampler wave stalled a quarter of the
// frame streams of the pipeline until it left).  One probe = one wave for `window_ticks` of the 100 MHz wall clock.
namespace {
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long *slot, int window) {
  if (threadIdx.x != 0) return;
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  unsigned long long w1;
  do {
    __builtin_amdgcn_s_sleep(8);
    w1 = wall_clock64();
  } while (w1 < w0 + window);
  const unsigned long long c1 = clock64();
  slot[0] = w0;
  slot[1] = c0;
  slot[2] = w1;
  slot[3] = c1;
}
__global__ void clock_stamp_kernel(unsigned long long *slot) {
  if (threadIdx.x == 0) *slot = wall_clock64();
}
}  // namespace

extern "C" int cra5_clock_probe(uint64_t *slot4_dev, int window_ticks, void *stream) {
  if (!slot4_dev || window_ticks <= 0 || window_ticks > 1000000) return CRA5_ERR_ARG;   // <= 10 ms per probe
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     reinterpret_cast<unsigned long long *>(slot4_dev), window_ticks);
  return (int)hipGetLastError();
}

extern "C" int cra5_clock_stamp(uint64_t *slot_dev, void *stream) {
  if (!slot_dev) return CRA5_ERR_ARG;
  hipLaunchKernelGGL(clock_stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     reinterpret_cast<unsigned long long *>(slot_dev));
  return (int)hipGetLastError();
}
