// Probe of ds_read_b64_tr_b16 on gfx950: which LDS halves does lane l receive, for a few address patterns?
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/tr_probe.hip -o build_variants/tr_probe ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned short *out, int mode) {
  __shared__ unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = lane * 8;                          // consecutive 8-byte slots
  else if (mode == 1) addr = (lane & 15) * 8 + (lane >> 4) * 1024;   // 16 lanes contiguous, groups 1 KB apart
  else if (mode == 2) addr = (lane & 15) * 64 + (lane >> 4) * 8;     // 16 lanes on rows of 64 B, groups step 8 B
  else addr = (lane & 15) * 72 + (lane >> 4) * 8;                    // rows of 72 B
  addr += (unsigned)(size_t)lds;  // LDS base offset (0 for the first shared array, kept for safety)
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
  unsigned short *d;
  hipMalloc(&d, 64 * 4 * 2);
  std::vector<unsigned short> h(256);
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    printf("mode %d (values are half-indexes into LDS)\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d: %5d %5d %5d %5d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : " |");
    }
  }
  return 0;
}
