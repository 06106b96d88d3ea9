// anyorder.hip — does hipExtAnyOrderLaunch let two independent kernels of ONE stream overlap on gfx950?  Two launches of 128 workgroups that
// spin for a fixed number of clock ticks each: back to back with the barrier bit they take 2 T, overlapped T.  Also: a third, ordinary launch
// behind them must still see both finished (it checks the flags they set).
// build: hipcc --offload-arch=gfx950 -O2 -o tools/anyorder tools/anyorder.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void spin(unsigned long long ticks, unsigned *done, unsigned slot) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) atomicAdd(&done[slot], 1u);
}
__global__ void check(const unsigned *done, unsigned expect, unsigned *bad) {
  if (threadIdx.x == 0 && (__atomic_load_n(&done[0], __ATOMIC_RELAXED) != expect || __atomic_load_n(&done[1], __ATOMIC_RELAXED) != expect)) atomicAdd(bad, 1u);
}
int main() {
  unsigned *done, *bad;
  CK(hipMalloc(&done, 8)); CK(hipMalloc(&bad, 4));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned long long ticks = 100ull * 100;   // wall_clock64: 100 MHz -> 100 us
  for (int mode = 0; mode < 3; ++mode) {          // 0: ordinary launches; 1: second launch any-order; 2: both any-order
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemsetAsync(done, 0, 8, s)); CK(hipMemsetAsync(bad, 0, 4, s));
      CK(hipEventRecord(e0, s));
      hipExtLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, nullptr, nullptr, mode == 2 ? hipExtAnyOrderLaunch : 0, ticks, done, 0u);
      hipExtLaunchKernelGGL(spin, dim3(128), dim3(256), 0, s, nullptr, nullptr, mode >= 1 ? hipExtAnyOrderLaunch : 0, ticks, done, 1u);
      hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, s, done, 128u, bad);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned b; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
      printf("mode %d rep %d: %.1f us  ordered-after-both=%s\n", mode, rep, ms * 1e3, b ? "NO" : "yes");
    }
  }
  return 0;
}
