// Minimal C++ driver of the C ABI (debug tool): create, fill, ntt, download a few words.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/homulator_hip.h"
#define CK(x) do { int s_ = (x); if (s_) { printf("FAIL %s -> %d: %s\n", #x, s_, hm_last_error(ctx)); return 1; } printf("ok %s\n", #x); fflush(stdout); } while (0)
int main() {
  hm_ctx *ctx = nullptr;
  hm_params p = {16, 4, 2, 0, nullptr, nullptr, nullptr};
  CK(hm_create(&ctx, &p));
  void *d = nullptr;
  CK(hm_malloc(ctx, 8ull * 65536 * 4, &d));
  uint32_t mods[4] = {0, 1, 2, 3};
  CK(hm_fill_uniform(ctx, (uint64_t *)d, nullptr, mods, 4, 1));
  CK(hm_sync(ctx));
  CK(hm_ntt(ctx, (uint64_t *)d, nullptr, (uint64_t *)d, nullptr, mods, 4, 0, nullptr));
  CK(hm_sync(ctx));
  CK(hm_ntt(ctx, (uint64_t *)d, nullptr, (uint64_t *)d, nullptr, mods, 4, 1, nullptr));
  CK(hm_sync(ctx));
  std::vector<uint64_t> h(8);
  CK(hm_memcpy_d2h(ctx, h.data(), d, 64));
  for (auto v : h) printf("%llu\n", (unsigned long long)v);
  hm_destroy(ctx);
  return 0;
}
