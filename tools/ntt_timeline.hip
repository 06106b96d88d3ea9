// Development tool: per-workgroup phase timeline of one NTT pass (s_memtime stamps), to see how the workgroups of a
// launch line up in time.  Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ihomulator_amd/csrc -Iinclude tools/ntt_timeline.hip homulator_amd/csrc/hm_params.cpp -o tools/ntt_timeline
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hm_ntt_core.h"
#include "hm_params.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
#define NSTAMP 12

__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }

template <int LOGR, bool STRIDED, bool INV, int MODE>
__global__ void __launch_bounds__(HM_THREADS) k_pass(HmNttArgs a, uint64_t *stamps) {
  __shared__ __attribute__((aligned(16))) uint64_t lds[HM_LDS_WORDS];
  uint64_t ts[NSTAMP];
  const uint64_t rt0 = wall_clock64();
  ts[0] = now();
  const uint32_t tiles = 1u << (a.logN - HM_TILE_LOG);
  const uint32_t b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
  const uint32_t pair = slot / (2u * tiles), within = slot % (2u * tiles);
  const uint32_t tile = within >> 1, entry = pair * 16u + (within & 1u) * 8u + xcd;
  if (entry >= a.n_limbs) return;
  const int tid = threadIdx.x;
  const HmLimb lb = a.limb[entry];
  if (lb.mod == HM_NTT_NONE) return;
  const size_t N = (size_t)1 << a.logN;
  const uint64_t q = a.mods[lb.mod].q;
  const HmTw *twl = a.tw + (size_t)lb.mod * N;
  const uint32_t s0 = STRIDED ? 0u : (a.logN - 8u);
  const uint32_t prefix0 = STRIDED ? 0u : (tile << (HM_TILE_LOG - LOGR));
  constexpr bool FIRST = (STRIDED != INV);
  const uint64_t *src = FIRST ? a.in + (size_t)lb.in * N : a.out + (size_t)lb.out * N;
  uint64_t *dst = a.out + (size_t)lb.out * N;
  HmTw sc = {0, 0};
  HmEpi ep = hm_epi_none();
  HmNttState st;
  hm_ntt_phase<LOGR, STRIDED, INV, MODE, 0>(st, tid, lds, src, dst, tile, twl, s0, prefix0, q, sc, ep);
  ts[1] = now();                       // loads issued
  __builtin_amdgcn_s_waitcnt(0);       // vmcnt(0) lgkmcnt(0): data and both twiddle sets arrived
  ts[2] = now();
  hm_ntt_phase<LOGR, STRIDED, INV, MODE, 1>(st, tid, lds, src, dst, tile, twl, s0, prefix0, q, sc, ep);
  ts[3] = now();
  __syncthreads();
  ts[4] = now();
  hm_ntt_phase<LOGR, STRIDED, INV, MODE, 2>(st, tid, lds, src, dst, tile, twl, s0, prefix0, q, sc, ep);
  ts[5] = now();
  __syncthreads();
  ts[6] = now();
  hm_ntt_phase<LOGR, STRIDED, INV, MODE, 3>(st, tid, lds, src, dst, tile, twl, s0, prefix0, q, sc, ep);
  ts[7] = now();
  __builtin_amdgcn_s_waitcnt(0);
  ts[8] = now();
  uint32_t hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  ts[9] = hwid;
  ts[10] = rt0;
  ts[11] = wall_clock64();
  if ((tid & 63) == 0) {
    uint64_t *o = stamps + ((size_t)blockIdx.x * (HM_THREADS / 64) + (tid >> 6)) * NSTAMP;
    for (int i = 0; i < NSTAMP; ++i) o[i] = ts[i];
  }
}

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? atoi(argv[1]) : 35;
  const int strided = argc > 2 ? atoi(argv[2]) : 0;
  hm::Params P;
  P.init(16, 45, 15, nullptr, nullptr, nullptr);
  const uint32_t N = P.N, M = P.L + P.K;
  HmTw *d_tw; HmMod *d_mods; uint64_t *d_in, *d_out, *d_st;
  CK(hipMalloc(&d_tw, sizeof(HmTw) * (size_t)M * N));
  std::vector<HmTw> tmp(N);
  for (uint32_t m = 0; m < M; ++m) { P.make_table(m, false, tmp.data()); CK(hipMemcpy(d_tw + (size_t)m * N, tmp.data(), sizeof(HmTw) * N, hipMemcpyHostToDevice)); }
  CK(hipMalloc(&d_mods, sizeof(HmMod) * M)); CK(hipMemcpy(d_mods, P.modc.data(), sizeof(HmMod) * M, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_in, 8ull * N * n)); CK(hipMalloc(&d_out, 8ull * N * n)); CK(hipMemset(d_in, 1, 8ull * N * n)); CK(hipMemset(d_out, 1, 8ull * N * n));
  HmNttArgs a; a.in = d_in; a.out = d_out; a.tw = d_tw; a.mods = d_mods; a.logN = 16;
  const uint32_t np = (n + 1) / 2, cnt = ((np + 7) / 8) * 16;   // singles paired with each other, like ntt_common
  a.n_limbs = cnt;
  a.entry = nullptr; a.minuend = a.addend = a.mix = nullptr;
  for (uint32_t e = 0; e < cnt; ++e) a.limb[e] = HmLimb{0, 0, (uint16_t)HM_NTT_NONE, 0};
  for (uint32_t g = 0; g < n; ++g) { const uint32_t kk = g / 2, which = g & 1, e = (kk / 8) * 16 + which * 8 + (kk % 8); a.limb[e] = HmLimb{(uint16_t)g, (uint16_t)g, (uint16_t)(g % M), 0}; }
  const uint32_t tiles = N >> HM_TILE_LOG, grid = cnt * tiles, waves = HM_THREADS / 64;
  CK(hipMalloc(&d_st, 8ull * NSTAMP * grid * waves)); CK(hipMemset(d_st, 0, 8ull * NSTAMP * grid * waves));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(d_st, 0, 8ull * NSTAMP * grid * waves));
    CK(hipEventRecord(e0));
    if (strided) hipLaunchKernelGGL((k_pass<8, true, false, 0>), dim3(grid), dim3(HM_THREADS), 0, 0, a, d_st);
    else hipLaunchKernelGGL((k_pass<8, false, false, 1>), dim3(grid), dim3(HM_THREADS), 0, 0, a, d_st);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  std::vector<uint64_t> st((size_t)NSTAMP * grid * waves);
  CK(hipMemcpy(st.data(), d_st, 8ull * st.size(), hipMemcpyDeviceToHost));
  uint64_t t0 = ~0ull, tend = 0; size_t nw = 0;
  double cyc = 0, rt = 0;
  for (size_t w = 0; w < (size_t)grid * waves; ++w) {
    const uint64_t *o = &st[w * NSTAMP]; if (!o[0]) continue; ++nw;
    t0 = std::min(t0, o[10]); tend = std::max(tend, o[11]); cyc += (double)(o[8] - o[0]); rt += (double)(o[11] - o[10]);
  }
  const double rt_us = 0.01;                       // s_memrealtime: 100 MHz
  const double tick_us = rt * rt_us / cyc;         // cycle counter calibrated against it over all waves
  printf("%s pass, %u limbs, %u workgroups (%zu waves stamped): event time %.1f us, realtime span %.2f us, cycle tick %.5f us (%.0f MHz)\n",
         strided ? "COL" : "ROW", n, grid, nw, ms * 1e3, (double)(tend - t0) * rt_us, tick_us, 1.0 / tick_us);
  const char *names[8] = {"issue loads", "wait loads", "round0+LDS wr", "barrier", "round1 (+tw2 req, LDS rd/wr)", "barrier", "round2 + store issue", "wait stores"};
  std::vector<double> dur[8], start, end, total;
  for (size_t w = 0; w < (size_t)grid * waves; ++w) {
    const uint64_t *o = &st[w * NSTAMP]; if (!o[0]) continue;
    for (int i = 0; i < 8; ++i) dur[i].push_back((double)(o[i + 1] - o[i]) * tick_us);
    total.push_back((double)(o[8] - o[0]) * tick_us);
    start.push_back((double)(o[10] - t0) * rt_us); end.push_back((double)(o[11] - t0) * rt_us);
  }
  auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
  printf("%-32s %8s %8s %8s\n", "phase (us)", "p10", "median", "p90");
  for (int i = 0; i < 8; ++i) printf("%-32s %8.2f %8.2f %8.2f\n", names[i], pct(dur[i], .1), pct(dur[i], .5), pct(dur[i], .9));
  printf("%-32s %8.2f %8.2f %8.2f\n", "wave total", pct(total, .1), pct(total, .5), pct(total, .9));
  printf("%-32s %8.2f %8.2f %8.2f  max %.2f\n", "wave start", pct(start, .1), pct(start, .5), pct(start, .9), pct(start, 1.0));
  printf("%-32s %8.2f %8.2f %8.2f  max %.2f\n", "wave end", pct(end, .1), pct(end, .5), pct(end, .9), pct(end, 1.0));
  printf("raw HW_ID samples: %08x %08x %08x\n", (unsigned)st[9], (unsigned)st[9 + NSTAMP * waves], (unsigned)st[9 + 2 * NSTAMP * waves]);
  // workgroups per CU: HW_ID bits: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9 layout)
  std::vector<int> percu(8 * 64, 0);
  for (size_t w = 0; w < (size_t)grid * waves; w += waves) { const uint64_t *o = &st[w * NSTAMP]; if (!o[0]) continue; const uint32_t h = (uint32_t)o[9]; const uint32_t cu = (h >> 8) & 15, sh = (h >> 12) & 1, se = (h >> 13) & 7; percu[(se * 2 + sh) * 16 + cu]++; }
  int hist[16] = {0}; for (int v : percu) if (v < 16) hist[v]++;
  printf("workgroups per (se,sh,cu) id [xcd not in HW_ID]: "); for (int i = 0; i < 16; ++i) if (hist[i]) printf("%d:%d ", i, hist[i]); printf("\n");
  return 0;
}
