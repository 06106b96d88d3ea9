// Microbenchmark: VALU cost of the lazy NTT butterfly variants on gfx950 (compute only, twiddles in registers).
// Build: hipcc -O3 --offload-arch=gfx950 -Ihomulator_amd/csrc tools/bflyrate.hip -o tools/bflyrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "hm_modarith.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
constexpr int ITERS = 512;

// ---- V0: round-1 form (inline-asm v_mad_u64_u32, subtract with borrow)
__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) {
  uint64_t d; asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc"); return d;
}
__device__ __forceinline__ uint64_t v0_shoup(uint64_t x, uint64_t w, uint64_t ws, uint64_t q) {
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), s0 = (uint32_t)ws, s1 = (uint32_t)(ws >> 32);
  const uint64_t a = mad64(x0, s1, 0), b = mad64(x1, s0, 0);
  const uint64_t h = mad64(x1, s1, a >> 32) + (b >> 32);
  return x * w - h * q;
}
__device__ __forceinline__ uint64_t v0_csub(uint64_t x, uint64_t m) { const uint64_t t = x - m; return (int32_t)(t >> 32) < 0 ? x : t; }
struct V0 { static __device__ __forceinline__ void f(uint64_t &X, uint64_t &Y, HmTw t, uint64_t q, uint64_t z, int) {
  const uint64_t q4 = 4 * q; const uint64_t x = v0_csub(X, q4); const uint64_t v = v0_shoup(Y, t.w, t.ws, q); X = x + v; Y = x - v + q4; } };

// ---- V2: fence-free: products kept 64-bit by an opaque zero addend, quotient carry by a 32-bit add pair,
// x folded into the multiply-accumulate chain, Y' = 2x + 4q - X'
__device__ __forceinline__ uint64_t v2_quot(uint64_t x, uint64_t ws, uint64_t z) {
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), s0 = (uint32_t)ws, s1 = (uint32_t)(ws >> 32);
  const uint64_t a = (uint64_t)x0 * s1 + z, b = (uint64_t)x1 * s0 + z;
  const uint64_t c = (uint64_t)(uint32_t)(a >> 32) + (uint32_t)(b >> 32);
  return (uint64_t)x1 * s1 + c;
}
template <int CS> struct V2 { static __device__ __forceinline__ void f(uint64_t &X, uint64_t &Y, HmTw t, uint64_t q, uint64_t z, int stage) {
  const uint64_t nq = z - q;  // z is an opaque zero: the compiler cannot fold these back into subtractions
  uint64_t x = X, B = 4 * q;
  if (CS == 0) { const uint64_t tt = X + (z - 4 * q); x = tt < X ? tt : X; x = (X >= 4 * q) ? tt : X; }
  if (CS == 1) { const uint64_t tt = X + (z - 4 * q); x = tt > X ? X : tt; }            // unsigned min
  if (CS == 2) { if (stage & 1) { const uint64_t tt = X + (z - 8 * q); x = tt > X ? X : tt; } B = 4 * q; }  // every other stage
  const uint64_t h = v2_quot(Y, t.ws, z);
  const uint64_t Xn = x + Y * t.w + h * nq;
  Y = ((x << 1) + B) - Xn;
  X = Xn; } };

template <class V> __global__ void __launch_bounds__(512) k(uint64_t *out, const uint64_t *in, uint64_t q, uint64_t zero) {
  uint64_t v[8]; HmTw tw[7];
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < 8; ++i) v[i] = in[tid * 8 + i] % (4 * q);
  for (int i = 0; i < 7; ++i) { tw[i].w = in[tid + i] % q; tw[i].ws = (uint64_t)(((unsigned __int128)tw[i].w << 64) / q); }
  const uint64_t z = zero;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int pb = 2 - j;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (e & (1 << pb)) continue;
        V::f(v[e], v[e | (1 << pb)], tw[(1 << j) - 1 + (e >> (3 - j))], q, z, j + it);
      }
    }
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s ^= v[i] % q;   // reduced: variants must agree
  out[tid] = s;
}

// ---- V3: the shipped forward butterflies (hm_modarith.h hm_bfly_fwd_k: word-wise Montgomery product, q = h 2^32 + 1, conditional
// subtraction every other stage)
__global__ void __launch_bounds__(512) k3(uint64_t *out, const uint64_t *in, uint64_t q) {
  uint64_t v[8]; HmW tw[7];
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < 8; ++i) v[i] = in[tid * 8 + i] % (2 * q);
  for (int i = 0; i < 7; ++i) tw[i] = in[tid + i] % q;
  const HmBflyMod m = hm_bfly_mod(q);
  for (int it = 0; it < ITERS / 2; ++it) {
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) {
      const int j = jj % 3, pb = 2 - j;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (e & (1 << pb)) continue;
        if (jj & 1) hm_bfly_fwd_k<1>(v[e], v[e | (1 << pb)], tw[(1 << j) - 1 + (e >> (3 - j))], m);
        else        hm_bfly_fwd_k<0>(v[e], v[e | (1 << pb)], tw[(1 << j) - 1 + (e >> (3 - j))], m);
      }
    }
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s ^= v[i] % q;
  out[tid] = s;
}

// ---- V4 (round 5 experiment): the two carry-word additions (mov + 64-bit add each) as multiply-adds by an opaque 1
__device__ __forceinline__ uint64_t mont_acc_mad1(uint64_t c, uint64_t x, uint64_t wt, const HmBflyMod &m, uint32_t one) {
  const uint32_t b0 = (uint32_t)x, b1 = (uint32_t)(x >> 32), w0 = (uint32_t)wt, w1 = (uint32_t)(wt >> 32);
  const uint64_t P = (uint64_t)b0 * w0 + m.z;
  uint32_t n0 = ~(uint32_t)P;
  uint64_t A = (uint64_t)b0 * w1 + m.cc;
  HM_PIN(n0); HM_PIN64(A);
  A = (uint64_t)b1 * w0 + A;
  A = (uint64_t)n0 * m.h + A;
  uint32_t ph = (uint32_t)(P >> 32);
  HM_PIN(ph);
  const uint64_t S = (uint64_t)ph * one + A;
  uint32_t n1 = ~(uint32_t)S;
  HM_PIN(n1);
  uint64_t B = (uint64_t)b1 * w1 + c;
  B = (uint64_t)n1 * m.h + B;
  uint32_t sh = (uint32_t)(S >> 32);
  HM_PIN(sh);
  return (uint64_t)sh * one + B;
}
template <int KIND>
__device__ __forceinline__ void bfly_fwd_mad1(uint64_t &X, uint64_t &Y, uint64_t w, const HmBflyMod &m, uint32_t one) {
  uint64_t x = X;
  if (KIND >= 1) x = hm_csub_neg(x, m.nq4);
  const uint64_t xn = mont_acc_mad1(x, Y, w, m, one);
  Y = ((x << 1) + m.q2) - xn;
  X = xn;
}
__global__ void __launch_bounds__(512) k4(uint64_t *out, const uint64_t *in, uint64_t q) {
  uint64_t v[8]; uint64_t tw[7];
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < 8; ++i) v[i] = in[tid * 8 + i] % (2 * q);
  for (int i = 0; i < 7; ++i) tw[i] = in[tid + i] % q;
  const HmBflyMod m = hm_bfly_mod(q);
  uint32_t one;
  asm volatile("s_mov_b32 %0, 1" : "=s"(one));
  for (int it = 0; it < ITERS / 2; ++it) {
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) {
      const int j = jj % 3, pb = 2 - j;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (e & (1 << pb)) continue;
        if (jj & 1) bfly_fwd_mad1<1>(v[e], v[e | (1 << pb)], tw[(1 << j) - 1 + (e >> (3 - j))], m, one);
        else        bfly_fwd_mad1<0>(v[e], v[e | (1 << pb)], tw[(1 << j) - 1 + (e >> (3 - j))], m, one);
      }
    }
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s ^= v[i] % q;
  out[tid] = s;
}
// ---- V2s: round 3's shipped form for comparison (Shoup product with the approximate quotient, 9 multiplies, subtraction of 8q every other stage)
template <int KIND>
__device__ __forceinline__ void shoup_bfly_fwd_k(uint64_t &X, uint64_t &Y, const HmTw &t, const HmBflyMod &m) {
  uint64_t x = X;
  if (KIND >= 1) x = hm_csub_neg(x, m.nq8);
  const uint64_t xn = hm_shoup_lazy4_acc(x, Y, t, m);
  Y = ((x << 1) + m.q4) - xn;
  X = xn;
}
__global__ void __launch_bounds__(512) k2s(uint64_t *out, const uint64_t *in, uint64_t q) {
  uint64_t v[8]; HmTw tw[7];
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < 8; ++i) v[i] = in[tid * 8 + i] % (4 * q);
  for (int i = 0; i < 7; ++i) { tw[i].w = in[tid + i] % q; tw[i].ws = (uint64_t)(((unsigned __int128)tw[i].w << 64) / q); }
  const HmBflyMod m = hm_bfly_mod(q);
  for (int it = 0; it < ITERS / 2; ++it) {
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) {
      const int j = jj % 3, pb = 2 - j;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (e & (1 << pb)) continue;
        if (jj & 1) shoup_bfly_fwd_k<1>(v[e], v[e | (1 << pb)], tw[(1 << j) - 1 + (e >> (3 - j))], m);
        else        shoup_bfly_fwd_k<0>(v[e], v[e | (1 << pb)], tw[(1 << j) - 1 + (e >> (3 - j))], m);
      }
    }
  }
  uint64_t s = 0; for (int i = 0; i < 8; ++i) s ^= v[i] % q;
  out[tid] = s;
}
template <class K> static int run3(K kern, const char *name, const char *key) {
  const int blocks = 256 * 4, threads = 512;
  const uint64_t q = 1152921092289986561ull;   // (2^28 - 97) 2^32 + 1
  uint64_t *d, *in; CK(hipMalloc(&d, (size_t)blocks * threads * 8)); CK(hipMalloc(&in, (size_t)blocks * threads * 8 * 8 + 64));
  CK(hipMemset(in, 0x5a, (size_t)blocks * threads * 8 * 8 + 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, in, q);  // ~70 ms: clocks settle under load
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, in, q);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; sum += ms;
  }
  const double wbfly_per_simd = (double)blocks * threads * ITERS * 12 / 64.0 / 1024.0;
  printf("%-40s %8.3f ms (mean %.3f)  %7.2f cyc@2.4GHz/wave-butterfly/SIMD  %s (full chip, sustained) = %.5f  -> %.3f us per 2^16 limb NTT\n",
         name, best, sum / 5, best * 1e-3 * 2.4e9 / wbfly_per_simd, key, sum / 5 * 1e6 / wbfly_per_simd / 1024.0,
         sum / 5 * 1e6 / wbfly_per_simd / 1024.0 * (32768.0 * 16 / 64) * 1e-3);
  CK(hipFree(d)); CK(hipFree(in));
  return 0;
}

template <class V> int run(const char *name, uint64_t *sum) {
  const int blocks = 256 * 4, threads = 512; // 4 WGs/CU x 8 waves = 8 waves/SIMD, like the NTT kernels
  const uint64_t q = 1152921504606584833ull; // 2^60 - 2^18 + 1 ... any 60-bit odd value works for the rate
  uint64_t *d, *in; CK(hipMalloc(&d, (size_t)blocks * threads * 8)); CK(hipMalloc(&in, (size_t)blocks * threads * 8 * 8 + 64));
  CK(hipMemset(in, 0x5a, (size_t)blocks * threads * 8 * 8 + 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, d, in, q, 0ull); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(threads), 0, 0, d, in, q, 0ull);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  uint64_t h[64]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  uint64_t s = 0; for (int i = 0; i < 64; ++i) s = s * 1000003 + h[i];
  *sum = s;
  const double bfly = (double)blocks * threads * ITERS * 12;
  const double cyc = best * 1e-3 * 2.4e9 * 1024.0 / (bfly / 64.0);
  printf("%-40s %8.3f ms  %7.2f cyc/wave-butterfly/SIMD  -> %.3f us per 2^16 limb NTT  checksum %016llx\n", name, best, cyc,
         cyc * (32768.0 * 16 / 64) / 1024.0 / 2.4e3, (unsigned long long)s);
  CK(hipFree(d)); CK(hipFree(in));
  return 0;
}
int main() {
  uint64_t s;
  run<V0>("V0 asm mad, sub/borrow", &s);
  run<V2<0>>("V2 chain, csub by compare", &s);
  run<V2<1>>("V2 chain, csub by unsigned min", &s);
  run<V2<2>>("V2 chain, csub every other stage", &s);
  run3(k2s, "V2s round 3: Shoup, csub every other stage", "shoup_wave_butterfly_ns");
  run3(k3, "V3 shipped: Montgomery, q = h 2^32 + 1", "wave_butterfly_ns");
  run3(k4, "V4 experiment: carry words added by multiply-add x 1", "mad1_wave_butterfly_ns");
  return 0;
}
