// hm_modarith.h — 64-bit modular arithmetic for the MI355X (gfx950) FHE datapath.
//
// Measured on MI355X (tools/mulrate.hip, profiles/r01_mulrate.txt): v_mad_u64_u32 issues at
// ~5 cycles per wave64 instruction (v_add_u32 ~3, v_mul_hi_u32 ~7.7), i.e. 32x32->64 multiplies are
// close to full rate on CDNA4, so everything here is written as 64x64->128 products that hipcc lowers
// to v_mad_u64_u32 chains; v_mul_hi_u32 is avoided.
//
// Conventions: every modulus q satisfies 2^(k-1) < q < 2^k with k <= 60.
//   * Shoup form for known constants w: ws = floor(w * 2^64 / q); lazy product in [0, 2q).
//   * Barrett for variable x variable: mu = floor(2^(k+63) / q), valid for z < 2^(k+63).
// The same header compiles with g++ (HM_EMULATE) for the host-side kernel emulator in tests/emu.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HM_HD __host__ __device__ __forceinline__
#else
#define HM_HD inline
#endif

typedef unsigned __int128 hm_u128;

// Per-modulus constants, one 64-byte record per mod id, read through the scalar cache.
struct HmMod {
  uint64_t q;       // modulus
  uint64_t mu;      // floor(2^(k+63) / q), in (2^63, 2^64)
  uint64_t r64;     // 2^64 mod q (folds the top word of a 128-bit accumulator)
  uint64_t r64s;    // Shoup companion of r64
  uint64_t ninv;    // N^-1 mod q
  uint64_t ninvs;   // Shoup companion of ninv
  uint32_t sh;      // k - 1
  uint32_t pad0;
  uint64_t pad1;
};

struct HmTw {  // one twiddle: value and its Shoup companion (16 B -> one dwordx4 load)
  uint64_t w, ws;
};

HM_HD uint64_t hm_mulhi(uint64_t a, uint64_t b) { return (uint64_t)(((hm_u128)a * b) >> 64); }

// x - q if x >= q (unsigned compare); keeps x otherwise.
HM_HD uint64_t hm_csub(uint64_t x, uint64_t q) { return x >= q ? x - q : x; }

// w * x mod q, lazily: result in [0, 2q) for ANY 64-bit x (w < q, ws = floor(w 2^64 / q)).
HM_HD uint64_t hm_shoup_lazy(uint64_t x, uint64_t w, uint64_t ws, uint64_t q) {
  return x * w - hm_mulhi(x, ws) * q;
}
HM_HD uint64_t hm_shoup(uint64_t x, uint64_t w, uint64_t ws, uint64_t q) {
  return hm_csub(hm_shoup_lazy(x, w, ws, q), q);
}

// z mod q for z < 2^(k+63): covers one product (2^2k), MAC2, and sums of up to 8 products at k = 60.
HM_HD uint64_t hm_barrett(hm_u128 z, const HmMod &m) {
  uint64_t zh = (uint64_t)(z >> m.sh);
  uint64_t r = (uint64_t)z - hm_mulhi(zh, m.mu) * m.q;  // in [0, 3q)
  r = hm_csub(r, 2 * m.q);
  return hm_csub(r, m.q);
}
HM_HD uint64_t hm_mulmod(uint64_t a, uint64_t b, const HmMod &m) { return hm_barrett((hm_u128)a * b, m); }

// full 128-bit accumulator (base conversion: up to 16 products of 2^2k): fold the top word first.
HM_HD uint64_t hm_barrett_wide(hm_u128 z, const HmMod &m) {
  uint64_t zl = (uint64_t)z, zh = (uint64_t)(z >> 64);
  hm_u128 f = (hm_u128)hm_shoup_lazy(zh, m.r64, m.r64s, m.q) + zl;  // < 2q + 2^64
  return hm_barrett(f, m);
}

HM_HD uint64_t hm_addmod(uint64_t a, uint64_t b, uint64_t q) { return hm_csub(a + b, q); }
HM_HD uint64_t hm_submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }

// a * b + c with 32-bit a, b and 64-bit c: one v_mad_u64_u32.  Written as inline asm on the device because
// hipcc turns `(uint64_t)a * b >> 32` into the quarter-rate v_mul_hi_u32 (7.7 cycles vs 5.1, measured).
HM_HD uint64_t hm_mad64(uint32_t a, uint32_t b, uint64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t d;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
  return d;
#else
  return (uint64_t)a * b + c;
#endif
}

// Approximate Shoup product for the butterflies: the quotient estimate drops the low partial products of
// x * ws (three v_mad_u64_u32, no quarter-rate v_mul_hi_u32), so it may be up to 2 too small and the lazy
// result lies in [0, 4q) instead of [0, 2q) — still w * x mod q exactly, for ANY 64-bit x.
HM_HD uint64_t hm_shoup_lazy4(uint64_t x, uint64_t w, uint64_t ws, uint64_t q) {
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), s0 = (uint32_t)ws, s1 = (uint32_t)(ws >> 32);
  // floor(x ws / 2^64) = x1 s1 + floor((x1 s0 + x0 s1 + floor(x0 s0 / 2^32)) / 2^32); estimate: the two
  // cross terms truncated separately and x0 s0 dropped, i.e. at most 2 below the true quotient
  const uint64_t a = hm_mad64(x0, s1, 0);
  const uint64_t b = hm_mad64(x1, s0, 0);
  const uint64_t h = hm_mad64(x1, s1, a >> 32) + (b >> 32);
  return x * w - h * q;
}

// x - m if x >= m, for x, m < 2^63: the sign of the wrapped difference decides (one 32-bit compare)
HM_HD uint64_t hm_csub63(uint64_t x, uint64_t m) {
  const uint64_t t = x - m;
  return (int32_t)(t >> 32) < 0 ? x : t;
}

// Harvey butterflies with the approximate product (q < 2^60, so 8q < 2^63).
// forward (Cooley-Tukey): X, Y in [0, 8q) -> X', Y' in [0, 8q); q4 = 4q
HM_HD void hm_bfly_fwd(uint64_t &X, uint64_t &Y, const HmTw &t, uint64_t q, uint64_t q4) {
  const uint64_t x = hm_csub63(X, q4);                    // [0, 4q)
  const uint64_t v = hm_shoup_lazy4(Y, t.w, t.ws, q);     // [0, 4q)
  X = x + v;
  Y = x - v + q4;
}
// inverse (Gentleman-Sande): X, Y in [0, 4q) -> X', Y' in [0, 4q)
HM_HD void hm_bfly_inv(uint64_t &X, uint64_t &Y, const HmTw &t, uint64_t q, uint64_t q4) {
  const uint64_t s = hm_csub63(X + Y, q4);
  const uint64_t d = X - Y + q4;
  X = s;
  Y = hm_shoup_lazy4(d, t.w, t.ws, q);
}
// [0, 8q) -> [0, q)
HM_HD uint64_t hm_reduce8(uint64_t x, uint64_t q) { return hm_csub63(hm_csub63(hm_csub63(x, 4 * q), 2 * q), q); }
