// hm_modarith.h — 64-bit modular arithmetic for the MI355X (gfx950) FHE datapath.
//
// Measured on MI355X (tools/mulrate.hip, profiles/r01_mulrate.txt): v_mad_u64_u32 issues at
// ~5 cycles per wave64 instruction (v_add_u32 ~3, v_mul_hi_u32 ~7.7), i.e. 32x32->64 multiplies are
// close to full rate on CDNA4, so everything here is written as 64x64->128 products that hipcc lowers
// to v_mad_u64_u32 chains; v_mul_hi_u32 is avoided.
//
// TWO arithmetic back-ends from one source (round 5), chosen per context by hm_create from the chain it is given (hm_dispatch.cpp):
//   HM_GENERIC = 0 ("mont32", libhm_m32.so): every modulus is q = h 2^32 + 1 below 2^60.  q^-1 = 1 mod 2^32, so a word-wise Montgomery
//       reduction step is ONE multiply: six 32-bit multiplies per butterfly, one-word twiddles (round 4).  The default chain.
//   HM_GENERIC = 1 ("generic", libhm_gen.so): any distinct primes = 1 mod 2N with 2^20 < q < 2^60 (SURVEY.md 8d's chain, a 36-bit-word
//       chain as the reference's configuration models, a chain an existing FHE library hands over).  Shoup form for known constants
//       (ws = floor(w 2^64 / q), nine multiplies per butterfly, two-word twiddles), Barrett for the key product.
// The passes, kernels and host code use the neutral names below (HmW, hm_tw_acc, hm_kmul, hm_kconst, HM_LAZY_Q) and never ask which.
//
// Conventions: every modulus q satisfies 2^(k-1) < q < 2^k with k <= 60.
//   * mont32: Montgomery form for the transform's twiddles and the per-launch constants: wt = w 2^64 mod q.
//   * Shoup form (generic: twiddles and constants; both: element-wise constants): ws = floor(w * 2^64 / q); lazy product in [0, 2q).
//   * Barrett for variable x variable: mu = floor(2^(k+63) / q), valid for z < 2^(k+63).
// The same header compiles with g++ (HM_EMULATE) for the host-side kernel emulator in tests/emu.
#pragma once
#include <stdint.h>
#ifndef HM_GENERIC
#define HM_GENERIC 0
#endif

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
  union {
    uint64_t r128;  // mont32: 2^128 mod q: takes a sum of Montgomery products (x y 2^-64) back to x y in one more product
    uint64_t ninvs; // generic: Shoup companion of ninv
  };
  uint32_t sh;      // k - 1
  uint32_t pad0;
  uint64_t nqinv;   // -q^-1 mod 2^64 (generic: Montgomery reduction of the base-conversion accumulators; mont32 needs only h = q >> 32)
};

struct HmTw {  // a constant in Shoup form: value and companion (epilogue / prologue / element-wise constants)
  uint64_t w, ws;
};
#if HM_GENERIC
typedef HmTw HmW;       // generic: a transform twiddle or twist constant with its Shoup companion, two words per table entry
#define HM_TW_WORDS 2
#else
typedef uint64_t HmW;   // mont32: a transform twiddle or twist constant in Montgomery form, w 2^64 mod q: ONE word per table entry (round 4)
#define HM_TW_WORDS 1
#endif

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
// the same without the final conditional subtractions: z mod q + {0, q, 2q}, in [0, 3q) — for sums that are reduced once at the end
HM_HD uint64_t hm_barrett_lazy(hm_u128 z, const HmMod &m) {
  const uint64_t zh = (uint64_t)(z >> m.sh);
  return (uint64_t)z - hm_mulhi(zh, m.mu) * m.q;
}
// [0, 16q) -> [0, q)   (q < 2^60)
HM_HD uint64_t hm_reduce16(uint64_t x, uint64_t q) {
  return hm_csub(hm_csub(hm_csub(hm_csub(x, 8 * q), 4 * q), 2 * q), q);
}

// full 128-bit accumulator (base conversion: up to 16 products of 2^2k): fold the top word first.
HM_HD uint64_t hm_barrett_wide(hm_u128 z, const HmMod &m) {
  uint64_t zl = (uint64_t)z, zh = (uint64_t)(z >> 64);
  hm_u128 f = (hm_u128)hm_shoup_lazy(zh, m.r64, m.r64s, m.q) + zl;  // < 2q + 2^64
  return hm_barrett(f, m);
}

// Montgomery reduction of a wide accumulator: z -> z * 2^-64 mod q, fully reduced.  Two word-wise steps (q = h 2^32 + 1: q^-1 = 1 mod
// 2^32, a step is one multiply: see hm_mont_acc below): t = (z + m1 q + m2 q 2^32) / 2^64 with m1, m2 <= 2^32, so
// t < (z >> 64) + q + h + 2.  About a third of the instructions of hm_barrett_wide; the 2^-64 is absorbed by constants
// stored as c * 2^64 mod q (base-conversion tables).
// TERMS = number of products y * w summed into z with y < 2^60 - 2^32 (any input modulus) and w < q: z >> 64 < TERMS * q (1 - 2^-28) / 16,
// i.e. below q - 16 h for up to 16 terms and below 2q - 32 h for up to 32: one conditional subtraction for TERMS <= 16, two for <= 32.
template <int TERMS = 32>
HM_HD uint64_t hm_redc_wide(hm_u128 z, const HmMod &m) {
  static_assert(TERMS <= 32, "accumulator bound");
  const uint64_t lo = (uint64_t)z, hi = (uint64_t)(z >> 64);
#if HM_GENERIC
  // any odd q: mq = lo * (-q^-1) makes z + mq q divisible by 2^64; the low words cancel and carry iff lo != 0.  z >> 64 < TERMS q / 16
  // (y < 2^60, w < q), the quotient adds less than q + 1: one conditional subtraction for TERMS <= 16, two for <= 32
  uint64_t t = hi + hm_mulhi(lo * m.nqinv, m.q) + (lo != 0);
#else
  const uint32_t h = (uint32_t)(m.q >> 32);
  const uint64_t c = (uint64_t)h + 1;
  const uint64_t S = (uint64_t)(~(uint32_t)lo) * h + (c + (uint32_t)(lo >> 32));
  uint64_t t = (uint64_t)(~(uint32_t)S) * h + (hi + c) + (uint32_t)(S >> 32);
#endif
  if (TERMS > 16) t = hm_csub(t, 2 * m.q);
  return hm_csub(t, m.q);
}

HM_HD uint64_t hm_addmod(uint64_t a, uint64_t b, uint64_t q) { return hm_csub(a + b, q); }
HM_HD uint64_t hm_submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }

// hipcc rewrites `(uint64_t)a * b >> 32` into the quarter-rate v_mul_hi_u32 (7.7 cycles against 5.1 for
// v_mad_u64_u32, measured) and `x + (-m)` back into a two-instruction subtract with borrow.  A zero the compiler
// cannot see through (an SGPR pair passed through an empty asm statement, no instruction emitted) keeps the
// cheaper forms: `a * b + z` stays one full-width v_mad_u64_u32, `x + (z - m)` one v_lshl_add_u64.
HM_HD uint64_t hm_opaque_zero() {
  uint64_t z = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+s"(z));
#endif
  return z;
}

// keeps a value as computed: the optimiser cannot look through the empty statement (no instruction is emitted)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HM_NO_PIN)
#define HM_PIN(x) asm("" : "+v"(x))
#define HM_PIN64(x) asm("" : "+v"(x))
#else
#define HM_PIN(x) ((void)0)
#define HM_PIN64(x) ((void)0)
#endif
// Per-modulus constants of the lazy butterflies and products (wave-uniform, SGPRs)
struct HmBflyMod {
  uint64_t q2, q4, nq, nq2, nq4, nq8, cc, z;
  uint32_t h;
};
HM_HD HmBflyMod hm_bfly_mod(uint64_t q) {
  HmBflyMod m;
  m.z = hm_opaque_zero();
  m.q2 = m.z + 2 * q;   // (opaque: (x << 1) + 2q stays ONE v_lshl_add_u64 instead of (x + q) << 1)
  m.q4 = 4 * q;
  m.nq = m.z - q;
  m.nq2 = m.z - 2 * q;
  m.nq4 = m.z - 4 * q;
  m.nq8 = m.z - 8 * q;
  m.h = (uint32_t)(q >> 32);
  m.cc = ((uint64_t)m.h + 1) * ((1ull << 32) + 1);
  return m;
}

// Approximate Shoup quotient (generic butterflies, the generic key multiply-accumulate's Barrett steps): floor(x ws / 2^64) = x1 s1 +
// floor((x1 s0 + x0 s1 + floor(x0 s0 / 2^32)) / 2^32); the estimate truncates the two cross terms separately and drops x0 s0 (three
// v_mad_u64_u32, no v_mul_hi_u32), so it is at most 2 below the true quotient and the lazy product
// w x - h q lies in [0, 4q) instead of [0, 2q) — still w x mod q exactly, for ANY 64-bit x.
HM_HD uint64_t hm_shoup_quot(uint64_t x, uint64_t ws, uint64_t z) {
  const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), s0 = (uint32_t)ws, s1 = (uint32_t)(ws >> 32);
  const uint64_t a = (uint64_t)x0 * s1 + z, b = (uint64_t)x1 * s0 + z;
  return (uint64_t)x1 * s1 + (a >> 32) + (b >> 32);
}
// c + w x - h q with nq = 2^64 - q: one multiply-accumulate chain seeded with c
HM_HD uint64_t hm_shoup_lazy4_acc(uint64_t c, uint64_t x, const HmTw &t, const HmBflyMod &m) {
  return c + x * t.w + hm_shoup_quot(x, t.ws, m.z) * m.nq;
}

// x - m if x >= m, given nm = 2^64 - m: when x < m the wrapped sum is larger than x, so the unsigned minimum decides
HM_HD uint64_t hm_csub_neg(uint64_t x, uint64_t nm) {
  const uint64_t t = x + nm;
  return t < x ? t : x;
}
// w 2^64 mod q on the host (table generation)
HM_HD uint64_t hm_to_mont(uint64_t w, uint64_t q) { return (uint64_t)(((hm_u128)w << 64) % q); }
HM_HD uint64_t hm_shoup_companion(uint64_t w, uint64_t q) { return (uint64_t)(((hm_u128)w << 64) / q); }

#if !HM_GENERIC
// ---------------------------------------------------------------------------------------------------
// mont32 (round 4): word-wise Montgomery products for moduli q = h 2^32 + 1.
// q^-1 = 1 mod 2^32, so a reduction step T -> (T - T0 q) / 2^32 = (T >> 32) - T0 h needs ONE multiply; written with the complement,
// (T + (~T0) q + q) / 2^32 = (T >> 32) + (~T0) h + (h + 1) (the low words always carry exactly 1; adding q more leaves the class).
// Two steps take the 124-bit product x wt to x wt 2^-64 mod q: SIX multiplies (four of the product, one per step) where the Shoup form
// needs nine (3 + 3 + 3), and no companion word per constant.  wt = w 2^64 mod q (tables are stored in this form).
//   c + v,  v = x w mod q + {0, q},  0 <= v = (x wt + m1 q + m2 q 2^32) / 2^64 with m1, m2 <= 2^32, so v <= floor(x wt / 2^64) + q + h + 1
//   < 1.5 q + 2^28 for x < 2^63 (no 64-bit sum overflows: b1 w0 < 2^63, b0 w1 < 2^60, (~P0) h < 2^60, cc < 2^60 + 2^33), c + v < 2^64.
//   (The 2^28 is the first step's m1 q / 2^64; every range below has half a q of slack.)
// cc = (h + 1)(2^32 + 1): the constants of both steps in one addend (the second step's sits 32 bits up, where the first step's low word
// cannot see it).  15 VALU instructions per forward butterfly against 17, 65.9 against 79.0 cycles per wave
// (tools/bflyrate.hip, profiles/r04_bflyrate.txt).
// ---------------------------------------------------------------------------------------------------
HM_HD uint64_t hm_mont_acc(uint64_t c, uint64_t x, uint64_t wt, const HmBflyMod &m) {
  const uint32_t b0 = (uint32_t)x, b1 = (uint32_t)(x >> 32), w0 = (uint32_t)wt, w1 = (uint32_t)(wt >> 32);
  const uint64_t P = (uint64_t)b0 * w0 + m.z;
  uint32_t n0 = ~(uint32_t)P;
  uint64_t A = (uint64_t)b0 * w1 + m.cc;
  HM_PIN(n0); HM_PIN64(A);   // (hipcc otherwise re-associates the complement and the constant into a longer sequence: 19 instead of 15 instructions)
  A = (uint64_t)b1 * w0 + A;
  A = (uint64_t)n0 * m.h + A;
  const uint64_t S = A + (uint32_t)(P >> 32);
  uint32_t n1 = ~(uint32_t)S;
  HM_PIN(n1);
  uint64_t B = (uint64_t)b1 * w1 + c;
  B = (uint64_t)n1 * m.h + B;
  return B + (uint32_t)(S >> 32);
}
// x k mod q, fully reduced, for a constant held in Montgomery form kt = k 2^64 mod q and any x below 2^63 (the per-launch constants of the
// transforms' prologues and epilogues: 15 instructions where the exact Shoup product took about 20, one of them a quarter-rate v_mul_hi_u32)
HM_HD uint64_t hm_mont_const_mul(uint64_t x, uint64_t kt, uint64_t q) {
  const HmBflyMod m = hm_bfly_mod(q);
  return hm_csub_neg(hm_mont_acc(0, x, kt, m), m.nq);
}
#endif

// ---------------------------------------------------------------------------------------------------
// The neutral surface both back-ends implement (the passes, kernels and launch code are written against it):
//   HM_LAZY_Q           the lazy product adds less than HM_LAZY_Q q to its seed: 2 (mont32: below 1.5q + 2^28) or 4 (generic: [0, 4q))
//   hm_tw_acc(c, x, w)  c + w x mod q, lazily (w: a table entry HmW); x below HM_X_MAX
//   hm_kmul(x, k, q)    x k mod q fully reduced, k a per-launch constant record made by hm_kconst on the host (HmTw)
//   hm_kconst(k, q)     host: the record of constant k (mont32: {k 2^64 mod q, 0}; generic: {k, Shoup companion})
//   hm_tw_entry(w, q)   host: the table entry of twiddle w
// ---------------------------------------------------------------------------------------------------
#if HM_GENERIC
#define HM_LAZY_Q 4
HM_HD uint64_t hm_tw_acc(uint64_t c, uint64_t x, const HmW &w, const HmBflyMod &m) { return hm_shoup_lazy4_acc(c, x, w, m); }
HM_HD uint64_t hm_kmul(uint64_t x, const HmTw &k, uint64_t q) { return hm_shoup(x, k.w, k.ws, q); }
HM_HD HmTw hm_kconst(uint64_t k, uint64_t q) { return HmTw{k, hm_shoup_companion(k, q)}; }
HM_HD HmW hm_tw_entry(uint64_t w, uint64_t q) { return HmTw{w, hm_shoup_companion(w, q)}; }
HM_HD uint64_t hm_tw_bits(const HmW &w) { return w.w ^ w.ws; }   // (ablation builds: keeps a twiddle load alive)
#else
#define HM_LAZY_Q 2
HM_HD uint64_t hm_tw_acc(uint64_t c, uint64_t x, const HmW &w, const HmBflyMod &m) { return hm_mont_acc(c, x, w, m); }
HM_HD uint64_t hm_kmul(uint64_t x, const HmTw &k, uint64_t q) { return hm_mont_const_mul(x, k.w, q); }
HM_HD HmTw hm_kconst(uint64_t k, uint64_t q) { return HmTw{hm_to_mont(k, q), 0}; }
HM_HD HmW hm_tw_entry(uint64_t w, uint64_t q) { return hm_to_mont(w, q); }
HM_HD uint64_t hm_tw_bits(const HmW &w) { return w; }
#endif

// Harvey-lazy butterflies on the lazy product.  With U = HM_LAZY_Q / 2 (1: mont32, 2: generic) every value of a pass stays below 8U q
// (mont32: 8q <= 2^63, the Montgomery product's operand range; generic: 16q < 2^64, the Shoup product takes any 64-bit operand) and
// the conditional subtraction is SCHEDULED over the stages (hm_fwd_kind):
//   kind 0: no subtraction          X < 6U q          ->  X', Y' < 8U q
//   kind 1: X -= 4U q if X >= 4U q  X < 8U q          ->  X', Y' < 6U q
//   kind 2: both 4U q and 2U q      X < 8U q          ->  X', Y' < 4U q   (last stage of a transform: hm_reduce_fwd follows)
// Y' = X - v + 2U q is computed as (2X + 2U q) - X'  (generic: 2X + 4q may wrap around 2^64; the difference is exact all the same).
template <int KIND>
HM_HD void hm_bfly_fwd_k(uint64_t &X, uint64_t &Y, const HmW &w, const HmBflyMod &m) {
  uint64_t x = X;
#if !defined(HM_ABL_NOCSUB)   // (timing-only ablation: butterflies without their conditional subtractions)
#if HM_GENERIC
  if (KIND >= 1) x = hm_csub_neg(x, m.nq8);
  if (KIND == 2) x = hm_csub_neg(x, m.nq4);
#else
  if (KIND >= 1) x = hm_csub_neg(x, m.nq4);
  if (KIND == 2) x = hm_csub_neg(x, m.nq2);
#endif
#endif
  const uint64_t xn = hm_tw_acc(x, Y, w, m);
#if HM_GENERIC
  Y = ((x << 1) + m.q4) - xn;
#else
  Y = ((x << 1) + m.q2) - xn;
#endif
  X = xn;
}
// inverse (Gentleman-Sande): X, Y in [0, 4q) -> X' in [0, 4q), Y' in [0, HM_LAZY_Q q)   (X + 4q - Y < 8q <= 2^63)
HM_HD void hm_bfly_inv(uint64_t &X, uint64_t &Y, const HmW &w, const HmBflyMod &m) {
  const uint64_t d = (X + m.q4) - Y;
#if defined(HM_ABL_NOCSUB)
  X = X + Y;
#else
  X = hm_csub_neg(X + Y, m.nq4);
#endif
  Y = hm_tw_acc(0, d, w, m);
}
// [0, 4q) -> [0, q)
HM_HD uint64_t hm_reduce4(uint64_t x, uint64_t q) { return hm_csub(hm_csub(x, 2 * q), q); }
// [0, 8q) -> [0, q)
HM_HD uint64_t hm_reduce8(uint64_t x, uint64_t q) {
  return hm_csub(hm_csub(hm_csub(x, 4 * q), 2 * q), q);
}
// what a forward transform hands out: below 2 HM_LAZY_Q q
HM_HD uint64_t hm_reduce_fwd(uint64_t x, uint64_t q) { return HM_GENERIC ? hm_reduce8(x, q) : hm_reduce4(x, q); }
