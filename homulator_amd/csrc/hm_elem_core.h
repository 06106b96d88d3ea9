// hm_elem_core.h — K2 automorphism, K3 element-wise engine, K4 base conversion, synthetic fill.
// Per-thread bodies (no cross-thread state) shared by the HIP kernels and the host emulator.
//
// Reference shapes (the reference carries address tokens only; the arithmetic is the build's):
//   K3 EWE   InsGen::GenEWE src/InsGen.cpp:77-125, "(op1 x op2) + (op3 x op4)" :90-95,
//            adder tree src/Components.cpp:8-57.  Upstream has no opcode (always tagged MULT);
//            the opcodes below are what the stages of src/Operation.cpp actually need.
//   K4 BCONV InsGen::GenBCONV src/InsGen.cpp:263-313 (an InLevel-deep MAC chain per output limb),
//            BCONVU 2x6 MAC array src/Components.cpp:268-295.
//   K2 AUTO  InsGen::GenAUTO src/InsGen.cpp:46-71, AUTOU src/Components.cpp:173-194.
#pragma once
#include "hm_modarith.h"
#include "hm_ntt_core.h"

enum HmEweOp {
  HM_EWE_MUL = 0,        // out = a*b
  HM_EWE_MAC2 = 1,       // out = a*b + c*d
  HM_EWE_MAC_ADD = 2,    // out = a*b + c
  HM_EWE_ADD = 3,        // out = a + c
  HM_EWE_SUB = 4,        // out = a - c
  HM_EWE_MUL_CONST = 5,  // out = a*k
  HM_EWE_SUB_SCALE = 6,  // out = (a - c)*k
  HM_EWE_COPY = 7,       // out = a
  HM_EWE_SUB_SCALE_ADD = 8,  // out = (a - c)*k + d   (ModDown finish fused with the final add)
  HM_EWE_NOPS
};

struct HmEweLimb {
  uint16_t a, b, c, d, out, mod;
};
struct HmEweArgs {
  const uint64_t *a, *b, *c, *d;
  uint64_t *out;
  const HmMod *mods;
  uint32_t logN, n_limbs, op;
  HmEweLimb limb[HM_MAX_LIMBS];
  HmTw k[HM_MAX_LIMBS];  // per-limb constant (Shoup form) for *_CONST / *_SCALE
};

template <int OP>
HM_HD uint64_t hm_ewe_one(uint64_t a, uint64_t b, uint64_t c, uint64_t d, const HmTw &k, const HmMod &m) {
  switch (OP) {
  case HM_EWE_MUL: return hm_mulmod(a, b, m);
  case HM_EWE_MAC2: return hm_barrett((hm_u128)a * b + (hm_u128)c * d, m);
  case HM_EWE_MAC_ADD: return hm_addmod(hm_mulmod(a, b, m), c, m.q);
  case HM_EWE_ADD: return hm_addmod(a, c, m.q);
  case HM_EWE_SUB: return hm_submod(a, c, m.q);
  case HM_EWE_MUL_CONST: return hm_shoup(a, k.w, k.ws, m.q);
  case HM_EWE_SUB_SCALE: return hm_shoup(a - c + m.q, k.w, k.ws, m.q);
  case HM_EWE_SUB_SCALE_ADD: return hm_addmod(hm_shoup(a - c + m.q, k.w, k.ws, m.q), d, m.q);
  default: return a;
  }
}

// operand usage per opcode (bit 0 = a, 1 = b, 2 = c, 3 = d)
HM_HD constexpr int hm_ewe_uses(int op) {
  return op == HM_EWE_MUL ? 3 : op == HM_EWE_MAC2 ? 15 : op == HM_EWE_MAC_ADD ? 7 :
         op == HM_EWE_ADD ? 5 : op == HM_EWE_SUB ? 5 : op == HM_EWE_MUL_CONST ? 1 :
         op == HM_EWE_SUB_SCALE ? 5 : op == HM_EWE_SUB_SCALE_ADD ? 13 : 1;
}

// tensor product of two ciphertexts in one pass over the four inputs (TensorCompute::computeD0/D1/D2,
// src/Operation.cpp:624-739): d0 = a*b, d1 = a*d + c*b, d2 = c*d with a = c00, b = c10, c = c01, d = c11
struct HmTensorLimb {
  uint16_t a, b, c, d, o0, o1, o2, mod;
};
struct HmTensorArgs {
  const uint64_t *a, *b, *c, *d;
  uint64_t *o0, *o1, *o2;
  const HmMod *mods;
  uint32_t logN, n_limbs;
  HmTensorLimb limb[HM_MAX_LIMBS];
};
// Round 4: on the word-wise Montgomery product (hm_mont_acc, q = h 2^32 + 1).  b and d are taken to Montgomery form once (a product
// with 2^128 mod q: b 2^64 mod q + {0, q}, below 1.5q + 2^28), then the four products are exact: x wt 2^-64 with wt = b 2^64 is x b.  Six
// products of 11 instructions + four subtractions where three Barrett reductions of full 128-bit products took about twice as many.
// A product with an operand below q and a constant below 1.5q + 2^28 comes out below 1.1q + 1.
HM_HD void hm_tensor_one(uint64_t a, uint64_t b, uint64_t c, uint64_t d, const HmMod &m, uint64_t &d0, uint64_t &d1, uint64_t &d2) {
#if HM_GENERIC
  d0 = hm_mulmod(a, b, m);   // generic: three Barrett reductions of full 128-bit products
  d1 = hm_barrett((hm_u128)a * d + (hm_u128)c * b, m);
  d2 = hm_mulmod(c, d, m);
#else
  const HmBflyMod bm = hm_bfly_mod(m.q);
  const uint64_t bt = hm_mont_acc(0, b, m.r128, bm), dt = hm_mont_acc(0, d, m.r128, bm);
  d0 = hm_csub_neg(hm_mont_acc(0, a, bt, bm), bm.nq);
  d2 = hm_csub_neg(hm_mont_acc(0, c, dt, bm), bm.nq);
  d1 = hm_mont_acc(hm_mont_acc(0, a, dt, bm), c, bt, bm);   // below 2.2q + 2
  d1 = hm_csub_neg(hm_csub_neg(d1, bm.nq2), bm.nq);
#endif
}

// ---- K5 inner product with the evaluation key (the reference's HPIP unit: InsGen::GenHPIP src/InsGen.cpp:356-406,
// HPIP src/Components.cpp:571-595; in the shipped configs it runs on the EWE as beta-1 MAC groups per key,
// KeySwitch::InnerProduceOperation src/Operation.cpp:294-414).  One pass: acc_k = sum_j x_j * y_{j,k}, k < n_out.
#define HM_IP_MAX_TERMS 4
#define HM_IP_MAX_OUT 2
#define HM_IP_MAX_LIMBS 64
struct HmIpLimb {
  uint16_t x[HM_IP_MAX_TERMS];
  uint16_t y[HM_IP_MAX_OUT][HM_IP_MAX_TERMS];
  uint16_t out[HM_IP_MAX_OUT];
  uint16_t mod, pad;
};
struct HmIpArgs {
  const uint64_t *x, *y;
  uint64_t *out;
  const HmMod *mods;
  uint32_t logN, n_limbs, n_terms, n_out;
  uint32_t x_galois;   // > 1: the x operands are read through the automorphism X -> X^x_galois (round 6, hm_inner_product_ex)
  HmIpLimb limb[HM_IP_MAX_LIMBS];
};

// ---- K4 base conversion: out[t][x] = sum_i in[i][x] * table[i][t] mod q_t (device table: hm_bconv_entry form)
// One launch carries up to HM_BCONV_MAX_PROB independent conversions (the beta digits of a ModUp, the two
// keys of a ModDown): grid = (N / (HM_BCONV_THREADS * HM_BCONV_CPT), output chunks, problems); two adjacent
// coefficients per thread (16-byte accesses; every scalar operand feeds two multiply-adds).
#define HM_BCONV_MAX_IN 32   // parameter set A converts from a 28-limb basis (alpha = 28)
#define HM_BCONV_MAX_OUT 64
#define HM_BCONV_MAX_PROB 256  // problems per launch (blockIdx.z); the records live in a device table cached by content
#ifndef HM_BCONV_CHUNK
#define HM_BCONV_CHUNK 8    // output limbs per block when the launch is small (hm_bconv_batch raises it for big ones)
#endif
#define HM_BCONV_THREADS 256
#ifndef HM_BCONV_CPT
#define HM_BCONV_CPT 2       // coefficients per thread
#endif
struct HmBconvProb {
  const uint64_t *in;
  uint64_t *out;
  const uint64_t *table;  // device, [n_out][HM_BCONV_ROW(n_in)], entries in hm_bconv_entry form, rows zero-padded
  const uint64_t *qn;     // device, [n_out] x {q, -q^-1 mod 2^64} of the output moduli (HmQn)
  uint32_t n_in, n_out;
  // 32-bit entries: indexed by the (wave-uniform) output counter, they must be scalar loads from the kernarg
  // segment; 16-bit entries made hipcc emit a VECTOR load per output, and the modulus record load that depends on it
  // was a second serialised global-memory round trip per output (the kernel was latency-bound on these two loads)
  uint32_t in_limb[HM_BCONV_MAX_IN];
  uint32_t out_limb[HM_BCONV_MAX_OUT];
  // optional epilogue (ep_a != nullptr): out_t = (ep_a[ep_a_limb[t]] - conv_t) * ep_k[t] [+ ep_b[ep_b_limb[t]]]
  const uint64_t *ep_a, *ep_b;
  const HmTw *ep_k;                      // device, [n_out]
  uint32_t ep_a_limb[HM_BCONV_MAX_OUT], ep_b_limb[HM_BCONV_MAX_OUT];
  uint32_t in_packed;                    // the inputs are stored in the split-30 packed form (hm_pack30)
};
struct HmBconvArgs {
  const HmBconvProb *prob;  // device, [n_prob]: read with scalar loads (wave-uniform index)
  uint32_t logN, n_prob;
  uint32_t chunk;           // output limbs per block (the host trades input re-reads against blocks in flight)
};
#if defined(__HIP_DEVICE_COMPILE__)
typedef const HmBconvProb __attribute__((address_space(4))) *HmConstProb;
#define HM_CONST_PROB(p) ((HmConstProb)(uintptr_t)(p))
#define HM_CONST_PROB_T(T, p) ((const T __attribute__((address_space(4))) *)(uintptr_t)(p))
#else
#define HM_CONST_PROB_T(T, p) ((const T *)(p))
typedef const HmBconvProb *HmConstProb;
#define HM_CONST_PROB(p) (p)
#endif

// Split-30 MAC: operands are below 2^60, so y = y1 2^30 + y0 and w = w1 2^30 + w0 with 30-bit halves; each
// of the four partial-product columns y_a w_b stays below 2^64 over 16 terms, so a MAC is four
// v_mad_u64_u32 into plain 64-bit accumulators with no carry chain; the columns are recombined once per
// output.  Device table entries are pre-split: low word = w0, high word = w1 (hm_bconv_pack).
HM_HD uint64_t hm_bconv_pack(uint64_t w) { return (w & 0x3FFFFFFFull) | ((w >> 30) << 32); }
// device form of a conversion factor w for output modulus m: Montgomery form (the accumulator is reduced by
// hm_redc_wide, which divides by 2^64), split-30 packed
HM_HD uint64_t hm_bconv_entry(uint64_t w, const HmMod &m) { return hm_bconv_pack(hm_mulmod(w, m.r64, m)); }

#if defined(__HIP_DEVICE_COMPILE__)
typedef const HmMod __attribute__((address_space(4))) *HmConstMod;
#define HM_CONST_MODS(p) ((HmConstMod)(uintptr_t)(p))
#else
typedef const HmMod *HmConstMod;
#define HM_CONST_MODS(p) (p)
#endif
// The conversion table is read through the SCALAR cache: every lane of a wave needs the same entries, so they are
// s_load'ed into SGPRs (asynchronously, no VGPRs, no LDS, no barrier) and feed v_mad_u64_u32 as scalar operands.
// hipcc only emits scalar loads for memory it knows to be constant, hence the constant-address-space view of the
// table pointer.  Device layout: [n_out][HM_BCONV_ROW(n_in)] entries (hm_bconv_entry form), a row = the factors of
// one output limb padded with zeros to a multiple of 8 entries, so that a row is fetched by 64-byte scalar loads
// (s_load_dwordx16) instead of one 8-byte load plus three address instructions per entry.
struct HmRow8 {
  uint64_t w[8];
};
#define HM_BCONV_ROW(n_in) ((((n_in) + 7u) / 8u) * 8u)
#if defined(__HIP_DEVICE_COMPILE__)
typedef const HmRow8 __attribute__((address_space(4))) *HmConstRow8;
#define HM_CONST_ROWS(p) ((HmConstRow8)(uintptr_t)(p))
#else
typedef const HmRow8 *HmConstRow8;
#define HM_CONST_ROWS(p) ((const HmRow8 *)(p))
#endif

// one output limb of one coefficient: sum_i y_i w_i with the split-30 columns, Montgomery-reduced
template <int N_IN>
HM_HD uint64_t hm_bconv_dot(const uint32_t (&yl)[N_IN], const uint32_t (&yh)[N_IN], const HmRow8 (&row)[(N_IN + 7) / 8], uint64_t q, uint64_t nqinv) {
  hm_u128 acc = 0;
  constexpr int GROUPS = (N_IN + 15) / 16;  // 16 terms per carry-free column group (each column stays below 2^64)
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) {
    uint64_t s00 = 0, s01 = 0, s10 = 0, s11 = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int i = g * 16 + j;
      if (i < N_IN) {
        const uint64_t w = row[i >> 3].w[i & 7];
        const uint32_t wl = (uint32_t)w, wh = (uint32_t)(w >> 32);
        s00 += (uint64_t)yl[i] * wl;
#if !defined(HM_ABL_BCONV_HALF)   // (timing-only ablation: half of the multiply-adds of a conversion)
        s01 += (uint64_t)yl[i] * wh;
        s10 += (uint64_t)yh[i] * wl;
#endif
        s11 += (uint64_t)yh[i] * wh;
      }
    }
    acc += (hm_u128)s00 + (((hm_u128)s01 + s10) << 30) + ((hm_u128)s11 << 60);   // 32 products of < 2^120: below 2^125
  }
  HmMod m;
  m.q = q;
  m.nqinv = nqinv;
  return hm_redc_wide<N_IN>(acc, m);
}

// the same sum for up to 16 consecutive inputs, NOT reduced: the carry-free split-30 columns of one group, recombined into a 128-bit value
// (round 6: the fused conversion + first pass takes digits of 16 .. 32 limbs in two such groups, each loaded, multiplied and dropped before
// the next one, so that at most 16 inputs x 2 coefficients x 2 halves = 64 VGPRs of inputs are live; hm_redc_wide<N_IN> reduces the sum of
// the groups).  row = the table entries of these inputs (a whole number of 8-entry groups: the chunk starts at a multiple of 8).
template <int CN>
HM_HD hm_u128 hm_bconv_cols(const uint32_t (&yl)[CN], const uint32_t (&yh)[CN], const HmRow8 (&row)[(CN + 7) / 8]) {
  static_assert(CN >= 1 && CN <= 16, "one carry-free column group");
  uint64_t s00 = 0, s01 = 0, s10 = 0, s11 = 0;
#pragma unroll
  for (int i = 0; i < CN; ++i) {
    const uint64_t w = row[i >> 3].w[i & 7];
    const uint32_t wl = (uint32_t)w, wh = (uint32_t)(w >> 32);
    s00 += (uint64_t)yl[i] * wl;
    s01 += (uint64_t)yl[i] * wh;
    s10 += (uint64_t)yh[i] * wl;
    s11 += (uint64_t)yh[i] * wh;
  }
  return (hm_u128)s00 + (((hm_u128)s01 + s10) << 30) + ((hm_u128)s11 << 60);
}

// per-output modulus constants, stored behind the table rows (HmBconvProb::qn): no load depends on another load
struct HmQn {
  uint64_t q, nqinv;
};
#if defined(__HIP_DEVICE_COMPILE__)
typedef const HmQn __attribute__((address_space(4))) *HmConstQn;
#define HM_CONST_QN(p) ((HmConstQn)(uintptr_t)(p))
#else
typedef const HmQn *HmConstQn;
#define HM_CONST_QN(p) ((const HmQn *)(p))
#endif

// CPT coefficients x, x+1, .. (CPT = 2: one 16-byte access per input / output limb) for outputs [t0, t1).  The inputs
// stay in registers for all outputs of the chunk; a row feeds 4 * N_IN * CPT multiply-adds as scalar operands.
// (Requesting the NEXT output's row ahead of the products changed nothing: hipcc sinks the loads to the end of the
// iteration, and the kernel is bound by VALU issue — per output and coefficient 4 * N_IN multiply-adds plus ~30
// instructions of column recombination and Montgomery reduction — not by the scalar-cache latency.)
// PACKED: the inputs are stored in the split-30 packed form (hm_pack30): the halves are taken as they are (a template parameter: chosen per
// value at run time the select costs what the packed form saves)
template <int N_IN, int CPT, bool PACKED = false, class PROB>
HM_HD void hm_bconv_thread(const PROB &p, uint32_t logN, uint32_t x, uint32_t t0, uint32_t t1) {
  const size_t N = (size_t)1 << logN;
  constexpr int NG = (N_IN + 7) / 8;  // 8-entry groups per row
  HmConstRow8 tab = HM_CONST_ROWS(p.table);
  HmConstQn qn = HM_CONST_QN(p.qn);
  uint32_t yl[CPT][N_IN], yh[CPT][N_IN];
#pragma unroll
  for (int i = 0; i < N_IN; ++i) {
    uint64_t v[2] = {0, 0};
    // uniform limb base + one lane offset: all N_IN loads issue back to back (with a 64-bit address per load hipcc
    // recycled the registers of loaded data for the next addresses and serialised the loads pairwise)
    if (CPT == 2) hm_bld2(p.in + (size_t)p.in_limb[i] * N, x << 3, v[0], v[1]);
    else v[0] = p.in[(size_t)p.in_limb[i] * N + x];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      if (PACKED) { yl[c][i] = (uint32_t)v[c]; yh[c][i] = (uint32_t)(v[c] >> 32); }
      else { yl[c][i] = (uint32_t)v[c] & 0x3FFFFFFFu; yh[c][i] = (uint32_t)(v[c] >> 30); }
    }
  }
  for (uint32_t t = t0; t < t1; ++t) {
    HmRow8 row[NG];
#if defined(HM_ABL_BCONV_NOTAB)   // timing-only ablation: one table row for every output (no per-output scalar loads)
    const HmQn m = qn[t0];
#pragma unroll
    for (int g = 0; g < NG; ++g) row[g] = tab[t0 * NG + g];
#else
    const HmQn m = qn[t];
#pragma unroll
    for (int g = 0; g < NG; ++g) row[g] = tab[t * NG + g];
#endif
    uint64_t r[2] = {0, 0};
#pragma unroll
    for (int c = 0; c < CPT; ++c) r[c] = hm_bconv_dot<N_IN>(yl[c], yh[c], row, m.q, m.nqinv);
    if (p.ep_a) {   // (wave-uniform) out = (a - conv) * k [+ b]
#if defined(__HIP_DEVICE_COMPILE__)
      typedef const HmTw __attribute__((address_space(4))) *ConstTw;
      const HmTw k = {((ConstTw)(uintptr_t)p.ep_k)[t].w, ((ConstTw)(uintptr_t)p.ep_k)[t].ws};
#else
      const HmTw k = p.ep_k[t];
#endif
      uint64_t av[2] = {0, 0}, bv[2] = {0, 0};
      if (CPT == 2) hm_bld2(p.ep_a + (size_t)p.ep_a_limb[t] * N, x << 3, av[0], av[1]);
      else av[0] = p.ep_a[(size_t)p.ep_a_limb[t] * N + x];
      if (p.ep_b) {
        if (CPT == 2) hm_bld2(p.ep_b + (size_t)p.ep_b_limb[t] * N, x << 3, bv[0], bv[1]);
        else bv[0] = p.ep_b[(size_t)p.ep_b_limb[t] * N + x];
      }
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        r[c] = hm_shoup(av[c] + m.q - r[c], k.w, k.ws, m.q);
        if (p.ep_b) r[c] = hm_addmod(r[c], bv[c], m.q);
      }
    }
    if (CPT == 2) hm_bst2(p.out + (size_t)p.out_limb[t] * N, x << 3, r[0], r[1]);
    else p.out[(size_t)p.out_limb[t] * N + x] = r[0];
  }
}

// (K2, the automorphism's index map hm_auto_src: hm_ntt_core.h — the transforms gather through it as well)

// ---- deterministic synthetic data (same definition as oracle/homoracle.c ho_fill_uniform)
HM_HD uint64_t hm_mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
HM_HD uint64_t hm_synth(uint64_t stream, uint32_t x, uint64_t q) {
  uint64_t z = hm_mix64(stream * 0xD1342543DE82EF95ull + (uint64_t)x * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull);
  return hm_mulhi(z, q);
}
