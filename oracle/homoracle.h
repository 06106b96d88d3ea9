/*
 * homoracle.h — CPU oracle for the RNS-CKKS hmult / hrotate + hybrid key-switch datapath.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under homulator_amd/ (the product) may include, link or call
 * this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as
 * the checker / the CPU number printed beside the GPU number.
 *
 * PARITY STATUS: "parity unpinned" for ARITHMETIC.  The reference (FHE-ACCELE/Homulator) is a
 * cycle-level timing simulator: it defines no modulus, twiddle, base-conversion table or expected
 * value anywhere.  What this file follows from the reference is STRUCTURE —
 * stage list, limb counts, digit partition, operand arity — cited per function as
 * `src/Operation.cpp:<lines>` / `src/InsGen.cpp:<lines>`.  The arithmetic follows the published
 * RNS-CKKS hybrid key-switching construction (the accelerators the reference models: SHARP
 * README.md:53, ARK include/Components.h:196) under the conventions frozen in SURVEY.md
 * Appendix A / DESIGN.md §2, and is pinned by independent big-integer known-answer tests
 * (tests/test_oracle_kat.py) rather than by reference vectors.  STRUCTURE parity is pinned by the
 * compiled reference itself (oracle/_ref, tests/golden/structural.json).
 */
#ifndef HOMORACLE_H
#define HOMORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ho_ctx ho_ctx;

/* Parameter set: N = 2^logN, L Q-primes q_0..q_{L-1}, K special primes p_0..p_{K-1} (the reference
 * sizes the special basis as exactly alpha limbs: src/Operation.cpp:160,193,297-304).
 * Primes: the L+K largest primes below 2^60 that are 1 mod 2^32 (hence 1 mod 2N), in descending order; the first L
 * are Q, the next K are P.  psi = smallest primitive 2N-th root of unity of each prime.
 * "mod id" m: m < L -> q_m ; m >= L -> p_{m-L}. */
ho_ctx *ho_create(uint32_t logN, uint32_t L, uint32_t K);
/* the same over a caller-given chain: L+K distinct primes = 1 mod 2N below 2^60 (first L = Q, next K = P); NULL = the default chain.
 * Returns NULL if a modulus does not qualify. */
ho_ctx *ho_create_chain(uint32_t logN, uint32_t L, uint32_t K, const uint64_t *moduli);
/* the `count` largest primes below 2^bits that are 1 mod 2N, descending (bits = 60: the chain of SURVEY.md 8(d) as written; bits = 36:
 * 36-bit words as the reference's configuration models).  Returns 0 if the range does not hold that many. */
int ho_chain_below(uint32_t logN, uint32_t bits, uint32_t count, uint64_t *out);
void ho_destroy(ho_ctx *);
uint32_t ho_N(const ho_ctx *);
uint32_t ho_L(const ho_ctx *);
uint32_t ho_K(const ho_ctx *);
uint64_t ho_modulus(const ho_ctx *, uint32_t mod_id);
uint64_t ho_psi(const ho_ctx *, uint32_t mod_id);
void ho_set_threads(int nthreads); /* OpenMP threads for the limb loops (cpu_baseline leg) */

/* scalar helpers (exposed so tests can KAT them against Python big ints) */
uint64_t ho_mulmod(uint64_t a, uint64_t b, uint64_t q);
uint64_t ho_powmod(uint64_t a, uint64_t e, uint64_t q);
uint64_t ho_invmod(uint64_t a, uint64_t q);

/* K1: negacyclic NTT of one limb, in place.  Forward = Cooley-Tukey, natural-order input,
 * bit-reversed-order output; inverse = Gentleman-Sande, bit-reversed in, natural out, scaled by
 * N^-1.  One reference "instruction" is one 256-coefficient batch of this (src/InsGen.cpp:17-44). */
void ho_ntt(const ho_ctx *, uint32_t mod_id, uint64_t *a, int inverse);
/* many limbs: a is [n][N], mod_ids[n] */
void ho_ntt_limbs(const ho_ctx *, const uint32_t *mod_ids, uint32_t n, uint64_t *a, int inverse);

/* K2: automorphism X -> X^g (g odd).  eval form = index permutation of the bit-reversed NTT
 * output (modulus independent); coef form = signed index map.  (src/InsGen.cpp:46-71) */
void ho_automorph_eval(const ho_ctx *, const uint64_t *in, uint64_t *out, uint32_t g);
void ho_automorph_coef(const ho_ctx *, uint32_t mod_id, const uint64_t *in, uint64_t *out, uint32_t g);

/* K3: element-wise engine, one limb.  The reference EWE is "(op1 x op2) + (op3 x op4)" with no
 * opcode (src/InsGen.cpp:77-125, 90-95); the build adds explicit opcodes. */
enum ho_ewe_op {
  HO_EWE_MUL = 0,       /* out = a*b                  */
  HO_EWE_MAC2 = 1,      /* out = a*b + c*d            */
  HO_EWE_MAC_ADD = 2,   /* out = a*b + c              */
  HO_EWE_ADD = 3,       /* out = a + c                */
  HO_EWE_SUB = 4,       /* out = a - c                */
  HO_EWE_MUL_CONST = 5, /* out = a * k                */
  HO_EWE_SUB_SCALE = 6, /* out = (a - c) * k          */
  HO_EWE_COPY = 7       /* out = a                    */
};
void ho_ewe(const ho_ctx *, int op, uint32_t mod_id, const uint64_t *a, const uint64_t *b,
            const uint64_t *c, const uint64_t *d, uint64_t k, uint64_t *out);

/* K4: fast base conversion.  in: [n_in][N] coefficient-form limbs x_i mod q_i (NOT pre-scaled).
 * scale step  y_i = x_i * [(Q_D/q_i)^-1]_{q_i}       (src/Operation.cpp:104-135, 447-487)
 * matmul step out_t = sum_i y_i * [Q_D/q_i]_t mod t   (src/Operation.cpp:137-188, 489-519;
 *                                                      src/InsGen.cpp:263-313)
 * ho_bconv_consts fills qhat_inv[n_in] and table[n_in][n_out] (row-major). */
void ho_bconv_consts(const ho_ctx *, const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids,
                     uint32_t n_out, uint64_t *qhat_inv, uint64_t *table);
void ho_bconv_scale(const ho_ctx *, const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids,
                    uint32_t n_out, const uint64_t *in, uint64_t *scaled);
void ho_bconv_matmul(const ho_ctx *, const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids,
                     uint32_t n_out, const uint64_t *scaled, uint64_t *out);

/* Hybrid key switch at level ell (ell active Q limbs), alpha = K.  d: [ell][N] eval form.
 * evk: [beta][2][E][N], E = ell + K, extended limb order = Q limbs then P limbs.
 * out0/out1: [ell][N].  Stage order = src/Operation.cpp:9-54.
 * If dump != NULL it receives named intermediates (see ho_ks_dump). */
typedef struct ho_ks_dump {
  uint64_t *modup_intt;   /* [ell][N]      ModUpINTTOut  (plain INTT, before scale)            */
  uint64_t *modup_decomp; /* [ell][N]      ModUpDecompOut (scaled)                             */
  uint64_t *ext;          /* [beta][E][N]  NTTOut_beta(j) (eval-form extended digits)          */
  uint64_t *ip;           /* [2][E][N]     InnerProduceOut_Key{k}                              */
  uint64_t *moddown_intt; /* [2][K][N]     INTTOut_ModDown_Key(k)                              */
  uint64_t *moddown_bconv;/* [2][ell][N]   ModdownBConvOut_Key{k} (coefficient form)           */
  uint64_t *moddown_ntt;  /* [2][ell][N]   NTTOut_ModDown_Key(k)                               */
} ho_ks_dump;
void ho_keyswitch(const ho_ctx *, uint32_t ell, const uint64_t *d, const uint64_t *evk,
                  uint64_t *out0, uint64_t *out1, const ho_ks_dump *dump);

/* Rescale one polynomial: x [ell][N] eval form -> out [ell-1][N]. (src/Operation.cpp:741-911) */
void ho_rescale(const ho_ctx *, uint32_t ell, const uint64_t *x, uint64_t *out);

/* HMULT (src/Operation.cpp:913-1023): ct = [2][ell][N] (c0 limbs then c1 limbs), eval form.
 * out [2][ell-1][N] if do_rescale else [2][ell][N]. */
void ho_hmult(const ho_ctx *, uint32_t ell, const uint64_t *ct1, const uint64_t *ct2,
              const uint64_t *evk, int do_rescale, uint64_t *out);
/* HROTATE (src/Operation.cpp:1271-1358): out [2][ell][N]. */
void ho_hrotate(const ho_ctx *, uint32_t ell, const uint64_t *ct, uint32_t galois,
                const uint64_t *evk, uint64_t *out);
/* HADD / PMULT / PADD (src/Operation.cpp:1114-1176, 1460-1523, 1625-1680) */
void ho_hadd(const ho_ctx *, uint32_t ell, const uint64_t *ct1, const uint64_t *ct2, uint64_t *out);
void ho_pmult(const ho_ctx *, uint32_t ell, const uint64_t *ct, const uint64_t *pt, uint64_t *out);
void ho_padd(const ho_ctx *, uint32_t ell, const uint64_t *ct, const uint64_t *pt, uint64_t *out);

/* Deterministic synthetic data: SplitMix64 stream, uniform in [0, modulus) by rejection-free
 * 128-bit multiply-shift (value = (x * q) >> 64).  Fills [n][N] limbs. */
void ho_fill_uniform(const ho_ctx *, const uint32_t *mod_ids, uint32_t n, uint64_t seed, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif
