/*
 * homoracle.c — CPU oracle (plain C, unsigned __int128) for the hmult / hrotate + hybrid
 * key-switch datapath.  TEST INFRASTRUCTURE ONLY — see homoracle.h for the rules and for the
 * "parity unpinned" statement.  Structure follows /root/reference (cited per function);
 * arithmetic follows SURVEY.md Appendix A.
 *
 * Build: make -C oracle   (gcc -O3 -march=native -fopenmp -shared -fPIC)
 */
#include "homoracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

struct ho_ctx {
  uint32_t logN, N, L, K;
  uint64_t *mod;      /* [L+K] */
  uint64_t *psi;      /* [L+K] */
  uint64_t *ninv;     /* [L+K] N^-1 mod m */
  uint64_t **w;       /* [L+K][N] psi^bitrev(i)           */
  uint64_t **ws;      /* Shoup companions floor(w*2^64/m) */
  uint64_t **wi;      /* [L+K][N] psi^-bitrev(i)          */
  uint64_t **wis;
};

static int g_threads = 1;
void ho_set_threads(int n) { g_threads = n < 1 ? 1 : n; }

/* ------------------------------------------------------------------ scalar arithmetic */
uint64_t ho_mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
static inline uint64_t addmod(uint64_t a, uint64_t b, uint64_t q) { uint64_t s = a + b; return s >= q ? s - q : s; }
static inline uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }

uint64_t ho_powmod(uint64_t a, uint64_t e, uint64_t q) {
  uint64_t r = 1 % q; a %= q;
  while (e) { if (e & 1) r = ho_mulmod(r, a, q); a = ho_mulmod(a, a, q); e >>= 1; }
  return r;
}
uint64_t ho_invmod(uint64_t a, uint64_t q) { return ho_powmod(a, q - 2, q); } /* q prime */

/* Barrett for the element-wise loops: ratio = floor(2^128 / q) as two words; q < 2^62.
 * Validated against ho_mulmod (%) by tests/test_oracle_kat.py. */
typedef struct { uint64_t q, r0, r1; } barrett_t;
static barrett_t barrett_make(uint64_t q) {
  barrett_t b; b.q = q;
  /* floor(2^128/q): long division of 2^128 by q */
  u128 hi = (((u128)1) << 64) / q;             /* floor(2^64/q) * ... first word */
  u128 rem = (((u128)1) << 64) % q;
  u128 lo = (rem << 64) / q;
  b.r1 = (uint64_t)hi; b.r0 = (uint64_t)lo;
  return b;
}
static inline uint64_t barrett_reduce128(u128 z, const barrett_t *b) {
  uint64_t z0 = (uint64_t)z, z1 = (uint64_t)(z >> 64);
  /* quotient estimate = floor(z * ratio / 2^128), dropping the lowest partial product */
  u128 t = ((u128)z0 * b->r0) >> 64;
  u128 m1 = (u128)z0 * b->r1;
  u128 m2 = (u128)z1 * b->r0;
  u128 mid = t + (uint64_t)m1 + (uint64_t)m2;
  uint64_t qhat = (uint64_t)(mid >> 64) + (uint64_t)(m1 >> 64) + (uint64_t)(m2 >> 64) + z1 * b->r1;
  uint64_t r = z0 - qhat * b->q;
  while (r >= b->q) r -= b->q;
  return r;
}
static inline uint64_t bmul(uint64_t a, uint64_t x, const barrett_t *b) { return barrett_reduce128((u128)a * x, b); }

static inline uint64_t shoup_make(uint64_t w, uint64_t q) { return (uint64_t)((((u128)w) << 64) / q); }
/* returns w*x mod q in [0,q) given ws = floor(w*2^64/q) */
static inline uint64_t shoup_mul(uint64_t x, uint64_t w, uint64_t ws, uint64_t q) {
  uint64_t h = (uint64_t)(((u128)x * ws) >> 64);
  uint64_t r = x * w - h * q;
  return r >= q ? r - q : r;
}

/* ------------------------------------------------------------------ parameter generation */
static int is_prime_u64(uint64_t n) {
  if (n < 2) return 0;
  static const uint64_t sp[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (int i = 0; i < 12; ++i) { if (n == sp[i]) return 1; if (n % sp[i] == 0) return 0; }
  uint64_t d = n - 1; int s = 0;
  while (!(d & 1)) { d >>= 1; ++s; }
  for (int i = 0; i < 12; ++i) { /* deterministic for n < 3.3e24 */
    uint64_t x = ho_powmod(sp[i], d, n);
    if (x == 1 || x == n - 1) continue;
    int comp = 1;
    for (int r = 1; r < s; ++r) { x = ho_mulmod(x, x, n); if (x == n - 1) { comp = 0; break; } }
    if (comp) return 0;
  }
  return 1;
}

static uint32_t bitrev(uint32_t x, uint32_t bits) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

/* smallest primitive 2N-th root of unity mod q */
static uint64_t min_primitive_root_2n(uint64_t q, uint32_t N) {
  uint64_t twoN = 2ull * N, e = (q - 1) / twoN, r = 0;
  for (uint64_t x = 2;; ++x) {
    r = ho_powmod(x, e, q);
    if (ho_powmod(r, N, q) == q - 1) break; /* r^N = -1  <=> order exactly 2N */
  }
  /* all primitive roots are r^k, k odd: take the minimum */
  uint64_t r2 = ho_mulmod(r, r, q), cur = r, best = r;
  for (uint32_t k = 1; k < N; ++k) { cur = ho_mulmod(cur, r2, q); if (cur < best) best = cur; }
  return best;
}

void ho_destroy(ho_ctx *c);
/* the `count` largest primes below 2^bits that are 1 mod 2N, descending: SURVEY.md 8(d)'s chain is bits = 60; bits = 36 gives a chain of
 * 36-bit words as the reference's configuration models (config/config_4.cfg:9 elementBitWidth, script/README.md:17-22).  Returns 0 when
 * the range does not hold that many. */
int ho_chain_below(uint32_t logN, uint32_t bits, uint32_t count, uint64_t *out) {
  if (bits < 21 || bits > 60 || logN + 1 >= bits) return 0;
  const uint64_t step = 2ull << logN;
  uint64_t cand = (1ull << bits) + 1;
  for (uint32_t m = 0; m < count;) {
    if (cand <= step) return 0;
    cand -= step;
    if (is_prime_u64(cand)) out[m++] = cand;
  }
  return 1;
}

/* moduli == NULL: the default chain, the L+K largest primes below 2^60 that are 1 mod 2^32 (hence 1 mod 2N): DESIGN.md section 2.
 * Otherwise L+K distinct primes = 1 mod 2N below 2^60, first L = Q, next K = P (checked). */
ho_ctx *ho_create_chain(uint32_t logN, uint32_t L, uint32_t K, const uint64_t *moduli) {
  if (logN < 2 || logN > 17 || L == 0) return NULL;
  ho_ctx *c = (ho_ctx *)calloc(1, sizeof(*c));
  c->logN = logN; c->N = 1u << logN; c->L = L; c->K = K;
  uint32_t M = L + K, N = c->N;
  c->mod = calloc(M, 8); c->psi = calloc(M, 8); c->ninv = calloc(M, 8);
  c->w = calloc(M, sizeof(void *)); c->ws = calloc(M, sizeof(void *));
  c->wi = calloc(M, sizeof(void *)); c->wis = calloc(M, sizeof(void *));
  if (moduli) {
    for (uint32_t m = 0; m < M; ++m) {
      const uint64_t q = moduli[m];
      int ok = !(q >> 60) && q > 2ull * N && (q - 1) % (2ull * N) == 0 && is_prime_u64(q);
      for (uint32_t j = 0; ok && j < m; ++j) ok = moduli[j] != q;
      if (!ok) { ho_destroy(c); return NULL; }
      c->mod[m] = q;
    }
  } else {
    uint64_t step = 1ull << 32, cand = (1ull << 60) + 1;
    for (uint32_t m = 0; m < M;) {
      cand -= step;
      if (is_prime_u64(cand)) c->mod[m++] = cand;
    }
  }
  for (uint32_t m = 0; m < M; ++m) {
    uint64_t q = c->mod[m];
    uint64_t psi = min_primitive_root_2n(q, N), psi_inv = ho_invmod(psi, q);
    c->psi[m] = psi; c->ninv[m] = ho_invmod(N, q);
    c->w[m] = malloc(8ull * N); c->ws[m] = malloc(8ull * N);
    c->wi[m] = malloc(8ull * N); c->wis[m] = malloc(8ull * N);
    uint64_t p = 1, pi = 1;
    for (uint32_t i = 0; i < N; ++i) { /* p = psi^i */
      uint32_t r = bitrev(i, logN);
      c->w[m][r] = p; c->wi[m][r] = pi;
      p = ho_mulmod(p, psi, q); pi = ho_mulmod(pi, psi_inv, q);
    }
    for (uint32_t i = 0; i < N; ++i) { c->ws[m][i] = shoup_make(c->w[m][i], q); c->wis[m][i] = shoup_make(c->wi[m][i], q); }
  }
  return c;
}

ho_ctx *ho_create(uint32_t logN, uint32_t L, uint32_t K) { return ho_create_chain(logN, L, K, NULL); }

void ho_destroy(ho_ctx *c) {
  if (!c) return;
  for (uint32_t m = 0; m < c->L + c->K; ++m) { free(c->w[m]); free(c->ws[m]); free(c->wi[m]); free(c->wis[m]); }
  free(c->w); free(c->ws); free(c->wi); free(c->wis); free(c->mod); free(c->psi); free(c->ninv); free(c);
}
uint32_t ho_N(const ho_ctx *c) { return c->N; }
uint32_t ho_L(const ho_ctx *c) { return c->L; }
uint32_t ho_K(const ho_ctx *c) { return c->K; }
uint64_t ho_modulus(const ho_ctx *c, uint32_t m) { return c->mod[m]; }
uint64_t ho_psi(const ho_ctx *c, uint32_t m) { return c->psi[m]; }

/* ------------------------------------------------------------------ K1: NTT
 * Reference shape: InsGen::GenNTT src/InsGen.cpp:17-44 (one instruction per 256-coefficient
 * batch), NTTU src/Components.cpp:380-436 (8 + 8 butterfly stages).  The maths is the build's. */
void ho_ntt(const ho_ctx *c, uint32_t m, uint64_t *a, int inverse) {
  const uint64_t q = c->mod[m]; const uint32_t N = c->N;
  if (!inverse) {
    const uint64_t *w = c->w[m], *ws = c->ws[m];
    uint32_t t = N;
    for (uint32_t mm = 1; mm < N; mm <<= 1) {
      t >>= 1;
      for (uint32_t i = 0; i < mm; ++i) {
        uint32_t j1 = 2 * i * t; uint64_t S = w[mm + i], Ss = ws[mm + i];
        for (uint32_t j = j1; j < j1 + t; ++j) {
          uint64_t U = a[j], V = shoup_mul(a[j + t], S, Ss, q);
          a[j] = addmod(U, V, q); a[j + t] = submod(U, V, q);
        }
      }
    }
  } else {
    const uint64_t *w = c->wi[m], *ws = c->wis[m];
    uint32_t t = 1;
    for (uint32_t mm = N; mm > 1; mm >>= 1) {
      uint32_t h = mm >> 1, j1 = 0;
      for (uint32_t i = 0; i < h; ++i) {
        uint64_t S = w[h + i], Ss = ws[h + i];
        for (uint32_t j = j1; j < j1 + t; ++j) {
          uint64_t U = a[j], V = a[j + t];
          a[j] = addmod(U, V, q); a[j + t] = shoup_mul(submod(U, V, q), S, Ss, q);
        }
        j1 += 2 * t;
      }
      t <<= 1;
    }
    uint64_t ni = c->ninv[m], nis = shoup_make(ni, q);
    for (uint32_t j = 0; j < N; ++j) a[j] = shoup_mul(a[j], ni, nis, q);
  }
}

void ho_ntt_limbs(const ho_ctx *c, const uint32_t *ids, uint32_t n, uint64_t *a, int inverse) {
#pragma omp parallel for num_threads(g_threads) schedule(dynamic)
  for (uint32_t i = 0; i < n; ++i) ho_ntt(c, ids[i], a + (size_t)i * c->N, inverse);
}

/* ------------------------------------------------------------------ K2: automorphism
 * Reference shape: InsGen::GenAUTO src/InsGen.cpp:46-71 (no Galois parameter exists upstream). */
void ho_automorph_eval(const ho_ctx *c, const uint64_t *in, uint64_t *out, uint32_t g) {
  const uint32_t N = c->N, lg = c->logN, mask = 2 * N - 1;
  for (uint32_t i = 0; i < N; ++i) {
    uint32_t e = (uint32_t)(((uint64_t)g * (2 * bitrev(i, lg) + 1)) & mask); /* odd exponent */
    out[i] = in[bitrev((e - 1) >> 1, lg)];
  }
}
void ho_automorph_coef(const ho_ctx *c, uint32_t m, const uint64_t *in, uint64_t *out, uint32_t g) {
  const uint32_t N = c->N, mask = 2 * N - 1; const uint64_t q = c->mod[m];
  for (uint32_t i = 0; i < N; ++i) {
    uint32_t e = (uint32_t)(((uint64_t)i * g) & mask);
    if (e < N) out[e] = in[i]; else out[e - N] = in[i] ? q - in[i] : 0;
  }
}

/* ------------------------------------------------------------------ K3: element-wise engine
 * Reference shape: InsGen::GenEWE src/InsGen.cpp:77-125; EWE adder tree src/Components.cpp:8-57. */
void ho_ewe(const ho_ctx *c, int op, uint32_t m, const uint64_t *a, const uint64_t *b, const uint64_t *cc,
            const uint64_t *d, uint64_t k, uint64_t *out) {
  const uint64_t q = c->mod[m]; const uint32_t N = c->N; barrett_t br = barrett_make(q);
  switch (op) {
  case HO_EWE_MUL: for (uint32_t i = 0; i < N; ++i) out[i] = bmul(a[i], b[i], &br); break;
  case HO_EWE_MAC2: for (uint32_t i = 0; i < N; ++i) out[i] = barrett_reduce128((u128)a[i] * b[i] + (u128)cc[i] * d[i], &br); break;
  case HO_EWE_MAC_ADD: for (uint32_t i = 0; i < N; ++i) out[i] = addmod(bmul(a[i], b[i], &br), cc[i], q); break;
  case HO_EWE_ADD: for (uint32_t i = 0; i < N; ++i) out[i] = addmod(a[i], cc[i], q); break;
  case HO_EWE_SUB: for (uint32_t i = 0; i < N; ++i) out[i] = submod(a[i], cc[i], q); break;
  case HO_EWE_MUL_CONST: for (uint32_t i = 0; i < N; ++i) out[i] = bmul(a[i], k, &br); break;
  case HO_EWE_SUB_SCALE: for (uint32_t i = 0; i < N; ++i) out[i] = bmul(submod(a[i], cc[i], q), k, &br); break;
  case HO_EWE_COPY: memcpy(out, a, 8ull * N); break;
  default: fprintf(stderr, "ho_ewe: bad opcode %d\n", op); abort();
  }
}

/* ------------------------------------------------------------------ K4: base conversion */
void ho_bconv_consts(const ho_ctx *c, const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids,
                     uint32_t n_out, uint64_t *qhat_inv, uint64_t *table) {
  for (uint32_t i = 0; i < n_in; ++i) {
    uint64_t qi = c->mod[in_ids[i]], prod = 1;
    for (uint32_t k = 0; k < n_in; ++k) if (k != i) prod = ho_mulmod(prod, c->mod[in_ids[k]] % qi, qi);
    qhat_inv[i] = ho_invmod(prod, qi);
    for (uint32_t t = 0; t < n_out; ++t) {
      uint64_t qt = c->mod[out_ids[t]], pr = 1;
      for (uint32_t k = 0; k < n_in; ++k) if (k != i) pr = ho_mulmod(pr, c->mod[in_ids[k]] % qt, qt);
      table[(size_t)i * n_out + t] = pr;
    }
  }
}
/* scale step: ModUpDecompFusionBConvStep1 src/Operation.cpp:104-135 / ModDownBConvStep1 :447-487 */
void ho_bconv_scale(const ho_ctx *c, const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids,
                    uint32_t n_out, const uint64_t *in, uint64_t *scaled) {
  uint64_t *qhi = malloc(8ull * n_in), *tab = malloc(8ull * n_in * (n_out ? n_out : 1));
  ho_bconv_consts(c, in_ids, n_in, out_ids, n_out, qhi, tab);
#pragma omp parallel for num_threads(g_threads)
  for (uint32_t i = 0; i < n_in; ++i)
    ho_ewe(c, HO_EWE_MUL_CONST, in_ids[i], in + (size_t)i * c->N, 0, 0, 0, qhi[i], scaled + (size_t)i * c->N);
  free(qhi); free(tab);
}
/* matmul step: ModUpBConvStep2 src/Operation.cpp:137-188 / ModDownBConvStep2 :489-519;
 * per output limb a d_j-deep MAC chain (InsGen::GenBCONV src/InsGen.cpp:263-313). */
void ho_bconv_matmul(const ho_ctx *c, const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids,
                     uint32_t n_out, const uint64_t *scaled, uint64_t *out) {
  const uint32_t N = c->N;
  uint64_t *qhi = malloc(8ull * n_in), *tab = malloc(8ull * n_in * n_out);
  ho_bconv_consts(c, in_ids, n_in, out_ids, n_out, qhi, tab);
#pragma omp parallel for num_threads(g_threads)
  for (uint32_t t = 0; t < n_out; ++t) {
    barrett_t br = barrett_make(c->mod[out_ids[t]]);
    uint64_t *o = out + (size_t)t * N;
    for (uint32_t x = 0; x < N; ++x) {
      u128 acc = 0; /* n_in <= 16 terms of < 2^120 each: no overflow */
      for (uint32_t i = 0; i < n_in; ++i) acc += (u128)scaled[(size_t)i * N + x] * tab[(size_t)i * n_out + t];
      o[x] = barrett_reduce128(acc, &br);
    }
  }
  free(qhi); free(tab);
}

/* ------------------------------------------------------------------ hybrid key switch
 * Stage order: KeySwitch::KeySwitch src/Operation.cpp:9-54.  beta = ceil(ell/alpha) (:22),
 * digit size d_j = min(alpha, ell - j*alpha) (:108-114), extended size E = ell + alpha (:159-161). */
void ho_keyswitch(const ho_ctx *c, uint32_t ell, const uint64_t *d, const uint64_t *evk, uint64_t *out0,
                  uint64_t *out1, const ho_ks_dump *dump) {
  const uint32_t N = c->N, K = c->K, L = c->L, E = ell + K, beta = (ell + K - 1) / K;
  const size_t LP = N;
  uint32_t *ext_ids = malloc(4 * E);
  for (uint32_t i = 0; i < ell; ++i) ext_ids[i] = i;
  for (uint32_t i = 0; i < K; ++i) ext_ids[ell + i] = L + i;

  /* ModUp_INTT  (KeySwitch::ModUpINTT :63-102) */
  uint64_t *coef = malloc(8 * LP * ell);
  memcpy(coef, d, 8 * LP * ell);
  ho_ntt_limbs(c, ext_ids, ell, coef, 1);
  if (dump && dump->modup_intt) memcpy(dump->modup_intt, coef, 8 * LP * ell);

  uint64_t *ext = malloc(8 * LP * E * beta);   /* NTTOut_beta(j) */
  uint64_t *scaled = malloc(8 * LP * K);
  for (uint32_t j = 0; j < beta; ++j) {
    uint32_t lo = j * K, dj = ell - lo < K ? ell - lo : K, n_out = E - dj;
    uint32_t *in_ids = malloc(4 * dj), *out_ids = malloc(4 * n_out);
    for (uint32_t i = 0; i < dj; ++i) in_ids[i] = lo + i;
    uint32_t o = 0;
    for (uint32_t t = 0; t < E; ++t) if (t < lo || t >= lo + dj) out_ids[o++] = ext_ids[t];
    /* ModUp_DecompOut<j>)  (:104-135) */
    ho_bconv_scale(c, in_ids, dj, out_ids, n_out, coef + LP * lo, scaled);
    if (dump && dump->modup_decomp) memcpy(dump->modup_decomp + LP * lo, scaled, 8 * LP * dj);
    /* ModUp_BCONV_(j)  (:137-188) */
    uint64_t *conv = malloc(8 * LP * n_out);
    ho_bconv_matmul(c, in_ids, dj, out_ids, n_out, scaled, conv);
    /* ModUp_NTT_(j)  (:190-292): converted limbs are NTT'd; the digit's own limbs are the
     * original eval-form input (SURVEY Appendix A, row ModUp_NTT) */
    ho_ntt_limbs(c, out_ids, n_out, conv, 0);
    uint64_t *ej = ext + LP * E * j;
    o = 0;
    for (uint32_t t = 0; t < E; ++t) {
      if (t >= lo && t < lo + dj) memcpy(ej + LP * t, d + LP * t, 8 * LP);
      else memcpy(ej + LP * t, conv + LP * (o++), 8 * LP);
    }
    free(conv); free(in_ids); free(out_ids);
  }
  if (dump && dump->ext) memcpy(dump->ext, ext, 8 * LP * E * beta);

  /* InnerProOut_(.)_Key<k>  (KeySwitch::InnerProduceOperation :294-414) */
  uint64_t *ip = malloc(8 * LP * E * 2);
#pragma omp parallel for num_threads(g_threads) collapse(2)
  for (uint32_t k = 0; k < 2; ++k)
    for (uint32_t t = 0; t < E; ++t) {
      barrett_t br = barrett_make(c->mod[ext_ids[t]]);
      uint64_t *o = ip + LP * (E * k + t);
      for (uint32_t x = 0; x < N; ++x) {
        u128 acc = 0;
        for (uint32_t j = 0; j < beta; ++j)
          acc += (u128)ext[LP * (E * j + t) + x] * evk[LP * ((size_t)(j * 2 + k) * E + t) + x];
        o[x] = barrett_reduce128(acc, &br);
      }
    }
  if (dump && dump->ip) memcpy(dump->ip, ip, 8 * LP * E * 2);

  /* ModDown (:417-590) */
  uint32_t *p_ids = ext_ids + ell, *q_ids = ext_ids;
  for (uint32_t k = 0; k < 2; ++k) {
    uint64_t *pk = malloc(8 * LP * K);
    memcpy(pk, ip + LP * (E * k + ell), 8 * LP * K);
    ho_ntt_limbs(c, p_ids, K, pk, 1);                                   /* ModDownINTTOut_Key(k) :417-445 */
    if (dump && dump->moddown_intt) memcpy(dump->moddown_intt + LP * K * k, pk, 8 * LP * K);
    ho_bconv_scale(c, p_ids, K, q_ids, ell, pk, scaled);                /* ModDownBConvStep1_Key(k) :447-487 */
    uint64_t *w = malloc(8 * LP * ell);
    ho_bconv_matmul(c, p_ids, K, q_ids, ell, scaled, w);                /* ModDown_BCONV_Key(k) :489-519 */
    if (dump && dump->moddown_bconv) memcpy(dump->moddown_bconv + LP * ell * k, w, 8 * LP * ell);
    ho_ntt_limbs(c, q_ids, ell, w, 0);                                  /* ModDownNTTOut_Key(k) :521-546 */
    if (dump && dump->moddown_ntt) memcpy(dump->moddown_ntt + LP * ell * k, w, 8 * LP * ell);
    uint64_t *out = k ? out1 : out0;
#pragma omp parallel for num_threads(g_threads)
    for (uint32_t i = 0; i < ell; ++i) {                                /* KeySwitchFinalOutput_Key(k) :548-590 */
      uint64_t qi = c->mod[i], pinv = 1;
      for (uint32_t p = 0; p < K; ++p) pinv = ho_mulmod(pinv, c->mod[L + p] % qi, qi);
      pinv = ho_invmod(pinv, qi);
      ho_ewe(c, HO_EWE_SUB_SCALE, i, ip + LP * (E * k + i), 0, w + LP * i, 0, pinv, out + LP * i);
    }
    free(pk); free(w);
  }
  free(ip); free(ext); free(scaled); free(coef); free(ext_ids);
}

/* ------------------------------------------------------------------ rescale
 * Rescale::NTTOps/SubOps/MulOps src/Operation.cpp:766-911 (the reference issues one NTT; the
 * maths needs ell-1). */
void ho_rescale(const ho_ctx *c, uint32_t ell, const uint64_t *x, uint64_t *out) {
  const uint32_t N = c->N, last = ell - 1; const uint64_t ql = c->mod[last];
  uint64_t *r = malloc(8ull * N);
  memcpy(r, x + (size_t)last * N, 8ull * N);
  ho_ntt(c, last, r, 1);
#pragma omp parallel for num_threads(g_threads)
  for (uint32_t i = 0; i < last; ++i) {
    uint64_t qi = c->mod[i], qlinv = ho_invmod(ql % qi, qi);
    uint64_t *t = malloc(8ull * N);
    for (uint32_t j = 0; j < N; ++j) t[j] = r[j] % qi;
    ho_ntt(c, i, t, 0);
    ho_ewe(c, HO_EWE_SUB_SCALE, i, x + (size_t)i * N, 0, t, 0, qlinv, out + (size_t)i * N);
    free(t);
  }
  free(r);
}

/* ------------------------------------------------------------------ ops
 * TensorCompute::computeD0/D1/D2 src/Operation.cpp:624-739; HMULT::HMULT :913-1023. */
void ho_hmult(const ho_ctx *c, uint32_t ell, const uint64_t *ct1, const uint64_t *ct2, const uint64_t *evk,
              int do_rescale, uint64_t *out) {
  const size_t LP = c->N; const size_t P = LP * ell;
  uint64_t *d0 = malloc(8 * P), *d1 = malloc(8 * P), *d2 = malloc(8 * P), *k0 = malloc(8 * P), *k1 = malloc(8 * P);
#pragma omp parallel for num_threads(g_threads)
  for (uint32_t i = 0; i < ell; ++i) {
    const uint64_t *a0 = ct1 + LP * i, *a1 = ct1 + P + LP * i, *b0 = ct2 + LP * i, *b1 = ct2 + P + LP * i;
    ho_ewe(c, HO_EWE_MUL, i, a0, b0, 0, 0, 0, d0 + LP * i);
    ho_ewe(c, HO_EWE_MAC2, i, a0, b1, a1, b0, 0, d1 + LP * i);
    ho_ewe(c, HO_EWE_MUL, i, a1, b1, 0, 0, 0, d2 + LP * i);
  }
  ho_keyswitch(c, ell, d2, evk, k0, k1, NULL);
  for (uint32_t i = 0; i < ell; ++i) { /* HMULT_Hadd_Key(k) :967-1005 */
    ho_ewe(c, HO_EWE_ADD, i, k0 + LP * i, 0, d0 + LP * i, 0, 0, d0 + LP * i);
    ho_ewe(c, HO_EWE_ADD, i, k1 + LP * i, 0, d1 + LP * i, 0, 0, d1 + LP * i);
  }
  if (do_rescale) { /* 2x Rescale :1008-1022 */
    ho_rescale(c, ell, d0, out);
    ho_rescale(c, ell, d1, out + LP * (ell - 1));
  } else {
    memcpy(out, d0, 8 * P); memcpy(out + P, d1, 8 * P);
  }
  free(d0); free(d1); free(d2); free(k0); free(k1);
}

/* HROTATE::HROTATE src/Operation.cpp:1271-1358 */
void ho_hrotate(const ho_ctx *c, uint32_t ell, const uint64_t *ct, uint32_t g, const uint64_t *evk, uint64_t *out) {
  const size_t LP = c->N; const size_t P = LP * ell;
  uint64_t *r0 = malloc(8 * P), *r1 = malloc(8 * P), *k0 = malloc(8 * P);
  for (uint32_t i = 0; i < ell; ++i) { /* AUTO_Key(k) :1302-1324 */
    ho_automorph_eval(c, ct + LP * i, r0 + LP * i, g);
    ho_automorph_eval(c, ct + P + LP * i, r1 + LP * i, g);
  }
  ho_keyswitch(c, ell, r1, evk, k0, out + P, NULL);
  for (uint32_t i = 0; i < ell; ++i) ho_ewe(c, HO_EWE_ADD, i, k0 + LP * i, 0, r0 + LP * i, 0, 0, out + LP * i);
  free(r0); free(r1); free(k0);
}

void ho_hadd(const ho_ctx *c, uint32_t ell, const uint64_t *a, const uint64_t *b, uint64_t *out) {
  for (uint32_t i = 0; i < 2 * ell; ++i)
    ho_ewe(c, HO_EWE_ADD, i % ell, a + (size_t)i * c->N, 0, b + (size_t)i * c->N, 0, 0, out + (size_t)i * c->N);
}
void ho_pmult(const ho_ctx *c, uint32_t ell, const uint64_t *ct, const uint64_t *pt, uint64_t *out) {
  for (uint32_t i = 0; i < 2 * ell; ++i)
    ho_ewe(c, HO_EWE_MUL, i % ell, ct + (size_t)i * c->N, pt + (size_t)(i % ell) * c->N, 0, 0, 0, out + (size_t)i * c->N);
}
void ho_padd(const ho_ctx *c, uint32_t ell, const uint64_t *ct, const uint64_t *pt, uint64_t *out) {
  for (uint32_t i = 0; i < ell; ++i)
    ho_ewe(c, HO_EWE_ADD, i, ct + (size_t)i * c->N, 0, pt + (size_t)i * c->N, 0, 0, out + (size_t)i * c->N);
  memcpy(out + (size_t)ell * c->N, ct + (size_t)ell * c->N, 8ull * ell * c->N);
}

/* ------------------------------------------------------------------ synthetic data */
static inline uint64_t mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z;
}
void ho_fill_uniform(const ho_ctx *c, const uint32_t *ids, uint32_t n, uint64_t seed, uint64_t *out) {
  for (uint32_t i = 0; i < n; ++i) {
    uint64_t q = c->mod[ids[i]], stream = seed + i;
    for (uint32_t x = 0; x < c->N; ++x) {
      uint64_t z = mix64(stream * 0xD1342543DE82EF95ull + (uint64_t)x * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull);
      out[(size_t)i * c->N + x] = (uint64_t)(((u128)z * q) >> 64);
    }
  }
}
