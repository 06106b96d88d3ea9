// hm_params.cpp — see hm_params.h.  Host code only (runs once per context).
#include "hm_params.h"
#include <stdexcept>
#include <string>

namespace hm {

typedef unsigned __int128 u128;

uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
uint64_t powmod(uint64_t a, uint64_t e, uint64_t q) {
  uint64_t r = 1 % q;
  a %= q;
  for (; e; e >>= 1) {
    if (e & 1) r = mulmod(r, a, q);
    a = mulmod(a, a, q);
  }
  return r;
}
uint64_t invmod(uint64_t a, uint64_t q) { return powmod(a % q, q - 2, q); }
uint64_t shoup(uint64_t w, uint64_t q) { return (uint64_t)((((u128)w) << 64) / q); }
uint32_t bitrev(uint32_t x, uint32_t bits) {
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; ++i, x >>= 1) r = (r << 1) | (x & 1);
  return r;
}

bool is_prime(uint64_t n) {  // deterministic Miller-Rabin for 64-bit n
  static const uint64_t bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  if (n < 2) return false;
  for (uint64_t b : bases) {
    if (n == b) return true;
    if (n % b == 0) return false;
  }
  uint64_t d = n - 1;
  int s = 0;
  while ((d & 1) == 0) { d >>= 1; ++s; }
  for (uint64_t b : bases) {
    uint64_t x = powmod(b, d, n);
    if (x == 1 || x == n - 1) continue;
    bool witness = true;
    for (int r = 1; r < s && witness; ++r) {
      x = mulmod(x, x, n);
      if (x == n - 1) witness = false;
    }
    if (witness) return false;
  }
  return true;
}

static uint64_t smallest_primitive_root(uint64_t q, uint32_t N) {
  const uint64_t e = (q - 1) / (2ull * N);
  uint64_t r = 0;
  for (uint64_t x = 2;; ++x) {
    r = powmod(x, e, q);
    if (powmod(r, N, q) == q - 1) break;
  }
  uint64_t step = mulmod(r, r, q), cur = r, best = r;
  for (uint32_t k = 1; k < N; ++k) {  // odd powers of r are exactly the primitive 2N-th roots
    cur = mulmod(cur, step, q);
    if (cur < best) best = cur;
  }
  return best;
}

std::vector<uint64_t> Params::chain_below(uint32_t logN, uint32_t bits, uint32_t count) {
  if (bits < 21 || bits > 60 || logN + 1 >= bits) throw std::invalid_argument("chain_below: bits must be in [21, 60] and above log2(2N)");
  std::vector<uint64_t> out;
  const uint64_t step = 2ull << logN;
  uint64_t cand = (1ull << bits) + 1;
  while (out.size() < count) {
    if (cand <= step) throw std::invalid_argument("chain_below: not enough primes = 1 mod 2N below 2^" + std::to_string(bits));
    cand -= step;
    if (is_prime(cand)) out.push_back(cand);
  }
  return out;
}

void Params::init(uint32_t logN_, uint32_t L_, uint32_t K_, const uint64_t *q, const uint64_t *p, const uint64_t *psi_in, bool forGeneric) {
  if (logN_ < 13 || logN_ > 17) throw std::invalid_argument("logN must be in [13,17] for the HIP backend");
  if (L_ == 0 || L_ + K_ > 4096) throw std::invalid_argument("bad limb counts");
  logN = logN_; N = 1u << logN; L = L_; K = K_;
  const uint32_t M = L + K;
  mod.assign(M, 0); psi.assign(M, 0); modc.assign(M, HmMod{});
  if (q) {
    for (uint32_t i = 0; i < L; ++i) mod[i] = q[i];
    for (uint32_t i = 0; i < K; ++i) mod[L + i] = p[i];
  } else {
    uint64_t cand = (1ull << 60) + 1;
    for (uint32_t m = 0; m < M;) {
      cand -= 1ull << 32;   // q = h 2^32 + 1: one multiply per Montgomery reduction step (hm_modarith.h), and 1 mod 2N for every N
      if (is_prime(cand)) mod[m++] = cand;
    }
  }
  mont32 = true;
  generic = forGeneric;
  for (uint32_t m = 0; m < M; ++m) {
    const uint64_t qm = mod[m];
    if (qm >> 60 || qm < (1ull << 20) || (qm - 1) % (2ull * N) != 0 || !is_prime(qm))
      throw std::invalid_argument("modulus " + std::to_string(qm) + " is not a prime = 1 mod 2N between 2^20 and 2^60");
    if ((qm & 0xffffffffull) != 1 || (qm >> 32) == 0) mont32 = false;
    for (uint32_t j = 0; j < m; ++j)
      if (mod[j] == qm) throw std::invalid_argument("duplicate modulus");
    psi[m] = psi_in ? psi_in[m] : smallest_primitive_root(qm, N);
    if (powmod(psi[m], N, qm) != qm - 1) throw std::invalid_argument("psi is not a primitive 2N-th root");
    HmMod &c = modc[m];
    uint32_t k = 64 - (uint32_t)__builtin_clzll(qm);
    c.q = qm;
    c.sh = k - 1;
    c.mu = (uint64_t)((((u128)1) << (k + 63)) / qm);
    c.r64 = (uint64_t)((((u128)1) << 64) % qm);
    c.r64s = shoup(c.r64, qm);
    {  // -q^-1 mod 2^64 by Newton iteration (q odd): x <- x (2 - q x) doubles the correct low bits, 3 -> 96
      uint64_t x = qm;
      for (int it = 0; it < 5; ++it) x *= 2 - qm * x;
      c.nqinv = 0 - x;
    }
    c.ninv = invmod(N, qm);
    if (forGeneric) c.ninvs = shoup(c.ninv, qm);
    else c.r128 = mulmod(c.r64, c.r64, qm);
  }
  if (!forGeneric && !mont32)
    throw std::invalid_argument("the chain holds a modulus that is not h 2^32 + 1: it needs the generic arithmetic back-end");
}

template <class ENTRY, class MAKE>
static void table_of(const Params &P, uint32_t m, bool inverse, ENTRY *out, MAKE make) {
  const uint64_t q = P.mod[m];
  const uint64_t base = inverse ? invmod(P.psi[m], q) : P.psi[m];
  uint64_t p = 1;
  for (uint32_t i = 0; i < P.N; ++i) {
    out[bitrev(i, P.logN)] = make(p, q);
    p = mulmod(p, base, q);
  }
}
template <class ENTRY, class MAKE>
static void twist_of(const Params &P, uint32_t m, bool inverse, ENTRY *out, MAKE make) {
  const uint64_t q = P.mod[m];
  const uint64_t base = inverse ? invmod(P.psi[m], q) : P.psi[m];
  const uint32_t rows = P.N >> 8, bits = P.logN - 8;
  for (uint32_t r = 0; r < rows; ++r) {
    const uint64_t a = powmod(base, 1 + 2ull * bitrev(r, bits), q);
    uint64_t p = 1;
    for (uint32_t k = 0; k < 3; ++k) {
      p = mulmod(p, a, q);
      out[3 * r + k] = make(p, q);
    }
  }
}
static uint64_t mont_entry(uint64_t w, uint64_t q) { return hm_to_mont(w, q); }   // Montgomery form: the butterflies' product is x wt 2^-64 (hm_mont_acc)
static HmTw shoup_entry(uint64_t w, uint64_t q) { return HmTw{w, shoup(w, q)}; }
void Params::make_table(uint32_t m, bool inverse, uint64_t *out) const { table_of(*this, m, inverse, out, mont_entry); }
void Params::make_table(uint32_t m, bool inverse, HmTw *out) const { table_of(*this, m, inverse, out, shoup_entry); }
void Params::make_twist(uint32_t m, bool inverse, uint64_t *out) const { twist_of(*this, m, inverse, out, mont_entry); }
void Params::make_twist(uint32_t m, bool inverse, HmTw *out) const { twist_of(*this, m, inverse, out, shoup_entry); }

void Params::bconv_consts(const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids, uint32_t n_out,
                          uint64_t *qhat_inv, uint64_t *table) const {
  for (uint32_t i = 0; i < n_in; ++i) {
    const uint64_t qi = mod[in_ids[i]];
    uint64_t prod = 1;
    for (uint32_t k = 0; k < n_in; ++k)
      if (k != i) prod = mulmod(prod, mod[in_ids[k]] % qi, qi);
    qhat_inv[i] = invmod(prod, qi);
    for (uint32_t t = 0; t < n_out; ++t) {
      const uint64_t qt = mod[out_ids[t]];
      uint64_t pr = 1;
      for (uint32_t k = 0; k < n_in; ++k)
        if (k != i) pr = mulmod(pr, mod[in_ids[k]] % qt, qt);
      table[(size_t)i * n_out + t] = pr;
    }
  }
}

}  // namespace hm
