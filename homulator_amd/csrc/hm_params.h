// hm_params.h — host-side parameter generation for the HIP backend: prime chain, roots of unity,
// twiddle tables (Montgomery form, one word per entry), per-modulus Barrett constants, base-conversion tables.
// The reference defines none of these; the rules are SURVEY.md §8d / Appendix A:
// the L+K largest primes below 2^60 congruent to 1 mod 2^32 (descending; first L = Q, next K = P),
// psi = smallest primitive 2N-th root of unity, tables in bit-reversed order.
#pragma once
#include <cstdint>
#include <vector>
#include "hm_modarith.h"

namespace hm {

uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q);
uint64_t powmod(uint64_t a, uint64_t e, uint64_t q);
uint64_t invmod(uint64_t a, uint64_t q);
bool is_prime(uint64_t n);
uint64_t shoup(uint64_t w, uint64_t q);
uint32_t bitrev(uint32_t x, uint32_t bits);

struct Params {
  uint32_t logN = 0, N = 0, L = 0, K = 0;
  std::vector<uint64_t> mod;   // [L+K]
  std::vector<uint64_t> psi;   // [L+K]
  std::vector<HmMod> modc;     // [L+K]

  // default chain (q == nullptr) or caller-supplied moduli / roots (psi may be nullptr)
  void init(uint32_t logN, uint32_t L, uint32_t K, const uint64_t *q, const uint64_t *p, const uint64_t *psi);
  // forward (inverse=false) or inverse twiddle table of one modulus, N entries w 2^64 mod q, bit-reversed order
  void make_table(uint32_t mod_id, bool inverse, HmW *out) const;
  // per-row constants of the ROW pass (hm_ntt_core.h): out[3 r + k - 1] = alpha_r^k (inverse: alpha_r^-k), k = 1..3,
  // alpha_r = psi^(1 + 2 brev(r)), r < N / 256
  void make_twist(uint32_t mod_id, bool inverse, HmW *out) const;
  // base conversion constants for an input basis -> output basis
  void bconv_consts(const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids, uint32_t n_out,
                    uint64_t *qhat_inv, uint64_t *table) const;
};

}  // namespace hm
