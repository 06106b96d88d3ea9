// hm_params.h — host-side parameter generation for the HIP backend: prime chain, roots of unity,
// twiddle tables (mont32: Montgomery form, one word per entry; generic: value + Shoup companion), per-modulus Barrett constants,
// base-conversion tables.  One build serves both arithmetic back-ends (hm_modarith.h) and the host layer; which tables a context
// needs is a run-time flag.
// The reference defines none of these; the rules are SURVEY.md §8d / Appendix A:
// default chain = the L+K largest primes below 2^60 congruent to 1 mod 2^32 (descending; first L = Q, next K = P; round 4) or, by name
// ("survey", SURVEY.md 8d as written), the L+K largest primes below 2^60 congruent to 1 mod 2N;
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
  bool mont32 = true;          // every modulus is h 2^32 + 1: the chain fits the word-wise Montgomery back-end
  bool generic = false;        // the per-modulus records were filled for the generic back-end (HmMod::ninvs instead of r128)

  // default chain (q == nullptr) or caller-supplied moduli / roots (psi may be nullptr).  Any distinct primes = 1 mod 2N with
  // 2^20 < q < 2^60 are accepted; `mont32` says whether they all are h 2^32 + 1.  forGeneric: fill the records for the generic
  // back-end (a mont32 chain may run on either).
  void init(uint32_t logN, uint32_t L, uint32_t K, const uint64_t *q, const uint64_t *p, const uint64_t *psi, bool forGeneric = false);
  // the `count` largest primes = 1 mod 2N below 2^bits (21 <= bits <= 60), descending: bits = 60 is the chain of SURVEY.md 8d as written,
  // bits = 36 a chain of 36-bit words as the reference's configuration models (config/config_4.cfg:9); throws if there are not that many
  static std::vector<uint64_t> chain_below(uint32_t logN, uint32_t bits, uint32_t count);
  // forward (inverse=false) or inverse twiddle table of one modulus, N entries, bit-reversed order: w 2^64 mod q (mont32) or
  // (w, Shoup companion) (generic)
  void make_table(uint32_t mod_id, bool inverse, uint64_t *out) const;
  void make_table(uint32_t mod_id, bool inverse, HmTw *out) const;
  // per-row constants of the ROW pass (hm_ntt_core.h): out[3 r + k - 1] = alpha_r^k (inverse: alpha_r^-k), k = 1..3,
  // alpha_r = psi^(1 + 2 brev(r)), r < N / 256
  void make_twist(uint32_t mod_id, bool inverse, uint64_t *out) const;
  void make_twist(uint32_t mod_id, bool inverse, HmTw *out) const;
  // base conversion constants for an input basis -> output basis
  void bconv_consts(const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids, uint32_t n_out,
                    uint64_t *qhat_inv, uint64_t *table) const;
};

}  // namespace hm
