// hm_emu.cpp — host emulator of the HIP kernels' per-thread phase functions (TEST INFRASTRUCTURE).
// Compiles homulator_amd/csrc/hm_*_core.h with g++ and runs each workgroup phase by phase, thread by
// thread, so that index maths, twiddle indexing, lazy-reduction ranges and the host-side parameter
// tables (hm_params.cpp) are checked against the oracle on the CPU before a GPU is involved.
// It is not a CPU backend: the product library never links or calls this.
// Built twice, like the kernels: libhm_emu.so (HM_GENERIC = 0: word-wise Montgomery on primes h 2^32 + 1) and libhm_emu_gen.so
// (-DHM_GENERIC=1: Shoup / Barrett arithmetic on any NTT-friendly chain).
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>
#include "../../homulator_amd/csrc/hm_elem_core.h"
#include "../../homulator_amd/csrc/hm_modarith.h"
#include "../../homulator_amd/csrc/hm_ntt_core.h"
#include "../../homulator_amd/csrc/hm_params.h"

struct Emu {
  hm::Params P;
  std::vector<std::vector<HmW>> fwd, inv, twf, twi;
};

// the two geometries of the passes (hm_ntt_core.h): 16 coefficients per thread (hm16) and 8 (hm8, N = 2^16 only)
struct G16 {
  typedef hm16::HmNttState State;
  static constexpr int EPT = 16;
  template <int LOGR> static constexpr int rounds() { return hm16::HmRounds<LOGR>::n; }
  template <int TL, int LOGR, bool STRIDED> static constexpr int ldsWords() { return hm16::HmLds<TL, LOGR, STRIDED>::WORDS; }
  template <int TL, int LOGR, bool STRIDED, bool INV, int MODE, int P, class... A> static void phase(A &&...a) { hm16::hm_ntt_phase<TL, LOGR, STRIDED, INV, MODE, P>(a...); }
};
struct G8 {
  typedef hm8::HmNttState State;
  static constexpr int EPT = 8;
  template <int LOGR> static constexpr int rounds() { return hm8::HmRounds<LOGR>::n; }
  template <int TL, int LOGR, bool STRIDED> static constexpr int ldsWords() { return hm8::HmLds<TL, LOGR, STRIDED>::WORDS; }
  template <int TL, int LOGR, bool STRIDED, bool INV, int MODE, int P, class... A> static void phase(A &&...a) { hm8::hm_ntt_phase<TL, LOGR, STRIDED, INV, MODE, P>(a...); }
};

// phase P of every thread of the workgroup, then phase P + 1, ... (a barrier on the GPU = the end of a phase loop here)
template <class G, int TL, int LOGR, bool STRIDED, bool INV, int MODE, int P>
static void run_phases(std::vector<typename G::State> &st, uint64_t *lds, const uint64_t *src, uint64_t *dst, uint32_t tile, const HmW *twl,
                       const HmW *twt, uint32_t s0, uint32_t prefix0, uint64_t q, HmTw sc, HmEpi ep) {
  for (int t = 0; t < (int)st.size(); ++t) G::template phase<TL, LOGR, STRIDED, INV, MODE, P>(st[t], t, lds, src, dst, tile, twl, twt, s0, prefix0, q, sc, ep);
  if constexpr (P < G::template rounds<LOGR>()) run_phases<G, TL, LOGR, STRIDED, INV, MODE, P + 1>(st, lds, src, dst, tile, twl, twt, s0, prefix0, q, sc, ep);
}

template <class G, int LOGR, bool STRIDED, bool INV, int MODE>
static void run_pass(const Emu &e, uint32_t mod, const uint64_t *src, uint64_t *dst, HmTw sc, HmEpi ep = hm_epi_none()) {
  constexpr int TL = HM_TL(STRIDED), THREADS = (1 << TL) / G::EPT;
  const uint32_t tiles = e.P.N >> TL;
  const uint64_t q = e.P.mod[mod];
  const HmW *twl = (INV ? e.inv : e.fwd)[mod].data();
  const uint32_t s0 = STRIDED ? 0u : (e.P.logN - HM_ROW_LOG);
  std::vector<uint64_t> lds(G::template ldsWords<TL, LOGR, STRIDED>());
  std::vector<typename G::State> st(THREADS);
  const HmW *twist = (INV ? e.twi : e.twf)[mod].data();
  // a pass may run in place (src == dst): every thread reads its elements before any thread writes its own
  for (uint32_t tile = 0; tile < tiles; ++tile) {
    const uint32_t prefix0 = STRIDED ? 0u : (tile << (TL - LOGR));
    const HmW *twt = twist + (size_t)prefix0 * 3;
    run_phases<G, TL, LOGR, STRIDED, INV, MODE, 0>(st, lds.data(), src, dst, tile, twl, twt, s0, prefix0, q, sc, ep);
  }
}

template <class G, int LOG1>
static void run_ntt(const Emu &e, uint32_t mod, const uint64_t *in, uint64_t *out, int inverse, HmTw sc) {
  if (!inverse) {
    run_pass<G, LOG1, true, false, 0>(e, mod, in, out, sc);
    run_pass<G, HM_ROW_LOG, false, false, 1>(e, mod, out, out, sc);
  } else {
    run_pass<G, HM_ROW_LOG, false, true, 0>(e, mod, in, out, sc);
    run_pass<G, LOG1, true, true, 2>(e, mod, out, out, sc);
  }
}

template <int N_IN>
static void emu_bconv_n(Emu &e, HmBconvProb &p, const std::vector<uint64_t> &tb, const uint32_t *out_ids) {
  const uint32_t row = HM_BCONV_ROW(p.n_in);
  std::vector<uint64_t> tt((size_t)row * p.n_out, 0);  // device format: [n_out][row], Montgomery form, packed, zero-padded rows
  std::vector<uint64_t> qn;                            // {q, -q^-1} per output
  for (uint32_t i = 0; i < p.n_in; ++i)
    for (uint32_t t = 0; t < p.n_out; ++t) tt[(size_t)t * row + i] = hm_bconv_entry(tb[(size_t)i * p.n_out + t], e.P.modc[out_ids[t]]);
  for (uint32_t t = 0; t < p.n_out; ++t) { qn.push_back(e.P.modc[out_ids[t]].q); qn.push_back(e.P.modc[out_ids[t]].nqinv); }
  p.table = tt.data();
  p.qn = qn.data();
  for (uint32_t t0 = 0; t0 < p.n_out; t0 += HM_BCONV_CHUNK) {
    uint32_t t1 = t0 + HM_BCONV_CHUNK < p.n_out ? t0 + HM_BCONV_CHUNK : p.n_out;
    for (uint32_t x = 0; x < e.P.N; x += HM_BCONV_CPT) {
      if (p.in_packed) hm_bconv_thread<N_IN, HM_BCONV_CPT, true>(p, e.P.logN, x, t0, t1);
      else hm_bconv_thread<N_IN, HM_BCONV_CPT, false>(p, e.P.logN, x, t0, t1);
    }
  }
}

// round 6: the arithmetic of the two-group conversion of k_bconv_col (digits of 16 .. 32 limbs, hm_bcol_units_wide): per output and coefficient the
// 128-bit sums of the group [0, 16) and of the group [16, n_in) (hm_bconv_cols: carry-free split-30 columns, recombined per group), added, ONE
// Montgomery reduction for all n_in terms (hm_redc_wide<N_IN>).  Same table format as the kernel (rows of 8-entry groups, Montgomery form,
// split-30 packed); packed != 0: the inputs arrive in the split-30 packed form.
template <int N_IN>
static void emu_bconv_wide_n(Emu &e, const uint32_t *out_ids, uint32_t n_out, const std::vector<uint64_t> &tb, const uint64_t *in, uint64_t *out, int packed) {
  constexpr int ROWS = (N_IN + 7) / 8, C1 = N_IN - 16;
  const uint32_t N = e.P.N;
  std::vector<HmRow8> rows((size_t)n_out * ROWS);
  for (auto &r : rows) for (auto &w : r.w) w = 0;
  for (uint32_t t = 0; t < n_out; ++t)
    for (int i = 0; i < N_IN; ++i) rows[(size_t)t * ROWS + i / 8].w[i % 8] = hm_bconv_entry(tb[(size_t)i * n_out + t], e.P.modc[out_ids[t]]);
  for (uint32_t t = 0; t < n_out; ++t) {
    const HmMod &m = e.P.modc[out_ids[t]];
    HmRow8 r0[2], r1[(C1 > 0 ? C1 + 7 : 8) / 8];
    r0[0] = rows[(size_t)t * ROWS]; r0[1] = rows[(size_t)t * ROWS + 1];
    for (int g = 0; g < (C1 + 7) / 8; ++g) r1[g] = rows[(size_t)t * ROWS + 2 + g];
    for (uint32_t x = 0; x < N; ++x) {
      uint32_t yl[16], yh[16], zl[C1 > 0 ? C1 : 1] = {0}, zh[C1 > 0 ? C1 : 1] = {0};
      for (int i = 0; i < N_IN; ++i) {
        const uint64_t v = in[(size_t)i * N + x];
        const uint32_t lo = packed ? (uint32_t)v : (uint32_t)v & 0x3FFFFFFFu, hi = packed ? (uint32_t)(v >> 32) : (uint32_t)(v >> 30);
        if (i < 16) { yl[i] = lo; yh[i] = hi; } else { zl[i - 16] = lo; zh[i - 16] = hi; }
      }
      hm_u128 acc = hm_bconv_cols<16>(yl, yh, r0);
      if constexpr (C1 > 0) acc += hm_bconv_cols<C1>(zl, zh, r1);
      out[(size_t)t * N + x] = hm_redc_wide<N_IN>(acc, m);
    }
  }
}

// the 256-point ROW pass's LDS image as the kernels index it, for tools/lds_banks.py's bank model (tests/test_emu_kernels.py pins the
// model to these): the LDS word of access unit a of thread tid in round R (ept = 16 / 8: geometry; inverse: the pass's direction)
template <int R> static int row_unit16(int tid, int a, bool inv) {
  int i0, i1, xb, d, c;
  hm16::HmRound<12, 8, false, R>::unit_split(tid, a, i0, i1, xb, d, c);
  return inv ? hm16::hm_lds_at<12, 8, false, true>(xb, d, c) : hm16::hm_lds_at<12, 8, false, false>(xb, d, c);
}
template <int R> static int row_unit8(int tid, int a, bool inv) {
  int i0, i1, xb, d, c;
  hm8::HmRound<12, 8, false, R>::unit_split(tid, a, i0, i1, xb, d, c);
  return inv ? hm8::hm_lds_at<12, 8, false, true>(xb, d, c) : hm8::hm_lds_at<12, 8, false, false>(xb, d, c);
}
// round 6: the transforms that read an operand through an automorphism (MODE 6: the inverse transform's input; MODE 7: the fused epilogue's addend),
// both geometries (ept = 16 / 8)
template <class G, int LOG1>
static void run_intt_auto(const Emu &e, uint32_t mod, const uint64_t *in, uint64_t *out, HmTw sc, uint32_t g) {
  HmEpi ep = hm_epi_none();
  ep.g = g; ep.logN = e.P.logN;
  run_pass<G, HM_ROW_LOG, false, true, 6>(e, mod, in, out, sc, ep);
  run_pass<G, LOG1, true, true, 2>(e, mod, out, out, sc);
}
extern "C" {
// q == nullptr: the default chain; otherwise a caller-chosen one (q: L moduli, p: K special moduli; primes = 1 mod 2N below 2^60).
// Returns nullptr if the chain does not fit this build's arithmetic.
static void *emu_make(uint32_t logN, uint32_t L, uint32_t K, const uint64_t *q, const uint64_t *p, bool tables) {
  Emu *e = new Emu;
  try {
    e->P.init(logN, L, K, q, p, nullptr, HM_GENERIC != 0);
  } catch (const std::exception &) {
    delete e;
    return nullptr;
  }
  if (!tables) return e;
  e->fwd.resize(L + K); e->inv.resize(L + K); e->twf.resize(L + K); e->twi.resize(L + K);
  for (uint32_t m = 0; m < L + K; ++m) {
    e->fwd[m].resize(e->P.N); e->inv[m].resize(e->P.N);
    e->twf[m].resize((e->P.N >> 8) * 3); e->twi[m].resize((e->P.N >> 8) * 3);
    e->P.make_twist(m, false, e->twf[m].data());
    e->P.make_twist(m, true, e->twi[m].data());
    e->P.make_table(m, false, e->fwd[m].data());
    e->P.make_table(m, true, e->inv[m].data());
  }
  return e;
}
void *emu_create(uint32_t logN, uint32_t L, uint32_t K) { return emu_make(logN, L, K, nullptr, nullptr, true); }
void *emu_create_chain(uint32_t logN, uint32_t L, uint32_t K, const uint64_t *q, const uint64_t *p) { return emu_make(logN, L, K, q, p, true); }
// only the modulus constants (the element-wise checks that need no twiddle tables)
void *emu_create_mods(uint32_t logN, uint32_t L, uint32_t K, const uint64_t *q, const uint64_t *p) { return emu_make(logN, L, K, q, p, false); }
int emu_generic(void) { return HM_GENERIC; }
void emu_destroy(void *h) { delete (Emu *)h; }
uint64_t emu_modulus(void *h, uint32_t m) { return ((Emu *)h)->P.mod[m]; }
uint64_t emu_psi(void *h, uint32_t m) { return ((Emu *)h)->P.psi[m]; }

int emu_ntt(void *h, uint32_t mod, const uint64_t *in, uint64_t *out, int inverse, uint64_t scale, int has_scale) {
  Emu &e = *(Emu *)h;
  uint64_t q = e.P.mod[mod], k = e.P.modc[mod].ninv;
  if (has_scale) k = hm::mulmod(k, scale, q);
  HmTw sc = hm_kconst(k, q);
  switch (e.P.logN - HM_ROW_LOG) {
  case 5: run_ntt<G16, 5>(e, mod, in, out, inverse, sc); break;
  case 6: run_ntt<G16, 6>(e, mod, in, out, inverse, sc); break;
  case 7: run_ntt<G16, 7>(e, mod, in, out, inverse, sc); break;
  case 8: run_ntt<G16, 8>(e, mod, in, out, inverse, sc); break;
  case 9: run_ntt<G16, 9>(e, mod, in, out, inverse, sc); break;
  default: return 1;
  }
  return 0;
}

// fused forward transform: out = (minuend - NTT(in [+ mix_k * mix])) * k [+ addend [* addend_k]]   (mix_k / addend_k = 0: none)
int emu_ntt_sub_scale(void *h, uint32_t mod, const uint64_t *in, const uint64_t *minuend, const uint64_t *addend, uint64_t *out,
                      uint64_t k, const uint64_t *mix, uint64_t mix_k, uint64_t addend_k) {
  Emu &e = *(Emu *)h;
  const uint64_t q = e.P.mod[mod];
  HmTw sc = hm_kconst(k, q);
  HmEpi ep = hm_epi_none();
  ep.a = minuend; ep.d = addend;
  if (addend_k) ep.dk = hm_kconst(addend_k, q);
  if (mix) { ep.b = mix; ep.bk = hm_kconst(mix_k, q); }
  switch (e.P.logN - HM_ROW_LOG) {
#define HM_CASE(n) case n: if (mix) run_pass<G16, n, true, false, 4>(e, mod, in, out, sc, ep); else run_pass<G16, n, true, false, 0>(e, mod, in, out, sc); break;
    HM_CASE(5) HM_CASE(6) HM_CASE(7) HM_CASE(8) HM_CASE(9)
#undef HM_CASE
  default: return 1;
  }
  run_pass<G16, HM_ROW_LOG, false, false, 3>(e, mod, out, out, sc, ep);
  return 0;
}
// the 8-coefficient geometry (hm8: 512-thread workgroups, radix-4 rounds), N = 2^16 and (round 6) N = 2^15
int emu_ntt8(void *h, uint32_t mod, const uint64_t *in, uint64_t *out, int inverse, uint64_t scale, int has_scale) {
  Emu &e = *(Emu *)h;
  if (e.P.logN != 16 && e.P.logN != 15) return 1;
  uint64_t q = e.P.mod[mod], k = e.P.modc[mod].ninv;
  if (has_scale) k = hm::mulmod(k, scale, q);
  if (e.P.logN == 16) run_ntt<G8, 8>(e, mod, in, out, inverse, hm_kconst(k, q));
  else run_ntt<G8, 7>(e, mod, in, out, inverse, hm_kconst(k, q));
  return 0;
}
int emu_ntt_sub_scale8(void *h, uint32_t mod, const uint64_t *in, const uint64_t *minuend, const uint64_t *addend, uint64_t *out,
                       uint64_t k, const uint64_t *mix, uint64_t mix_k, uint64_t addend_k) {
  Emu &e = *(Emu *)h;
  if (e.P.logN != 16 && e.P.logN != 15) return 1;
  const uint64_t q = e.P.mod[mod];
  HmTw sc = hm_kconst(k, q);
  HmEpi ep = hm_epi_none();
  ep.a = minuend; ep.d = addend;
  if (addend_k) ep.dk = hm_kconst(addend_k, q);
  if (mix) { ep.b = mix; ep.bk = hm_kconst(mix_k, q); }
  if (e.P.logN == 16) { if (mix) run_pass<G8, 8, true, false, 4>(e, mod, in, out, sc, ep); else run_pass<G8, 8, true, false, 0>(e, mod, in, out, sc); }
  else { if (mix) run_pass<G8, 7, true, false, 4>(e, mod, in, out, sc, ep); else run_pass<G8, 7, true, false, 0>(e, mod, in, out, sc); }
  run_pass<G8, HM_ROW_LOG, false, false, 3>(e, mod, out, out, sc, ep);
  return 0;
}
int emu_intt_auto(void *h, uint32_t mod, const uint64_t *in, uint64_t *out, uint32_t galois, int ept) {
  Emu &e = *(Emu *)h;
  if (in == out || (ept == 8 && e.P.logN != 16 && e.P.logN != 15)) return 1;
  const uint64_t q = e.P.mod[mod];
  const HmTw sc = hm_kconst(e.P.modc[mod].ninv, q);
  if (ept == 8) { if (e.P.logN == 16) run_intt_auto<G8, 8>(e, mod, in, out, sc, galois); else run_intt_auto<G8, 7>(e, mod, in, out, sc, galois); return 0; }
  switch (e.P.logN - HM_ROW_LOG) {
  case 5: run_intt_auto<G16, 5>(e, mod, in, out, sc, galois); break;
  case 6: run_intt_auto<G16, 6>(e, mod, in, out, sc, galois); break;
  case 7: run_intt_auto<G16, 7>(e, mod, in, out, sc, galois); break;
  case 8: run_intt_auto<G16, 8>(e, mod, in, out, sc, galois); break;
  case 9: run_intt_auto<G16, 9>(e, mod, in, out, sc, galois); break;
  default: return 1;
  }
  return 0;
}
int emu_ntt_sub_scale_auto(void *h, uint32_t mod, const uint64_t *in, const uint64_t *minuend, const uint64_t *addend, uint64_t *out, uint64_t k,
                           uint64_t addend_k, uint32_t galois, int ept) {
  Emu &e = *(Emu *)h;
  if (ept == 8 && e.P.logN != 16 && e.P.logN != 15) return 1;
  const uint64_t q = e.P.mod[mod];
  const HmTw sc = hm_kconst(k, q);
  HmEpi ep = hm_epi_none();
  ep.a = minuend; ep.d = addend; ep.g = galois; ep.logN = e.P.logN;
  if (addend_k) ep.dk = hm_kconst(addend_k, q);
  if (ept == 8) {
    if (e.P.logN == 16) run_pass<G8, 8, true, false, 0>(e, mod, in, out, sc); else run_pass<G8, 7, true, false, 0>(e, mod, in, out, sc);
    run_pass<G8, HM_ROW_LOG, false, false, 7>(e, mod, out, out, sc, ep);
    return 0;
  }
  switch (e.P.logN - HM_ROW_LOG) {
#define HM_CASE(n) case n: run_pass<G16, n, true, false, 0>(e, mod, in, out, sc); break;
    HM_CASE(5) HM_CASE(6) HM_CASE(7) HM_CASE(8) HM_CASE(9)
#undef HM_CASE
  default: return 1;
  }
  run_pass<G16, HM_ROW_LOG, false, false, 7>(e, mod, out, out, sc, ep);
  return 0;
}
void emu_tensor(void *h, uint32_t mod, const uint64_t *a, const uint64_t *b, const uint64_t *c, const uint64_t *d, uint64_t *o0,
                uint64_t *o1, uint64_t *o2) {
  Emu &e = *(Emu *)h;
  for (uint32_t x = 0; x < e.P.N; ++x) hm_tensor_one(a[x], b[x], c[x], d[x], e.P.modc[mod], o0[x], o1[x], o2[x]);
}

void emu_ewe(void *h, int op, uint32_t mod, const uint64_t *a, const uint64_t *b, const uint64_t *c,
             const uint64_t *d, uint64_t kk, uint64_t *out) {
  Emu &e = *(Emu *)h;
  const HmMod m = e.P.modc[mod];
  HmTw k = {kk, hm::shoup(kk, m.q)};
  for (uint32_t x = 0; x < e.P.N; ++x) {
    uint64_t va = a ? a[x] : 0, vb = b ? b[x] : 0, vc = c ? c[x] : 0, vd = d ? d[x] : 0, r = 0;
    switch (op) {
    case 0: r = hm_ewe_one<0>(va, vb, vc, vd, k, m); break;
    case 1: r = hm_ewe_one<1>(va, vb, vc, vd, k, m); break;
    case 2: r = hm_ewe_one<2>(va, vb, vc, vd, k, m); break;
    case 3: r = hm_ewe_one<3>(va, vb, vc, vd, k, m); break;
    case 4: r = hm_ewe_one<4>(va, vb, vc, vd, k, m); break;
    case 5: r = hm_ewe_one<5>(va, vb, vc, vd, k, m); break;
    case 6: r = hm_ewe_one<6>(va, vb, vc, vd, k, m); break;
    case 7: r = hm_ewe_one<7>(va, vb, vc, vd, k, m); break;
    case 8: r = hm_ewe_one<8>(va, vb, vc, vd, k, m); break;
    }
    out[x] = r;
  }
}

// packed != 0: the inputs are first brought into the split-30 packed form (hm_pack30) and the conversion is told so (in_packed)
void emu_bconv_form(void *h, const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids, uint32_t n_out,
                    const uint64_t *in, uint64_t *out, int packed);
void emu_bconv(void *h, const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids, uint32_t n_out,
               const uint64_t *in, uint64_t *out) { emu_bconv_form(h, in_ids, n_in, out_ids, n_out, in, out, 0); }
void emu_bconv_form(void *h, const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids, uint32_t n_out,
                    const uint64_t *in, uint64_t *out, int packed) {
  Emu &e = *(Emu *)h;
  std::vector<uint64_t> qh(n_in), tb((size_t)n_in * n_out);
  e.P.bconv_consts(in_ids, n_in, out_ids, n_out, qh.data(), tb.data());
  std::vector<uint64_t> pk;
  if (packed) {
    pk.assign(in, in + (size_t)n_in * e.P.N);
    for (uint64_t &v : pk) v = hm_pack30(v);
    in = pk.data();
  }
  HmBconvProb p{};
  p.in_packed = packed ? 1u : 0u;
  p.in = in; p.out = out; p.table = tb.data(); p.n_in = n_in; p.n_out = n_out;
  for (uint32_t i = 0; i < n_in; ++i) p.in_limb[i] = i;
  for (uint32_t t = 0; t < n_out; ++t) p.out_limb[t] = t;
  switch (n_in) {
#define HM_CASE(n) case n: emu_bconv_n<n>(e, p, tb, out_ids); break;
    HM_CASE(1) HM_CASE(2) HM_CASE(3) HM_CASE(4) HM_CASE(5) HM_CASE(6) HM_CASE(7) HM_CASE(8)
    HM_CASE(9) HM_CASE(10) HM_CASE(11) HM_CASE(12) HM_CASE(13) HM_CASE(14) HM_CASE(15) HM_CASE(16)
    HM_CASE(17) HM_CASE(18) HM_CASE(19) HM_CASE(20) HM_CASE(21) HM_CASE(22) HM_CASE(23) HM_CASE(24)
    HM_CASE(25) HM_CASE(26) HM_CASE(27) HM_CASE(28) HM_CASE(29) HM_CASE(30) HM_CASE(31) HM_CASE(32)
#undef HM_CASE
  }
}
int emu_bconv_two_groups(void *h, const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids, uint32_t n_out, const uint64_t *in, uint64_t *out, int packed) {
  Emu &e = *(Emu *)h;
  std::vector<uint64_t> qh(n_in), tb((size_t)n_in * n_out);
  e.P.bconv_consts(in_ids, n_in, out_ids, n_out, qh.data(), tb.data());
  switch (n_in) {
#define HM_CASE(n) case n: emu_bconv_wide_n<n>(e, out_ids, n_out, tb, in, out, packed); break;
    HM_CASE(16) HM_CASE(17) HM_CASE(18) HM_CASE(19) HM_CASE(20) HM_CASE(21) HM_CASE(22) HM_CASE(23) HM_CASE(24)
    HM_CASE(25) HM_CASE(26) HM_CASE(27) HM_CASE(28) HM_CASE(29) HM_CASE(30) HM_CASE(31) HM_CASE(32)
#undef HM_CASE
  default: return 1;
  }
  return 0;
}
void emu_bconv_consts(void *h, const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids, uint32_t n_out,
                      uint64_t *qh, uint64_t *tb) {
  ((Emu *)h)->P.bconv_consts(in_ids, n_in, out_ids, n_out, qh, tb);
}
void emu_automorph(void *h, const uint64_t *in, uint64_t *out, uint32_t g) {
  Emu &e = *(Emu *)h;
  for (uint32_t i = 0; i < e.P.N; ++i) out[i] = in[hm_auto_src(i, g, e.P.logN)];
}
void emu_fill(void *h, uint32_t mod, uint64_t stream, uint64_t *out) {
  Emu &e = *(Emu *)h;
  for (uint32_t x = 0; x < e.P.N; ++x) out[x] = hm_synth(stream, x, e.P.mod[mod]);
}

// the fused kernel's key multiply-accumulate (hm_mac_add, lazy form) element by element: acc[i] (+)= x[i] * y[i], `fold` as for the third
// and later terms; returns the raw lazy accumulators (the test checks range and congruence), and the final reduction separately
int emu_mac(void *h, uint32_t mod, uint64_t *acc, const uint64_t *x, const uint64_t *y, uint32_t n, int fold) {
  Emu &e = *(Emu *)h;
  const HmMod &m = e.P.modc[mod];
  const hm16::HmMacMod mm = hm16::hm_mac_mod(m);
  for (uint32_t i = 0; i < n; ++i) hm16::hm_mac_add(acc[i], x[i], y[i], m, mm, fold != 0);
  return 0;
}
uint64_t emu_mac_final(void *h, uint32_t mod, uint64_t acc) { return hm16::hm_mac_final(acc, ((Emu *)h)->P.modc[mod]); }
// the word-wise Montgomery product itself (hm_mont_acc): out[i] = c[i] + x[i] * wt[i] * 2^-64 mod q + {0, q}; and the per-launch constant
// product (hm_mont_const_mul), the wide reduction (hm_redc_wide) and one forward / inverse butterfly of each kind
#if !HM_GENERIC
void emu_mont_acc(void *h, uint32_t mod, const uint64_t *c, const uint64_t *x, const uint64_t *wt, uint64_t *out, uint32_t n) {
  const HmBflyMod m = hm_bfly_mod(((Emu *)h)->P.mod[mod]);
  for (uint32_t i = 0; i < n; ++i) out[i] = hm_mont_acc(c[i], x[i], wt[i], m);
}
uint64_t emu_mont_const_mul(void *h, uint32_t mod, uint64_t x, uint64_t k) {
  const uint64_t q = ((Emu *)h)->P.mod[mod];
  return hm_mont_const_mul(x, hm_to_mont(k, q), q);
}
#endif
uint64_t emu_redc_wide(void *h, uint32_t mod, uint64_t lo, uint64_t hi, int terms) {
  const HmMod &m = ((Emu *)h)->P.modc[mod];
  const hm_u128 z = ((hm_u128)hi << 64) | lo;
  return terms <= 16 ? hm_redc_wide<16>(z, m) : hm_redc_wide<32>(z, m);
}
void emu_bfly(void *h, uint32_t mod, int kind, uint64_t *X, uint64_t *Y, uint64_t w) {   // kind 0..2: forward; 3: inverse
  const uint64_t q = ((Emu *)h)->P.mod[mod];
  const HmBflyMod m = hm_bfly_mod(q);
  const HmW wt = hm_tw_entry(w, q);
  if (kind == 0) hm_bfly_fwd_k<0>(*X, *Y, wt, m);
  else if (kind == 1) hm_bfly_fwd_k<1>(*X, *Y, wt, m);
  else if (kind == 2) hm_bfly_fwd_k<2>(*X, *Y, wt, m);
  else hm_bfly_inv(*X, *Y, wt, m);
}
int emu_row_lds_word(int ept, int R, int tid, int a, int inverse) {
  if (ept == 16) return R == 0 ? row_unit16<0>(tid, a, inverse) : R == 1 ? row_unit16<1>(tid, a, inverse) : row_unit16<2>(tid, a, inverse);
  return R == 0 ? row_unit8<0>(tid, a, inverse) : R == 1 ? row_unit8<1>(tid, a, inverse) : R == 2 ? row_unit8<2>(tid, a, inverse) : row_unit8<3>(tid, a, inverse);
}
}
