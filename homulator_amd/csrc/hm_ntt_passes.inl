// hm_ntt_passes.inl — the per-thread phases of a transform pass for ONE geometry (HM_EPT coefficients per thread); included by
// hm_ntt_core.h once per geometry, inside that geometry's namespace.  No include guard on purpose.
// LDS words of a pass: the tile, then the staged shared twiddles (HM_TW_WORDS words each: one in Montgomery form, two with a Shoup companion)
template <int TL, int LOGR, bool STRIDED>
struct HmLds {
  static constexpr int TILE = 1 << TL, THREADS = TILE / HM_EPT;
  static constexpr int NTW = STRIDED ? (1 << LOGR) : 128;  // staged entries: w[0 .. NTW) of the modulus
  static constexpr int WORDS = TILE + (HM_TW_IN_LDS(STRIDED) ? HM_TW_WORDS * NTW : 0);
  static_assert(NTW % 2 == 0, "staged in 16-byte units");
};

// ---------------------------------------------------------------------------------------------------
// Per-thread phases.  A round = NB butterfly stages on local bits [K, K+NB) of x, done in registers by the
// thread that owns the 2^NB elements of a group.  The first round of a pass reads its elements straight
// from global memory and the last one writes straight back; only the exchanges between rounds go through
// LDS (one barrier each).  Twiddles that come from global memory are requested one round ahead, so that their
// latency overlaps the butterflies.  HmNttState is the per-thread register state (the host emulator keeps one per
// thread).
// ---------------------------------------------------------------------------------------------------
#define HM_MAX_TW 14  // twiddles of one round of one thread: a pair of radix-8 groups 7, four radix-4 groups 12
struct HmNttState {
  uint64_t v[HM_EPT];
  HmW tw[4][HM_MAX_TW];
  HmW tws[3];  // twist constants of the thread's row (ROW pass)
};

// Round schedule per sub-transform length: bits are consumed from the top for the forward transform
// (K descending) and from the bottom for the inverse.  {NB, K} lists, forward order.
template <int LOGR> struct HmRounds;
#if HM_EPT == 16
template <> struct HmRounds<5> { static constexpr int n = 2; static constexpr int nb[4] = {3, 2, 0, 0}; static constexpr int k[4] = {2, 0, 0, 0}; };
template <> struct HmRounds<6> { static constexpr int n = 2; static constexpr int nb[4] = {3, 3, 0, 0}; static constexpr int k[4] = {3, 0, 0, 0}; };
template <> struct HmRounds<7> { static constexpr int n = 3; static constexpr int nb[4] = {3, 2, 2, 0}; static constexpr int k[4] = {4, 2, 0, 0}; };
template <> struct HmRounds<8> { static constexpr int n = 3; static constexpr int nb[4] = {3, 3, 2, 0}; static constexpr int k[4] = {5, 2, 0, 0}; };
template <> struct HmRounds<9> { static constexpr int n = 3; static constexpr int nb[4] = {3, 3, 3, 0}; static constexpr int k[4] = {6, 3, 0, 0}; };
#else   // 8 coefficients per thread: a pair of radix-4 groups per round, four rounds of a 256-point pass; the 128-point COL pass of
        // N = 2^15 (round 6) ends on a radix-2 round (four groups per thread)
template <> struct HmRounds<7> { static constexpr int n = 4; static constexpr int nb[4] = {2, 2, 2, 1}; static constexpr int k[4] = {5, 3, 1, 0}; };
template <> struct HmRounds<8> { static constexpr int n = 4; static constexpr int nb[4] = {2, 2, 2, 2}; static constexpr int k[4] = {6, 4, 2, 0}; };
#endif

// geometry of round R of a pass
template <int TL, int LOGR, bool STRIDED, int R>
struct HmRound {
  static constexpr int NB = HmRounds<LOGR>::nb[R], K = HmRounds<LOGR>::k[R];
  static constexpr int E = 1 << NB;
  static constexpr int THREADS = (1 << TL) / HM_EPT;
  static constexpr int LOGC = TL - LOGR, C = 1 << LOGC;
  static constexpr int NG = HM_EPT / E;             // groups per thread
  static constexpr int XR = (1 << LOGR) >> NB;      // groups per sub-transform
  // groups 2v, 2v+1 of a thread are neighbours in memory and share their twiddles.  Not so in the CONTIG round on the
  // lowest bits (K == 0): there the E elements of ONE group are contiguous.
  static constexpr bool PAIRED = STRIDED || K >= 1;
  static constexpr int SETS = PAIRED ? NG / 2 : NG;  // twiddle sets of a thread, E - 1 twiddles each
  static_assert(SETS * (E - 1) <= HM_MAX_TW, "twiddle registers");
  static_assert(C >= 2, "a pair needs two columns");
  // group u of thread tid: sub-transform c, group index xr inside it
  static HM_HD void group(int tid, int u, int &c, int &xr) {
    if (STRIDED) {
      const int pid = tid + THREADS * (u >> 1);
      c = ((pid & (C / 2 - 1)) << 1) | (u & 1);
      xr = pid >> (LOGC - 1);
    } else if (K >= 1) {
      const int pid = tid + THREADS * (u >> 1);
      xr = ((pid & (XR / 2 - 1)) << 1) | (u & 1);
      c = pid / (XR / 2);
    } else {
      constexpr int LPR = XR / NG;  // lanes per row
      xr = (tid & (LPR - 1)) + LPR * u;
      c = tid / LPR;
    }
  }
  static HM_HD void coords(int tid, int u, int &c, int &hi, int &xb) {
    int xr;
    group(tid, u, c, xr);
    const int lo = xr & ((1 << K) - 1);
    hi = xr >> K;
    xb = (hi << (K + NB)) | lo;
  }
  static constexpr int twslot(int u) { return (PAIRED ? (u >> 1) : u) * (E - 1); }
  // access unit a < 8 of thread tid: register indices of its two words and the coordinates of the first one
  static HM_HD void unit(int tid, int a, int &i0, int &i1, int &x, int &c) {
    int hi, xb;
    if (PAIRED) {
      const int v = a / E, e = a % E;
      i0 = (2 * v) * E + e;
      i1 = (2 * v + 1) * E + e;
      coords(tid, 2 * v, c, hi, xb);
      x = xb | (e << K);
    } else {
      const int u = a / (E / 2), h = a % (E / 2);
      i0 = u * E + 2 * h;
      i1 = i0 + 1;
      coords(tid, u, c, hi, xb);
      x = xb | (2 * h);
    }
  }
  // the same, x split into the thread's part xb and the unit's part d (a constant after unrolling; xb & d == 0): the LDS index is
  // XOR-linear in x, so hm_lds_at takes idx(xb, c) ^ idx(d, 0) — one index per group and thread, one XOR with a constant per unit
  static HM_HD void unit_split(int tid, int a, int &i0, int &i1, int &xb, int &d, int &c) {
    int hi;
    if (PAIRED) {
      const int v = a / E, e = a % E;
      i0 = (2 * v) * E + e;
      i1 = (2 * v + 1) * E + e;
      coords(tid, 2 * v, c, hi, xb);
      d = e << K;
    } else {
      const int u = a / (E / 2), h = a % (E / 2);
      i0 = u * E + 2 * h;
      i1 = i0 + 1;
      coords(tid, 0, c, hi, xb);   // group u of a thread is XR / NG groups after its group 0: bits above the lane's
      d = ((XR / NG * u) << NB) | 2 * h;
    }
  }
  static HM_HD uint32_t gidx(uint32_t tile, int x, int c) {
    if (STRIDED) return ((uint32_t)x << HM_ROW_LOG) + (tile << LOGC) + (uint32_t)c;
    return (tile << TL) + ((uint32_t)c << LOGR) + (uint32_t)x;
  }
  // the same index for access unit a, split into a wave-uniform part (tile, element number: scalar registers, folded
  // into the base pointer) and the thread's part (one 32-bit register per pair / group): global accesses then take the
  // `scalar base + 32-bit lane offset` form instead of a 64-bit address register pair per unit
  static HM_HD uint32_t guni(uint32_t tile, int a) {
    if (PAIRED) {
      const int e = a % E;
      return STRIDED ? ((uint32_t)e << (K + HM_ROW_LOG)) + (tile << LOGC) : (tile << TL) + ((uint32_t)e << K);
    }
    return (tile << TL) + 2u * (uint32_t)(a % (E / 2));
  }
  static HM_HD uint32_t gthr(int tid, int a) {
    int c, hi, xb;
    coords(tid, PAIRED ? 2 * (a / E) : a / (E / 2), c, hi, xb);
    return STRIDED ? ((uint32_t)xb << HM_ROW_LOG) + (uint32_t)c : ((uint32_t)c << LOGR) + (uint32_t)xb;
  }
};

// Forward transform: which kind of butterfly (hm_bfly_fwd_k) local stage sigma of a pass runs.  Bounds in units of q, written for
// mont32 (U = HM_LAZY_Q / 2 = 1; the generic back-end's lazy product is twice as wide and every figure doubles): every value
// stays below 8q <= 2^63, the Montgomery product's operand range; the COL pass starts from reduced data (1), its first two stages need
// no subtraction (3, 5); from there on the stages alternate so that the pass ENDS on a subtracting stage (6 out); the ROW pass
// (8 stages, 6 in) alternates 0 / 1 and ends with kind 2 (4 out).  hm_fwd_bound replays the bounds at compile time: every stage is
// checked below.
constexpr int hm_fwd_kind(bool strided, int logr, int sigma) {
  if (strided) return sigma <= 1 ? 0 : (((logr - 1 - sigma) & 1) == 0 ? 1 : 0);
  return sigma == logr - 1 ? 2 : (sigma & 1);
}
constexpr int hm_fwd_bound(bool strided, int logr, int upto) {  // bound (in q) of the values entering local stage `upto`
  constexpr int U = HM_LAZY_Q / 2;
  int b = strided ? 1 : 6 * U;
  for (int s = 0; s < upto; ++s) {
    const int k = hm_fwd_kind(strided, logr, s);
    if (k == 0 ? b > 6 * U : b > 8 * U) return 1000;            // the stage's input condition
    b = (k == 0 ? b : k == 1 ? 4 * U : 2 * U) + 2 * U;
  }
  return b;
}
static_assert(hm_fwd_bound(true, 5, 5) == 3 * HM_LAZY_Q && hm_fwd_bound(true, 6, 6) == 3 * HM_LAZY_Q && hm_fwd_bound(true, 7, 7) == 3 * HM_LAZY_Q &&
              hm_fwd_bound(true, 8, 8) == 3 * HM_LAZY_Q && hm_fwd_bound(true, 9, 9) == 3 * HM_LAZY_Q, "COL pass hands over values below 6q (generic: 12q)");
static_assert(hm_fwd_bound(false, 8, 8) == 2 * HM_LAZY_Q, "ROW pass ends below 4q (generic: 8q)");

// Where the twiddles of a pass come from.  exec(i) = the i-th round executed.
template <int LOGR, bool STRIDED, bool INV>
struct HmPass {
  static constexpr int n = HmRounds<LOGR>::n;
  static constexpr int exec(int i) { return INV ? n - 1 - i : i; }
  // the ROW round on the lowest two bits runs with the shared twiddles (data twisted by alpha^(j mod 4))
#if defined(HM_NO_TWIST)   // ablation: every ROW round with the row's private twiddles (as much twiddle traffic as data)
  static constexpr int twistRound = -1;
#else
  static constexpr int twistRound = (!STRIDED && LOGR == 8) ? n - 1 : -1;
#endif
  static constexpr bool shared(int R) { return STRIDED || R == twistRound; }
  static constexpr bool fromLds(int R) { return HM_TW_IN_LDS(STRIDED) && shared(R) && R != exec(0); }
  static constexpr bool anyLds() { return fromLds(exec(1)) || (n >= 3 && fromLds(exec(2))) || (n >= 4 && fromLds(exec(3))); }
};

// request the twiddles of round R: sub-stage j needs 2^j of them, indexed by the top j bits of e.
// SHARED ROW round: the row-independent factor w[h]; otherwise the full table entry.
template <int TL, int LOGR, bool STRIDED, int R, bool SHARED>
HM_HD void hm_ph_load_tw(HmNttState &st, int tid, const HmW *twl, uint32_t s0, uint32_t prefix0) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
#pragma unroll
  for (int v = 0; v < G::SETS; ++v) {
    int c, hi, xb;
    G::coords(tid, G::PAIRED ? 2 * v : v, c, hi, xb);
    const uint32_t prefix = (STRIDED || SHARED) ? 0u : (prefix0 + (uint32_t)c);
#pragma unroll
    for (int j = 0; j < G::NB; ++j) {
      const int sigma = LOGR - G::K - G::NB + j;  // local stage index
      const uint32_t twbase = ((!STRIDED && SHARED) ? 0u : (1u << (s0 + sigma))) + (prefix << sigma) + ((uint32_t)hi << j);
#pragma unroll
      for (int t = 0; t < (1 << j); ++t) {
#if defined(HM_ABL_NOTW)
#if HM_GENERIC
        st.tw[R][v * (G::E - 1) + (1 << j) - 1 + t] = HmTw{(uint64_t)(twbase + t) * 0x9E3779B97F4A7C15ull >> 5, (uint64_t)(twbase + t) * 0xD1342543DE82EF95ull};
#else
        st.tw[R][v * (G::E - 1) + (1 << j) - 1 + t] = (uint64_t)(twbase + t) * 0x9E3779B97F4A7C15ull >> 5;
#endif
#else
        st.tw[R][v * (G::E - 1) + (1 << j) - 1 + t] = twl[twbase + (uint32_t)t];
#endif
      }
    }
  }
}
// the three twist constants of the thread's row (the K == 0 ROW round: all groups of a thread lie in one row)
template <int TL, int LOGR, bool STRIDED, int R>
HM_HD void hm_ph_load_twist(HmNttState &st, int tid, const HmW *twist_tile) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
  int c, xr;
  G::group(tid, 0, c, xr);
#pragma unroll
  for (int k = 0; k < 3; ++k) st.tws[k] = twist_tile[c * 3 + k];
}
// copy the shared twiddles w[0 .. NTW) of the modulus into LDS (visible after the next barrier)
template <int TL, int LOGR, bool STRIDED>
HM_HD void hm_ph_stage_tw(int tid, uint64_t *lds, const HmW *twl) {
  using LD = HmLds<TL, LOGR, STRIDED>;
  const uint64_t *tww = reinterpret_cast<const uint64_t *>(twl);   // the table as words: 16 bytes = two Montgomery entries or one Shoup pair
  constexpr int UNITS = LD::NTW * HM_TW_WORDS / 2;
#pragma unroll
  for (int i = 0; i < (UNITS + LD::THREADS - 1) / LD::THREADS; ++i) {   // 16 bytes per thread and step
    const int k = tid + LD::THREADS * i;
    if (k < UNITS) hm_st2(lds + LD::TILE + 2 * k, tww[2 * k], tww[2 * k + 1]);
  }
}

template <int TL, int LOGR, bool STRIDED, int R, int AUX = 0>
HM_HD void hm_ph_load_global(HmNttState &st, int tid, const uint64_t *g, uint32_t tile) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
#pragma unroll
  for (int a = 0; a < HM_UNITS; ++a) {
    int i0, i1, x, c;
    G::unit(tid, a, i0, i1, x, c);
#if defined(HM_ABL_NOMEM)   // timing-only ablation build (tools/ablate.sh): no data traffic
    st.v[i0] = (uint64_t)G::gidx(tile, x, c) * 0x9E3779B97F4A7C15ull >> 5; st.v[i1] = st.v[i0] ^ 0x5555;
#else
    hm_gld2<G, AUX>(g, tile, tid, a, st.v[i0], st.v[i1]);
#endif
  }
}
// MODE 6 (round 6): the input is read through an automorphism (ep.g; hm_gld2_auto): INTT(auto(x)) without auto(x) in memory
template <int TL, int LOGR, bool STRIDED, int R, int AUX = 0>
HM_HD void hm_ph_load_global_auto(HmNttState &st, int tid, const uint64_t *g, uint32_t tile, const HmEpi &ep) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
  static_assert(!STRIDED, "units of the COL pass are pairs of columns, not of neighbours in memory");
#pragma unroll
  for (int a = 0; a < HM_UNITS; ++a) {
    int i0, i1, x, c;
    G::unit(tid, a, i0, i1, x, c);
    hm_gld2_auto<AUX>(g, G::gidx(tile, x, c), ep.g, ep.logN, st.v[i0], st.v[i1]);
  }
}
// MODE 4: the same with the linear prologue x = in + k * mix (both reduced; x reduced)
template <int TL, int LOGR, bool STRIDED, int R, int AUX = 0>
HM_HD void hm_ph_load_global_mix(HmNttState &st, int tid, const uint64_t *g, uint32_t tile, uint64_t q, HmEpi ep) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
#pragma unroll
  for (int a = 0; a < HM_UNITS; ++a) {
    int i0, i1, x, c;
    G::unit(tid, a, i0, i1, x, c);
    uint64_t p0, p1, b0, b1;
    hm_gld2<G, AUX>(g, tile, tid, a, p0, p1);
    hm_gld2<G, AUX>(ep.b, tile, tid, a, b0, b1);
    st.v[i0] = hm_addmod(p0, hm_kmul(b0, ep.bk, q), q);
    st.v[i1] = hm_addmod(p1, hm_kmul(b1, ep.bk, q), q);
  }
}

// MODE 5 (ROW pass of the fused transform x key inner product): the last pass keeps its results in registers (lazy, below
// 2 HM_LAZY_Q q) and stores nothing; hm_ph_mac consumes them.
// the epilogue of one coefficient.  MODE 0 / 4 / 6: store as is (lazy values, hand-off between the two passes; 4 = first
// pass with the mix prologue, 6 = inverse first pass whose input is read through an automorphism); 1: forward final, reduce [0, 2 HM_LAZY_Q q) -> [0,q); 2: inverse final, multiply by the per-limb
// constant and reduce to [0,q); 3: forward final fused with out = (minuend - x) * k [+ addend [* ak]]; 7: the same with the addend read through an
// automorphism (hm_ph_store_global gathers it)
// (sc, ep.dk, ep.bk: constant records made by hm_kconst on the host; hm_kmul multiplies by them)
template <int MODE>
HM_HD uint64_t hm_epilogue(uint64_t a, uint64_t va, uint64_t vd, uint64_t q, HmTw sc, const HmEpi &ep) {
  if (MODE == 1) return hm_reduce_fwd(a, q);
  if (MODE == 2) { a = hm_kmul(a, sc, q); return ep.pack ? hm_pack30(a) : a; }
  if (MODE == 3 || MODE == 7) {  // a in [0, 2 HM_LAZY_Q q): minuend - a + 2 HM_LAZY_Q q stays positive and below 9q < 2^64 (mont32: 5q < 2^63); the product reduces it
    a = hm_kmul(va + 2 * HM_LAZY_Q * q - a, sc, q);
    if (ep.d) a = hm_addmod(a, ep.dk.w ? hm_kmul(vd, ep.dk, q) : vd, q);
  }
  return a;
}
template <int TL, int LOGR, bool STRIDED, int R, int MODE, int AUX = 0, int CH = HM_EPI_CHUNK>
HM_HD void hm_ph_store_global(const HmNttState &st, int tid, uint64_t *g, uint32_t tile, uint64_t q, HmTw sc, HmEpi ep) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
  // MODE 3: the epilogue operands are requested two units at a time, right before they are used (with a scheduling
  // fence in between): prefetching all of them before the last round costs 32 registers
  // CH = units per chunk (HM_EPI_CHUNK in the two-kernel transform, 1 inside the one-launch transform, whose second pass has less room)
#pragma unroll
  for (int a2 = 0; a2 < HM_UNITS; a2 += CH) {
    uint64_t ea[2 * CH], ed[2 * CH];
#pragma unroll
    for (int k = 0; k < 2 * CH; ++k) ea[k] = ed[k] = 0;
    if (MODE == 3 || MODE == 7) {
#pragma unroll
      for (int k = 0; k < CH && a2 + k < HM_UNITS; ++k) {
        int i0, i1, x, c;
        G::unit(tid, a2 + k, i0, i1, x, c);
#if !defined(HM_ABL_EPI_NOA)   // (timing-only ablations: the fused epilogue without its minuend / addend loads)
        hm_gld2<G>(ep.a, tile, tid, a2 + k, ea[2 * k], ea[2 * k + 1]);
#endif
#if !defined(HM_ABL_EPI_NOD)
        if constexpr (MODE == 7) {   // the addend through an automorphism (ep.g; 1 = as stored)
          if (ep.d) hm_gld2_auto<0>(ep.d, G::gidx(tile, x, c), ep.g, ep.logN, ed[2 * k], ed[2 * k + 1]);
        } else if (ep.d) hm_gld2<G>(ep.d, tile, tid, a2 + k, ed[2 * k], ed[2 * k + 1]);
#endif
      }
    }
#pragma unroll
    for (int k = 0; k < CH && a2 + k < HM_UNITS; ++k) {
      int i0, i1, x, c;
      G::unit(tid, a2 + k, i0, i1, x, c);
#if defined(HM_ABL_NOMEM)
      if (st.v[i0] == 0x123456789ull)   // never true in practice: keeps the values alive without the store traffic
#endif
      hm_gst2<G, AUX>(g, tile, tid, a2 + k, hm_epilogue<MODE>(st.v[i0], ea[2 * k], ed[2 * k], q, sc, ep),
                 hm_epilogue<MODE>(st.v[i1], ea[2 * k + 1], ed[2 * k + 1], q, sc, ep));
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (MODE == 3 || MODE == 7) __builtin_amdgcn_sched_barrier(0);
#endif
  }
}

// LDS word index of access unit (xb | d, c): see HmRound::unit_split.  INV: the pass's direction picks the swizzle (hm_lds_idx)
template <int TL, int LOGR, bool STRIDED, bool INV>
HM_HD int hm_lds_at(int xb, int d, int c) {
  constexpr int SWZ = HM_EPT == 8 ? 2 : INV ? 1 : 0;
  return hm_lds_idx<TL, LOGR, STRIDED, SWZ>(xb, c) ^ hm_lds_idx<TL, LOGR, STRIDED, SWZ>(d, 0);
}
template <int TL, int LOGR, bool STRIDED, int R, bool INV = false>
HM_HD void hm_ph_load_lds(HmNttState &st, int tid, const uint64_t *lds) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
#pragma unroll
  for (int a = 0; a < HM_UNITS; ++a) {
    int i0, i1, xb, d, c;
    G::unit_split(tid, a, i0, i1, xb, d, c);
    hm_ld2(lds + hm_lds_at<TL, LOGR, STRIDED, INV>(xb, d, c), st.v[i0], st.v[i1]);
  }
}
// MODE 4 with the tile in LDS: x = tile + k * mix, the mix operand straight from global memory
template <int TL, int LOGR, bool STRIDED, int R, int AUX = 0, bool INV = false>
HM_HD void hm_ph_load_lds_mix(HmNttState &st, int tid, const uint64_t *lds, uint32_t tile, uint64_t q, HmEpi ep) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
#pragma unroll
  for (int a = 0; a < HM_UNITS; ++a) {
    int i0, i1, xb, d, c;
    G::unit_split(tid, a, i0, i1, xb, d, c);
    uint64_t p0, p1, b0, b1;
    hm_gld2<G, AUX>(ep.b, tile, tid, a, b0, b1);
    hm_ld2(lds + hm_lds_at<TL, LOGR, STRIDED, INV>(xb, d, c), p0, p1);
    st.v[i0] = hm_addmod(p0, hm_kmul(b0, ep.bk, q), q);
    st.v[i1] = hm_addmod(p1, hm_kmul(b1, ep.bk, q), q);
  }
}
template <int TL, int LOGR, bool STRIDED, int R, bool INV = false>
HM_HD void hm_ph_store_lds(const HmNttState &st, int tid, uint64_t *lds) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
#pragma unroll
  for (int a = 0; a < HM_UNITS; ++a) {
    int i0, i1, xb, d, c;
    G::unit_split(tid, a, i0, i1, xb, d, c);
    hm_st2(lds + hm_lds_at<TL, LOGR, STRIDED, INV>(xb, d, c), st.v[i0], st.v[i1]);
  }
}

template <int TL, int LOGR, bool STRIDED, int R, bool INV>
HM_HD void hm_ph_compute(HmNttState &st, uint64_t q) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
#if defined(HM_ABL_NOCOMPUTE)
  st.v[0] ^= hm_tw_bits(st.tw[R][0]);  // keeps the twiddle loads alive
  return;
#endif
  const HmBflyMod m = hm_bfly_mod(q);  // lazy ranges: forward [0, 8q) inside a pass, inverse [0, 4q)
#pragma unroll
  for (int jj = 0; jj < G::NB; ++jj) {
    const int j = INV ? (G::NB - 1 - jj) : jj;  // sub-stage j combines e-bit (NB-1-j)
    const int pb = G::NB - 1 - j;
#pragma unroll
    for (int u = 0; u < G::NG; ++u) {
#pragma unroll
      for (int e = 0; e < G::E; ++e) {
        if (e & (1 << pb)) continue;
        const HmW t = st.tw[R][G::twslot(u) + (1 << j) - 1 + (e >> (G::NB - j))];
        constexpr int sigma = LOGR - G::K - G::NB;  // + j: the local stage (j is a constant once unrolled)
        if (INV) hm_bfly_inv(st.v[u * G::E + e], st.v[u * G::E + (e | (1 << pb))], t, m);
        else if (hm_fwd_kind(STRIDED, LOGR, sigma + j) == 0) hm_bfly_fwd_k<0>(st.v[u * G::E + e], st.v[u * G::E + (e | (1 << pb))], t, m);
        else if (hm_fwd_kind(STRIDED, LOGR, sigma + j) == 1) hm_bfly_fwd_k<1>(st.v[u * G::E + e], st.v[u * G::E + (e | (1 << pb))], t, m);
        else hm_bfly_fwd_k<2>(st.v[u * G::E + e], st.v[u * G::E + (e | (1 << pb))], t, m);
      }
    }
  }
}
// multiply element j of the row by tws[(j mod 4) - 1] (the K == 0, NB == 2 ROW round: v[4 u + e] is element 4 hi + e).
// Lazy Montgomery product (constants in Montgomery form): input below 2^63 (every value of a pass is below 8q), result in
// [0, 1.5q + 2^28) — inside the input range of both butterfly forms.
HM_HD void hm_ph_twist(HmNttState &st, uint64_t q) {
#if defined(HM_ABL_NOCOMPUTE)
  st.v[1] ^= hm_tw_bits(st.tws[0]) ^ hm_tw_bits(st.tws[1]) ^ hm_tw_bits(st.tws[2]);
  return;
#endif
  const HmBflyMod m = hm_bfly_mod(q);
#pragma unroll
  for (int u = 0; u < HM_EPT / 4; ++u)
#pragma unroll
    for (int k = 1; k < 4; ++k) st.v[4 * u + k] = hm_tw_acc(0, st.v[4 * u + k], st.tws[k - 1], m);
}

// The whole pass of one thread, phase by phase.  `sync` is __syncthreads() on the GPU; the emulator calls the
// phases itself (see tests/emu/hm_emu.cpp) in the same order, one phase for all threads at a time.
// A pass of n rounds (n = 2, 3; 4 in the 8-coefficient geometry) has n + 1 phases; exec(i) = the i-th round executed (the
// inverse walks the rounds from the last to the first):
//   P0:      stage shared tw -> LDS, tw[exec 0], global -> v, [twist constants], tw[exec 1] (if global, early)
//   P1:      compute exec 0, [inverse twist], [tw[exec 1] late], v -> LDS                                    | barrier
//   Pi+1:    tw[exec i+1] (if global), [twist constants for the last round], tw[exec i] (if LDS), LDS -> v,
//            compute exec i, v -> LDS                        for the middle rounds 0 < i < n - 1              | barrier
//   Pn:      tw[exec n-1] (if LDS), LDS -> v, [forward twist], compute exec n-1, v -> global (MODE 5: stays in registers)
// twl = table of the modulus; twist_tile = twist constants of the tile's first row (ROW pass)
// LDAUX / STAUX: cache-policy bits of the pass's data loads / stores (hm_gld2)
// pre() (phase 0 only): called between the first round's twiddle requests and everything that touches LDS or the pass's input — the
// one-launch transform waits there for the other workgroups' hand-off, with the second pass's first twiddles already on their way
// FROMREG (round 5): the pass's input is already in st.v, in the layout of its first executed round (the inner-product kernel hands its
// reduced accumulators to the inverse ROW pass this way: the forward ROW pass's last round and the inverse ROW pass's first are the same
// round); phase 0 then requests twiddles only
template <int TL, int LOGR, bool STRIDED, bool INV, int MODE, int PHASE, int LDAUX = 0, int STAUX = 0, int EPICH = HM_EPI_CHUNK, class PRE = HmNoPre, bool FROMREG = false>
HM_HD void hm_ntt_phase(HmNttState &st, int tid, uint64_t *lds, const uint64_t *src, uint64_t *dst, uint32_t tile,
                        const HmW *twl, const HmW *twist_tile, uint32_t s0, uint32_t prefix0, uint64_t q, HmTw sc, HmEpi ep, PRE pre = PRE()) {
  using PS = HmPass<LOGR, STRIDED, INV>;
  constexpr int n = PS::n;
  static_assert(PHASE >= 0 && PHASE <= n, "a pass of n rounds has phases 0 .. n");
  constexpr int TWR = PS::twistRound;
  constexpr int iTW = TWR < 0 ? -1 : (INV ? n - 1 - TWR : TWR);   // execution index of the twisted round
  const HmW *ltw = reinterpret_cast<const HmW *>(lds + (1 << TL));   // the staged shared twiddles
  // register pressure: the inverse ROW pass has 120 registers of loads in flight in its first phase (data, the first
  // round's shared twiddles, the twist constants and the NEXT round's private twiddles); inside the one-launch transform
  // that no longer fits 128.  The next round's twiddles are then requested after the first round's butterflies instead
  // (they still have the LDS exchange and the barrier to arrive in).
  constexpr bool LATE_TW1 = ((INV && HM_LATE_TW1) || MODE == 5) && !STRIDED;   // MODE 5: 64 accumulator registers are live beside the pass
  if constexpr (PHASE == 0) {
    constexpr int r0 = PS::exec(0), r1 = PS::exec(1);
    hm_ph_load_tw<TL, LOGR, STRIDED, r0, PS::shared(r0)>(st, tid, twl, s0, prefix0);
    pre();
    if (PS::anyLds()) hm_ph_stage_tw<TL, LOGR, STRIDED>(tid, lds, twl);
    if (FROMREG) { static_assert(!FROMREG || MODE != 4, "the mix prologue reads its operands from memory"); }
    else if (MODE == 4) hm_ph_load_global_mix<TL, LOGR, STRIDED, r0, LDAUX>(st, tid, src, tile, q, ep);
    else if constexpr (MODE == 6) hm_ph_load_global_auto<TL, LOGR, STRIDED, r0, LDAUX>(st, tid, src, tile, ep);
    else hm_ph_load_global<TL, LOGR, STRIDED, r0, LDAUX>(st, tid, src, tile);
    // the twist constants are requested one phase ahead of the twisted round (phase iTW + 1)
    if (iTW >= 0 && iTW <= 1) hm_ph_load_twist<TL, LOGR, STRIDED, (TWR >= 0 ? TWR : 0)>(st, tid, twist_tile);
    if (!PS::fromLds(r1) && !LATE_TW1) hm_ph_load_tw<TL, LOGR, STRIDED, r1, PS::shared(r1)>(st, tid, twl, s0, prefix0);
  } else if constexpr (PHASE == 1) {
    constexpr int r0 = PS::exec(0), r1 = PS::exec(1);
    hm_ph_compute<TL, LOGR, STRIDED, r0, INV>(st, q);
    if (INV && r0 == TWR) hm_ph_twist(st, q);
    if (!PS::fromLds(r1) && LATE_TW1) {
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_sched_barrier(0);
#endif
      hm_ph_load_tw<TL, LOGR, STRIDED, r1, PS::shared(r1)>(st, tid, twl, s0, prefix0);
    }
    hm_ph_store_lds<TL, LOGR, STRIDED, r0, INV>(st, tid, lds);
  } else if constexpr (PHASE < n) {   // a middle round
    constexpr int i = PHASE - 1, ri = PS::exec(i), rn = PS::exec(i + 1);
    if (!PS::fromLds(rn)) hm_ph_load_tw<TL, LOGR, STRIDED, rn, PS::shared(rn)>(st, tid, twl, s0, prefix0);   // next round's, a round ahead
    if (iTW == i + 1 && iTW >= 2) hm_ph_load_twist<TL, LOGR, STRIDED, (TWR >= 0 ? TWR : 0)>(st, tid, twist_tile);  // used in the next phase
    if (PS::fromLds(ri)) hm_ph_load_tw<TL, LOGR, STRIDED, ri, PS::shared(ri)>(st, tid, ltw, s0, prefix0);
    hm_ph_load_lds<TL, LOGR, STRIDED, ri, INV>(st, tid, lds);
    hm_ph_compute<TL, LOGR, STRIDED, ri, INV>(st, q);
    hm_ph_store_lds<TL, LOGR, STRIDED, ri, INV>(st, tid, lds);
  } else {                            // the last round
    constexpr int rl = PS::exec(n - 1);
    // fused epilogue (MODE 3) / register hand-over (MODE 5): the twist constants are dead before the last round's twiddles are
    // read from LDS (the other order keeps 12 more registers alive and spilled inside the one-launch transform)
    constexpr bool TWIST_FIRST = (MODE == 3 || MODE == 5 || MODE == 7) && !INV && rl == TWR && PS::fromLds(rl);
    if (PS::fromLds(rl) && !TWIST_FIRST) hm_ph_load_tw<TL, LOGR, STRIDED, rl, PS::shared(rl)>(st, tid, ltw, s0, prefix0);
    hm_ph_load_lds<TL, LOGR, STRIDED, rl, INV>(st, tid, lds);
    if (!INV && rl == TWR) hm_ph_twist(st, q);
    if (TWIST_FIRST) {
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_sched_barrier(0);
#endif
      hm_ph_load_tw<TL, LOGR, STRIDED, rl, PS::shared(rl)>(st, tid, ltw, s0, prefix0);
    }
    hm_ph_compute<TL, LOGR, STRIDED, rl, INV>(st, q);
    if (MODE == 5) return;
#if defined(__HIP_DEVICE_COMPILE__)
    if (TWIST_FIRST && HM_EPI_FENCE) __builtin_amdgcn_sched_barrier(0);   // the epilogue's operand loads start after the last butterflies
#endif
    hm_ph_store_global<TL, LOGR, STRIDED, rl, MODE, STAUX, EPICH>(st, tid, dst, tile, q, sc, ep);
  }
}
// all phases of a pass with `sync()` between them (the GPU passes __syncthreads, the emulator runs the phases itself)
template <int TL, int LOGR, bool STRIDED, bool INV, int MODE, int LDAUX = 0, int STAUX = 0, int EPICH = HM_EPI_CHUNK, bool FROMREG = false, class SYNC, class PRE = HmNoPre>
HM_HD void hm_ntt_pass_phases(HmNttState &st, int tid, uint64_t *lds, const uint64_t *src, uint64_t *dst, uint32_t tile,
                              const HmW *twl, const HmW *twist_tile, uint32_t s0, uint32_t prefix0, uint64_t q, HmTw sc, HmEpi ep, SYNC sync, PRE pre = PRE()) {
  constexpr int n = HmRounds<LOGR>::n;
  hm_ntt_phase<TL, LOGR, STRIDED, INV, MODE, 0, LDAUX, STAUX, EPICH, PRE, FROMREG>(st, tid, lds, src, dst, tile, twl, twist_tile, s0, prefix0, q, sc, ep, pre);
  hm_ntt_phase<TL, LOGR, STRIDED, INV, MODE, 1, LDAUX, STAUX, EPICH>(st, tid, lds, src, dst, tile, twl, twist_tile, s0, prefix0, q, sc, ep);
  sync();
  hm_ntt_phase<TL, LOGR, STRIDED, INV, MODE, 2, LDAUX, STAUX, EPICH>(st, tid, lds, src, dst, tile, twl, twist_tile, s0, prefix0, q, sc, ep);
  if constexpr (n >= 3) {
    sync();
    hm_ntt_phase<TL, LOGR, STRIDED, INV, MODE, 3, LDAUX, STAUX, EPICH>(st, tid, lds, src, dst, tile, twl, twist_tile, s0, prefix0, q, sc, ep);
  }
  if constexpr (n >= 4) {
    sync();
    hm_ntt_phase<TL, LOGR, STRIDED, INV, MODE, 4, LDAUX, STAUX, EPICH>(st, tid, lds, src, dst, tile, twl, twist_tile, s0, prefix0, q, sc, ep);
  }
}

// ---------------------------------------------------------------------------------------------------
// K1 x K5: the transform's last pass multiplied into the evaluation key (the reference's HPIP unit as a fused NTT-epilogue x
// evk MAC: HPIP src/Components.cpp:571-668, InsGen::GenHPIP src/InsGen.cpp:356-406, KeySwitch::InnerProduceOperation
// src/Operation.cpp:294-414).  After the ROW pass of digit j (MODE 5) a thread holds 16 coefficients of ext_j in the layout
// of the pass's last round; hm_ph_mac adds ext_j * evk_{j,k} for both keys to the thread's accumulators, which live in
// registers across the digits: the extended digit never exists in HBM.  Accumulators are lazy (below 8q, x may be any value below
// 2^63); hm_ph_mac_store reduces once.
// ---------------------------------------------------------------------------------------------------
// Accumulator forms: uint64_t — every product reduced lazily before it is added (64 registers for both keys); hm_u128 — the raw 128-bit
// products are summed (12 instructions per product: x < 2q and y < q keep 4 terms below 2^123 once x is brought below 2q) and reduced
// once per output by hm_barrett, at the price of 128 accumulator registers (two waves per SIMD).
//
// The lazy product (round 4): the transform's word-wise Montgomery product (hm_mont_acc, q = h 2^32 + 1) with the key word as the
// constant: acc += x y 2^-64 mod q + {0, q}, six multiplies and 11 instructions where Barrett's quotient from two approximate high
// products took twelve and 19.  x < 4q (the transform's lazy output; any x below 2^63 is allowed), y < q: a term adds less than
// 1.5q + 2^28, so up to five terms stay below 8q <= 2^63 with no folding.  hm_mac_final multiplies the sum by 2^128 mod q the same way
// (sum 2^-64 2^128 2^-64 = sum) and subtracts q once: 16 instructions per output.
#if HM_GENERIC
// generic: Barrett's quotient from two APPROXIMATE high products and no 128-bit shift (round 3).  With X = 2x (x < 8q < 2^63) and
// Y = y 2^(64-k) (y < q < 2^k) the high word of X Y IS floor(x y / 2^(k-1)); hm_shoup_quot gives it and then floor(zh mu / 2^64) from
// three v_mad_u64_u32 each, at most 2 below the true value.  The estimate never exceeds floor(x y / q) and is at most
// 2 (Barrett) + 2 (zh) + 2 (second product) = 6 below it, so x y - qe q lies in [0, 7q): about 19 instructions per product.  Two
// products fit a word (14q < 16q <= 2^64); from the third term on the accumulator is first brought below 8q (one conditional
// subtraction per accumulator and digit), so any number of terms stays below 15q; hm_mac_final reduces from below 16q.
struct HmMacMod {
  uint64_t mu, nq, nq8, z;
  uint32_t ysh;
};
HM_HD HmMacMod hm_mac_mod(const HmMod &m) {
  HmMacMod r;
  r.z = hm_opaque_zero();
  r.mu = m.mu;
  r.nq = r.z - m.q;
  r.nq8 = r.z - 8 * m.q;
  r.ysh = 63 - m.sh;
  return r;
}
HM_HD void hm_mac_add(uint64_t &acc, uint64_t x, uint64_t y, const HmMod &, const HmMacMod &mm, bool fold) {
  if (fold) acc = hm_csub_neg(acc, mm.nq8);                        // [0, 15q) -> [0, 8q)
  const uint64_t zh = hm_shoup_quot(x << 1, y << mm.ysh, mm.z);    // floor(x y / 2^(k-1)) - {0, 1, 2}
  const uint64_t qe = hm_shoup_quot(zh, mm.mu, mm.z);              // floor(x y / q) - {0 .. 6}
  acc = acc + x * y + qe * mm.nq;                                  // + (x y mod q) + {0 .. 6} q, exact in the low word
}
HM_HD uint64_t hm_mac_final(uint64_t acc, const HmMod &m) { return hm_reduce16(acc, m.q); }
// [0, 8q) -> [0, 2q): the wide accumulators take transform outputs below 2q, so that the sum stays inside hm_barrett's range
HM_HD void hm_ph_below_2q(HmNttState &st, uint64_t q) {
  const HmBflyMod m = hm_bfly_mod(q);
#pragma unroll
  for (int i = 0; i < HM_EPT; ++i) st.v[i] = hm_csub_neg(hm_csub_neg(st.v[i], m.nq4), m.nq2);
}
#else
struct HmMacMod {
  HmBflyMod b;
  uint64_t r128;
};
HM_HD HmMacMod hm_mac_mod(const HmMod &m) {
  HmMacMod r;
  r.b = hm_bfly_mod(m.q);
  r.r128 = m.r128;
  return r;
}
HM_HD void hm_mac_add(uint64_t &acc, uint64_t x, uint64_t y, const HmMod &, const HmMacMod &mm, bool) { acc = hm_mont_acc(acc, x, y, mm.b); }
HM_HD uint64_t hm_mac_final(uint64_t acc, const HmMod &m) {   // acc < 8q
  const HmBflyMod b = hm_bfly_mod(m.q);
  return hm_csub_neg(hm_mont_acc(0, acc, m.r128, b), b.nq);   // [0, 1.5q + 2^28) -> [0, q)
}
// [0, 4q) -> [0, 2q): the wide accumulators take transform outputs below 2q, so that the sum stays inside hm_barrett's range
HM_HD void hm_ph_below_2q(HmNttState &st, uint64_t q) {
  const HmBflyMod m = hm_bfly_mod(q);
#pragma unroll
  for (int i = 0; i < HM_EPT; ++i) st.v[i] = hm_csub_neg(st.v[i], m.nq2);
}
#endif
HM_HD void hm_mac_add(hm_u128 &acc, uint64_t x, uint64_t y, const HmMod &, const HmMacMod &, bool) { acc += (hm_u128)x * y; }
HM_HD uint64_t hm_mac_final(hm_u128 acc, const HmMod &m) { return hm_barrett(acc, m); }   // 4 terms x (x < 2q) x (y < q) < 2^123
template <int TL, int LOGR, int R, int OUTS, int CH = 2, class ACC = uint64_t>
HM_HD void hm_ph_mac(const HmNttState &st, ACC (&acc)[OUTS][HM_EPT], int tid, const uint64_t *const (&y)[OUTS], uint32_t tile, const HmMod &m, uint32_t term) {
  using G = HmRound<TL, LOGR, false, R>;
  const HmMacMod mm = hm_mac_mod(m);
  const bool fold = term >= 2;   // wave-uniform
#pragma unroll
  for (int a2 = 0; a2 < HM_UNITS; a2 += CH) {
    uint64_t e[OUTS][2 * CH];
#pragma unroll
    for (int k = 0; k < OUTS; ++k)
#pragma unroll
#if defined(HM_ABL_NIP_NOKEYLOAD)   // timing-only ablation: the products without the key loads
      for (int c = 0; c < CH; ++c) { e[k][2 * c] = (uint64_t)(uintptr_t)y[k] + a2; e[k][2 * c + 1] = e[k][2 * c] + tid; }
#else
      for (int c = 0; c < CH; ++c) hm_gld2<G>(y[k], tile, tid, a2 + c, e[k][2 * c], e[k][2 * c + 1]);
#endif
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      int i0, i1, x, cc;
      G::unit(tid, a2 + c, i0, i1, x, cc);
#pragma unroll
      for (int k = 0; k < OUTS; ++k) {
        hm_mac_add(acc[k][i0], st.v[i0], e[k][2 * c], m, mm, fold);
        hm_mac_add(acc[k][i1], st.v[i1], e[k][2 * c + 1], m, mm, fold);
      }
    }
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
}
// the same with the key words already in registers (requested before the digit's transform by hm_ph_key_load: two waves per
// SIMD leave room for them, and their latency hides behind the butterflies)
template <int TL, int LOGR, int R, int OUTS>
HM_HD void hm_ph_key_load(uint64_t (&e)[OUTS][HM_EPT], int tid, const uint64_t *const (&y)[OUTS], uint32_t tile) {
  using G = HmRound<TL, LOGR, false, R>;
#pragma unroll
  for (int k = 0; k < OUTS; ++k)
#pragma unroll
    for (int a = 0; a < HM_UNITS; ++a) {
      int i0, i1, x, c;
      G::unit(tid, a, i0, i1, x, c);
      hm_gld2<G>(y[k], tile, tid, a, e[k][i0], e[k][i1]);
    }
}
template <int OUTS, class ACC>
HM_HD void hm_ph_mac_regs(const HmNttState &st, ACC (&acc)[OUTS][HM_EPT], const uint64_t (&e)[OUTS][HM_EPT], const HmMod &m, uint32_t term) {
  const HmMacMod mm = hm_mac_mod(m);
  const bool fold = term >= 2;
#pragma unroll
  for (int k = 0; k < OUTS; ++k)
#pragma unroll
    for (int i = 0; i < HM_EPT; ++i) hm_mac_add(acc[k][i], st.v[i], e[k][i], m, mm, fold);
}
template <int TL, int LOGR, int R, int OUTS, class ACC, int AUX = 0>
HM_HD void hm_ph_mac_store(const ACC (&acc)[OUTS][HM_EPT], int tid, uint64_t *const (&out)[OUTS], uint32_t tile, const HmMod &m) {
  using G = HmRound<TL, LOGR, false, R>;
#pragma unroll
  for (int k = 0; k < OUTS; ++k)
#pragma unroll
    for (int a = 0; a < HM_UNITS; ++a) {
      int i0, i1, x, c;
      G::unit(tid, a, i0, i1, x, c);
      hm_gst2<G, AUX>(out[k], tile, tid, a, hm_mac_final(acc[k][i0], m), hm_mac_final(acc[k][i1], m));
    }
}

