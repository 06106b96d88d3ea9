// hm_ntt_core.h — K1: negacyclic NTT / INTT over one RNS limb as two global passes of LDS-tiled
// radix-8/4/2 rounds.  Replaces the reference's NTTU timing model (src/Components.cpp:380-436:
// 8 butterfly stages, transpose, 8 butterfly stages) and InsGen::GenNTT (src/InsGen.cpp:17-44)
// with real arithmetic; the ordering convention is SURVEY.md Appendix A (1).
//
// Decomposition of N = 2^logN coefficients, index i = x1 * 256 + x2:
//   pass COL ("strided"):  sub-transforms of length R1 = N/256 over x1, one per column x2
//                          (global stages 0 .. LOG1-1, twiddles shared by all columns)
//   pass ROW ("contig"):   sub-transforms of length 256 over x2, one per row x1
//                          (global stages LOG1 .. logN-1, twiddles private to the row)
// forward = COL then ROW (Cooley-Tukey, natural in, bit-reversed out);
// inverse = ROW then COL (Gentleman-Sande, bit-reversed in, natural out, then * scale).
//
// A workgroup of HM_THREADS = 512 threads owns a tile of HM_TILE = 4096 coefficients (32 KiB) in LDS and runs
// the rounds LDS -> registers -> LDS with one barrier per round; the phase functions below carry no
// register state across barriers, so the host emulator (tests/emu) can run them thread by thread.
#pragma once
#include "hm_modarith.h"

#define HM_TILE 4096
#define HM_TILE_LOG 12
#ifndef HM_THREADS
#define HM_THREADS 512
#endif
#define HM_MAX_LIMBS 128
#ifndef HM_ROW_LOG
#define HM_ROW_LOG 8  // log2 of the contiguous sub-transform length (pass ROW)
#endif

struct HmLimb {  // one limb-poly of an automorphism / fill launch: limb indices into the in/out bases, modulus id
  uint16_t in, out, mod, aux;
};

// One limb-poly of a transform launch.  The constants of a launch form a table in device memory (cached by content: the
// plans of this datapath repeat the same launches), so a launch carries up to HM_NTT_MAX_ENTRIES limb-polys instead of
// 128 (the kernel-argument segment is 4 KiB): the limb-polys of a batch go into one or two grids instead of 128-entry
// pieces with their ramp-up and tail.
// Fused forward transform (ModDown finish, rescale, or both merged):
//   first pass, MODE 4:  x = in + mixk * mix         (coefficient domain, before the first butterfly)
//   last pass,  MODE 3:  out = (minuend - NTT(x)) * sc [+ addend * ak]
struct HmNttEntry {                  // the part of a record that is not needed to start loading (device table)
  uint16_t alimb, mixlimb;           // MODE 3 addend limb (HM_NTT_NONE: no addend), MODE 4 operand limb
  uint16_t pad[2];
  uint64_t pad1;
  HmTw sc;                           // inverse: N^-1 * extra scale; fused forward: the epilogue constant k
  HmTw ak;                           // fused forward: addend constant (w == 0: none)
  HmTw mixk;                         // MODE 4 prologue constant
};
#define HM_NTT_NONE 0xFFFFu
#define HM_NTT_MAX_ENTRIES 448       // records in the kernel-argument segment (8 bytes each, 4 KiB limit)

struct HmNttArgs {
  const uint64_t *in;
  uint64_t *out;
  const HmTw *tw;      // [n_mod][N] forward or inverse table (chosen by the host)
  const HmMod *mods;   // [n_mod]
  const HmNttEntry *entry;                // [n_limbs], device
  const uint64_t *minuend, *addend, *mix; // bases of the MODE 3 / MODE 4 operands (addend may be null)
  uint32_t logN;
  uint32_t n_limbs;    // entries, a multiple of 16 (pairs x 8 XCDs, see hm_block_map)
  // what a workgroup needs before it can issue its first load rides in the kernel arguments (a dependent read of the
  // device table at workgroup start cost 5 % on the whole op): in / out limbs, modulus id (HM_NTT_NONE: empty slot) and,
  // in aux, the MODE 3 minuend limb
  HmLimb limb[HM_NTT_MAX_ENTRIES];
};
struct HmEpi {  // the prologue / epilogue operands of one limb-poly, resolved by the kernel
  const uint64_t *a, *d;     // minuend, addend (MODE 3)
  HmTw dk;                   // addend constant; dk.w == 0: none
  const uint64_t *b;         // mix operand (MODE 4)
  HmTw bk;
};
HM_HD HmEpi hm_epi_none() { return HmEpi{nullptr, nullptr, HmTw{0, 0}, nullptr, HmTw{0, 0}}; }

// LDS image of a tile.  STRIDED: [x][c] with the C columns contiguous (a plain copy of C-element
// row segments).  CONTIG: [c][x] with 4 words of padding per 32 so that stride-32 column reads of the
// middle round spread over all 64 banks.
template <int LOGR, bool STRIDED>
HM_HD int hm_lds_idx(int x, int c) {
  if (STRIDED) return (x << (HM_TILE_LOG - LOGR)) | c;
  int i = (c << LOGR) | x;
  return i + ((i >> 5) << 2);
}
#define HM_LDS_WORDS (HM_TILE + (HM_TILE >> 5) * 4)

// global index of tile-linear element `lin` (the order in which the tile is copied)
template <int LOGR, bool STRIDED>
HM_HD uint32_t hm_tile_gidx(uint32_t tile, uint32_t lin) {
  if (STRIDED) {  // lin = x * C + c ; global = x * 256 + tile * C + c
    const int LOGC = HM_TILE_LOG - LOGR;
    uint32_t x = lin >> LOGC, c = lin & ((1u << LOGC) - 1);
    return (x << HM_ROW_LOG) + (tile << LOGC) + c;
  }
  return (tile << HM_TILE_LOG) + lin;
}
template <int LOGR, bool STRIDED>
HM_HD int hm_tile_lidx(uint32_t lin) {
  if (STRIDED) return (int)lin;
  return (int)(lin + ((lin >> 5) << 2));
}

// ---------------------------------------------------------------------------------------------------
// Per-thread phases.  A round = NB butterfly stages on local bits [K, K+NB) of x, done in registers by the
// thread that owns the 2^NB elements of a group.  The first round of a pass reads its elements straight
// from global memory and the last one writes straight back; only the exchanges between rounds go through
// LDS (one barrier each).  Twiddles of round r+1 are requested before round r is computed, so that their
// HBM latency overlaps the butterflies.  HmNttState is the per-thread register state (the host emulator
// keeps one per thread).
// ---------------------------------------------------------------------------------------------------
#define HM_MAX_GPT 2  // groups per thread in one round (HM_TILE >> NB) / HM_THREADS
struct HmNttState {
  uint64_t v[HM_MAX_GPT][8];
  HmTw tw[3][HM_MAX_GPT][7];
};

// Round schedule per sub-transform length: bits are consumed from the top for the forward transform
// (K descending) and from the bottom for the inverse.  {NB, K} lists, forward order.
template <int LOGR> struct HmRounds;
template <> struct HmRounds<5> { static constexpr int n = 2; static constexpr int nb[3] = {3, 2, 0}; static constexpr int k[3] = {2, 0, 0}; };
template <> struct HmRounds<6> { static constexpr int n = 2; static constexpr int nb[3] = {3, 3, 0}; static constexpr int k[3] = {3, 0, 0}; };
template <> struct HmRounds<7> { static constexpr int n = 3; static constexpr int nb[3] = {3, 2, 2}; static constexpr int k[3] = {4, 2, 0}; };
template <> struct HmRounds<8> { static constexpr int n = 3; static constexpr int nb[3] = {3, 3, 2}; static constexpr int k[3] = {5, 2, 0}; };
template <> struct HmRounds<9> { static constexpr int n = 3; static constexpr int nb[3] = {3, 3, 3}; static constexpr int k[3] = {6, 3, 0}; };

// geometry of round R of a pass
template <int LOGR, bool STRIDED, int R>
struct HmRound {
  static constexpr int NB = HmRounds<LOGR>::nb[R], K = HmRounds<LOGR>::k[R];
  static constexpr int E = 1 << NB;
  static constexpr int LOGC = HM_TILE_LOG - LOGR, C = 1 << LOGC;
  static constexpr int GPT = (HM_TILE >> NB) / HM_THREADS;
  static constexpr int XR = (1 << LOGR) >> NB;  // groups per sub-transform
  static_assert(GPT >= 1 && GPT <= HM_MAX_GPT, "tile / thread geometry");
  // group u of thread tid: column c, high/low parts of the group index
  static HM_HD void coords(int tid, int u, int &c, int &hi, int &xb) {
    const int gid = tid + HM_THREADS * u;
    int xr;
    if (STRIDED) { c = gid & (C - 1); xr = gid >> LOGC; }
    else         { xr = gid & (XR - 1); c = gid / XR; }
    const int lo = xr & ((1 << K) - 1);
    hi = xr >> K;
    xb = (hi << (K + NB)) | lo;
  }
  static HM_HD uint32_t gidx(uint32_t tile, int x, int c) {
    if (STRIDED) return ((uint32_t)x << HM_ROW_LOG) + (tile << LOGC) + (uint32_t)c;
    return (tile << HM_TILE_LOG) + ((uint32_t)c << LOGR) + (uint32_t)x;
  }
};

// request the twiddles of round R: sub-stage j needs 2^j of them, indexed by the top j bits of e
template <int LOGR, bool STRIDED, int R>
HM_HD void hm_ph_load_tw(HmNttState &st, int tid, const HmTw *twl, uint32_t s0, uint32_t prefix0) {
  using G = HmRound<LOGR, STRIDED, R>;
#pragma unroll
  for (int u = 0; u < G::GPT; ++u) {
    int c, hi, xb;
    G::coords(tid, u, c, hi, xb);
    const uint32_t prefix = STRIDED ? 0u : (prefix0 + (uint32_t)c);
#pragma unroll
    for (int j = 0; j < G::NB; ++j) {
      const int sigma = LOGR - G::K - G::NB + j;  // local stage index
      const uint32_t twbase = (1u << (s0 + sigma)) + (prefix << sigma) + ((uint32_t)hi << j);
#pragma unroll
      for (int t = 0; t < (1 << j); ++t) st.tw[R][u][(1 << j) - 1 + t] = twl[twbase + (uint32_t)t];
    }
  }
}

template <int LOGR, bool STRIDED, int R>
HM_HD void hm_ph_load_global(HmNttState &st, int tid, const uint64_t *g, uint32_t tile) {
  using G = HmRound<LOGR, STRIDED, R>;
#pragma unroll
  for (int u = 0; u < G::GPT; ++u) {
    int c, hi, xb;
    G::coords(tid, u, c, hi, xb);
#pragma unroll
    for (int e = 0; e < G::E; ++e) st.v[u][e] = g[G::gidx(tile, xb | (e << G::K), c)];
  }
}
// MODE 4: the same with the linear prologue x = in + k * mix (both reduced; x reduced)
template <int LOGR, bool STRIDED, int R>
HM_HD void hm_ph_load_global_mix(HmNttState &st, int tid, const uint64_t *g, uint32_t tile, uint64_t q, HmEpi ep) {
  using G = HmRound<LOGR, STRIDED, R>;
#pragma unroll
  for (int u = 0; u < G::GPT; ++u) {
    int c, hi, xb;
    G::coords(tid, u, c, hi, xb);
#pragma unroll
    for (int e = 0; e < G::E; ++e) {
      const uint32_t gi = G::gidx(tile, xb | (e << G::K), c);
      st.v[u][e] = hm_addmod(g[gi], hm_shoup(ep.b[gi], ep.bk.w, ep.bk.ws, q), q);
    }
  }
}

// MODE 0 / 4: store as is (lazy values, hand-off between the two passes; 4 = first pass with the mix prologue);
// 1: forward final, reduce [0,8q) ->
// [0,q); 2: inverse final, multiply by the per-limb constant and reduce to [0,q); 3: forward final fused with
// out = (minuend - x) * k [+ addend]
template <int LOGR, bool STRIDED, int R, int MODE>
HM_HD void hm_ph_store_global(const HmNttState &st, int tid, uint64_t *g, uint32_t tile, uint64_t q, HmTw sc, HmEpi ep) {
  using G = HmRound<LOGR, STRIDED, R>;
#pragma unroll
  for (int u = 0; u < G::GPT; ++u) {
    int c, hi, xb;
    G::coords(tid, u, c, hi, xb);
    // MODE 3: the epilogue operands of one group are requested right before they are used (group by group, with a
    // scheduling fence in between): prefetching all of them before the last round cost 32 registers and a workgroup per CU
    uint64_t ea[G::E], ed[G::E];
    if (MODE == 3) {
#pragma unroll
      for (int e = 0; e < G::E; ++e) {
        const uint32_t gi = G::gidx(tile, xb | (e << G::K), c);
        ea[e] = ep.a[gi];
        ed[e] = ep.d ? ep.d[gi] : 0;
      }
    }
#pragma unroll
    for (int e = 0; e < G::E; ++e) {
      uint64_t a = st.v[u][e];
      if (MODE == 1) a = hm_reduce8(a, q);
      else if (MODE == 2) a = hm_shoup(a, sc.w, sc.ws, q);
      const uint32_t gi = G::gidx(tile, xb | (e << G::K), c);
      if (MODE == 3) {  // a in [0, 8q): minuend - a + 8q stays positive and below 2^64; the product reduces it
        const uint64_t va = ea[e], vd = ed[e];
        a = hm_shoup(va + 8 * q - a, sc.w, sc.ws, q);
        if (ep.d) a = hm_addmod(a, ep.dk.w ? hm_shoup(vd, ep.dk.w, ep.dk.ws, q) : vd, q);
      }
      g[gi] = a;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
#endif
  }
}

template <int LOGR, bool STRIDED, int R>
HM_HD void hm_ph_load_lds(HmNttState &st, int tid, const uint64_t *lds) {
  using G = HmRound<LOGR, STRIDED, R>;
#pragma unroll
  for (int u = 0; u < G::GPT; ++u) {
    int c, hi, xb;
    G::coords(tid, u, c, hi, xb);
#pragma unroll
    for (int e = 0; e < G::E; ++e) st.v[u][e] = lds[hm_lds_idx<LOGR, STRIDED>(xb | (e << G::K), c)];
  }
}
template <int LOGR, bool STRIDED, int R>
HM_HD void hm_ph_store_lds(const HmNttState &st, int tid, uint64_t *lds) {
  using G = HmRound<LOGR, STRIDED, R>;
#pragma unroll
  for (int u = 0; u < G::GPT; ++u) {
    int c, hi, xb;
    G::coords(tid, u, c, hi, xb);
#pragma unroll
    for (int e = 0; e < G::E; ++e) lds[hm_lds_idx<LOGR, STRIDED>(xb | (e << G::K), c)] = st.v[u][e];
  }
}

template <int LOGR, bool STRIDED, int R, bool INV>
HM_HD void hm_ph_compute(HmNttState &st, uint64_t q) {
  using G = HmRound<LOGR, STRIDED, R>;
  const HmBflyMod m = hm_bfly_mod(q);  // lazy ranges: forward [0, 8q), inverse [0, 4q)
#pragma unroll
  for (int u = 0; u < G::GPT; ++u) {
#pragma unroll
    for (int jj = 0; jj < G::NB; ++jj) {
      const int j = INV ? (G::NB - 1 - jj) : jj;  // sub-stage j combines e-bit (NB-1-j)
      const int pb = G::NB - 1 - j;
#pragma unroll
      for (int e = 0; e < G::E; ++e) {
        if (e & (1 << pb)) continue;
        const HmTw t = st.tw[R][u][(1 << j) - 1 + (e >> (G::NB - j))];
        if (INV) hm_bfly_inv(st.v[u][e], st.v[u][e | (1 << pb)], t, m);
        else     hm_bfly_fwd(st.v[u][e], st.v[u][e | (1 << pb)], t, m);
      }
    }
  }
}

// The whole pass of one thread, phase by phase.  `sync` is __syncthreads() on the GPU; the emulator calls the
// phases itself (see tests/emu/hm_emu.cpp) in the same order, one phase for all threads at a time.
// Phase list (forward; the inverse walks the rounds from the last to the first):
//   P0: tw[r0], global -> v, tw[r1]          P1: compute r0, v -> LDS            | barrier
//   P2: tw[r2] (3 rounds), LDS -> v, compute r1, then  (2 rounds) v -> global    or  v -> LDS | barrier
//   P3: LDS -> v, compute r2, v -> global
template <int LOGR, bool STRIDED, bool INV, int MODE, int PHASE>
HM_HD void hm_ntt_phase(HmNttState &st, int tid, uint64_t *lds, const uint64_t *src, uint64_t *dst, uint32_t tile,
                        const HmTw *twl, uint32_t s0, uint32_t prefix0, uint64_t q, HmTw sc, HmEpi ep) {
  using RS = HmRounds<LOGR>;
  constexpr int n = RS::n;
  constexpr int r0 = INV ? n - 1 : 0, r1 = INV ? n - 2 : 1, r2 = INV ? 0 : 2;  // r2 only when n == 3
  if (PHASE == 0) {
    hm_ph_load_tw<LOGR, STRIDED, r0>(st, tid, twl, s0, prefix0);
    if (MODE == 4) hm_ph_load_global_mix<LOGR, STRIDED, r0>(st, tid, src, tile, q, ep);
    else hm_ph_load_global<LOGR, STRIDED, r0>(st, tid, src, tile);
    hm_ph_load_tw<LOGR, STRIDED, r1>(st, tid, twl, s0, prefix0);
  } else if (PHASE == 1) {
    hm_ph_compute<LOGR, STRIDED, r0, INV>(st, q);
    hm_ph_store_lds<LOGR, STRIDED, r0>(st, tid, lds);
  } else if (PHASE == 2) {
    if (n == 3) hm_ph_load_tw<LOGR, STRIDED, (n == 3 ? r2 : r1)>(st, tid, twl, s0, prefix0);
    hm_ph_load_lds<LOGR, STRIDED, r1>(st, tid, lds);
    hm_ph_compute<LOGR, STRIDED, r1, INV>(st, q);
    if (n == 3) hm_ph_store_lds<LOGR, STRIDED, r1>(st, tid, lds);
    else hm_ph_store_global<LOGR, STRIDED, r1, MODE>(st, tid, dst, tile, q, sc, ep);
  } else if (PHASE == 3) {
    if (n == 3) {
      hm_ph_load_lds<LOGR, STRIDED, (n == 3 ? r2 : r1)>(st, tid, lds);
      hm_ph_compute<LOGR, STRIDED, (n == 3 ? r2 : r1), INV>(st, q);
      hm_ph_store_global<LOGR, STRIDED, (n == 3 ? r2 : r1), MODE>(st, tid, dst, tile, q, sc, ep);
    }
  }
}
