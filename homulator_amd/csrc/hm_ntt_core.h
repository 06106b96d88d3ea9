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
// A workgroup of 256 threads owns a tile of HM_TILE = 4096 coefficients (32 KiB) in LDS and runs
// the rounds LDS -> registers -> LDS with one barrier per round; the phase functions below carry no
// register state across barriers, so the host emulator (tests/emu) can run them thread by thread.
#pragma once
#include "hm_modarith.h"

#define HM_TILE 4096
#define HM_TILE_LOG 12
#define HM_THREADS 256
#define HM_MAX_LIMBS 128

struct HmLimb {  // one limb-poly of a launch: limb indices into the in/out bases, modulus id
  uint16_t in, out, mod, aux;
};

struct HmNttArgs {
  const uint64_t *in;
  uint64_t *out;
  const HmTw *tw;      // [n_mod][N] forward or inverse table (chosen by the host)
  const HmMod *mods;   // [n_mod]
  uint32_t logN;
  uint32_t n_limbs;
  HmLimb limb[HM_MAX_LIMBS];
};
struct HmScale {  // per-limb epilogue constant of the inverse transform: c = N^-1 * extra, Shoup form
  HmTw c[HM_MAX_LIMBS];
};

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
    return (x << 8) + (tile << LOGC) + c;
  }
  return (tile << HM_TILE_LOG) + lin;
}
template <int LOGR, bool STRIDED>
HM_HD int hm_tile_lidx(uint32_t lin) {
  if (STRIDED) return (int)lin;
  return (int)(lin + ((lin >> 5) << 2));
}

// ---- phase: global -> LDS, 16 B per lane
template <int LOGR, bool STRIDED>
HM_HD void hm_tile_load(int tid, uint64_t *lds, const uint64_t *g, uint32_t tile) {
#pragma unroll
  for (int it = 0; it < HM_TILE / 2 / HM_THREADS; ++it) {
    uint32_t lin = 2u * (uint32_t)(it * HM_THREADS + tid);
    uint32_t gi = hm_tile_gidx<LOGR, STRIDED>(tile, lin);
    int li = hm_tile_lidx<LOGR, STRIDED>(lin);
    uint64_t a = g[gi], b = g[gi + 1];
    lds[li] = a;
    lds[li + 1] = b;
  }
}

// ---- phase: LDS -> global with the pass epilogue
// MODE 0: store as is (lazy values, internal hand-off between the two passes)
// MODE 1: forward final: reduce [0,4q) -> [0,q)
// MODE 2: inverse final: multiply by the per-limb constant, reduce to [0,q)
template <int LOGR, bool STRIDED, int MODE>
HM_HD void hm_tile_store(int tid, const uint64_t *lds, uint64_t *g, uint32_t tile, uint64_t q, HmTw sc) {
#pragma unroll
  for (int it = 0; it < HM_TILE / 2 / HM_THREADS; ++it) {
    uint32_t lin = 2u * (uint32_t)(it * HM_THREADS + tid);
    uint32_t gi = hm_tile_gidx<LOGR, STRIDED>(tile, lin);
    int li = hm_tile_lidx<LOGR, STRIDED>(lin);
    uint64_t a = lds[li], b = lds[li + 1];
    if (MODE == 1) {
      a = hm_csub(hm_csub(a, 2 * q), q);
      b = hm_csub(hm_csub(b, 2 * q), q);
    } else if (MODE == 2) {
      a = hm_shoup(a, sc.w, sc.ws, q);
      b = hm_shoup(b, sc.w, sc.ws, q);
    }
    g[gi] = a;
    g[gi + 1] = b;
  }
}

// ---- phase: one round = NB butterfly stages on local bits [K, K+NB) of x, in registers.
// twl = twiddle table of this limb; s0 = first global stage of the pass (0 for COL, LOG1 for ROW);
// prefix0 = global row index of tile column 0 (ROW pass) or 0 (COL pass: twiddles do not depend on c).
template <int LOGR, bool STRIDED, int NB, int K, bool INV>
HM_HD void hm_ntt_round(int tid, uint64_t *lds, const HmTw *twl, uint32_t s0, uint32_t prefix0, uint64_t q) {
  constexpr int R = 1 << LOGR, LOGC = HM_TILE_LOG - LOGR, C = 1 << LOGC;
  constexpr int E = 1 << NB;
  constexpr int GROUPS = HM_TILE >> NB;
  constexpr int XR = R >> NB;  // groups per sub-transform
  const uint64_t q2 = 2 * q;
#pragma unroll
  for (int u = 0; u < GROUPS / HM_THREADS; ++u) {
    int gid = tid + HM_THREADS * u;
    int c, xr;
    if (STRIDED) { c = gid & (C - 1); xr = gid >> LOGC; }
    else         { xr = gid & (XR - 1); c = gid / XR; }
    int lo = xr & ((1 << K) - 1), hi = xr >> K;
    int xb = (hi << (K + NB)) | lo;
    uint64_t v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = lds[hm_lds_idx<LOGR, STRIDED>(xb | (e << K), c)];
    uint32_t prefix = STRIDED ? 0u : (prefix0 + (uint32_t)c);
#pragma unroll
    for (int jj = 0; jj < NB; ++jj) {
      const int j = INV ? (NB - 1 - jj) : jj;   // sub-stage: combines e-bit (NB-1-j)
      const int sigma = LOGR - K - NB + j;      // local stage index
      const uint32_t twbase = (1u << (s0 + sigma)) + (prefix << sigma) + ((uint32_t)hi << j);
      const int pb = NB - 1 - j;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        if (e & (1 << pb)) continue;
        HmTw t = twl[twbase + (uint32_t)(e >> (NB - j))];
        if (INV) hm_bfly_inv(v[e], v[e | (1 << pb)], t, q, q2);
        else     hm_bfly_fwd(v[e], v[e | (1 << pb)], t, q, q2);
      }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) lds[hm_lds_idx<LOGR, STRIDED>(xb | (e << K), c)] = v[e];
  }
}

// Round schedule per sub-transform length: bits are consumed from the top for the forward
// transform (K descending) and from the bottom for the inverse.  {NB, K} lists, forward order.
template <int LOGR> struct HmRounds;
template <> struct HmRounds<5> { static constexpr int n = 2; static constexpr int nb[3] = {3, 2, 0}; static constexpr int k[3] = {2, 0, 0}; };
template <> struct HmRounds<6> { static constexpr int n = 2; static constexpr int nb[3] = {3, 3, 0}; static constexpr int k[3] = {3, 0, 0}; };
template <> struct HmRounds<7> { static constexpr int n = 3; static constexpr int nb[3] = {3, 2, 2}; static constexpr int k[3] = {4, 2, 0}; };
template <> struct HmRounds<8> { static constexpr int n = 3; static constexpr int nb[3] = {3, 3, 2}; static constexpr int k[3] = {5, 2, 0}; };
template <> struct HmRounds<9> { static constexpr int n = 3; static constexpr int nb[3] = {3, 3, 3}; static constexpr int k[3] = {6, 3, 0}; };

#if defined(__HIPCC__)
#define HM_SYNC() __syncthreads()
#else
#define HM_SYNC()
#endif
