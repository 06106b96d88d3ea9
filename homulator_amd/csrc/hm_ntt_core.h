// hm_ntt_core.h — K1: negacyclic NTT / INTT over one RNS limb as two global passes of LDS-tiled
// radix-8/4 rounds.  Replaces the reference's NTTU timing model (src/Components.cpp:380-436:
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
// A workgroup of 2^TL / 16 threads owns a tile of 2^TL coefficients (TL = 12: 256 threads, 32 KiB) in LDS; a thread
// holds 16 of them: two (radix-8 round) or four (radix-4 round) butterfly groups.  The groups of a thread come in
// PAIRS that are adjacent in the memory-contiguous coordinate (two neighbouring columns in the COL pass, two
// neighbouring x2 in the ROW pass), so that
//   * every global and LDS access moves 16 bytes per lane (8-byte accesses run at about half the rate on gfx950),
//   * the two groups of a pair use the same twiddles: one twiddle fetch serves 16 coefficients.
// The LDS image is XOR-swizzled (no padding) so that the 16-byte accesses of all three rounds are conflict-free.
//
// Twiddles.  The table of a modulus is w[k] = psi^brev(k), k < N.  Stage s uses w[2^s + g], g < 2^s.
//   * COL pass: the 2^LOG1 - 1 twiddles w[1 .. 2^LOG1) are the same for every column.  They are staged through LDS
//     once per workgroup; only the first round (which runs before the first barrier) reads them from global memory.
//   * ROW pass, local stage sigma < 8 of row r (global stage LOG1 + sigma, butterfly block h < 2^sigma):
//         w[2^(LOG1+sigma) + r 2^sigma + h]  =  alpha_r^(2^(7-sigma)) * w[h],      alpha_r = psi^(1 + 2 brev(r)).
//     The second factor is shared by all rows; the first is a per-row constant of the stage.  The last two local
//     stages hold 75 % of a row's twiddles (64 + 128 of 255, 16 bytes each with the Shoup companion: as many bytes
//     as the data itself over a whole pass).  For them the per-row factor is applied to the DATA instead: element j
//     of the row is multiplied by alpha_r^(j mod 4) before the last forward round (three Shoup products per four
//     coefficients; `twist` table, 3 constants per row), after which the two stages are plain butterflies with the
//     shared twiddles w[h] (LDS-resident: the first 128 entries of the table).  The inverse pass runs its first
//     round with the shared inverse twiddles and multiplies by alpha_r^-(j mod 4) afterwards.  Bit-identical to the
//     table-driven transform: everything is exact arithmetic mod q.
#pragma once
#include "hm_modarith.h"

#define HM_EPT 16                      // coefficients per thread
#define HM_UNITS (HM_EPT / 2)          // 16-byte access units per thread
#define HM_MAX_THREADS 256
// Tile size per pass (log2 of the coefficients a workgroup owns; threads = tile / 16).  More, smaller workgroups per CU
// overlap better: a workgroup alternates between waiting for memory and computing, and with k of them resident a CU
// keeps both busy about k / (k + 1) of the time (measured: 3 x 512 threads 0.55 us per limb-NTT, 4 x 256 0.51).
// ROW tiles are contiguous in memory at any size; COL tiles of 2048 have 64-byte row segments.
#ifndef HM_TL_COL
#define HM_TL_COL 12
#endif
#ifndef HM_TL_ROW
#define HM_TL_ROW 12
#endif
#define HM_TL(STRIDED) ((STRIDED) ? HM_TL_COL : HM_TL_ROW)
#define HM_MAX_LIMBS 128
#ifndef HM_ROW_LOG
#define HM_ROW_LOG 8  // log2 of the contiguous sub-transform length (pass ROW)
#endif
// shared twiddles of the later rounds come from an LDS copy (0: from global memory), per pass
#ifndef HM_TW_LDS_COL
#define HM_TW_LDS_COL 1
#endif
#ifndef HM_TW_LDS_ROW
#define HM_TW_LDS_ROW 1
#endif
#define HM_TW_IN_LDS(STRIDED) ((STRIDED) ? HM_TW_LDS_COL : HM_TW_LDS_ROW)
#ifndef HM_EPI_CHUNK
#define HM_EPI_CHUNK 2
#endif
#ifndef HM_EPI_FENCE
#define HM_EPI_FENCE 1
#endif
#ifndef HM_LATE_TW1
#define HM_LATE_TW1 1  // inverse ROW pass: second round's twiddles requested after the first round's butterflies (see hm_ntt_phase)
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
  const HmTw *twist;   // [n_mod][N / 256][3] per-row constants alpha_r^k (forward) or alpha_r^-k (inverse), k = 1..3
  const HmMod *mods;   // [n_mod]
  const HmNttEntry *entry;                // [n_limbs], device
  const uint64_t *minuend, *addend, *mix; // bases of the MODE 3 / MODE 4 operands (addend may be null)
  uint32_t logN;
  uint32_t n_limbs;    // entries, a multiple of 8 G (groups x 8 XCDs, see hm_block_map)
  uint32_t logG;       // log2 of the same-modulus group size G (1 .. 3)
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

// 16-byte accesses (two adjacent words; p is 16-byte aligned)
HM_HD void hm_ld2(const uint64_t *p, uint64_t &a, uint64_t &b) {
#if defined(__HIP_DEVICE_COMPILE__)
  const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(p);
  a = t.x; b = t.y;
#else
  a = p[0]; b = p[1];
#endif
}
HM_HD void hm_st2(uint64_t *p, uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  ulonglong2 t; t.x = a; t.y = b;
  *reinterpret_cast<ulonglong2 *>(p) = t;
#else
  p[0] = a; p[1] = b;
#endif
}

// Global access of unit a of round G: buffer instructions take a wave-uniform descriptor (the limb-poly's base), a scalar
// offset (tile, element number) and ONE 32-bit lane offset per pair / group — no 64-bit address register pair per unit
// (flat addressing cost 16 VGPRs per operand stream and pushed the pipelined kernel into scratch).
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned hm_u32x4 __attribute__((ext_vector_type(4)));
HM_HD __amdgpu_buffer_rsrc_t hm_rsrc(const uint64_t *base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t *>(base), 0, -1, 0x00020000);
}
#endif
// 16 bytes at uniform base + 32-bit lane byte offset (buffer form on the device: one address VGPR for all limbs)
HM_HD void hm_bld2(const uint64_t *base, uint32_t lane_bytes, uint64_t &v0, uint64_t &v1) {
#if defined(__HIP_DEVICE_COMPILE__)
  const hm_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(hm_rsrc(base), lane_bytes, 0, 0);
  v0 = (uint64_t)t.x | ((uint64_t)t.y << 32);
  v1 = (uint64_t)t.z | ((uint64_t)t.w << 32);
#else
  hm_ld2(base + (lane_bytes >> 3), v0, v1);
#endif
}
HM_HD void hm_bst2(uint64_t *base, uint32_t lane_bytes, uint64_t v0, uint64_t v1) {
#if defined(__HIP_DEVICE_COMPILE__)
  hm_u32x4 t;
  t.x = (unsigned)v0; t.y = (unsigned)(v0 >> 32); t.z = (unsigned)v1; t.w = (unsigned)(v1 >> 32);
  __builtin_amdgcn_raw_buffer_store_b128(t, hm_rsrc(base), lane_bytes, 0, 0);  // soffset stays 0: see hm_gst2
#else
  hm_st2(base + (lane_bytes >> 3), v0, v1);
#endif
}
// AUX = cache-policy bits of the buffer instruction (gfx950: 1 = sc0, 2 = nt, 16 = sc1).  The second pass of the one-launch
// transform reads the first pass's hand-off with sc1 loads: they bypass this CU's vector L1 (never refreshed by another CU's
// stores) and are served by the XCD's L2, where the producing workgroups' plain stores left the lines.
template <class G, int AUX = 0>
HM_HD void hm_gld2(const uint64_t *g, uint32_t tile, int tid, int a, uint64_t &v0, uint64_t &v1) {
#if defined(__HIP_DEVICE_COMPILE__)
  const hm_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(hm_rsrc(g), G::gthr(tid, a) << 3, G::guni(tile, a) << 3, AUX);
  v0 = (uint64_t)t.x | ((uint64_t)t.y << 32);
  v1 = (uint64_t)t.z | ((uint64_t)t.w << 32);
#else
  hm_ld2(g + G::guni(tile, a) + G::gthr(tid, a), v0, v1);
#endif
}
template <class G, int AUX = 0>
HM_HD void hm_gst2(uint64_t *g, uint32_t tile, int tid, int a, uint64_t v0, uint64_t v1) {
#if defined(__HIP_DEVICE_COMPILE__)
  hm_u32x4 t;
  t.x = (unsigned)v0; t.y = (unsigned)(v0 >> 32); t.z = (unsigned)v1; t.w = (unsigned)(v1 >> 32);
  // NO scalar offset on stores: a 16-byte buffer store with an SGPR soffset, followed at once by a VALU write to its data
  // registers, stored the new values for the lanes read last on gfx950 (hipcc only guards the immediate-soffset form of
  // this hazard: seen as 16 wrong coefficients in random tiles, different ones every run).  The uniform part goes into
  // the descriptor's base instead (two scalar adds).
  __builtin_amdgcn_raw_buffer_store_b128(t, hm_rsrc(g + G::guni(tile, a)), G::gthr(tid, a) << 3, 0, AUX);
#else
  hm_st2(g + G::guni(tile, a) + G::gthr(tid, a), v0, v1);
#endif
}

// LDS image of a tile: word index of coefficient (x, c), x = position inside the sub-transform, c = which sub-transform.
// STRIDED: [x][c] with the C columns contiguous; CONTIG: [c][x].  Swizzles (bijections that keep word pairs together):
//   CONTIG (256-point rows): bits 2..4 ^= x[7:5], bit 1 ^= x[5] — the middle round reads 16-byte units at a 256-byte
//     stride (x = hi*32 + e*4 + ..) and the last one 32 contiguous bytes per lane: both become conflict-free;
//   STRIDED with 16 columns: row x swaps with its neighbour when x[2] is set (8 columns: two bits, 4 columns: three) —
//     the last round's rows are 4 apart (x = 4 xr + e), i.e. 512 bytes with 16 columns: without the swizzle two lanes
//     of a 16-lane group share a bank.
template <int TL, int LOGR, bool STRIDED>
HM_HD int hm_lds_idx(int x, int c) {
  if (STRIDED) {
    constexpr int LOGC = TL - LOGR;
    int w = (x << LOGC) | c;
    if (LOGC <= 4) w ^= ((x >> 2) & ((1 << (5 - LOGC)) - 1)) << LOGC;  // a row is 2^(LOGC+3) bytes; 256 bytes span all banks
    return w;
  }
  int w = (c << LOGR) | x;
  if (LOGR == 8) w ^= (((x >> 5) & 7) << 2) ^ (((x >> 5) & 1) << 1);
  return w;
}
// LDS words of a pass: the tile, then the staged shared twiddles (2 words each)
template <int TL, int LOGR, bool STRIDED>
struct HmLds {
  static constexpr int TILE = 1 << TL, THREADS = TILE / HM_EPT;
  static constexpr int NTW = STRIDED ? (1 << LOGR) : 128;  // staged entries: w[0 .. NTW) of the modulus
  static constexpr int WORDS = TILE + (HM_TW_IN_LDS(STRIDED) ? 2 * NTW : 0);
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
  HmTw tw[3][HM_MAX_TW];
  HmTw tws[3];  // twist constants of the thread's row (ROW pass)
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

// Forward transform: which kind of butterfly (hm_bfly_fwd_k) local stage sigma of a pass runs.  Bounds in units of q:
// the COL pass starts from reduced data (1), its first two stages need no subtraction (5, 9); from there on the stages
// alternate so that the pass ENDS on a subtracting stage (12 out); the ROW pass (8 stages, 12 in) alternates 0 / 1 and
// ends with kind 2 (8 out).  hm_fwd_bound replays the bounds at compile time: every stage is checked below.
constexpr int hm_fwd_kind(bool strided, int logr, int sigma) {
  if (strided) return sigma <= 1 ? 0 : (((logr - 1 - sigma) & 1) == 0 ? 1 : 0);
  return sigma == logr - 1 ? 2 : (sigma & 1);
}
constexpr int hm_fwd_bound(bool strided, int logr, int upto) {  // bound (in q) of the values entering local stage `upto`
  int b = strided ? 1 : 12;
  for (int s = 0; s < upto; ++s) {
    const int k = hm_fwd_kind(strided, logr, s);
    if (k == 0 ? b > 12 : b > 16) return 1000;                  // the stage's input condition
    b = (k == 0 ? b : k == 1 ? 8 : 4) + 4;
  }
  return b;
}
static_assert(hm_fwd_bound(true, 5, 5) == 12 && hm_fwd_bound(true, 6, 6) == 12 && hm_fwd_bound(true, 7, 7) == 12 &&
              hm_fwd_bound(true, 8, 8) == 12 && hm_fwd_bound(true, 9, 9) == 12, "COL pass hands over values below 12q");
static_assert(hm_fwd_bound(false, 8, 8) == 8, "ROW pass ends below 8q");

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
  static constexpr bool anyLds() { return fromLds(exec(1)) || (n == 3 && fromLds(exec(2))); }
};

// request the twiddles of round R: sub-stage j needs 2^j of them, indexed by the top j bits of e.
// SHARED ROW round: the row-independent factor w[h]; otherwise the full table entry.
template <int TL, int LOGR, bool STRIDED, int R, bool SHARED>
HM_HD void hm_ph_load_tw(HmNttState &st, int tid, const HmTw *twl, uint32_t s0, uint32_t prefix0) {
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
        st.tw[R][v * (G::E - 1) + (1 << j) - 1 + t] = HmTw{(uint64_t)(twbase + t) * 0x9E3779B97F4A7C15ull >> 5, (uint64_t)(twbase + t) * 0xD1342543DE82EF95ull};
#else
        st.tw[R][v * (G::E - 1) + (1 << j) - 1 + t] = twl[twbase + (uint32_t)t];
#endif
      }
    }
  }
}
// the three twist constants of the thread's row (the K == 0 ROW round: all groups of a thread lie in one row)
template <int TL, int LOGR, bool STRIDED, int R>
HM_HD void hm_ph_load_twist(HmNttState &st, int tid, const HmTw *twist_tile) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
  int c, xr;
  G::group(tid, 0, c, xr);
#pragma unroll
  for (int k = 0; k < 3; ++k) st.tws[k] = twist_tile[c * 3 + k];
}
// copy the shared twiddles w[0 .. NTW) of the modulus into LDS (visible after the next barrier)
template <int TL, int LOGR, bool STRIDED>
HM_HD void hm_ph_stage_tw(int tid, uint64_t *lds, const HmTw *twl) {
  using LD = HmLds<TL, LOGR, STRIDED>;
#pragma unroll
  for (int i = 0; i < (LD::NTW + LD::THREADS - 1) / LD::THREADS; ++i) {
    const int k = tid + LD::THREADS * i;
    if (k < LD::NTW) {
      const HmTw t = twl[k];
      hm_st2(lds + LD::TILE + 2 * k, t.w, t.ws);
    }
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
    st.v[i0] = hm_addmod(p0, hm_shoup(b0, ep.bk.w, ep.bk.ws, q), q);
    st.v[i1] = hm_addmod(p1, hm_shoup(b1, ep.bk.w, ep.bk.ws, q), q);
  }
}

// MODE 5 (ROW pass of the fused transform x key inner product): the last pass keeps its results in registers (lazy, below 8q)
// and stores nothing; hm_ph_mac consumes them.
// the epilogue of one coefficient.  MODE 0 / 4: store as is (lazy values, hand-off between the two passes; 4 = first
// pass with the mix prologue); 1: forward final, reduce [0,8q) -> [0,q); 2: inverse final, multiply by the per-limb
// constant and reduce to [0,q); 3: forward final fused with out = (minuend - x) * k [+ addend [* ak]]
template <int MODE>
HM_HD uint64_t hm_epilogue(uint64_t a, uint64_t va, uint64_t vd, uint64_t q, HmTw sc, const HmEpi &ep) {
  if (MODE == 1) return hm_reduce8(a, q);
  if (MODE == 2) return hm_shoup(a, sc.w, sc.ws, q);
  if (MODE == 3) {  // a in [0, 8q): minuend - a + 8q stays positive and below 2^64; the product reduces it
    a = hm_shoup(va + 8 * q - a, sc.w, sc.ws, q);
    if (ep.d) a = hm_addmod(a, ep.dk.w ? hm_shoup(vd, ep.dk.w, ep.dk.ws, q) : vd, q);
  }
  return a;
}
template <int TL, int LOGR, bool STRIDED, int R, int MODE, int AUX = 0, int CH = HM_EPI_CHUNK>
HM_HD void hm_ph_store_global(const HmNttState &st, int tid, uint64_t *g, uint32_t tile, uint64_t q, HmTw sc, HmEpi ep) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
  // MODE 3: the epilogue operands are requested two units at a time, right before they are used (with a scheduling
  // fence in between): prefetching all of them before the last round costs 32 registers
  // CH = units per chunk: 2 in the two-kernel transform, 1 inside the one-launch transform (whose second pass has less room)
#pragma unroll
  for (int a2 = 0; a2 < HM_UNITS; a2 += CH) {
    uint64_t ea[4] = {0, 0, 0, 0}, ed[4] = {0, 0, 0, 0};
    if (MODE == 3) {
#pragma unroll
      for (int k = 0; k < CH; ++k) {
        int i0, i1, x, c;
        G::unit(tid, a2 + k, i0, i1, x, c);
        hm_gld2<G>(ep.a, tile, tid, a2 + k, ea[2 * k], ea[2 * k + 1]);
        if (ep.d) hm_gld2<G>(ep.d, tile, tid, a2 + k, ed[2 * k], ed[2 * k + 1]);
      }
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      int i0, i1, x, c;
      G::unit(tid, a2 + k, i0, i1, x, c);
#if defined(HM_ABL_NOMEM)
      if (st.v[i0] == 0x123456789ull)   // never true in practice: keeps the values alive without the store traffic
#endif
      hm_gst2<G, AUX>(g, tile, tid, a2 + k, hm_epilogue<MODE>(st.v[i0], ea[2 * k], ed[2 * k], q, sc, ep),
                 hm_epilogue<MODE>(st.v[i1], ea[2 * k + 1], ed[2 * k + 1], q, sc, ep));
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
#endif
  }
}

template <int TL, int LOGR, bool STRIDED, int R>
HM_HD void hm_ph_load_lds(HmNttState &st, int tid, const uint64_t *lds) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
#pragma unroll
  for (int a = 0; a < HM_UNITS; ++a) {
    int i0, i1, x, c;
    G::unit(tid, a, i0, i1, x, c);
    hm_ld2(lds + hm_lds_idx<TL, LOGR, STRIDED>(x, c), st.v[i0], st.v[i1]);
  }
}
template <int TL, int LOGR, bool STRIDED, int R>
HM_HD void hm_ph_store_lds(const HmNttState &st, int tid, uint64_t *lds) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
#pragma unroll
  for (int a = 0; a < HM_UNITS; ++a) {
    int i0, i1, x, c;
    G::unit(tid, a, i0, i1, x, c);
    hm_st2(lds + hm_lds_idx<TL, LOGR, STRIDED>(x, c), st.v[i0], st.v[i1]);
  }
}

template <int TL, int LOGR, bool STRIDED, int R, bool INV>
HM_HD void hm_ph_compute(HmNttState &st, uint64_t q) {
  using G = HmRound<TL, LOGR, STRIDED, R>;
#if defined(HM_ABL_NOCOMPUTE)
  st.v[0] ^= st.tw[R][0].w;  // keeps the twiddle loads alive
  return;
#endif
  const HmBflyMod m = hm_bfly_mod(q);  // lazy ranges: forward [0, 8q), inverse [0, 4q)
#pragma unroll
  for (int jj = 0; jj < G::NB; ++jj) {
    const int j = INV ? (G::NB - 1 - jj) : jj;  // sub-stage j combines e-bit (NB-1-j)
    const int pb = G::NB - 1 - j;
#pragma unroll
    for (int u = 0; u < G::NG; ++u) {
#pragma unroll
      for (int e = 0; e < G::E; ++e) {
        if (e & (1 << pb)) continue;
        const HmTw t = st.tw[R][G::twslot(u) + (1 << j) - 1 + (e >> (G::NB - j))];
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
// Lazy product: any 64-bit input, result in [0, 4q) — inside the input range of both butterfly forms.
HM_HD void hm_ph_twist(HmNttState &st, uint64_t q) {
#if defined(HM_ABL_NOCOMPUTE)
  st.v[1] ^= st.tws[0].w ^ st.tws[1].w ^ st.tws[2].w;
  return;
#endif
  const HmBflyMod m = hm_bfly_mod(q);
#pragma unroll
  for (int u = 0; u < HM_EPT / 4; ++u)
#pragma unroll
    for (int k = 1; k < 4; ++k) st.v[4 * u + k] = hm_shoup_lazy4_acc(0, st.v[4 * u + k], st.tws[k - 1], m);
}

// The whole pass of one thread, phase by phase.  `sync` is __syncthreads() on the GPU; the emulator calls the
// phases itself (see tests/emu/hm_emu.cpp) in the same order, one phase for all threads at a time.
// Phase list (forward; the inverse walks the rounds from the last to the first):
//   P0: stage shared tw -> LDS, tw[r0], global -> v, tw[r1] (if global)   P1: compute r0, v -> LDS        | barrier
//   P2: tw[r2] (if global), tw[r1] (if LDS), LDS -> v, compute r1, then (2 rounds) v -> global  or  v -> LDS | barrier
//   P3: tw[r2] (if LDS), LDS -> v, compute r2, v -> global
// twl = table of the modulus; twist_tile = twist constants of the tile's first row (ROW pass)
// LDAUX / STAUX: cache-policy bits of the pass's data loads / stores (hm_gld2)
template <int TL, int LOGR, bool STRIDED, bool INV, int MODE, int PHASE, int LDAUX = 0, int STAUX = 0, int EPICH = HM_EPI_CHUNK>
HM_HD void hm_ntt_phase(HmNttState &st, int tid, uint64_t *lds, const uint64_t *src, uint64_t *dst, uint32_t tile,
                        const HmTw *twl, const HmTw *twist_tile, uint32_t s0, uint32_t prefix0, uint64_t q, HmTw sc, HmEpi ep) {
  using PS = HmPass<LOGR, STRIDED, INV>;
  constexpr int n = PS::n;
  constexpr int r0 = PS::exec(0), r1 = PS::exec(1), r2 = n == 3 ? PS::exec(2) : PS::exec(1);  // r2 only when n == 3
  constexpr int TWR = PS::twistRound;
  const HmTw *ltw = reinterpret_cast<const HmTw *>(lds + (1 << TL));
  // register pressure: the inverse ROW pass has 120 registers of loads in flight in its first phase (data, the first
  // round's shared twiddles, the twist constants and the NEXT round's private twiddles); inside the one-launch transform
  // that no longer fits 128.  The next round's twiddles are then requested after the first round's butterflies instead
  // (they still have the LDS exchange and the barrier to arrive in).
  constexpr bool LATE_TW1 = ((INV && HM_LATE_TW1) || MODE == 5) && !STRIDED;   // MODE 5: 64 accumulator registers are live beside the pass
  if (PHASE == 0) {
    if (PS::anyLds()) hm_ph_stage_tw<TL, LOGR, STRIDED>(tid, lds, twl);
    hm_ph_load_tw<TL, LOGR, STRIDED, r0, PS::shared(r0)>(st, tid, twl, s0, prefix0);
    if (MODE == 4) hm_ph_load_global_mix<TL, LOGR, STRIDED, r0, LDAUX>(st, tid, src, tile, q, ep);
    else hm_ph_load_global<TL, LOGR, STRIDED, r0, LDAUX>(st, tid, src, tile);
    if (TWR >= 0 && (INV || n == 2)) hm_ph_load_twist<TL, LOGR, STRIDED, (TWR >= 0 ? TWR : 0)>(st, tid, twist_tile);
    if (!PS::fromLds(r1) && !LATE_TW1) hm_ph_load_tw<TL, LOGR, STRIDED, r1, PS::shared(r1)>(st, tid, twl, s0, prefix0);
  } else if (PHASE == 1) {
    hm_ph_compute<TL, LOGR, STRIDED, r0, INV>(st, q);
    if (INV && r0 == TWR) hm_ph_twist(st, q);
    if (!PS::fromLds(r1) && LATE_TW1) {
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_sched_barrier(0);
#endif
      hm_ph_load_tw<TL, LOGR, STRIDED, r1, PS::shared(r1)>(st, tid, twl, s0, prefix0);
    }
    hm_ph_store_lds<TL, LOGR, STRIDED, r0>(st, tid, lds);
  } else if (PHASE == 2) {
    if (n == 3 && !PS::fromLds(r2)) hm_ph_load_tw<TL, LOGR, STRIDED, r2, PS::shared(r2)>(st, tid, twl, s0, prefix0);
    if (TWR >= 0 && !INV && n == 3) hm_ph_load_twist<TL, LOGR, STRIDED, (TWR >= 0 ? TWR : 0)>(st, tid, twist_tile);  // used in the next phase
    if (PS::fromLds(r1)) hm_ph_load_tw<TL, LOGR, STRIDED, r1, PS::shared(r1)>(st, tid, ltw, s0, prefix0);
    hm_ph_load_lds<TL, LOGR, STRIDED, r1>(st, tid, lds);
    if (!INV && n == 2 && r1 == TWR) hm_ph_twist(st, q);
    hm_ph_compute<TL, LOGR, STRIDED, r1, INV>(st, q);
    if (n == 3) hm_ph_store_lds<TL, LOGR, STRIDED, r1>(st, tid, lds);
    else hm_ph_store_global<TL, LOGR, STRIDED, r1, MODE, STAUX, EPICH>(st, tid, dst, tile, q, sc, ep);
  } else if (PHASE == 3) {
    if (n == 3) {
      // fused epilogue (MODE 3): the twist constants are dead before the last round's twiddles are read from LDS (the
      // other order keeps 12 more registers alive and spilled inside the one-launch transform)
      constexpr bool TWIST_FIRST = (MODE == 3 || MODE == 5) && !INV && r2 == TWR && PS::fromLds(r2);
      if (PS::fromLds(r2) && !TWIST_FIRST) hm_ph_load_tw<TL, LOGR, STRIDED, r2, PS::shared(r2)>(st, tid, ltw, s0, prefix0);
      hm_ph_load_lds<TL, LOGR, STRIDED, r2>(st, tid, lds);
      if (!INV && r2 == TWR) hm_ph_twist(st, q);
      if (TWIST_FIRST) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_sched_barrier(0);
#endif
        hm_ph_load_tw<TL, LOGR, STRIDED, r2, PS::shared(r2)>(st, tid, ltw, s0, prefix0);
      }
      hm_ph_compute<TL, LOGR, STRIDED, r2, INV>(st, q);
      if (MODE == 5) return;
#if defined(__HIP_DEVICE_COMPILE__)
      if (TWIST_FIRST && HM_EPI_FENCE) __builtin_amdgcn_sched_barrier(0);   // the epilogue's operand loads start after the last butterflies
#endif
      hm_ph_store_global<TL, LOGR, STRIDED, r2, MODE, STAUX, EPICH>(st, tid, dst, tile, q, sc, ep);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// K1 x K5: the transform's last pass multiplied into the evaluation key (the reference's HPIP unit as a fused NTT-epilogue x
// evk MAC: HPIP src/Components.cpp:571-668, InsGen::GenHPIP src/InsGen.cpp:356-406, KeySwitch::InnerProduceOperation
// src/Operation.cpp:294-414).  After the ROW pass of digit j (MODE 5) a thread holds 16 coefficients of ext_j in the layout
// of the pass's last round; hm_ph_mac adds ext_j * evk_{j,k} for both keys to the thread's accumulators, which live in
// registers across the digits: the extended digit never exists in HBM.  Accumulators are lazy: every product is reduced
// to [0, 3q) (hm_barrett_lazy; x may be any value below 8q), up to 4 terms stay below 12q < 2^64; hm_ph_mac_store reduces once.
// ---------------------------------------------------------------------------------------------------
// Accumulator forms: uint64_t — every product reduced to [0, 3q) before it is added (about 40 instructions per product, 64
// registers for both keys); hm_u128 — the raw 128-bit products are summed (12 instructions per product: x < 2q and y < q keep 4
// terms below 2^123 once x is brought below 2q) and reduced once per output by hm_barrett, at the price of 128 accumulator
// registers (two waves per SIMD).
HM_HD void hm_mac_add(uint64_t &acc, uint64_t x, uint64_t y, const HmMod &m) { acc += hm_barrett_lazy((hm_u128)x * y, m); }
HM_HD void hm_mac_add(hm_u128 &acc, uint64_t x, uint64_t y, const HmMod &) { acc += (hm_u128)x * y; }
HM_HD uint64_t hm_mac_final(uint64_t acc, const HmMod &m) { return hm_reduce16(acc, m.q); }
HM_HD uint64_t hm_mac_final(hm_u128 acc, const HmMod &m) { return hm_barrett(acc, m); }   // 4 terms x (x < 2q) x (y < q) < 2^123
// [0, 8q) -> [0, 2q): the wide accumulators take transform outputs below 2q, so that the sum stays inside hm_barrett's range
HM_HD void hm_ph_below_2q(HmNttState &st, uint64_t q) {
  const HmBflyMod m = hm_bfly_mod(q);
#pragma unroll
  for (int i = 0; i < HM_EPT; ++i) st.v[i] = hm_csub_neg(hm_csub_neg(st.v[i], m.nq4), m.z - 2 * q);
}
template <int TL, int LOGR, int R, int OUTS, int CH = 2, class ACC = uint64_t>
HM_HD void hm_ph_mac(const HmNttState &st, ACC (&acc)[OUTS][HM_EPT], int tid, const uint64_t *const (&y)[OUTS], uint32_t tile, const HmMod &m) {
  using G = HmRound<TL, LOGR, false, R>;
#pragma unroll
  for (int a2 = 0; a2 < HM_UNITS; a2 += CH) {
    uint64_t e[OUTS][2 * CH];
#pragma unroll
    for (int k = 0; k < OUTS; ++k)
#pragma unroll
      for (int c = 0; c < CH; ++c) hm_gld2<G>(y[k], tile, tid, a2 + c, e[k][2 * c], e[k][2 * c + 1]);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      int i0, i1, x, cc;
      G::unit(tid, a2 + c, i0, i1, x, cc);
#pragma unroll
      for (int k = 0; k < OUTS; ++k) {
        hm_mac_add(acc[k][i0], st.v[i0], e[k][2 * c], m);
        hm_mac_add(acc[k][i1], st.v[i1], e[k][2 * c + 1], m);
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
HM_HD void hm_ph_mac_regs(const HmNttState &st, ACC (&acc)[OUTS][HM_EPT], const uint64_t (&e)[OUTS][HM_EPT], const HmMod &m) {
#pragma unroll
  for (int k = 0; k < OUTS; ++k)
#pragma unroll
    for (int i = 0; i < HM_EPT; ++i) hm_mac_add(acc[k][i], st.v[i], e[k][i], m);
}
template <int TL, int LOGR, int R, int OUTS, class ACC>
HM_HD void hm_ph_mac_store(const ACC (&acc)[OUTS][HM_EPT], int tid, uint64_t *const (&out)[OUTS], uint32_t tile, const HmMod &m) {
  using G = HmRound<TL, LOGR, false, R>;
#pragma unroll
  for (int k = 0; k < OUTS; ++k)
#pragma unroll
    for (int a = 0; a < HM_UNITS; ++a) {
      int i0, i1, x, c;
      G::unit(tid, a, i0, i1, x, c);
      hm_gst2<G>(out[k], tile, tid, a, hm_mac_final(acc[k][i0], m), hm_mac_final(acc[k][i1], m));
    }
}
