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
// access units of the fused epilogue's operands (minuend, addend: 16 bytes per lane each) requested together before they are used.  2 units =
// 4 loads in flight per lane beside the pass's own; 4 units spill in the 128-register ROW kernel (profiles/r03_knobs.txt)
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
  uint16_t pack;                     // MODE 2 (round 5): the output is stored in the split-30 packed form of the base conversions' inputs
  uint16_t pad;
  uint32_t galois;                   // MODE 6: the input, MODE 7: the addend is read through the automorphism X -> X^galois (1 = as stored)
  uint32_t pad1;
  HmTw sc;                           // inverse: N^-1 * extra scale; fused forward: the epilogue constant k
  HmTw ak;                           // fused forward: addend constant (w == 0: none)
  HmTw mixk;                         // MODE 4 prologue constant
};
#define HM_NTT_NONE 0xFFFFu
#define HM_NTT_MAX_ENTRIES 448       // records in the kernel-argument segment (8 bytes each, 4 KiB limit)

struct HmNttArgs {
  const uint64_t *in;
  uint64_t *out;
  const HmW *tw;       // [n_mod][N] forward or inverse table (chosen by the host), Montgomery form
  const HmW *twist;    // [n_mod][N / 256][3] per-row constants alpha_r^k (forward) or alpha_r^-k (inverse), k = 1..3
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
  uint32_t pack;             // MODE 2: store hm_pack30(value) (wave-uniform)
  uint32_t g, logN;          // MODE 6 / 7: the Galois element the input / the addend is gathered through, and the ring size its index map needs
};
HM_HD HmEpi hm_epi_none() { return HmEpi{nullptr, nullptr, HmTw{0, 0}, nullptr, HmTw{0, 0}, 0, 0, 0}; }
// Split-30 packed form (round 5): x < 2^60 stored as (x mod 2^30) | ((x >> 30) << 32) — the two 30-bit halves a base conversion multiplies
// with, one per dword.  The inverse transforms that feed ONLY base conversions (ModUp_DecompOut, ModDownBConvStep1: src/Operation.cpp:
// 104-135, 447-487) store this form (two instructions per value, once), and every conversion workgroup that reads the value (one per pair
// of output limbs: 18 readers per value in a 35-output ModUp digit) takes the halves as they are instead of shifting and masking again.
HM_HD uint64_t hm_pack30(uint64_t x) { return (x & 0x3FFFFFFFull) | ((x >> 30) << 32); }
HM_HD uint64_t hm_unpack30(uint64_t p) { return (p & 0x3FFFFFFFull) | ((p >> 32) << 30); }

// ---- K2 automorphism in evaluation form: out[i] = in[pi_g(i)] (bit-reversed NTT layout)
HM_HD uint32_t hm_brev(uint32_t x, uint32_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(x) >> (32 - bits);
#else
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
#endif
}
HM_HD uint32_t hm_auto_src(uint32_t i, uint32_t g, uint32_t logN) {
  uint32_t mask = (2u << logN) - 1;
  uint32_t e = (g * (2 * hm_brev(i, logN) + 1)) & mask;
  return hm_brev((e - 1) >> 1, logN);
}
// The map is affine in the natural index and both sides are stored bit-reversed, so every aligned block of 2^k outputs comes from ONE aligned
// block of 2^k inputs, for every k: a 4096-coefficient tile from one tile, a 16-byte access unit (outputs 2m, 2m + 1) from one aligned pair
// of inputs — in order, or swapped (hm_auto_src(2m + 1) = hm_auto_src(2m) ^ 1).  A transform can therefore read its input (MODE 6), or its
// epilogue's addend (MODE 7), THROUGH the automorphism with the same 16-byte loads it uses anyway: hrotate's automorphism launch
// (InsGen::GenAUTO, src/InsGen.cpp:46-71) folds into the ModUp INTT and the final add (round 6).

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

// the access unit whose first word is coefficient i (even) of limb-poly `g`, read through the automorphism X -> X^galois: one 16-byte load from
// the aligned pair that holds both sources, the words swapped when the pair arrives in the other order (hm_auto_src above)
template <int AUX = 0>
HM_HD void hm_gld2_auto(const uint64_t *g, uint32_t i, uint32_t galois, uint32_t logN, uint64_t &v0, uint64_t &v1) {
  const uint32_t s = hm_auto_src(i, galois, logN);
  uint64_t a, b;
#if defined(__HIP_DEVICE_COMPILE__)
  const hm_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(hm_rsrc(g), (int)((s & ~1u) << 3), 0, AUX);
  a = (uint64_t)t.x | ((uint64_t)t.y << 32);
  b = (uint64_t)t.z | ((uint64_t)t.w << 32);
#else
  hm_ld2(g + (s & ~1u), a, b);
#endif
  const bool swap = s & 1u;
  v0 = swap ? b : a;
  v1 = swap ? a : b;
}

// LDS image of a tile: word index of coefficient (x, c), x = position inside the sub-transform, c = which sub-transform.
// STRIDED: [x][c] with the C columns contiguous; CONTIG: [c][x].  Swizzles (bijections that keep word pairs together):
//   CONTIG (256-point rows): XORs of bits of x into bits 1..4 of the word index, every source bit above its target (so the map
//     inverts).  An image lives inside one pass, so each pass takes the set that suits its accesses (SWZ):
//       0  forward pass, 16 coefficients per thread: bits 2..4 ^= x[7:5], bit 1 ^= x[5] — the middle round reads 16-byte units at a
//          256-byte stride (x = hi*32 + e*4 + ..) and the last one 32 contiguous bytes per lane: both conflict-free;
//       1  inverse pass, 16 per thread: bit 1 ^= x[2], bit 2 ^= x[4] ^ x[5], bits 3, 4 ^= x[6], x[7].  The inverse pass WRITES the
//          32-bytes-per-lane round first, and 16-byte writes are banked differently from reads (8 groups of 8 lanes over 32
//          banks, against 4 groups of 16 over 64): under set 0 those writes were 2-way, 25 % of the pass's LDS cycles
//          (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of k_ntt_row<true, 0> in profiles/r05_pmc_*, r06_pmc_*);
//       2  8 per thread (four radix-4 rounds), both directions: bit 1 ^= x[2], bits 2, 3 ^= x[4], x[5], bit 4 ^= x[5] ^ x[7]; set 0
//          cost it 25 % (forward) and 36 % (inverse) extra LDS cycles.
//     Sets 1 and 2 come from an exhaustive search over such XOR sets against the chip's banking rules (tools/lds_banks.py, which
//     also reproduces the counters above): every read and write of every round is conflict-free.  (Set 1 serves the forward pass
//     as well, but costs it two more registers: the generic transform x key kernel then spills.)
//   STRIDED with 16 columns: row x swaps with its neighbour when x[2] is set (8 columns: two bits, 4 columns: three) —
//     the last round's rows are 4 apart (x = 4 xr + e), i.e. 512 bytes with 16 columns: without the swizzle two lanes
//     of a 16-lane group share a bank.
// The index is XOR-linear in x: idx(a | b, c) = idx(a, c) ^ idx(b, 0) for a & b == 0 (hm_lds_at relies on it).
template <int TL, int LOGR, bool STRIDED, int SWZ>
HM_HD int hm_lds_idx(int x, int c) {
  if (STRIDED) {
    constexpr int LOGC = TL - LOGR;
    int w = (x << LOGC) | c;
    if (LOGC <= 4) w ^= ((x >> 2) & ((1 << (LOGC <= 4 ? 5 - LOGC : 0)) - 1)) << LOGC;  // a row is 2^(LOGC+3) bytes; 256 bytes span all banks
    return w;
  }
  int w = (c << LOGR) | x;
  if (LOGR == 8) {
#if defined(HM_ABL_OLD_SWIZZLE)
    constexpr int S = 0;
#else
    constexpr int S = SWZ;
#endif
    if (S == 0) w ^= (((x >> 5) & 7) << 2) ^ (((x >> 5) & 1) << 1);
    if (S == 1) w ^= ((x >> 1) & 2) ^ ((((x >> 4) ^ (x >> 5)) & 1) << 2) ^ (((x >> 6) & 3) << 3);
    if (S == 2) w ^= ((x >> 1) & 2) ^ (((x >> 4) & 3) << 2) ^ ((((x >> 5) ^ (x >> 7)) & 1) << 4);
  }
  return w;
}

// ---------------------------------------------------------------------------------------------------
// The passes exist in two geometries (hm_ntt_passes.inl is included once per geometry, each in its own namespace):
//   hm16: 16 coefficients per thread, 256-thread workgroups, radix-8 rounds — the throughput geometry (large launches: four
//         workgroups per CU overlap each other's memory and compute phases);
//   hm8:   8 coefficients per thread, 512-thread workgroups on the same 4096-coefficient tiles, radix-4 rounds — twice the waves,
//         half the serial work per wave: for launches of up to ~128 limb-polys (a single op's stages, the 50-limb sweep of the
//         extended basis, the per-rank launches of a sharded run), which do not fill the chip with the wide geometry and are
//         bound by the latency of ONE workgroup's pass.  N = 2^16 (both passes of length 256) and, since round 6, N = 2^15 (a 128-point
//         COL pass: three radix-4 rounds + a radix-2 round).
// ---------------------------------------------------------------------------------------------------
#define HM_EPT 16                      // coefficients per thread
#define HM_UNITS (HM_EPT / 2)          // 16-byte access units per thread
struct HmNoPre { HM_HD void operator()() const {} };
namespace hm16 {
#include "hm_ntt_passes.inl"
}
#undef HM_EPT
#define HM_EPT 8
namespace hm8 {
#include "hm_ntt_passes.inl"
}
#undef HM_EPT
#define HM_EPT 16
using namespace hm16;   // unqualified names = the throughput geometry
