// hm_bcol.h — base conversion fused into the first pass of the transform that consumes it: kernel body, launch records and the
// kernel table.  The instantiations (input-basis size x ring size x mix prologue) are compiled in several translation units
// (hm_bcol_part.hip with -DHM_BCOL_PART=k, in parallel); hm_backend.hip looks kernels up through hm_bcol_kernel_for().
#pragma once
#include "hm_elem_core.h"
#include "hm_ntt_core.h"
#include "hm_caps.h"   // HM_BCOL_MAX_IN, HM_BCOL_MAX_IN_MIX
#ifndef HM_NTT_MIN_WAVES
#define HM_NTT_MIN_WAVES 4
#endif

// ---- base conversion fused into the first pass of the transform that consumes it (round 3) --------------------------------
// The COL workgroup of (conversion p, output limb o, column tile t) computes its 4096 coefficients of output o from the N_IN input
// tiles (the workgroups of the same (p, t) for the other outputs sit in neighbouring dispatch slots of one XCD and find the inputs in
// L2), then runs the COL pass on them: the converted limb-poly never exists in HBM, only the first pass's hand-off does.
template <int V> struct HmIc { static constexpr int value = V; };
#define HM_BCOL_ONE_GROUP 15    // up to here every input of an access unit is held at once (16: hipcc leaves the arrays in scratch, 1 KB per lane)
struct HmBcolProb {
  const uint64_t *in;
  const uint64_t *table, *qn;
  uint32_t n_in, n_out;
  uint32_t in_limb[HM_BCONV_MAX_IN];
  uint32_t out_limb[HM_BCONV_MAX_OUT];   // where the hand-off of output o goes (limb of `out`)
  uint32_t out_mod[HM_BCONV_MAX_OUT];    // its modulus id (shared twiddles)
  uint32_t mix_limb[HM_BCONV_MAX_OUT];   // MIX: the operand added to output o before the transform (limb of HmBcolArgs::mix), constant in mixk
  const HmTw *mixk;                      // MIX: device, [n_out]
  uint32_t in_packed;                    // the inputs are stored in the split-30 packed form (hm_pack30): no shift / mask per input and workgroup
  // round 6 (scalar-register diet): ONE buffer descriptor for all inputs of a conversion — base = the lowest input limb-poly, input i at
  // byte offset in_off[i] from it (a scalar operand of the load).  A descriptor per input (four scalar registers each: 60 for a 15-limb
  // digit, 112 for 28) was hoisted out of the unit loop together with the table rows and spilled into vector-register lanes: 390
  // v_readlane / v_writelane in 5 531 vector instructions of k_bconv_col2<15>, 1 444 in 9 045 of k_bconv_col2<28>.  The host checks that
  // the inputs of a conversion lie within 4 GiB of each other (the digits of a plan are neighbours in the pool); a conversion whose inputs
  // are further apart runs as conversion + first pass (bconv_col_launch).
  const uint64_t *in_base;
  uint32_t in_off[HM_BCONV_MAX_IN];
};
// the table pointer made opaque once per access unit: the rows of an output are then requested again for every unit (scalar loads that hit
// the scalar cache, issued while the unit's vector loads are in flight) instead of being held — 64 scalar registers for two outputs of a
// 15-limb digit — across the whole unit loop
#ifndef HM_BCOL_ROWS_PER_UNIT
#define HM_BCOL_ROWS_PER_UNIT 1
#endif
#ifndef HM_BCOL_ONE_DESC
#define HM_BCOL_ONE_DESC 1
#endif
template <class PROB>
__device__ __forceinline__ const uint64_t *hm_bcol_table(const PROB &p) {
  const uint64_t *t = p.table;
#if defined(__HIP_DEVICE_COMPILE__)
  if (HM_BCOL_ROWS_PER_UNIT) asm volatile("" : "+s"(t));
#endif
  return t;
}
// input i of access unit u: 16 bytes per lane
template <class G0, class PROB>
__device__ __forceinline__ void hm_bcol_load_in(const PROB &p, int i, uint32_t tile, int tid, int u, size_t N, uint64_t &v0, uint64_t &v1) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (HM_BCOL_ONE_DESC) {
    const hm_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(hm_rsrc(p.in_base), G0::gthr(tid, u) << 3, (G0::guni(tile, u) << 3) + p.in_off[i], 0);
    v0 = (uint64_t)t.x | ((uint64_t)t.y << 32);
    v1 = (uint64_t)t.z | ((uint64_t)t.w << 32);
    return;
  }
#endif
  hm_gld2<G0>(p.in + (size_t)p.in_limb[i] * N, tile, tid, u, v0, v1);
}
struct HmBcolArgs {
  const HmBcolProb *prob;   // device
  uint64_t *out;
  const HmW *tw;
  uint32_t logN, n_prob, max_out;   // max_out: output GROUPS (of NOUT limbs) per (conversion, tile)
  const uint64_t *mix;
  uint32_t tile0, logTiles;         // the column tiles this launch works on: [tile0, tile0 + 2^logTiles) (all of them, or a rank's column slice)
};
// One workgroup = (conversion, NOUT consecutive output limbs, column tile).  NOUT = 2 (round 4): the N_IN input access units of a thread
// are loaded and split ONCE and multiplied into both outputs (half the L2 requests, the shift / mask work of the split amortised), then
// the COL rounds run for one output after the other on the one LDS tile (the second output waits in registers).  MIX (round 4: the
// ModDown conversion inside the merged ModDown + rescale transform): x = conv + k * mix before the first butterfly, the MODE 4 prologue.
// Any ring size (LOG1 = log2 N - 8).
// (Helpers are plain forceinline functions with everything passed by value: a lambda that captures the constant-address-space problem
// record by reference loses the address space, its scalar loads become vector loads and every buffer access a waterfall loop.)
template <int N_IN, bool MIX, class G0>
__device__ __forceinline__ void hm_bcol_convert(const uint32_t (&yl)[2][N_IN], const uint32_t (&yh)[2][N_IN], const uint64_t *table, const uint64_t *qn,
                                                const HmTw *mixk, const uint64_t *mixp, uint32_t o, uint32_t tile, int tid, int u, uint64_t &r0v, uint64_t &r1v) {
  constexpr int NG = (N_IN + 7) / 8;
  HmRow8 row[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) row[g] = HM_CONST_ROWS(table)[o * NG + g];
  const HmQn m = HM_CONST_QN(qn)[o];
  r0v = hm_bconv_dot<N_IN>(yl[0], yh[0], row, m.q, m.nqinv);
  r1v = hm_bconv_dot<N_IN>(yl[1], yh[1], row, m.q, m.nqinv);
  if (MIX) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const HmTw __attribute__((address_space(4))) *ConstTw;
    const HmTw mk = {((ConstTw)(uintptr_t)mixk)[o].w, ((ConstTw)(uintptr_t)mixk)[o].ws};
#else
    const HmTw mk = mixk[o];
#endif
    uint64_t b0, b1;
    hm_gld2<G0>(mixp, tile, tid, u, b0, b1);
    r0v = hm_addmod(r0v, hm_kmul(b0, mk, m.q), m.q);
    r1v = hm_addmod(r1v, hm_kmul(b1, mk, m.q), m.q);
  }
}
template <int TL, int LOG1>
__device__ __forceinline__ void hm_bcol_rounds(HmNttState &st, int tid, uint64_t *lds, uint64_t q, const HmW *twl, uint64_t *dst, uint32_t tile) {
  using PS = HmPass<LOG1, true, false>;
  constexpr int n = PS::n, r0 = PS::exec(0);
  const HmW *ltw = reinterpret_cast<const HmW *>(lds + (1 << TL));
  const HmTw sc = {0, 0};
  const HmEpi ep = hm_epi_none();
  hm_ph_load_tw<TL, LOG1, true, r0, true>(st, tid, ltw, 0, 0);
  hm_ph_compute<TL, LOG1, true, r0, false>(st, q);
  hm_ph_store_lds<TL, LOG1, true, r0>(st, tid, lds);
  __syncthreads();
  hm_ntt_phase<TL, LOG1, true, false, 0, 2>(st, tid, lds, nullptr, dst, tile, twl, nullptr, 0, 0, q, sc, ep);
  if constexpr (n >= 3) {
    __syncthreads();
    hm_ntt_phase<TL, LOG1, true, false, 0, 3>(st, tid, lds, nullptr, dst, tile, twl, nullptr, 0, 0, q, sc, ep);
  }
}
// conversion of a thread's HM_UNITS access units for the NOUT outputs of the workgroup (a function template, not a lambda: a lambda that
// captures the constant-address-space problem record by reference loses the address space)
template <int N_IN, int NOUT, bool MIX, bool PACKED, class G0, class PROB>
__device__ __forceinline__ void hm_bcol_units(const PROB &p, HmNttState &st0, HmNttState &st1, const uint64_t *mix0, const uint64_t *mix1, uint32_t o0, uint32_t o1,
                                              uint32_t tile, int tid, size_t N) {
#pragma unroll
  for (int u = 0; u < HM_UNITS; ++u) {   // (hm_bcol_part.hip is compiled with a raised pragma-unroll threshold: this loop must be unrolled)
    int i0, i1, x, c;
    G0::unit(tid, u, i0, i1, x, c);
    uint32_t yl[2][N_IN], yh[2][N_IN];
#pragma unroll
    for (int i = 0; i < N_IN; ++i) {
      uint64_t v0, v1;
      hm_bcol_load_in<G0>(p, i, tile, tid, u, N, v0, v1);
#if defined(HM_ABL_BCOL_PACKED)   // timing-only ablation: inputs taken as if already stored split (no shift / mask per input and output)
      yl[0][i] = (uint32_t)v0; yh[0][i] = (uint32_t)(v0 >> 32);
      yl[1][i] = (uint32_t)v1; yh[1][i] = (uint32_t)(v1 >> 32);
#else
      if (PACKED) {
        yl[0][i] = (uint32_t)v0; yh[0][i] = (uint32_t)(v0 >> 32);
        yl[1][i] = (uint32_t)v1; yh[1][i] = (uint32_t)(v1 >> 32);
      } else {
        yl[0][i] = (uint32_t)v0 & 0x3FFFFFFFu; yh[0][i] = (uint32_t)(v0 >> 30);
        yl[1][i] = (uint32_t)v1 & 0x3FFFFFFFu; yh[1][i] = (uint32_t)(v1 >> 30);
      }
#endif
    }
    const uint64_t *table = hm_bcol_table(p);
    hm_bcol_convert<N_IN, MIX, G0>(yl, yh, table, p.qn, p.mixk, mix0, o0, tile, tid, u, st0.v[i0], st0.v[i1]);
    // an odd basis' last group converts its one output twice (wave-uniform; the copy is never transformed or stored)
    if (NOUT == 2) hm_bcol_convert<N_IN, MIX, G0>(yl, yh, table, p.qn, p.mixk, mix1, o1, tile, tid, u, st1.v[i0], st1.v[i1]);
#if defined(__HIP_DEVICE_COMPILE__)
    // the results are "used" here: otherwise the products are sunk below the barrier to their first real use and ALL units' inputs are
    // loaded (and spilled) up front
    asm volatile("" : "+v"(st0.v[i0]), "+v"(st0.v[i1]));
    if (NOUT == 2) asm volatile("" : "+v"(st1.v[i0]), "+v"(st1.v[i1]));
#endif
    __builtin_amdgcn_sched_barrier(0);   // one unit's loads in flight at a time (N_IN x 16 bytes per lane)
  }
}
// ---- digits of 16 .. 32 limbs (round 6: parameter set A and the `motivation` sweep convert from up to 28 limbs, script/README.md:17-22,
// src/Operation.cpp:137-188) ----------------------------------------------------------------------------------------------------------
// The inputs of an access unit are taken in two groups, [0, 16) and [16, N_IN): a group is loaded, split, multiplied into the 128-bit sums
// of the workgroup's NOUT outputs and dropped before the next one is requested, so that the live inputs never exceed what the 15-limb form
// holds (16 x 2 coefficients x 2 halves = 64 VGPRs; all N_IN at once left the arrays in scratch: 1 KB per lane at 16).  A group's table
// entries are whole 8-entry rows (the second group starts at entry 16), read through the scalar cache as before; one reduction per output
// (hm_redc_wide<N_IN>: two conditional subtractions above 16 terms).
template <int C0, int CN, int NOUT, bool PACKED, class G0, class PROB>
__device__ __forceinline__ void hm_bcol_group(const PROB &p, const uint64_t *table, uint32_t rowsPerOut, uint32_t o0, uint32_t o1, uint32_t tile, int tid, int u, size_t N,
                                              hm_u128 (&acc)[2][2]) {
  uint32_t yl[2][CN], yh[2][CN];
#pragma unroll
  for (int i = 0; i < CN; ++i) {
    uint64_t v0, v1;
    hm_bcol_load_in<G0>(p, C0 + i, tile, tid, u, N, v0, v1);
    if (PACKED) {
      yl[0][i] = (uint32_t)v0; yh[0][i] = (uint32_t)(v0 >> 32);
      yl[1][i] = (uint32_t)v1; yh[1][i] = (uint32_t)(v1 >> 32);
    } else {
      yl[0][i] = (uint32_t)v0 & 0x3FFFFFFFu; yh[0][i] = (uint32_t)(v0 >> 30);
      yl[1][i] = (uint32_t)v1 & 0x3FFFFFFFu; yh[1][i] = (uint32_t)(v1 >> 30);
    }
  }
  constexpr int NG = (CN + 7) / 8;
#pragma unroll
  for (int k = 0; k < NOUT; ++k) {
    const uint32_t o = k ? o1 : o0;
    HmRow8 row[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) row[g] = HM_CONST_ROWS(table)[o * rowsPerOut + C0 / 8 + g];
    acc[k][0] += hm_bconv_cols<CN>(yl[0], yh[0], row);
    acc[k][1] += hm_bconv_cols<CN>(yl[1], yh[1], row);
  }
}
template <int N_IN, int NOUT, bool PACKED, class G0, class PROB>
__device__ __forceinline__ void hm_bcol_units_wide(const PROB &p, HmNttState &st0, HmNttState &st1, uint32_t o0, uint32_t o1, uint32_t tile, int tid, size_t N) {
  static_assert(N_IN >= 16 && N_IN <= 32, "two input groups");
  constexpr uint32_t ROWS = (N_IN + 7) / 8;   // 8-entry table rows per output (HM_BCONV_ROW)
#pragma unroll
  for (int u = 0; u < HM_UNITS; ++u) {
    int i0, i1, x, c;
    G0::unit(tid, u, i0, i1, x, c);
    hm_u128 acc[2][2] = {{0, 0}, {0, 0}};
    hm_bcol_group<0, (N_IN < 16 ? N_IN : 16), NOUT, PACKED, G0>(p, hm_bcol_table(p), ROWS, o0, o1, tile, tid, u, N, acc);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]));   // the first group's sums are complete before the second group's inputs are requested
    if (NOUT == 2) asm volatile("" : "+v"(acc[1][0]), "+v"(acc[1][1]));
    __builtin_amdgcn_sched_barrier(0);
#endif
    if constexpr (N_IN > 16) hm_bcol_group<16, N_IN - 16, NOUT, PACKED, G0>(p, hm_bcol_table(p), ROWS, o0, o1, tile, tid, u, N, acc);
    HmMod m0;
    { const HmQn m = HM_CONST_QN(p.qn)[o0]; m0.q = m.q; m0.nqinv = m.nqinv; }
    st0.v[i0] = hm_redc_wide<N_IN>(acc[0][0], m0);
    st0.v[i1] = hm_redc_wide<N_IN>(acc[0][1], m0);
    if (NOUT == 2) {
      HmMod m1;
      { const HmQn m = HM_CONST_QN(p.qn)[o1]; m1.q = m.q; m1.nqinv = m.nqinv; }
      st1.v[i0] = hm_redc_wide<N_IN>(acc[1][0], m1);
      st1.v[i1] = hm_redc_wide<N_IN>(acc[1][1], m1);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(st0.v[i0]), "+v"(st0.v[i1]));
    if (NOUT == 2) asm volatile("" : "+v"(st1.v[i0]), "+v"(st1.v[i1]));
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
}
template <int N_IN, int LOG1, int NOUT, bool MIX, bool PACKED>
__device__ __forceinline__ void hm_bconv_col_body(const HmBcolArgs &a) {
  constexpr int TL = HM_TL_COL;
  using PS = HmPass<LOG1, true, false>;
  using G0 = HmRound<TL, LOG1, true, PS::exec(0)>;
  __shared__ __attribute__((aligned(16))) uint64_t lds[HmLds<TL, LOG1, true>::WORDS];
  // blocks b, b + 8 share an XCD; inside an XCD: (conversion, tile) pairs, each with its max_out output groups in consecutive slots
  const uint32_t b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
  // (the division runs on the vector unit: its results are made scalar again by hand, or every buffer access below becomes a waterfall loop)
  const uint32_t sdiv = __builtin_amdgcn_readfirstlane(slot / a.max_out);
  const uint32_t pair = sdiv * 8u + xcd, og = slot - sdiv * a.max_out;
  const uint32_t logT = a.logTiles;
  const uint32_t pi = pair >> logT, tile = a.tile0 + (pair & ((1u << logT) - 1u));
  if (pi >= a.n_prob) return;
  const auto &p = HM_CONST_PROB_T(HmBcolProb, a.prob)[pi];
  const uint32_t o0 = og * NOUT;
  if (o0 >= p.n_out) return;
  const int tid = threadIdx.x;
  const size_t N = (size_t)1 << a.logN;
  hm_ph_stage_tw<TL, LOG1, true>(tid, lds, a.tw + (size_t)p.out_mod[o0] * N);   // every round reads its shared twiddles from LDS (after the barrier behind the conversion)
  // (two named states and straight-line code for the two outputs: a loop over an array of states with barriers inside is not unrolled
  // and the states end up in scratch)
  HmNttState st0, st1;
  const bool two = NOUT == 2 && o0 + 1 < p.n_out;   // wave-uniform: the last group of an odd basis has one output
  const uint32_t o1 = two ? o0 + 1 : o0;
  const uint64_t *mix0 = MIX ? a.mix + (size_t)p.mix_limb[o0] * N : nullptr, *mix1 = MIX ? a.mix + (size_t)p.mix_limb[o1] * N : nullptr;
  // (the inputs' form is a template parameter of the KERNEL: two copies of the unit loop in one kernel shared one scalar-register budget,
  // and the packed copy reloaded spilled table words 1 262 times per workgroup — measured +0 % instead of the -16 % of the form on its own)
  if constexpr (N_IN > HM_BCOL_ONE_GROUP) {
    static_assert(N_IN <= HM_BCOL_ONE_GROUP || !MIX, "the mix prologue is built for digits of up to 15 limbs (the opt-in fused ModDown conversion)");
    hm_bcol_units_wide<N_IN, NOUT, PACKED, G0>(p, st0, st1, o0, o1, tile, tid, N);
  } else hm_bcol_units<N_IN, NOUT, MIX, PACKED, G0>(p, st0, st1, mix0, mix1, o0, o1, tile, tid, N);
  __syncthreads();
  hm_bcol_rounds<TL, LOG1>(st0, tid, lds, HM_CONST_QN(p.qn)[o0].q, a.tw + (size_t)p.out_mod[o0] * N, a.out + (size_t)p.out_limb[o0] * N, tile);
  if (NOUT == 2 && two) {
    __syncthreads();   // the first output's last round has read the tile and its twiddles
    hm_ph_stage_tw<TL, LOG1, true>(tid, lds, a.tw + (size_t)p.out_mod[o1] * N);
    __syncthreads();
    hm_bcol_rounds<TL, LOG1>(st1, tid, lds, HM_CONST_QN(p.qn)[o1].q, a.tw + (size_t)p.out_mod[o1] * N, a.out + (size_t)p.out_limb[o1] * N, tile);
  }
}
// PACKED: the inputs of every conversion of the launch are stored in the split-30 packed form (HmBcolProb::in_packed; the host launches
// packed and plain conversions separately)
template <int N_IN, int LOG1, bool MIX, bool PACKED>
__global__ void __launch_bounds__((1 << HM_TL_COL) / HM_EPT) __attribute__((amdgpu_waves_per_eu(HM_NTT_MIN_WAVES))) k_bconv_col(HmBcolArgs a) {
  hm_bconv_col_body<N_IN, LOG1, 1, MIX, PACKED>(a);
}
// Two outputs wait in registers beside the split inputs.  Round 6: without the spilled scalar registers (which lived in vector-register lanes) the
// 15-limb form needs 136 VGPRs; held to 128 (8 of them in scratch: four 8-byte loads and stores per thread) it runs four workgroups per CU instead
// of three: +0.4 % on the op, +0.7 % one at a time (gpurun_out/r06_bcol4_ab: three interleaved rounds, every round ahead).  The two-group form
// (16 .. 32 limbs: 149 VGPRs) keeps three — the launch bound cannot depend on a template parameter, hence a kernel template of its own.
#ifndef HM_BCOL2_WAVES
#define HM_BCOL2_WAVES 4
#endif
#ifndef HM_BCOL2_WAVES_WIDE
#define HM_BCOL2_WAVES_WIDE 3
#endif
template <int N_IN, int LOG1, bool MIX, bool PACKED>
__global__ void __launch_bounds__((1 << HM_TL_COL) / HM_EPT) __attribute__((amdgpu_waves_per_eu(HM_BCOL2_WAVES))) k_bconv_col2(HmBcolArgs a) {
  static_assert(N_IN <= HM_BCOL_ONE_GROUP, "digits of 16 .. 32 limbs: k_bconv_col2w");
  hm_bconv_col_body<N_IN, LOG1, 2, MIX, PACKED>(a);
}
template <int N_IN, int LOG1, bool MIX, bool PACKED>
__global__ void __launch_bounds__((1 << HM_TL_COL) / HM_EPT) __attribute__((amdgpu_waves_per_eu(HM_BCOL2_WAVES_WIDE))) k_bconv_col2w(HmBcolArgs a) {
  static_assert(N_IN > HM_BCOL_ONE_GROUP, "digits of up to 15 limbs: k_bconv_col2");
  hm_bconv_col_body<N_IN, LOG1, 2, MIX, PACKED>(a);
}


typedef void (*hm_bcol_kernel)(HmBcolArgs);
// nullptr: no such kernel (ring sizes other than 2^15 and 2^16 convert with hm_bconv_batch first; packed inputs with the mix prologue:
// the opt-in fused ModDown conversion takes plain inputs)
hm_bcol_kernel hm_bcol_kernel_for(uint32_t n_in, uint32_t logN, uint32_t n_out_per_wg, bool mix, bool packed);
