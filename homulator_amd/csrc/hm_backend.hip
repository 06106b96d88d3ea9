// hm_backend.hip — HIP kernels (gfx950) + context + the C ABI of include/homulator_hip.h.
// There is no CPU fallback in this library: every compute entry point launches a HIP kernel.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <tuple>
#include <type_traits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/homulator_hip.h"
#include "hm_elem_core.h"
#include "hm_modarith.h"
#include "hm_ntt_core.h"
#include "hm_params.h"
#include "hm_caps.h"

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------

// XCD-aware block -> (limb entry, tile) map.  Blocks are dealt round-robin over the 8 XCDs, so blocks b and
// b+8 share an L2.  All tiles of one limb-poly get the same b % 8, and both passes of a transform use the
// same map.  Entries come in GROUPS of G = 2^logG (e, e+8, .., e+8(G-1) inside a block of 8G) that the host fills with
// limb-polys of the SAME modulus whenever the launch has them (the two keys of a ModDown, the digits of a ModUp, c0/c1 of
// a rescale, the ops of a batch): the group's blocks for one tile sit in adjacent dispatch slots of one XCD, so all but
// the first find the row twiddles (0.27 MB per limb-NTT, a quarter of the data) in L2 instead of fetching them again.
__device__ __forceinline__ bool hm_block_map(uint32_t tiles_per_limb, uint32_t n_entries, uint32_t logG, uint32_t &entry, uint32_t &tile) {
  const uint32_t b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
  const uint32_t per = tiles_per_limb << logG;
  const uint32_t grp = slot / per, within = slot % per;
  tile = within >> logG;
  entry = grp * (8u << logG) + (within & ((1u << logG) - 1u)) * 8u + xcd;
  return entry < n_entries;
}

// Synchronisation between the rounds of a pass.  COL pass: the exchange crosses the waves of the workgroup, every one is a barrier.
// ROW pass: a 256-point row lives on 16 (hm16) or 32 (hm8) lanes of ONE wave in every round (HmRound::group: c = tid / lanes per row),
// so a wave only ever reads tile words it wrote itself; LDS operations of a wave execute in order, so after the first barrier (behind
// which the staged shared twiddles are visible) the later exchanges need no barrier at all, only that the compiler keeps the order
// (HM_ROW_WAVE_SYNC = 1: bit-exact, 96 GPU tests; measured without difference — 4 800 hmult/s either way — so the barriers stay).
#ifndef HM_ROW_WAVE_SYNC
#define HM_ROW_WAVE_SYNC 0
#endif
template <bool STRIDED>
__device__ __forceinline__ void hm_pass_sync(int nth) {
  if (STRIDED || !HM_ROW_WAVE_SYNC || nth == 0) __syncthreads();
  else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
}

// One pass of a transform over tile `tile` of limb-poly `entry`, in the workgroup's LDS buffer.
// the two geometries of the passes (hm_ntt_core.h)
struct Geo16 {
  static constexpr int EPT = 16;
  typedef hm16::HmNttState State;
  template <int TL, int LOGR, bool STRIDED> static constexpr int ldsWords() { return hm16::HmLds<TL, LOGR, STRIDED>::WORDS; }
  template <int TL, int LOGR, bool STRIDED, bool INV, int MODE, int LDAUX, int STAUX, int EPICH, class... A>
  static __device__ __forceinline__ void phases(A &&...a) { hm16::hm_ntt_pass_phases<TL, LOGR, STRIDED, INV, MODE, LDAUX, STAUX, EPICH>(a...); }
};
struct Geo8 {
  static constexpr int EPT = 8;
  typedef hm8::HmNttState State;
  template <int TL, int LOGR, bool STRIDED> static constexpr int ldsWords() { return hm8::HmLds<TL, LOGR, STRIDED>::WORDS; }
  template <int TL, int LOGR, bool STRIDED, bool INV, int MODE, int LDAUX, int STAUX, int EPICH, class... A>
  static __device__ __forceinline__ void phases(A &&...a) { hm8::hm_ntt_pass_phases<TL, LOGR, STRIDED, INV, MODE, LDAUX, STAUX, EPICH>(a...); }
};
template <int LOGR, bool STRIDED, bool INV, int MODE, int LDAUX = 0, int STAUX = 0, int EPICH = HM_EPI_CHUNK, class GEO = Geo16, class PRE = HmNoPre>
__device__ __forceinline__ void hm_ntt_pass_run(const HmNttArgs &a, uint64_t *lds, uint32_t entry, uint32_t tile, int tid, PRE pre = PRE()) {
  constexpr int TL = HM_TL(STRIDED);
  const HmLimb lb = a.limb[entry];
  const uint32_t mod = lb.mod;
  const size_t N = (size_t)1 << a.logN;
  const uint64_t q = HM_CONST_MODS(a.mods)[mod].q;   // through the scalar cache (read-only for the lifetime of the context)
  const HmW *twl = a.tw + (size_t)mod * N;
  const uint32_t s0 = STRIDED ? 0u : (a.logN - HM_ROW_LOG);
  const uint32_t prefix0 = STRIDED ? 0u : (tile << (TL - LOGR));
  const HmW *twt = STRIDED ? nullptr : a.twist + ((size_t)mod * (N >> HM_ROW_LOG) + prefix0) * 3;  // the tile's first row
  // the first pass of a transform reads `in`, the second works in place on `out`
  constexpr bool FIRST = (STRIDED != INV);
  const uint64_t *src = FIRST ? a.in + (size_t)lb.in * N : a.out + (size_t)lb.out * N;
  uint64_t *dst = a.out + (size_t)lb.out * N;

  HmTw sc = {0, 0};
  HmEpi ep = hm_epi_none();
  // the per-limb constants are wave-uniform: read through the scalar cache into SGPRs (as a plain global pointer hipcc
  // loaded them with vector loads and kept 12-16 VGPRs of uniform values alive through the whole pass)
  typedef const HmNttEntry __attribute__((address_space(4))) *ConstEntry;
  const ConstEntry entries = (ConstEntry)(uintptr_t)a.entry;
  if constexpr (MODE == 2) { sc.w = entries[entry].sc.w; sc.ws = entries[entry].sc.ws; ep.pack = entries[entry].pack; }
  if constexpr (MODE == 3 || MODE == 7) {
    const auto &en = entries[entry];
    sc.w = en.sc.w; sc.ws = en.sc.ws;
    ep.a = a.minuend + (size_t)lb.aux * N;
    ep.d = a.addend && en.alimb != HM_NTT_NONE ? a.addend + (size_t)en.alimb * N : nullptr;  // per limb-poly
    ep.dk.w = en.ak.w; ep.dk.ws = en.ak.ws;
    if constexpr (MODE == 7) { ep.g = en.galois; ep.logN = a.logN; }
  }
  if constexpr (MODE == 6) { ep.g = entries[entry].galois; ep.logN = a.logN; }
  if constexpr (MODE == 4) {
    const auto &en = entries[entry];
    ep.b = a.mix + (size_t)en.mixlimb * N;
    ep.bk.w = en.mixk.w; ep.bk.ws = en.mixk.ws;
  }
  typename GEO::State st;
  int nsync = 0;
  GEO::template phases<TL, LOGR, STRIDED, INV, MODE, LDAUX, STAUX, EPICH>(st, tid, lds, src, dst, tile, twl, twt, s0, prefix0, q, sc, ep,
                                                                           [&] { hm_pass_sync<STRIDED>(nsync++); }, pre);
}

template <int LOGR, bool STRIDED, bool INV, int MODE, class GEO = Geo16>
__device__ __forceinline__ void hm_ntt_pass_body(const HmNttArgs &a) {
  constexpr int TL = HM_TL(STRIDED);
  __shared__ __attribute__((aligned(16))) uint64_t lds[GEO::template ldsWords<TL, LOGR, STRIDED>()];
  uint32_t entry, tile;
  if (!hm_block_map(1u << (a.logN - TL), a.n_limbs, a.logG, entry, tile)) return;
  if (a.limb[entry].mod == HM_NTT_NONE) return;
#if defined(HM_ABL_EMPTY)   // timing-only ablation: launch, dispatch and the block map only
  return;
#endif
  hm_ntt_pass_run<LOGR, STRIDED, INV, MODE, 0, 0, HM_EPI_CHUNK, GEO>(a, lds, entry, tile, threadIdx.x);
}

// Minimum waves per SIMD the register allocator must leave room for (HIP's second launch-bound argument; it is ignored
// when it depends on a template parameter, hence one kernel template per pass).  4 workgroups of 256 threads per CU
// (36 KiB of LDS each) = 4 waves per SIMD: up to 128 VGPRs.
#ifndef HM_NTT_MIN_WAVES
#define HM_NTT_MIN_WAVES 4
#endif
#ifndef HM_NTT_MIN_WAVES_COL
#define HM_NTT_MIN_WAVES_COL HM_NTT_MIN_WAVES
#endif
// MODE 0: first pass / plain hand-off; 1: forward final; 2: inverse final (x scale); 3: forward final with the fused
// epilogue; 4: forward first pass with the mix prologue; 6: inverse first pass whose input is read through an automorphism; 7: MODE 3 with the
// addend read through an automorphism (round 6: hrotate's automorphism launch folded into the ModUp INTT and the final add)
template <int LOGR, bool INV, int MODE>
__global__ void __launch_bounds__((1 << HM_TL_COL) / HM_EPT) __attribute__((amdgpu_waves_per_eu(HM_NTT_MIN_WAVES_COL))) k_ntt_col(HmNttArgs a) {
  hm_ntt_pass_body<LOGR, true, INV, MODE>(a);
}
template <bool INV, int MODE>
__global__ void __launch_bounds__((1 << HM_TL_ROW) / HM_EPT) __attribute__((amdgpu_waves_per_eu(HM_NTT_MIN_WAVES))) k_ntt_row(HmNttArgs a) {
  hm_ntt_pass_body<HM_ROW_LOG, false, INV, MODE>(a);
}

// The small-launch geometry: 512-thread workgroups, 8 coefficients per thread (hm8), N = 2^16 and (round 6) N = 2^15.  A launch of up to
// ~128 limb-polys is ONE round of workgroups, so its time is the latency of one workgroup's pass; here every wave does half the serial work
// and the launch brings twice the waves (a single op's stages, the 50-limb sweep of the extended basis, a sharded run's per-rank launches).
template <int LOG1, bool INV, int MODE>
__global__ void __launch_bounds__((1 << HM_TL_COL) / 8) k_ntt_col8(HmNttArgs a) {
  hm_ntt_pass_body<LOG1, true, INV, MODE, Geo8>(a);
}
template <bool INV, int MODE>
__global__ void __launch_bounds__((1 << HM_TL_ROW) / 8) k_ntt_row8(HmNttArgs a) {
  hm_ntt_pass_body<HM_ROW_LOG, false, INV, MODE, Geo8>(a);
}

// ---- both passes of a transform in ONE launch (k_ntt_fused8: the default for launches of up to `ntt_fused_small` limb-polys, N = 2^16) ----
// The workgroups of one limb-poly (N / 4096 of them, all dealt to one XCD by hm_block_map) run the first pass on their tile, store the
// hand-off, meet at a per-limb counter and run the second pass on what the others stored.  What the MI355X does with the hand-off
// (profiles/r04_l2_handoff.txt; round 3's reading of the same counters — "write-through, no allocation on a store" — was an eviction
// result and is withdrawn): the XCD's L2 is WRITE-BACK for plain stores and the second pass's loads of the hand-off HIT as long as at most
// ~3 MiB are live per XCD; a line that a store allocated is still filled from the fabric once when it is first read, unless it was
// resident before the store, i.e. the transform runs IN PLACE (50-limb sweep: 104.5 MB of fabric traffic out of place, 79 MB in place).
// The gain of the one-launch form is launch and latency (28-32 us against 34-36 us for the 50-limb sweep as two kernels); from ~128
// limb-polys the rendezvous costs more than it saves (it idles a limb-poly's slots until its slowest workgroup has arrived), so larger
// launches stay two kernels.  The wide-geometry form of rounds 3 / 4 (k_ntt_fused: 256-thread workgroups, any ring size) lost every
// A/B against this one and left the tree in round 5 (profiles/r04_fused_small.txt keeps its numbers).
//
// Correct for ANY placement: every workgroup that has to wait publishes its XCC id (HW_REG_XCC_ID); if the limb's workgroups
// turn out to sit on several XCDs (hipcc / the dispatcher promise nothing), all of them take the agent-scope path instead:
// release (L2 write-back), a second rendezvous, acquire — slow, still right.  The rendezvous needs the limb's workgroups
// co-resident: they are neighbours in dispatch order (consecutive slots of one XCD hold a limb-poly), so with
// workgroups dispatched in order the oldest unfinished limb of an XCD is always fully dispatched (hm_create checks that an XCD holds
// at least one limb-poly's workgroups at the kernel's occupancy); the spin is bounded all the same, and a timeout is reported through
// the context (the next synchronising call fails, the form is switched off) instead of hanging the GPU.
// The rendezvous runs on XCD-LOCAL atomics.  An agent-scope atomic or load is a round trip to memory (the L2s of
// the eight XCDs are not coherent with each other): arrival + polls cost round 3's one-launch transform 16 us at 50 limb-polys.
// All workgroups of a limb-poly run on ONE XCD (hm_block_map; checked per workgroup against HW_REG_XCC_ID), so their
// counter can live in that XCD's L2: returning atomics WITHOUT the agent-scope bit (`global_atomic_add_x2 ... sc0`) execute there.  One
// 128-byte line per limb-poly (a line shared with a limb-poly of another XCD would be modified in two L2s at once): low word = arrivals,
// high word = leavers; the last leaver swaps it back to zero.  Should the dispatcher ever spread a limb-poly's workgroups over XCDs, no
// copy of the counter reaches the member count; the waiting workgroups find out through an agent-scope mask of XCC
// ids and everybody takes the agent-scope path (never seen outside the test hook: counter `ntt_cross_xcd`).
struct HmLimbSync {
  unsigned long long w;        // XCD-local atomics only
  unsigned long long pad[15];
};
struct HmNttSync {
  HmLimbSync fast[HM_NTT_MAX_ENTRIES];
  unsigned long long arrive2[HM_NTT_MAX_ENTRIES];  // rendezvous of the agent-scope path
  unsigned done[HM_NTT_MAX_ENTRIES];               // agent-scope path: workgroups that left the limb-poly; the last one zeroes the agent-scope words
  unsigned xccmask[HM_NTT_MAX_ENTRIES];            // agent-scope: XCC ids of the workgroups that had to wait (two bits set: the limb-poly is spread over XCDs)
  unsigned stats[4];                               // [0] limb-polys that took the agent-scope path (diagnostic)
};
#define HM_SPIN_LIMIT (1u << 22)
#ifndef HM_OPAQUE_TID2
#define HM_OPAQUE_TID2 1
#endif
#ifndef HM_FUSED_IN_AUX
#define HM_FUSED_IN_AUX 0    // cache policy of the first pass's input loads (2 = nt)
#endif
#ifndef HM_FUSED_OUT_AUX
#define HM_FUSED_OUT_AUX 0   // cache policy of the second pass's output stores (2 = nt)
#endif
#ifndef HM_FUSED_MID_AUX
#define HM_FUSED_MID_AUX 16  // the hand-off loads: sc1 (past the CU's vector L1, which another CU's stores never refresh)
#endif
__device__ __forceinline__ unsigned hm_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v & 7u;
}
// returning atomics executed in the L2 of the XCD the wave runs on (no sc1: not agent scope)
__device__ __forceinline__ unsigned long long hm_l2_add(unsigned long long *p, unsigned long long v) {
  unsigned long long old;
  asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(p), "v"(v) : "memory");
  return old;
}
__device__ __forceinline__ void hm_l2_zero(unsigned long long *p) {
  unsigned long long old, z = 0;
  asm volatile("global_atomic_swap_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(p), "v"(z) : "memory");
}
// not returning: the issuing wave does not wait for it (same-address operations of one wave reach the L2 in order)
__device__ __forceinline__ void hm_l2_add_noret(unsigned long long *p, unsigned long long v) {
  asm volatile("global_atomic_add_x2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
// The rendezvous of a limb-poly's workgroups, in two halves.  hm_limb_arrive: this workgroup's hand-off stores have reached the L2 (every
// storing wave waits for its own, then the barrier), one lane adds to the limb-poly's XCD-local counter.  It runs BEFORE the second pass
// requests anything, so nothing but the stores is waited for and the arrival is not held up by the twiddle loads (round 5; it used to sit
// behind them).  hm_limb_wait (inside the second pass's first phase, behind its first twiddle requests, which travel while the workgroup
// waits): all `members` workgroups have arrived.  Returns 1 if they all run on this XCD and met on the XCD-local counter, 0 if the
// limb-poly is spread over several XCDs (then no copy of the counter ever reaches `members`).  Which XCD a block id lands on is the
// dispatcher's business (blocks b and b + 8 share one, but not necessarily XCD b mod 8), so a spread is found by publication: a workgroup
// whose first poll fails ORs its XCC id into the limb-poly's agent-scope mask (not returning: no round trip for the workgroup, which is
// waiting anyway; the last arriver never gets that far) and looks at the mask every 16th spin.
__device__ __forceinline__ void hm_limb_arrive(HmNttSync *ws, uint32_t entry, bool withhold) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its stores have reached L2
  __syncthreads();
  if (threadIdx.x == 0 && !withhold) hm_l2_add_noret(&ws->fast[entry].w, 1ull);
}
__device__ __forceinline__ uint32_t hm_limb_wait(HmNttSync *ws, unsigned *err, uint32_t entry, uint32_t members, uint32_t *lds_flag, uint32_t spin_limit,
    uint32_t pretend_spread = 0, uint32_t tile = 0) {
  if (threadIdx.x == 0) {
    unsigned long long *w = &ws->fast[entry].w;
    uint32_t fast = 0;
    unsigned spins = 0;
    for (;;) {
      // (pretend_spread: test hook — the workgroups behave as if those of odd tiles ran on another XCD, whose copy of the counter this one
      // never sees complete: exercises the agent-scope path on hardware that never spreads a limb-poly)
      if ((uint32_t)hm_l2_add(w, 0ull) == members && !pretend_spread) { fast = 1; break; }
      if (spins == 0) (void)__hip_atomic_fetch_or(&ws->xccmask[entry], pretend_spread ? 1u << (tile & 1u) : 1u << hm_xcc_id(), __ATOMIC_RELAXED,
          __HIP_MEMORY_SCOPE_AGENT);
      else if ((spins & 15u) == 15u) {
        const unsigned m = __hip_atomic_load(&ws->xccmask[entry], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m & (m - 1u)) break;   // two XCDs: everybody takes the agent-scope path
      }
      __builtin_amdgcn_s_sleep(1);
      // 2: timed out, leave without a second wait
      if (++spins > spin_limit) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); fast = 2; break; }
    }
    *lds_flag = fast;
  }
  __syncthreads();
  return *lds_flag;
}
// the agent-scope path: make the hand-off visible to every XCD
__device__ __forceinline__ void hm_limb_publish_everywhere(HmNttSync *ws, unsigned *err, uint32_t entry, uint32_t members, uint32_t spin_limit) {
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&ws->arrive2[entry], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(&ws->arrive2[entry], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < members) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > spin_limit) { __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}
// leave the limb-poly: the last leaver of an XCD-local counter zeroes it (arrivals are final by then: in either mode every workgroup has
// passed a rendezvous) and, when everybody met there, the agent-scope mask; on the agent-scope path the last workgroup out zeroes the
// agent-scope words for the next launch
__device__ __forceinline__ void hm_limb_leave(HmNttSync *ws, uint32_t entry, uint32_t members, uint32_t fast) {
  if (threadIdx.x == 0) {
    unsigned long long *w = &ws->fast[entry].w;
    const unsigned long long old = hm_l2_add(w, 1ull << 32);
    if ((uint32_t)(old >> 32) + 1u == (uint32_t)old) {
      hm_l2_zero(w);
      if (fast) __hip_atomic_store(&ws->xccmask[entry], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!fast && __hip_atomic_fetch_add(&ws->done[entry], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
      __hip_atomic_store(&ws->arrive2[entry], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ws->xccmask[entry], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ws->done[entry], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&ws->stats[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
struct HmNttFusedArgs {
  HmNttSync *ws;
  unsigned *err;   // host-visible word: 0 = fine, 1 / 2 = a rendezvous timed out
  uint32_t pretend_spread;   // test hook (hm_set_option "ntt_fused_test_spread"): take the agent-scope path as if the limb-poly were spread over XCDs
  uint32_t spin_limit;       // HM_SPIN_LIMIT; the time-out test hook passes a short one
  uint32_t withhold;         // test hook (hm_set_option "ntt_fused_test_timeout"): tile 0 of every limb-poly never arrives
#if defined(HM_FUSED_TRACE)   // development builds only (tools/ablate.sh trace "-DHM_FUSED_TRACE"): 8 time stamps per workgroup, 100 MHz wall clock
  unsigned long long *trace;
#endif
};
#if defined(HM_FUSED_TRACE)
#define HM_STAMP(i) do { if (f.trace && threadIdx.x == 0) f.trace[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define HM_STAMP(i) do { } while (0)
#endif
// MODE_A: first pass (0, or 4 = mix prologue); MODE_B: last pass (1 forward, 3 fused epilogue, 2 inverse)
// IN_AUX: cache policy of the first pass's input loads.  Out of place they are read once and only crowd the L2 that should keep the
// hand-off: non-temporal loads (2) take 2-4 us off a 50-64 limb-poly launch; in place the hand-off lands on the very lines the input
// loads brought in, and plain loads (0) are 2 us faster (tools/ntt_fused_small_ab.py)
template <int LOG1, bool INV, int MODE_A, int MODE_B, class GEO, int IN_AUX = HM_FUSED_IN_AUX>
__device__ __forceinline__ void hm_ntt_fused_body(const HmNttArgs &a, const HmNttFusedArgs &f) {
  static_assert(HM_TL_COL == HM_TL_ROW, "the one-launch transform keeps a workgroup on tile t of both passes");
  constexpr int TL = HM_TL_ROW;
  constexpr int W1 = GEO::template ldsWords<TL, LOG1, true>(), W2 = GEO::template ldsWords<TL, HM_ROW_LOG, false>();
  __shared__ __attribute__((aligned(16))) uint64_t lds[(W1 > W2 ? W1 : W2) + 2];
  uint32_t entry, tile;
  const uint32_t members = 1u << (a.logN - TL);
  if (!hm_block_map(members, a.n_limbs, a.logG, entry, tile)) return;
  if (a.limb[entry].mod == HM_NTT_NONE) return;
  HM_STAMP(0);
#if defined(HM_FUSED_TRACE)
  if (f.trace && threadIdx.x == 0) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); f.trace[(size_t)blockIdx.x * 8 + 6] =
      ((unsigned long long)hm_xcc_id() << 32) | hw; f.trace[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)entry << 32) | tile; }
#endif
  uint32_t *flag = reinterpret_cast<uint32_t *>(lds + (W1 > W2 ? W1 : W2));
  if constexpr (!INV) hm_ntt_pass_run<LOG1, true, false, MODE_A, IN_AUX, 0, HM_EPI_CHUNK, GEO>(a, lds, entry, tile, threadIdx.x);
  else hm_ntt_pass_run<HM_ROW_LOG, false, true, MODE_A, IN_AUX, 0, HM_EPI_CHUNK, GEO>(a, lds, entry, tile, threadIdx.x);
  // a thread id the compiler cannot connect with the first pass's: otherwise lane offsets of the second pass are computed
  // up front and kept (spilled) through the first
  int tid2 = threadIdx.x;
  if (HM_OPAQUE_TID2) asm volatile("" : "+v"(tid2));
  __builtin_assume(tid2 >= 0 && tid2 < (1 << HM_TL_ROW) / GEO::EPT);
  // the arrival is published as soon as the hand-off is stored; the wait sits INSIDE the second pass's first phase, behind the requests for
  // its first round's twiddles (they do not depend on the hand-off and arrive while the workgroup waits) and in front of everything that
  // touches LDS or the hand-off
  HM_STAMP(1);   // first pass stored (this workgroup's last store issued)
  hm_limb_arrive(f.ws, entry, f.withhold && tile == 0);
  HM_STAMP(2);   // stores in the L2, arrival published
  uint32_t fast = 0;
  auto meet = [&] {
    HM_STAMP(3);   // second pass's first twiddle requests issued: starts to wait
    fast = hm_limb_wait(f.ws, f.err, entry, members, flag, f.spin_limit, f.pretend_spread, tile);
    if (!fast) hm_limb_publish_everywhere(f.ws, f.err, entry, members, f.spin_limit);
    HM_STAMP(4);   // all siblings have arrived
  };
  if constexpr (!INV) hm_ntt_pass_run<HM_ROW_LOG, false, false, MODE_B, HM_FUSED_MID_AUX, HM_FUSED_OUT_AUX, 1, GEO>(a, lds, entry, tile, tid2, meet);
  else hm_ntt_pass_run<LOG1, true, true, MODE_B, HM_FUSED_MID_AUX, HM_FUSED_OUT_AUX, HM_EPI_CHUNK, GEO>(a, lds, entry, tile, tid2, meet);
  if (fast == 2) return;   // timed out: the host zeroes the words (check_device_error)
  HM_STAMP(5);   // second pass stored
  hm_limb_leave(f.ws, entry, members, fast);
}
// the small-launch geometry (512-thread workgroups, 8 coefficients per thread; LOG1 = 8: N = 2^16, 16 workgroups per limb-poly; LOG1 = 7,
// round 6: N = 2^15, 8 workgroups per limb-poly)
template <int LOG1, bool INV, int MODE_A, int MODE_B, bool NTIN>
__global__ void __launch_bounds__((1 << HM_TL_ROW) / 8) k_ntt_fused8(HmNttArgs a, HmNttFusedArgs f) {
  hm_ntt_fused_body<LOG1, INV, MODE_A, MODE_B, Geo8, NTIN ? 2 : 0>(a, f);
}

// (Rounds 4 / 5 also built both passes as ONE persistent launch fed from per-XCD work queues — k_ntt_queue8: items of a limb-poly on one XCD
// by construction, the hand-off in that XCD's L2, round 5 with every control word on XCD-local atomics and pulls ahead.  It keeps the
// hand-off resident only at a look-ahead where second passes wait for first passes (-24 % fabric traffic, +27 % time) and loses 14 % at
// the look-ahead where nobody waits, because the L2 then no longer holds the hand-off: profiles/r05_ntt_queue.txt.  Removed in round 5.)

// ---- K1 x K5: last transform pass x evaluation key, both keys, all digits of one extended limb in one workgroup ---------
#define HM_NIP_MAX_TERMS 4
static_assert(HM_GENERIC || HM_NIP_MAX_TERMS <= 5, "mont32 lazy key products: five terms of less than 1.5q + 2^28 stay below 8q (hm_mac_add)");
#define HM_NIP_MAX_OUT 2
#define HM_NIP_MAX_LIMBS 4096   // per launch: the records live in a device table (cached by content), read through the scalar cache
struct HmNipLimb {                       // 32 bytes
  uint16_t x[HM_NIP_MAX_TERMS];          // per digit: limb of the first pass's hand-off (transformed digit) or of the evaluation-form operand
  uint16_t y[HM_NIP_MAX_OUT][HM_NIP_MAX_TERMS];
  uint16_t out[HM_NIP_MAX_OUT];
  uint16_t mod;                          // HM_NTT_NONE: empty slot
  uint16_t coeff_mask;                   // bit j: digit j goes through the transform (its x limb is relative to `hand`); bit 15 (HM_NIP_INV_OUT):
};                                       // the outputs leave as the first pass of their INVERSE transform (round 5)
#define HM_NIP_INV_OUT 0x8000u
struct HmNipArgs {
  const uint64_t *hand;   // first-pass hand-off (written by k_ntt_col just before)
  const uint64_t *x;      // evaluation-form operands (a digit's own limbs)
  const uint64_t *y;      // evaluation key
  uint64_t *out;
  const HmW *tw, *twist;
  const HmW *tw_inv, *twist_inv;   // inverse tables (HM_NIP_INV_OUT limbs)
  const HmMod *mods;
  uint32_t logN, n_limbs, logG, n_terms;
  const HmNipLimb *limb;  // device, [n_limbs]
  uint32_t x_galois;      // the XG kernels: evaluation-form operands are read through X -> X^x_galois
};
typedef const HmNipLimb __attribute__((address_space(4))) *HmConstNipLimb;
#ifndef HM_NIP_WAVES
#define HM_NIP_WAVES 3   // 64 accumulator registers on top of the pass: 168 VGPRs, three workgroups per CU
#endif
#ifndef HM_NIP_WIDE
#define HM_NIP_WIDE 0      // 1: 128-bit accumulators, one reduction per output (128 registers: HM_NIP_WAVES = 2)
#endif
#ifndef HM_NIP_PREFETCH
#define HM_NIP_PREFETCH 0  // 1: all key words of a digit requested before its transform (needs HM_NIP_WAVES = 2: 64 more registers)
#endif
#ifndef HM_NIP_MAC_CH
#define HM_NIP_MAC_CH 2   // access units of key words in flight per multiply-accumulate step (2: +0.6 % over 1; 4 needs two waves per SIMD and gains nothing)
#endif
// cache policy (hm_gld2's AUX bits: 2 = nt) of the once-read operand loads and of the output stores: streamed data marked non-temporal
// leaves the L2 to the key words, which the ops of a batch share
#ifndef HM_NIP_LD_AUX
#define HM_NIP_LD_AUX 0
#endif
#ifndef HM_NIP_ST_AUX
#define HM_NIP_ST_AUX 0
#endif
template <int K> struct HmNipKey { static constexpr int value = K; };
namespace hm16 {
#include "hm_nip_body.inl"
}
#undef HM_EPT
#define HM_EPT 8
namespace hm8 {
#include "hm_nip_body.inl"
}
#undef HM_EPT
#define HM_EPT 16
template <int OUTS, int INVOUT, bool XG = false>
__global__ void __launch_bounds__((1 << HM_TL_ROW) / HM_EPT) __attribute__((amdgpu_waves_per_eu(HM_NIP_WAVES))) k_ntt_row_ip(HmNipArgs a) {
  hm16::hm_nip_body<OUTS, INVOUT, HM_TL_ROW, XG>(a);
}
// The same in the small-launch geometry (round 5): one op at a time the launch is 800 workgroups on 768 slots (155 VGPRs: three 256-thread
// workgroups per CU) and pays a second, nearly empty round.  512-thread workgroups with 8 coefficients per thread halve the serial work of
// a workgroup and the accumulator registers per thread: the last round's idle time halves with them.  N = 2^16.
template <int OUTS, int INVOUT, bool XG = false>
__global__ void __launch_bounds__((1 << HM_TL_ROW) / 8) k_ntt_row_ip8(HmNipArgs a) {
  hm8::hm_nip_body<OUTS, INVOUT, HM_TL_ROW, XG>(a);
}
// (The same on HALF tiles — hm_nip_body's TLR = 11: 1 792 equal pieces of work for one op instead of 896 — was built and measured level
// within the noise, +0.5 % / +1.2 % / -1.7 % one op at a time on three boxes, and is not instantiated: profiles/r05_late_ab.txt.)

#include "hm_bcol.h"
__global__ void __launch_bounds__(256) k_tensor(HmTensorArgs a) {
  const uint32_t N = 1u << a.logN;
  const uint32_t per_limb = N / 512;
  const uint32_t entry = blockIdx.x / per_limb, chunk = blockIdx.x % per_limb;
  if (entry >= a.n_limbs) return;
  const HmTensorLimb lb = a.limb[entry];
  const HmMod m = a.mods[lb.mod];
  const size_t x = (size_t)chunk * 512 + 2 * threadIdx.x;
  const ulonglong2 va = *reinterpret_cast<const ulonglong2 *>(a.a + (size_t)lb.a * N + x);
  const ulonglong2 vb = *reinterpret_cast<const ulonglong2 *>(a.b + (size_t)lb.b * N + x);
  const ulonglong2 vc = *reinterpret_cast<const ulonglong2 *>(a.c + (size_t)lb.c * N + x);
  const ulonglong2 vd = *reinterpret_cast<const ulonglong2 *>(a.d + (size_t)lb.d * N + x);
  uint64_t x0, x1, x2, y0, y1, y2;
  hm_tensor_one(va.x, vb.x, vc.x, vd.x, m, x0, x1, x2);
  hm_tensor_one(va.y, vb.y, vc.y, vd.y, m, y0, y1, y2);
  const ulonglong2 r0 = {x0, y0}, r1 = {x1, y1}, r2 = {x2, y2};
  *reinterpret_cast<ulonglong2 *>(a.o0 + (size_t)lb.o0 * N + x) = r0;
  *reinterpret_cast<ulonglong2 *>(a.o1 + (size_t)lb.o1 * N + x) = r1;
  *reinterpret_cast<ulonglong2 *>(a.o2 + (size_t)lb.o2 * N + x) = r2;
}

// 16-byte units per thread of the element-wise kernel, 512 coefficients apart, all loads in flight before the first use.  1: 89 600 workgroups for a batch of
// ten hadd; 2 / 4 measured padd +3 / +5 %, pmult +1 %, hadd within the noise (tools/r06_ewe_ab.sh): not worth a second shape of the kernel's launch
#ifndef HM_EWE_UNITS
#define HM_EWE_UNITS 1
#endif
template <int OP>
__global__ void __launch_bounds__(256) k_ewe(HmEweArgs a) {
  const uint32_t N = 1u << a.logN;
  const uint32_t per_limb = N / (512 * HM_EWE_UNITS);  // blocks per limb: 256 threads x HM_EWE_UNITS 16-byte units, 512 coefficients apart
  uint32_t entry = blockIdx.x / per_limb, chunk = blockIdx.x % per_limb;
  if (entry >= a.n_limbs) return;
  const HmEweLimb lb = a.limb[entry];
  const HmMod m = a.mods[lb.mod];
  const HmTw k = a.k[entry];
  const size_t x0 = (size_t)chunk * (512 * HM_EWE_UNITS) + 2 * threadIdx.x;
  constexpr int uses = hm_ewe_uses(OP);
  ulonglong2 va[HM_EWE_UNITS], vb[HM_EWE_UNITS], vc[HM_EWE_UNITS], vd[HM_EWE_UNITS];
#pragma unroll
  for (int u = 0; u < HM_EWE_UNITS; ++u) {   // every load of the thread in flight before the first use
    const size_t x = x0 + 512 * u;
    va[u] = vb[u] = vc[u] = vd[u] = ulonglong2{0, 0};
    if (uses & 1) va[u] = *reinterpret_cast<const ulonglong2 *>(a.a + (size_t)lb.a * N + x);
    if (uses & 2) vb[u] = *reinterpret_cast<const ulonglong2 *>(a.b + (size_t)lb.b * N + x);
    if (uses & 4) vc[u] = *reinterpret_cast<const ulonglong2 *>(a.c + (size_t)lb.c * N + x);
    if (uses & 8) vd[u] = *reinterpret_cast<const ulonglong2 *>(a.d + (size_t)lb.d * N + x);
  }
#pragma unroll
  for (int u = 0; u < HM_EWE_UNITS; ++u) {
    ulonglong2 r;
    r.x = hm_ewe_one<OP>(va[u].x, vb[u].x, vc[u].x, vd[u].x, k, m);
    r.y = hm_ewe_one<OP>(va[u].y, vb[u].y, vc[u].y, vd[u].y, k, m);
    *reinterpret_cast<ulonglong2 *>(a.out + (size_t)lb.out * N + x0 + 512 * u) = r;
  }
}

template <int TERMS, int OUTS>
__global__ void __launch_bounds__(256) k_inner_product(HmIpArgs a) {
  const uint32_t N = 1u << a.logN;
  const uint32_t per_limb = N / 512;
  const uint32_t entry = blockIdx.x / per_limb, chunk = blockIdx.x % per_limb;
  if (entry >= a.n_limbs) return;
  const HmIpLimb &lb = a.limb[entry];
  const HmMod m = a.mods[lb.mod];
  const size_t off = (size_t)chunk * 512 + 2 * threadIdx.x;
  ulonglong2 vx[TERMS], vy[OUTS][TERMS];
  // x through an automorphism (wave-uniform choice): the aligned pair that holds this unit's two sources, swapped when it arrives in the other order
  size_t offx = off;
  bool swap = false;
  if (a.x_galois > 1) {
    const uint32_t s = hm_auto_src((uint32_t)off, a.x_galois, a.logN);
    offx = s & ~1u; swap = s & 1u;
  }
#pragma unroll
  for (int j = 0; j < TERMS; ++j) {
    vx[j] = *reinterpret_cast<const ulonglong2 *>(a.x + (size_t)lb.x[j] * N + offx);
    if (swap) { const unsigned long long t = vx[j].x; vx[j].x = vx[j].y; vx[j].y = t; }
#pragma unroll
    for (int k = 0; k < OUTS; ++k) vy[k][j] = *reinterpret_cast<const ulonglong2 *>(a.y + (size_t)lb.y[k][j] * N + off);
  }
#pragma unroll
  for (int k = 0; k < OUTS; ++k) {
    hm_u128 s0 = 0, s1 = 0;
#pragma unroll
    for (int j = 0; j < TERMS; ++j) {
      s0 += (hm_u128)vx[j].x * vy[k][j].x;
      s1 += (hm_u128)vx[j].y * vy[k][j].y;
    }
    const ulonglong2 r = {hm_barrett(s0, m), hm_barrett(s1, m)};  // TERMS <= 4 products below 2^120: within 2^(k+63)
    *reinterpret_cast<ulonglong2 *>(a.out + (size_t)lb.out[k] * N + off) = r;
  }
}

// One kernel per input-basis size: a single kernel switching over n_in is allocated for its largest case (140 VGPRs
// at 32 inputs) and hipcc left the per-case input arrays in scratch; every problem of a launch has the same n_in.
template <int N_IN>
__global__ void __launch_bounds__(HM_BCONV_THREADS) k_bconv(HmBconvArgs a) {
  const auto &p = HM_CONST_PROB(a.prob)[blockIdx.z];
  const uint32_t t0 = blockIdx.y * a.chunk;
  if (t0 >= p.n_out) return;
  const uint32_t t1 = min(t0 + a.chunk, p.n_out);
  const uint32_t x = (blockIdx.x * HM_BCONV_THREADS + threadIdx.x) * HM_BCONV_CPT;
  if (x >= (1u << a.logN)) return;
  if (p.in_packed) hm_bconv_thread<N_IN, HM_BCONV_CPT, true>(p, a.logN, x, t0, t1);   // (wave-uniform: two copies of the body)
  else hm_bconv_thread<N_IN, HM_BCONV_CPT, false>(p, a.logN, x, t0, t1);
}
typedef void (*hm_bconv_kernel)(HmBconvArgs);
static const hm_bconv_kernel k_bconv_by_n_in[HM_BCONV_MAX_IN + 1] = {
    nullptr,
#define HM_K(n) k_bconv<n>,
    HM_K(1) HM_K(2) HM_K(3) HM_K(4) HM_K(5) HM_K(6) HM_K(7) HM_K(8) HM_K(9) HM_K(10) HM_K(11) HM_K(12) HM_K(13) HM_K(14)
    HM_K(15) HM_K(16) HM_K(17) HM_K(18) HM_K(19) HM_K(20) HM_K(21) HM_K(22) HM_K(23) HM_K(24) HM_K(25) HM_K(26)
    HM_K(27) HM_K(28) HM_K(29) HM_K(30) HM_K(31) HM_K(32)
#undef HM_K
};

// strided chunk copy used to pack / unpack the exchange buffers: chunk c copies `len` words
#define HM_MAX_CHUNKS 512
struct HmChunkArgs {
  const uint64_t *src;
  uint64_t *dst;
  uint32_t len, n_chunks;
  uint32_t src_off[HM_MAX_CHUNKS / 2], dst_off[HM_MAX_CHUNKS / 2];  // in units of `len` words... see launch
};
__global__ void __launch_bounds__(256) k_chunk_copy(HmChunkArgs a) {
  const uint32_t per = a.len / 512;  // blocks per chunk (256 threads x 2 words)
  const uint32_t chunk = blockIdx.x / per, part = blockIdx.x % per;
  if (chunk >= a.n_chunks) return;
  const size_t x = (size_t)part * 512 + 2 * threadIdx.x;
  const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(a.src + (size_t)a.src_off[chunk] * a.len + x);
  *reinterpret_cast<ulonglong2 *>(a.dst + (size_t)a.dst_off[chunk] * a.len + x) = v;
}

// column-block copy (round 4: the exchange in the transposed domain): block c copies the `cw` words [col, col + cw) of every row of a
// limb-poly-shaped array (rows of `stride` words) to another one: limb layout (stride 256) <-> compact staging (stride cw)
struct HmColArgs {
  const uint64_t *src;
  uint64_t *dst;
  uint32_t rows, cw, src_stride, dst_stride, n_chunks;
  // offset of the block's first element in 16-byte units (limb 65535 of N = 2^16 is past 2^32 words)
  uint32_t src_off[HM_MAX_CHUNKS / 2], dst_off[HM_MAX_CHUNKS / 2];
};
__global__ void __launch_bounds__(256) k_col_copy(HmColArgs a) {
  const uint32_t per = (a.rows * a.cw / 2 + 255) / 256;   // blocks per chunk, two words per thread
  const uint32_t chunk = blockIdx.x / per, part = blockIdx.x % per;
  if (chunk >= a.n_chunks) return;
  const uint32_t e = (part * 256 + threadIdx.x) * 2;      // element of the block
  if (e >= a.rows * a.cw) return;
  const uint32_t r = e / a.cw, x = e % a.cw;
  const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(a.src + 2 * (size_t)a.src_off[chunk] + (size_t)r * a.src_stride + x);
  *reinterpret_cast<ulonglong2 *>(a.dst + 2 * (size_t)a.dst_off[chunk] + (size_t)r * a.dst_stride + x) = v;
}

struct HmAutoArgs {
  const uint64_t *in;
  uint64_t *out;
  uint32_t logN, n_limbs, g;
  HmLimb limb[HM_MAX_LIMBS];
};
// 1024 outputs per workgroup, four per thread 256 apart: a wave's 64 consecutive outputs come from ONE aligned block of 64 inputs
// (the permutation is affine in the natural index and both sides are stored bit-reversed), so every read touches whole lines;
// the four gathers of a thread are in flight together (13.4 against 17.0 us per hrotate at batch 10: 5.5 TB/s)
__global__ void __launch_bounds__(256) k_automorph(HmAutoArgs a) {
  const uint32_t N = 1u << a.logN;
  const uint32_t per_limb = N / 1024;
  const uint32_t entry = blockIdx.x / per_limb;
  const uint32_t i0 = (blockIdx.x % per_limb) * 1024 + threadIdx.x;
  const HmLimb lb = a.limb[entry];
  const uint64_t *in = a.in + (size_t)lb.in * N;
  uint64_t v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = in[hm_auto_src(i0 + 256 * k, a.g, a.logN)];
  uint64_t *out = a.out + (size_t)lb.out * N + i0;
#pragma unroll
  for (int k = 0; k < 4; ++k) out[256 * k] = v[k];
}

struct HmFillArgs {
  uint64_t *out;
  const HmMod *mods;
  uint64_t seed;
  uint32_t logN, n_limbs;
  HmLimb limb[HM_MAX_LIMBS];
};
__global__ void __launch_bounds__(256) k_fill(HmFillArgs a) {
  const uint32_t N = 1u << a.logN;
  const uint32_t per_limb = N / 256;
  uint32_t entry = blockIdx.x / per_limb;
  uint32_t x = (blockIdx.x % per_limb) * 256 + threadIdx.x;
  const HmLimb lb = a.limb[entry];
  a.out[(size_t)lb.out * N + x] = hm_synth(a.seed + lb.aux, x, a.mods[lb.mod].q);
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct hm_ctx {
  hm::Params P;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_done = nullptr;
  HmW *d_tw_fwd = nullptr, *d_tw_inv = nullptr;     // [L+K][N] table entries (mont32: one word in Montgomery form; generic: value + Shoup companion)
  HmW *d_twist_fwd = nullptr, *d_twist_inv = nullptr;  // [L+K][N/256][3], hm::Params::make_twist
  HmMod *d_mods = nullptr;
  std::map<std::vector<uint32_t>, uint64_t *> bconv_tables;  // key: n_in, in_ids..., out_ids...
  std::map<std::string, void *> ntt_tables;                  // launch tables (device_table), key: their bytes
  int live_graphs = 0;                                        // captured graphs that reference the tables: no eviction while > 0
  // ... the graphs themselves: hm_destroy detaches them (a late hm_graph_destroy must not touch a freed context)
  std::vector<struct hm_graph *> graphs;
  bool capturing = false;
  std::string err;
  // one-launch transforms (k_ntt_fused): rendezvous words in HBM, a host-visible error word, and the switch
  HmNttSync *ntt_ws = nullptr;
  unsigned *err_host = nullptr, *err_dev = nullptr;
  bool small_ept8 = true;      // launches of at most small_limbs entries use the 8-coefficient geometry (N = 2^16)
  uint32_t small_mode = 3;     // which passes of a small launch use it: bit 0 COL, bit 1 ROW
  uint32_t small_limbs = 64;   // measured (tools/ntt_small_ab.py): 2-3 us per launch faster up to ~64 entries, equal at 115, slower from 128
  uint32_t fused_test_spread = 0;   // test hook: one-launch transforms take the agent-scope path
#if defined(HM_FUSED_TRACE)
  unsigned long long *fused_trace = nullptr;   // option "ntt_fused_trace" = a device address (development builds)
#endif
  uint32_t fused_test_timeout = 0;  // test hook: tile 0 of every limb-poly withholds its arrival and the spins are short: the rendezvous times out
  uint32_t fused_slots_per_xcd = 0;  // workgroups of the one-launch transform an XCD holds at once (hm_create: occupancy x CUs per XCD)
  bool fused_broken = false;        // a rendezvous timed out: no one-launch transforms any more, graphs that hold one refuse to replay
  bool capture_has_fused = false;   // the capture in progress recorded a one-launch transform
  // launches of up to this many entries (N = 2^16) run as ONE launch in the small-launch geometry (k_ntt_fused8): 2-5 us faster than two kernels up to ~100
  // limb-polys, slower from 128 (tools/ntt_fused_small_ab.py); 0 = off
  uint32_t fused_small = 96;
  uint32_t bcol_outs = 0;   // output limbs per workgroup of the fused conversion + first pass (1 | 2; 0 = by launch size)
  // k_bconv: blocks a launch should have at least before its outputs are cut into fewer, larger chunks (a block re-reads its inputs per chunk)
  uint32_t bconv_blocks = 3072;
  uint32_t bcol_merge = 1;  // small launches: the digits of a call run ONE kernel, the widest digit's (bconv_col_launch)
  // hm_replicate_limbs of a list with ONE owner (the rescale residues of a batch) and at least this many bytes, on >= 4 ranks: the owner scatters
  // one chunk to every peer and the peers exchange their chunks (each link carries 2 / (W - 1) of the list instead of all of it); 0 = never
  uint64_t replicate_split_bytes = 2u << 20;
  // limb-polys per launch pair of a two-kernel transform (at most HM_NTT_MAX_ENTRIES): the hand-off between the passes of a launch pair is
  // ntt_launch_entries x N x 8 bytes; what is written and read between a hand-off line's store and its load decides whether the load is
  // served by the Infinity Cache (256 MiB) or by HBM
  uint32_t ntt_launch_entries = HM_NTT_MAX_ENTRIES;
  uint32_t nip_small = 64;  // transform x key launches of at most this many limb records (N = 2^16) run in the small-launch geometry (k_ntt_row_ip8); 0 = off
  int n_cu = 256;
  // multi-GPU
  int rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  hm_exchange_fn ext_fn = nullptr;
  void *ext_user = nullptr;
  uint64_t *stage_send = nullptr, *stage_recv = nullptr;
  size_t stage_words = 0;
  // exchange / compute overlap inside one op: the exchanges run on a stream of their own (hm_exchange_stream), ordered against the
  // compute stream by marks (events)
  hipStream_t xstream = nullptr;
  bool xasync = false;
  hipEvent_t xdep = nullptr;
  std::vector<hipEvent_t> xmarks;
};
// the stream an exchange runs on; with the exchange stream on, it first waits for everything enqueued on the compute stream so far
static hm_status exchange_stream_begin(hm_ctx *c, hipStream_t *S) {
  *S = c->stream;
  if (!c->xasync) return HM_OK;
  hipError_t e = hipEventRecord(c->xdep, c->stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(c->xstream, c->xdep, 0);
  if (e != hipSuccess) { c->err = std::string("exchange stream: ") + hipGetErrorString(e); return HM_ERR_HIP; }
  *S = c->xstream;
  return HM_OK;
}

// RCCL is loaded lazily so that the library itself never depends on it being present
struct RcclApi {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;   // optional (counter "comm_ranks_seen"; a test double need not have it)
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static const char *rccl_load() {
  if (g_rccl.h) return nullptr;
  // prefer an RCCL that is already mapped into the process (PyTorch ships and loads its own librccl.so): paging in a
  // second ~0.5 GB copy took minutes on a cold box
  // HOMULATOR_RCCL_LIB: another build of RCCL, or the test double of tests/mock_rccl (ranks as threads on one GPU)
  void *h = nullptr;
  if (const char *path = getenv("HOMULATOR_RCCL_LIB")) {
    h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return "HOMULATOR_RCCL_LIB could not be loaded";
  }
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return "librccl.so not found";
#define HM_SYM(field, name) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); if (!g_rccl.field) return "RCCL symbol missing: " name;
  HM_SYM(GetUniqueId, "ncclGetUniqueId") HM_SYM(CommInitRank, "ncclCommInitRank") HM_SYM(CommDestroy, "ncclCommDestroy")
  HM_SYM(Send, "ncclSend") HM_SYM(Recv, "ncclRecv") HM_SYM(GroupStart, "ncclGroupStart") HM_SYM(GroupEnd, "ncclGroupEnd")
  HM_SYM(GetErrorString, "ncclGetErrorString")
#undef HM_SYM
  g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(h, "ncclCommCount"));
  g_rccl.h = h;
  return nullptr;
}

static thread_local std::string g_create_err;

static hm_status fail(hm_ctx *c, hm_status st, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_err = buf;
  return st;
}
#define HM_HIP(c, expr)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail((c), HM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

extern "C" const char *hm_version(void) { return "homulator-hip 0.1 (gfx950)"; }
extern "C" hm_status hm_capability(uint32_t logN, const char *name, uint64_t *value) {
  if (!name || !value) return HM_ERR_ARG;
  return hm_cap_by_name(logN, name, value) ? HM_ERR_ARG : HM_OK;
}
extern "C" const char *hm_last_error(const hm_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

extern "C" hm_status hm_create(hm_ctx **out, const hm_params *p) {
  if (!out || !p) return fail(nullptr, HM_ERR_ARG, "hm_create: null argument");
  *out = nullptr;
  std::unique_ptr<hm_ctx> c(new hm_ctx);
  try {
    c->P.init(p->logN, p->L, p->K, p->q, p->p, p->psi, HM_GENERIC != 0);   // (the mont32 build refuses a chain with a modulus that is not h 2^32 + 1)
  } catch (const std::exception &e) {
    return fail(nullptr, HM_ERR_ARG, "hm_create: %s", e.what());
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, HM_ERR_HIP, "hm_create: no HIP device available (%s); this backend has no CPU fallback",
                hipGetErrorString(e));
  if (p->device < 0 || p->device >= ndev) return fail(nullptr, HM_ERR_ARG, "hm_create: device %d out of range", p->device);
  c->device = p->device;
  hm_ctx *cc = c.get();
  HM_HIP(nullptr, hipSetDevice(cc->device));
  HM_HIP(nullptr, hipStreamCreateWithFlags(&cc->stream, hipStreamNonBlocking));
  HM_HIP(nullptr, hipEventCreate(&cc->ev0));
  HM_HIP(nullptr, hipEventCreate(&cc->ev1));
  const uint32_t M = cc->P.L + cc->P.K, N = cc->P.N;
  HM_HIP(nullptr, hipMalloc(&cc->d_tw_fwd, sizeof(HmW) * (size_t)M * N));
  HM_HIP(nullptr, hipMalloc(&cc->d_tw_inv, sizeof(HmW) * (size_t)M * N));
  HM_HIP(nullptr, hipMalloc(&cc->d_mods, sizeof(HmMod) * M));
  const size_t twistRow = (size_t)(N >> HM_ROW_LOG) * 3;
  HM_HIP(nullptr, hipMalloc(&cc->d_twist_fwd, sizeof(HmW) * M * twistRow));
  HM_HIP(nullptr, hipMalloc(&cc->d_twist_inv, sizeof(HmW) * M * twistRow));
  std::vector<HmW> tmp(N);
  for (uint32_t m = 0; m < M; ++m) {
    cc->P.make_twist(m, false, tmp.data());
    HM_HIP(nullptr, hipMemcpy(cc->d_twist_fwd + m * twistRow, tmp.data(), sizeof(HmW) * twistRow, hipMemcpyHostToDevice));
    cc->P.make_twist(m, true, tmp.data());
    HM_HIP(nullptr, hipMemcpy(cc->d_twist_inv + m * twistRow, tmp.data(), sizeof(HmW) * twistRow, hipMemcpyHostToDevice));
    cc->P.make_table(m, false, tmp.data());
    HM_HIP(nullptr, hipMemcpy(cc->d_tw_fwd + (size_t)m * N, tmp.data(), sizeof(HmW) * N, hipMemcpyHostToDevice));
    cc->P.make_table(m, true, tmp.data());
    HM_HIP(nullptr, hipMemcpy(cc->d_tw_inv + (size_t)m * N, tmp.data(), sizeof(HmW) * N, hipMemcpyHostToDevice));
  }
  HM_HIP(nullptr, hipMemcpy(cc->d_mods, cc->P.modc.data(), sizeof(HmMod) * M, hipMemcpyHostToDevice));
  HM_HIP(nullptr, hipMalloc(&cc->ntt_ws, sizeof(HmNttSync)));
  HM_HIP(nullptr, hipMemset(cc->ntt_ws, 0, sizeof(HmNttSync)));
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cc->device) == hipSuccess && prop.multiProcessorCount > 0) cc->n_cu = prop.multiProcessorCount;
  }
  // Co-residency guard of the one-launch transform: its rendezvous needs the N / 4096 = 16 workgroups of a limb-poly resident on one XCD
  // at a time.  The occupancy the runtime reports for the register-heaviest variant times the CUs of an XCD (an eighth of what the device
  // shows: under a CU mask or a partition mode this is conservative) must cover that; otherwise the form is switched off for the context
  // instead of degrading into time-outs (counter "ntt_fused_slots_per_xcd").
  {
    uint32_t slots = ~0u;
    auto probe = [&](auto kern) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, (1 << HM_TL_ROW) / 8, 0) != hipSuccess) nb = 0;
      slots = std::min(slots, (uint32_t)std::max(nb, 0) * (uint32_t)std::max(1, cc->n_cu / 8));
    };
    auto probeAll = [&](auto L1) {
      constexpr int LOG1 = decltype(L1)::value;
      probe(k_ntt_fused8<LOG1, false, 0, 1, true>); probe(k_ntt_fused8<LOG1, false, 0, 1, false>);
      probe(k_ntt_fused8<LOG1, false, 0, 3, true>); probe(k_ntt_fused8<LOG1, false, 0, 3, false>);
      probe(k_ntt_fused8<LOG1, false, 4, 3, true>); probe(k_ntt_fused8<LOG1, false, 4, 3, false>);
      probe(k_ntt_fused8<LOG1, true, 0, 2, true>);  probe(k_ntt_fused8<LOG1, true, 0, 2, false>);
      probe(k_ntt_fused8<LOG1, true, 6, 2, true>);  probe(k_ntt_fused8<LOG1, false, 0, 7, true>); probe(k_ntt_fused8<LOG1, false, 0, 7, false>);
    };
    if (cc->P.logN == 15) probeAll(std::integral_constant<int, 7>()); else probeAll(std::integral_constant<int, 8>());
    cc->fused_slots_per_xcd = slots;
    if (slots < (cc->P.N >> HM_TL_ROW) || !hm_caps(cc->P.logN).small_geometry) cc->fused_small = 0;
  }
  if (const char *e = getenv("HOMULATOR_REPLICATE_SPLIT")) cc->replicate_split_bytes = strtoull(e, nullptr, 10);
  if (const char *e = getenv("HOMULATOR_NTT_LAUNCH_ENTRIES")) cc->ntt_launch_entries = (uint32_t)std::max(8, atoi(e));
  if (const char *e = getenv("HOMULATOR_NIP_SMALL")) cc->nip_small = (uint32_t)std::max(0, atoi(e));
  if (const char *e = getenv("HOMULATOR_BCOL_OUTS")) cc->bcol_outs = (uint32_t)std::min(2, std::max(0, atoi(e)));
  if (const char *e = getenv("HOMULATOR_BCOL_MERGE")) cc->bcol_merge = atoi(e) != 0;
  if (const char *e = getenv("HOMULATOR_BCONV_BLOCKS")) cc->bconv_blocks = (uint32_t)std::max(1, atoi(e));
  HM_HIP(nullptr, hipHostMalloc(reinterpret_cast<void **>(&cc->err_host), 64, hipHostMallocMapped));
  memset(cc->err_host, 0, 64);
  HM_HIP(nullptr, hipHostGetDevicePointer(reinterpret_cast<void **>(&cc->err_dev), cc->err_host, 0));
  if (const char *e = getenv("HOMULATOR_NTT_FUSED_SMALL")) cc->fused_small = (uint32_t)std::min(HM_NTT_MAX_ENTRIES, std::max(0, atoi(e)));
  // the guard above wins over the environment
  if (cc->fused_slots_per_xcd < (cc->P.N >> HM_TL_ROW) || !hm_caps(cc->P.logN).small_geometry) cc->fused_small = 0;
  if (const char *e = getenv("HOMULATOR_NTT_SMALL_LIMBS")) { cc->small_limbs = (uint32_t)atoi(e); cc->small_ept8 = cc->small_limbs != 0; }
  *out = c.release();
  return HM_OK;
}

static void detach_graphs(hm_ctx *c);
extern "C" void hm_destroy(hm_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  detach_graphs(c);
  for (auto &kv : c->bconv_tables) (void)hipFree(kv.second);
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  (void)hipFree(c->stage_send);
  (void)hipFree(c->stage_recv);
  for (auto &kv : c->ntt_tables) (void)hipFree(kv.second);
  (void)hipFree(c->d_tw_fwd);
  (void)hipFree(c->d_tw_inv);
  (void)hipFree(c->d_twist_fwd);
  (void)hipFree(c->d_twist_inv);
  (void)hipFree(c->d_mods);
  (void)hipFree(c->ntt_ws);
  (void)hipHostFree(c->err_host);
  if (c->xstream) { (void)hipStreamSynchronize(c->xstream); (void)hipStreamDestroy(c->xstream); }
  if (c->xdep) (void)hipEventDestroy(c->xdep);
  for (hipEvent_t e : c->xmarks) (void)hipEventDestroy(e);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" hm_status hm_get_modulus(const hm_ctx *c, uint32_t m, uint64_t *q) {
  if (!c || !q || m >= c->P.L + c->P.K) return HM_ERR_ARG;
  *q = c->P.mod[m];
  return HM_OK;
}
extern "C" hm_status hm_get_psi(const hm_ctx *c, uint32_t m, uint64_t *psi) {
  if (!c || !psi || m >= c->P.L + c->P.K) return HM_ERR_ARG;
  *psi = c->P.psi[m];
  return HM_OK;
}

extern "C" hm_status hm_malloc(hm_ctx *c, size_t bytes, void **dptr) {
  if (!c || !dptr) return HM_ERR_ARG;
  HM_HIP(c, hipSetDevice(c->device));
  HM_HIP(c, hipMalloc(dptr, bytes));
  return HM_OK;
}
extern "C" hm_status hm_free(hm_ctx *c, void *dptr) {
  if (!c) return HM_ERR_ARG;
  HM_HIP(c, hipStreamSynchronize(c->stream));
  HM_HIP(c, hipFree(dptr));
  return HM_OK;
}
static hm_status check_device_error(hm_ctx *c);
extern "C" hm_status hm_memcpy_h2d(hm_ctx *c, void *dst, const void *src, size_t bytes) {
  if (!c) return HM_ERR_ARG;
  HM_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  HM_HIP(c, hipStreamSynchronize(c->stream));
  return HM_OK;
}
extern "C" hm_status hm_memcpy_d2h(hm_ctx *c, void *dst, const void *src, size_t bytes) {
  if (!c) return HM_ERR_ARG;
  HM_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  HM_HIP(c, hipStreamSynchronize(c->stream));
  return check_device_error(c);   // a launch whose rendezvous timed out must not be read back as success
}
extern "C" hm_status hm_memcpy_d2d(hm_ctx *c, void *dst, const void *src, size_t bytes) {
  if (!c) return HM_ERR_ARG;
  HM_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return HM_OK;
}
// a rendezvous of a one-launch transform timed out (the limb's workgroups were never co-resident): the outputs of that
// launch are invalid.  Reported at the next synchronisation; the context falls back to two-kernel transforms.
static hm_status check_device_error(hm_ctx *c) {
  if (c->err_host && *c->err_host) {
    const unsigned code = *c->err_host;
    *c->err_host = 0;
    c->fused_small = 0;       // the default one-launch form too: every later launch would stall and fail the same way
    c->fused_broken = true;   // graphs captured with one-launch transforms inside refuse to replay (hm_graph_launch)
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemsetAsync(c->ntt_ws, 0, sizeof(HmNttSync), c->stream);
    (void)hipStreamSynchronize(c->stream);
    const char *why = code == 1 ? "XCD-local rendezvous: the workgroups of a limb-poly were not co-resident"
                    : code == 2 ? "agent-scope rendezvous of a limb-poly spread over XCDs"
                                : "unknown";
    return fail(c, HM_ERR_HIP,
        "one-launch transform: wait %u timed out (%s); results of the last launches are invalid, the context now uses two-kernel transforms "
        "and refuses to replay graphs that hold one-launch transforms", code, why);
  }
  return HM_OK;
}
extern "C" hm_status hm_sync(hm_ctx *c) {
  if (!c) return HM_ERR_ARG;
  HM_HIP(c, hipStreamSynchronize(c->stream));
  if (c->xstream) HM_HIP(c, hipStreamSynchronize(c->xstream));
  return check_device_error(c);
}
extern "C" hm_status hm_exchange_stream(hm_ctx *c, int enable) {
  if (!c) return HM_ERR_ARG;
  HM_HIP(c, hipSetDevice(c->device));
  if (enable && !c->xstream) {
    HM_HIP(c, hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
    HM_HIP(c, hipEventCreateWithFlags(&c->xdep, hipEventDisableTiming));
  }
  if (!enable && c->xasync) {   // leaving the mode: the compute stream continues behind everything the exchange stream holds
    HM_HIP(c, hipEventRecord(c->xdep, c->xstream));
    HM_HIP(c, hipStreamWaitEvent(c->stream, c->xdep, 0));
  }
  c->xasync = enable != 0;
  return HM_OK;
}
extern "C" hm_status hm_exchange_mark(hm_ctx *c, uint32_t slot) {
  if (!c || slot >= 256) return HM_ERR_ARG;
  HM_HIP(c, hipSetDevice(c->device));
  while (c->xmarks.size() <= slot) {
    hipEvent_t e;
    HM_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->xmarks.push_back(e);
  }
  HM_HIP(c, hipEventRecord(c->xmarks[slot], c->xasync ? c->xstream : c->stream));
  return HM_OK;
}
extern "C" hm_status hm_exchange_wait(hm_ctx *c, uint32_t slot) {
  if (!c) return HM_ERR_ARG;
  if (slot >= c->xmarks.size()) return fail(c, HM_ERR_ARG, "hm_exchange_wait: mark %u was never set", slot);
  HM_HIP(c, hipStreamWaitEvent(c->stream, c->xmarks[slot], 0));
  return HM_OK;
}
extern "C" void *hm_stream(hm_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" hm_status hm_wait_for(hm_ctx *c, hm_ctx *producer) {
  if (!c || !producer) return HM_ERR_ARG;
  if (c == producer) return HM_OK;  // same stream: already ordered
  if (c->device != producer->device) return fail(c, HM_ERR_ARG, "hm_wait_for: contexts on different devices");
  HM_HIP(c, hipSetDevice(c->device));
  if (!producer->ev_done) HM_HIP(c, hipEventCreateWithFlags(&producer->ev_done, hipEventDisableTiming));
  HM_HIP(c, hipEventRecord(producer->ev_done, producer->stream));
  HM_HIP(c, hipStreamWaitEvent(c->stream, producer->ev_done, 0));
  return HM_OK;
}

extern "C" hm_status hm_set_option(hm_ctx *c, const char *name, uint64_t value) {
  if (!c || !name) return HM_ERR_ARG;
  if (!strcmp(name, "ntt_fused_test_spread")) { c->fused_test_spread = value != 0; return HM_OK; }
#if defined(HM_FUSED_TRACE)
  if (!strcmp(name, "ntt_fused_trace")) { c->fused_trace = reinterpret_cast<unsigned long long *>((uintptr_t)value); return HM_OK; }
#endif
  if (!strcmp(name, "ntt_fused_test_timeout")) { c->fused_test_timeout = value != 0; return HM_OK; }
  if (!strcmp(name, "ntt_fused_small")) {
    if (value > HM_NTT_MAX_ENTRIES) return fail(c, HM_ERR_ARG, "hm_set_option: ntt_fused_small above %d", HM_NTT_MAX_ENTRIES);
    if (value && c->fused_broken) return fail(c, HM_ERR_UNSUPPORTED,
        "hm_set_option: a rendezvous of this context has timed out: one-launch transforms stay off");
    if (value && !hm_caps(c->P.logN).small_geometry) return fail(c, HM_ERR_UNSUPPORTED, "hm_set_option: no one-launch transform at N = 2^%u", c->P.logN);
    if (value && c->fused_slots_per_xcd < (c->P.N >> HM_TL_ROW)) return fail(c, HM_ERR_UNSUPPORTED,
        "hm_set_option: an XCD holds %u workgroups of the one-launch transform, a limb-poly needs %u", c->fused_slots_per_xcd, c->P.N >> HM_TL_ROW);
    c->fused_small = (uint32_t)value;
    return HM_OK;
  }
  if (!strcmp(name, "bconv_col_merge")) { c->bcol_merge = value != 0; return HM_OK; }
  if (!strcmp(name, "bconv_blocks")) { c->bconv_blocks = (uint32_t)std::max<uint64_t>(1, value); return HM_OK; }
  if (!strcmp(name, "bconv_col_outs")) { if (value > 2) return fail(c, HM_ERR_ARG, "hm_set_option: bconv_col_outs is 0 (by launch size), 1 or 2");
      c->bcol_outs = (uint32_t)value; return HM_OK; }
  if (!strcmp(name, "replicate_split_bytes")) {
    // the threshold decides the SHAPE of a collective (one exchange or scatter + exchange): every rank must hold the same value.  It is
    // compared across the ranks once, when the communicator is made (verify_replicate_split); after that it is fixed.
    if ((c->comm || c->ext_fn) && value != c->replicate_split_bytes)
      return fail(c, HM_ERR_UNSUPPORTED,
          "hm_set_option: replicate_split_bytes is fixed once the communicator exists (set it, the same on every rank, before hm_comm_init_*)");
    c->replicate_split_bytes = value;
    return HM_OK;
  }
  if (!strcmp(name, "ntt_launch_entries")) { c->ntt_launch_entries = (uint32_t)std::max<uint64_t>(8, value); return HM_OK; }
  if (!strcmp(name, "nip_small_limbs")) { c->nip_small = (uint32_t)value; return HM_OK; }
  if (!strcmp(name, "ntt_small_mode")) { c->small_mode = (uint32_t)value & 3u; return HM_OK; }
  if (!strcmp(name, "ntt_small_limbs")) { c->small_ept8 = value != 0; c->small_limbs = (uint32_t)value; return HM_OK; }
  return fail(c, HM_ERR_ARG, "hm_set_option: unknown option %s", name);
}
extern "C" hm_status hm_get_counter(hm_ctx *c, const char *name, uint64_t *value) {
  if (!c || !name || !value) return HM_ERR_ARG;
  // arithmetic back-end of this context: 0 = mont32 (word-wise Montgomery on q = h 2^32 + 1), 1 = generic
  if (!strcmp(name, "arith")) { *value = HM_GENERIC; return HM_OK; }
  if (!strcmp(name, "ntt_fused_slots_per_xcd")) { *value = c->fused_slots_per_xcd; return HM_OK; }
  // the capability table of this context's ring size (hm_caps.h): what the host layer plans its fusions from
  if (!strncmp(name, "cap_", 4) && !hm_cap_by_name(c->P.logN, name, value)) return HM_OK;
  if (!strcmp(name, "ntt_fused_small")) { *value = c->fused_small; return HM_OK; }   // 0: the one-launch form is off (option, guard, or after a time-out)
  // the communicator as the library sees it: ranks_seen = ncclCommCount over RCCL (what the wire was set up for), the caller's figure otherwise
  if (!strcmp(name, "comm_world")) { *value = (uint64_t)c->world; return HM_OK; }
  if (!strcmp(name, "comm_transport")) { *value = c->comm ? 1 : c->ext_fn ? 2 : 0; return HM_OK; }
  if (!strcmp(name, "comm_ranks_seen")) {
    int n = c->world;
    if (c->comm && g_rccl.CommCount && g_rccl.CommCount(c->comm, &n) != ncclSuccess) return fail(c, HM_ERR_COMM, "ncclCommCount failed");
    *value = (uint64_t)n;
    return HM_OK;
  }
  if (!strcmp(name, "ntt_cross_xcd")) {
    HM_HIP(c, hipStreamSynchronize(c->stream));
    unsigned v = 0;
    HM_HIP(c, hipMemcpy(&v, &c->ntt_ws->stats[0], sizeof v, hipMemcpyDeviceToHost));
    *value = v;
    return check_device_error(c);
  }
  return fail(c, HM_ERR_ARG, "hm_get_counter: unknown counter %s", name);
}

struct hm_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  bool has_fused = false;    // holds one-launch transforms: unusable once a rendezvous of the owner has timed out
  hm_ctx *owner = nullptr;   // its kernel nodes keep device addresses of the owner's launch tables: destroy graphs before their context
};
static void detach_graphs(hm_ctx *c) {
  for (hm_graph *g : c->graphs) g->owner = nullptr;
  c->graphs.clear();
}
extern "C" hm_status hm_capture_begin(hm_ctx *c) {
  if (!c) return HM_ERR_ARG;
  if (c->ext_fn) return fail(c, HM_ERR_UNSUPPORTED, "hm_capture_begin: an external exchange transport cannot be captured");
  HM_HIP(c, hipSetDevice(c->device));
  c->capturing = true;  // kernel nodes of the graph keep the device addresses of the launch tables
  c->capture_has_fused = false;
  hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { c->capturing = false; return fail(c, HM_ERR_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(e)); }
  return HM_OK;
}
extern "C" hm_status hm_capture_end(hm_ctx *c, hm_graph **out) {
  if (!c || !out) return HM_ERR_ARG;
  hm_graph *g = new hm_graph;
  c->capturing = false;
  hipError_t e = hipStreamEndCapture(c->stream, &g->graph);
  if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return fail(c, HM_ERR_HIP, "hm_capture_end: %s", hipGetErrorString(e));
  }
  g->owner = c;
  g->has_fused = c->capture_has_fused;
  c->graphs.push_back(g);
  c->live_graphs++;   // the launch-table cache is pinned while this graph lives (released in hm_graph_destroy)
  *out = g;
  return HM_OK;
}
extern "C" hm_status hm_graph_launch(hm_ctx *c, hm_graph *g) {
  if (!c || !g) return HM_ERR_ARG;
  if (g->owner != c) return fail(c, HM_ERR_ARG, "hm_graph_launch: the graph was captured from another (or a destroyed) context");
  if (g->has_fused && c->fused_broken) return fail(c, HM_ERR_UNSUPPORTED,
      "hm_graph_launch: the graph holds one-launch transforms and a rendezvous of this context has timed out: capture the plan again "
      "(it now uses two-kernel transforms)");
  HM_HIP(c, hipGraphLaunch(g->exec, c->stream));
  return HM_OK;
}
extern "C" void hm_graph_destroy(hm_graph *g) {
  if (!g) return;
  if (g->owner) {
    if (g->owner->live_graphs > 0) g->owner->live_graphs--;
    auto &v = g->owner->graphs;
    v.erase(std::remove(v.begin(), v.end(), g), v.end());
  }
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
}

extern "C" hm_status hm_timer_start(hm_ctx *c) {
  if (!c) return HM_ERR_ARG;
  HM_HIP(c, hipEventRecord(c->ev0, c->stream));
  return HM_OK;
}
extern "C" hm_status hm_timer_stop(hm_ctx *c, uint64_t *ns) {
  if (!c || !ns) return HM_ERR_ARG;
  HM_HIP(c, hipEventRecord(c->ev1, c->stream));
  HM_HIP(c, hipEventSynchronize(c->ev1));
  if (hm_status st = check_device_error(c)) return st;
  float ms = 0;
  HM_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *ns = (uint64_t)((double)ms * 1e6);
  return HM_OK;
}

// ------------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------------
static inline uint32_t limb_at(const uint32_t *l, uint32_t i) { return l ? l[i] : i; }

static hm_status check_limbs(hm_ctx *c, const char *what, const uint32_t *l, uint32_t n) {
  for (uint32_t i = 0; l && i < n; ++i)
    if (l[i] > 0xFFFFu) return fail(c, HM_ERR_ARG, "%s: limb index %u exceeds 65535", what, l[i]);
  return HM_OK;
}
static hm_status check_mods(hm_ctx *c, const char *what, const uint32_t *m, uint32_t n) {
  if (!m) return fail(c, HM_ERR_ARG, "%s: mod_ids is null", what);
  for (uint32_t i = 0; i < n; ++i)
    if (m[i] >= c->P.L + c->P.K) return fail(c, HM_ERR_ARG, "%s: mod id %u out of range", what, m[i]);
  return HM_OK;
}

// The kernels of a transform, by form: 0 = forward, 1 = forward with the fused epilogue (MODE 3), 2 = forward with the mix prologue and
// the epilogue (MODE 4 + 3), 3 = inverse, 4 = inverse with the input read through an automorphism (MODE 6), 5 = form 1 with the addend read
// through an automorphism (MODE 7).  `first` / `second` = the two pass kernels in the order they run (forward: COL then ROW,
// inverse: ROW then COL) in the 16-coefficient geometry; `first8` / `second8` = the same in the small-launch geometry and `one` = both
// passes in one launch ([0] out of place: non-temporal input loads, [1] in place); the small-launch forms exist where hm_caps says so
// (N = 2^15 and 2^16).
typedef void (*hm_ntt_kernel)(HmNttArgs);
typedef void (*hm_ntt_one_kernel)(HmNttArgs, HmNttFusedArgs);
struct HmNttKernels {
  hm_ntt_kernel first, second, first8, second8;
  hm_ntt_one_kernel one[2];
};
#define HM_NTT_FORMS 6
template <int LOG1>
static const HmNttKernels &ntt_kernels(int form) {
  static const HmNttKernels wide[HM_NTT_FORMS] = {
      {k_ntt_col<LOG1, false, 0>, k_ntt_row<false, 1>, nullptr, nullptr, {nullptr, nullptr}},
      {k_ntt_col<LOG1, false, 0>, k_ntt_row<false, 3>, nullptr, nullptr, {nullptr, nullptr}},
      {k_ntt_col<LOG1, false, 4>, k_ntt_row<false, 3>, nullptr, nullptr, {nullptr, nullptr}},
      {k_ntt_row<true, 0>, k_ntt_col<LOG1, true, 2>, nullptr, nullptr, {nullptr, nullptr}},
      {k_ntt_row<true, 6>, k_ntt_col<LOG1, true, 2>, nullptr, nullptr, {nullptr, nullptr}},
      {k_ntt_col<LOG1, false, 0>, k_ntt_row<false, 7>, nullptr, nullptr, {nullptr, nullptr}}};
  return wide[form];
}
// the ring sizes of the reference's configurations (config/config_4.cfg: N = 2^16, config_4_N15.cfg: N = 2^15) have all three forms
template <int LOG1>
static const HmNttKernels &ntt_kernels_all(int form) {
  static const HmNttKernels both[HM_NTT_FORMS] = {
      {k_ntt_col<LOG1, false, 0>, k_ntt_row<false, 1>, k_ntt_col8<LOG1, false, 0>, k_ntt_row8<false, 1>, {k_ntt_fused8<LOG1, false, 0, 1, true>,
          k_ntt_fused8<LOG1, false, 0, 1, false>}},
      {k_ntt_col<LOG1, false, 0>, k_ntt_row<false, 3>, k_ntt_col8<LOG1, false, 0>, k_ntt_row8<false, 3>, {k_ntt_fused8<LOG1, false, 0, 3, true>,
          k_ntt_fused8<LOG1, false, 0, 3, false>}},
      {k_ntt_col<LOG1, false, 4>, k_ntt_row<false, 3>, k_ntt_col8<LOG1, false, 4>, k_ntt_row8<false, 3>, {k_ntt_fused8<LOG1, false, 4, 3, true>,
          k_ntt_fused8<LOG1, false, 4, 3, false>}},
      {k_ntt_row<true, 0>, k_ntt_col<LOG1, true, 2>, k_ntt_row8<true, 0>, k_ntt_col8<LOG1, true, 2>, {k_ntt_fused8<LOG1, true, 0, 2, true>,
          k_ntt_fused8<LOG1, true, 0, 2, false>}},
      // (a gathered input is never read in place: one variant of the one-launch form serves)
      {k_ntt_row<true, 6>, k_ntt_col<LOG1, true, 2>, k_ntt_row8<true, 6>, k_ntt_col8<LOG1, true, 2>, {k_ntt_fused8<LOG1, true, 6, 2, true>,
          k_ntt_fused8<LOG1, true, 6, 2, true>}},
      {k_ntt_col<LOG1, false, 0>, k_ntt_row<false, 7>, k_ntt_col8<LOG1, false, 0>, k_ntt_row8<false, 7>, {k_ntt_fused8<LOG1, false, 0, 7, true>,
          k_ntt_fused8<LOG1, false, 0, 7, false>}}};
  return both[form];
}
template <> const HmNttKernels &ntt_kernels<8>(int form) { return ntt_kernels_all<8>(form); }
template <> const HmNttKernels &ntt_kernels<7>(int form) { return ntt_kernels_all<7>(form); }

// The launch-size thresholds of the small-launch forms are stated in limb-polys of N = 2^16 (16 workgroups each); what decides is the number
// of workgroups a launch brings, so a smaller ring admits proportionally more limb-polys (N = 2^15: twice as many; at most one launch's records)
static uint32_t small_entries(const hm_ctx *c, uint32_t limbs16) {
  if (!hm_caps(c->P.logN).small_geometry) return 0;
  return std::min<uint32_t>(HM_NTT_MAX_ENTRIES, limbs16 << (16 - std::min(16u, c->P.logN)));
}
static uint32_t fused_small_entries(const hm_ctx *c) { return small_entries(c, c->fused_small); }

template <int LOG1>
static void launch_ntt(hm_ctx *c, const HmNttArgs &a, int form, bool firstPassOnly, bool secondPassOnly = false) {
  // n_limbs = entries, a multiple of 8 (one group per XCD); one workgroup per 4096-coefficient tile of each pass (HM_TL_COL == HM_TL_ROW)
  const HmNttKernels &K = ntt_kernels<LOG1>(form);
  const bool inverse = form == 3 || form == 4;
  const dim3 grid(a.n_limbs * (c->P.N >> HM_TL_ROW)), block16((1 << HM_TL_ROW) / HM_EPT), block8((1 << HM_TL_ROW) / 8);
  // small launches (one round of workgroups on the chip): the 8-coefficient geometry halves the serial work per wave (small_mode:
  // bit 0 = the COL pass, bit 1 = the ROW pass; the hand-off between the passes is the same in both geometries)
  const bool small = K.first8 && c->small_ept8 && a.n_limbs <= small_entries(c, c->small_limbs);
  const bool col8 = small && (c->small_mode & 1), row8 = small && (c->small_mode & 2);
  const bool first8 = inverse ? row8 : col8, second8 = inverse ? col8 : row8;
  if (secondPassOnly) {   // the hand-off was written by a fused conversion + first pass (bconv_col_launch)
    hipLaunchKernelGGL(second8 ? K.second8 : K.second, grid, second8 ? block8 : block16, 0, c->stream, a);
    return;
  }
  if (firstPassOnly) {    // forward COL pass alone: hm_ntt_inner_product runs the ROW pass inside its own kernel
    hipLaunchKernelGGL(K.first, grid, block16, 0, c->stream, a);
    return;
  }
  // Both passes in one launch behind an XCD-local rendezvous.  Progress: a kernel's workgroups are dispatched in order, so it has at
  // most ONE partially dispatched limb-poly per XCD whose workgroups wait for siblings that have no slot yet (logG == 0: at most 15 of an
  // XCD's slots; hm_create checked that an XCD holds at least 16); every other resident workgroup belongs to a complete limb-poly and
  // finishes.  Up to eight such kernels in flight on one GPU (contexts, instances; HIP drives four hardware queues by default) cannot
  // starve one another; the spins are bounded all the same.
  if (K.one[0] && a.n_limbs <= fused_small_entries(c) && a.logG == 0) {
#if defined(HM_FUSED_TRACE)
    const HmNttFusedArgs f = {c->ntt_ws, c->err_dev, c->fused_test_spread, c->fused_test_timeout ? 1u << 10 : HM_SPIN_LIMIT, c->fused_test_timeout,
        c->fused_trace};
#else
    const HmNttFusedArgs f = {c->ntt_ws, c->err_dev, c->fused_test_spread, c->fused_test_timeout ? 1u << 10 : HM_SPIN_LIMIT, c->fused_test_timeout};
#endif
    if (c->capturing) c->capture_has_fused = true;
    bool inPlace = a.in == a.out;   // every limb-poly transformed onto itself: the input loads keep their lines for the hand-off
    for (uint32_t e = 0; inPlace && e < a.n_limbs; ++e) inPlace = a.limb[e].mod == HM_NTT_NONE || a.limb[e].in == a.limb[e].out;
    hipLaunchKernelGGL(K.one[inPlace ? 1 : 0], grid, block8, 0, c->stream, a, f);
    return;
  }
  hipLaunchKernelGGL(first8 ? K.first8 : K.first, grid, first8 ? block8 : block16, 0, c->stream, a);
  hipLaunchKernelGGL(second8 ? K.second8 : K.second, grid, second8 ? block8 : block16, 0, c->stream, a);
}

// Device copy of a launch's table (NTT entry constants, base-conversion problem records), cached by content; uploaded
// (synchronously) on first use.  Tables are never freed while a graph captured from this context may still replay them
// (hm_capture_begin pins the cache); a miss while the stream is capturing is an error, not a hidden synchronisation.
static hm_status device_table(hm_ctx *c, const void *data, size_t bytes, const void **out) {
  const std::string key(static_cast<const char *>(data), bytes);
  auto it = c->ntt_tables.find(key);
  if (it == c->ntt_tables.end()) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
      return fail(c, HM_ERR_UNSUPPORTED, "a launch table is missing while the stream is capturing: run the plan once before hm_capture_begin");
    // callers that never repeat a launch (tests): start over rather than grow without bound
    if (c->ntt_tables.size() >= 1024 && c->live_graphs == 0 && !c->capturing) {
      HM_HIP(c, hipStreamSynchronize(c->stream));
      for (auto &kv : c->ntt_tables) (void)hipFree(kv.second);
      c->ntt_tables.clear();
    }
    void *dev = nullptr;
    HM_HIP(c, hipMalloc(&dev, bytes));
    HM_HIP(c, hipMemcpy(dev, data, bytes, hipMemcpyHostToDevice));
    it = c->ntt_tables.emplace(key, dev).first;
  }
  *out = it->second;
  return HM_OK;
}
static hm_status ntt_table(hm_ctx *c, const std::vector<HmNttEntry> &tab, const HmNttEntry **out) {
  const void *p = nullptr;
  hm_status st = device_table(c, tab.data(), sizeof(HmNttEntry) * tab.size(), &p);
  *out = static_cast<const HmNttEntry *>(p);
  return st;
}

// operands of the fused forward transform: x = NTT(in [+ mix_k * mix]); out = (minuend - x) * k [+ addend [* addend_k]]
struct NttFused {
  const uint64_t *minuend = nullptr;
  const uint32_t *minuend_limbs = nullptr;
  const uint64_t *addend = nullptr;
  const uint32_t *addend_limbs = nullptr;
  const uint64_t *addend_k = nullptr;
  const uint64_t *mix = nullptr;
  const uint32_t *mix_limbs = nullptr;
  const uint64_t *mix_k = nullptr;
  const uint8_t *outPacked = nullptr;   // inverse: per limb-poly, store the split-30 packed form (hm_pack30)
  const uint32_t *inGalois = nullptr;   // inverse (round 6): per limb-poly, the input is read through X -> X^g (0 / 1: as stored)
  const uint32_t *addGalois = nullptr;  // fused forward (round 6): per limb-poly, the addend is read through X -> X^g (0 / 1: as stored)
  bool firstPassOnly = false;   // forward transform: run the COL pass only (the hand-off stays in `out`)
  bool secondPassOnly = false;  // forward transform: the hand-off is already in `out` (a fused conversion wrote it): run the ROW pass only
};

// common body of hm_ntt / hm_ntt_sub_scale / hm_ntt_mix_sub_scale.  `k`: inverse -> optional extra scale; fused forward ->
// the mandatory per-limb constant of the epilogue
static hm_status ntt_common(hm_ctx *c, const char *what, const uint64_t *in, const uint32_t *in_limbs, uint64_t *out,
                            const uint32_t *out_limbs, const uint32_t *mod_ids, uint32_t n, int inverse, const uint64_t *k,
                            const NttFused &f) {
  const bool fused = f.minuend != nullptr;
  hm_status st;
  if ((st = check_limbs(c, what, in_limbs, n)) || (st = check_limbs(c, what, out_limbs, n)) ||
      (st = check_limbs(c, what, f.minuend_limbs, n)) ||
      (st = check_limbs(c, what, f.mix_limbs, n)) || (st = check_mods(c, what, mod_ids, n)))
    return st;
  for (uint32_t g = 0; f.addend_limbs && g < n; ++g)
    if (f.addend_limbs[g] != HM_NO_LIMB && f.addend_limbs[g] >= 0xFFFFu) return fail(c, HM_ERR_ARG, "%s: addend limb index %u exceeds 65534", what,
        f.addend_limbs[g]);
  for (uint32_t g = 0; g < n; ++g) {
    const uint64_t q = c->P.mod[mod_ids[g]];
    if ((k && k[g] >= q) || (f.addend_k && f.addend_k[g] >= q) || (f.mix_k && f.mix_k[g] >= q))
      return fail(c, HM_ERR_ARG, "%s: constant [%u] is not reduced", what, g);
    if (f.addend_k && f.addend_k[g] == 0) return fail(c, HM_ERR_ARG, "%s: addend_k[%u] is zero (pass addend = NULL instead)", what, g);
  }
  // operands read through an automorphism (round 6): odd Galois elements below 2N; 0 and 1 mean "as stored"
  bool anyInGalois = false, anyAddGalois = false;
  for (uint32_t g = 0; g < n; ++g) {
    const uint32_t gi = f.inGalois ? f.inGalois[g] : 0, ga = f.addGalois ? f.addGalois[g] : 0;
    if ((gi && (!(gi & 1) || gi >= 2 * c->P.N)) || (ga && (!(ga & 1) || ga >= 2 * c->P.N)))
      return fail(c, HM_ERR_ARG, "%s: Galois element [%u] is not an odd number below 2N", what, g);
    anyInGalois |= gi > 1; anyAddGalois |= ga > 1;
  }
  if (anyInGalois && in == out) {   // a gathered input is read from OTHER tiles than the one a workgroup writes: no limb-poly the call writes may be one it gathers from
    std::vector<char> written;
    for (uint32_t g = 0; g < n; ++g) { const uint32_t l = limb_at(out_limbs, g); if (l >= written.size()) written.resize((size_t)l + 1, 0); written[l] = 1; }
    for (uint32_t g = 0; g < n; ++g) {
      const uint32_t l = limb_at(in_limbs, g);
      if (f.inGalois[g] > 1 && l < written.size() && written[l]) return fail(c, HM_ERR_ARG,
          "%s: limb-poly [%u] is read through an automorphism from a limb the call also writes (not in place, and not onto another entry's source)", what, g);
    }
  }
  if (anyInGalois && (!inverse || f.secondPassOnly)) return fail(c, HM_ERR_ARG, "%s: only the inverse transform reads its input through an automorphism", what);
  if (anyAddGalois && (!fused || f.mix || !f.addend)) return fail(c, HM_ERR_UNSUPPORTED,
      "%s: the addend is read through an automorphism by the fused forward transform without the mix prologue only", what);
  if (anyAddGalois && f.addend == out) {   // ... from other positions than the ones a workgroup writes: no output limb-poly may be a gathered addend
    std::vector<char> written;
    for (uint32_t g = 0; g < n; ++g) { const uint32_t l = limb_at(out_limbs, g); if (l >= written.size()) written.resize((size_t)l + 1, 0); written[l] = 1; }
    for (uint32_t g = 0; g < n; ++g) {
      if (f.addGalois[g] <= 1 || (f.addend_limbs && f.addend_limbs[g] == HM_NO_LIMB)) continue;
      const uint32_t l = limb_at(f.addend_limbs, g);
      if (l < written.size() && written[l]) return fail(c, HM_ERR_ARG, "%s: the addend of limb-poly [%u] is read through an automorphism from a limb the call writes", what, g);
    }
  }
  HM_HIP(c, hipSetDevice(c->device));
  // group limb-polys that share a modulus (see hm_block_map): G = the largest of 8, 4, 2 for which at least 7 of 8 limb-polys
  // of the call fall into full same-modulus groups (a batch of 10 ops x 2 keys has 20 limb-polys per modulus, a 50-limb sweep
  // of the extended basis one: G = 2 then costs nothing); leftovers of a modulus share groups with other leftovers
  std::map<uint32_t, std::vector<int>> byMod;
  for (uint32_t i = 0; i < n; ++i) byMod[mod_ids[i]].push_back((int)i);
  uint32_t logG = 1;
  // a call that will run as ONE launch (k_ntt_fused8) takes single limb-polys as groups: a kernel then has at most 15 workgroups per XCD
  // waiting for siblings that have no slot yet (launch_ntt), and the 50-limb sweep 56 entries instead of 64
  if (fused_small_entries(c) && !f.firstPassOnly && !f.secondPassOnly && (n + 7) / 8 * 8 <= fused_small_entries(c)) logG = 0;
#ifndef HM_NTT_MAX_LOGG
#define HM_NTT_MAX_LOGG 3
#endif
  for (uint32_t lg = HM_NTT_MAX_LOGG; lg >= 2; --lg) {
    size_t full = 0;
    for (auto &kv : byMod) full += kv.second.size() >> lg << lg;
    if (full * 8 >= (size_t)n * 7 && n >= (64u << lg)) { logG = lg; break; }
  }
  const uint32_t G = 1u << logG;
  std::vector<std::vector<int>> groups;  // indices into the caller's lists, -1 = empty
  {
    std::vector<int> rest;
    for (auto &kv : byMod) {
      auto &v = kv.second;
      size_t i = 0;
      for (; i + G <= v.size(); i += G) groups.emplace_back(v.begin() + i, v.begin() + i + G);
      rest.insert(rest.end(), v.begin() + i, v.end());   // leftovers of one modulus stay adjacent: they still share among themselves
    }
    for (size_t i = 0; i < rest.size(); i += G) {
      std::vector<int> g(rest.begin() + i, rest.begin() + std::min(rest.size(), i + G));
      g.resize(G, -1);
      groups.push_back(g);
    }
  }
  // As few launches as the kernel-argument segment allows (HM_NTT_MAX_ENTRIES records), of equal size; the constants
  // of a launch live in a device table cached by content (plans repeat their launches)
  // whole blocks of 8 groups (one per XCD)
  const uint32_t maxGroups = std::max(8u, std::min<uint32_t>(HM_NTT_MAX_ENTRIES, c->ntt_launch_entries) / G / 8 * 8);
  const uint32_t nLaunch = ((uint32_t)groups.size() + maxGroups - 1) / maxGroups;
  const uint32_t perLaunch = nLaunch ? (((uint32_t)groups.size() + nLaunch - 1) / nLaunch + 7) / 8 * 8 : 0;
  for (uint32_t base = 0; base < groups.size(); base += perLaunch) {
    const uint32_t ng = std::min<uint32_t>(perLaunch, (uint32_t)groups.size() - base);
    const uint32_t cnt = ((ng + 7) / 8) * 8 * G;  // entries: blocks of 8 groups = 8G entries
    std::vector<HmNttEntry> tab(cnt);
    memset(tab.data(), 0, sizeof(HmNttEntry) * cnt);
    HmNttArgs a;
    for (uint32_t e = 0; e < cnt; ++e) a.limb[e] = HmLimb{0, 0, (uint16_t)HM_NTT_NONE, 0};
    for (uint32_t kk = 0; kk < ng; ++kk) {
      for (uint32_t which = 0; which < G; ++which) {
        const int gi = groups[base + kk][which];
        if (gi < 0) continue;
        const uint32_t g = (uint32_t)gi, e = (kk / 8) * 8 * G + which * 8 + (kk % 8), m = mod_ids[g];
        const uint64_t q = c->P.mod[m];
        HmNttEntry &t = tab[e];
        a.limb[e] = HmLimb{(uint16_t)limb_at(in_limbs, g), (uint16_t)limb_at(out_limbs, g), (uint16_t)m, 0};
        if (inverse) {
          uint64_t v = c->P.modc[m].ninv;
          if (k) v = hm::mulmod(v, k[g], q);
          t.sc = hm_kconst(v, q);
          t.pack = f.outPacked && f.outPacked[g] ? 1 : 0;
          if (anyInGalois) t.galois = f.inGalois[g] ? f.inGalois[g] : 1u;
        } else if (fused) {
          if (anyAddGalois) t.galois = f.addGalois[g] ? f.addGalois[g] : 1u;
          t.sc = hm_kconst(k[g], q);
          a.limb[e].aux = (uint16_t)limb_at(f.minuend_limbs, g);
          t.alimb = f.addend_limbs && f.addend_limbs[g] == HM_NO_LIMB ? (uint16_t)HM_NTT_NONE : (uint16_t)limb_at(f.addend_limbs, g);
          if (f.addend_k) t.ak = hm_kconst(f.addend_k[g], q);
          if (f.mix) {
            t.mixlimb = (uint16_t)limb_at(f.mix_limbs, g);
            t.mixk = hm_kconst(f.mix_k[g], q);
          }
        }
      }
    }
    const HmNttEntry *dtab = nullptr;
    if (inverse || fused) {   // the plain forward transform needs no per-limb constants
      if ((st = ntt_table(c, tab, &dtab))) return st;
    }
    a.in = in; a.out = out;
    a.tw = inverse ? c->d_tw_inv : c->d_tw_fwd;
    a.twist = inverse ? c->d_twist_inv : c->d_twist_fwd;
    a.mods = c->d_mods;
    a.entry = dtab;
    a.minuend = f.minuend; a.addend = f.addend; a.mix = f.mix;
    a.logN = c->P.logN; a.n_limbs = cnt; a.logG = logG;
    const int form = inverse ? (anyInGalois ? 4 : 3) : fused ? (f.mix ? 2 : anyAddGalois ? 5 : 1) : 0;
    switch (c->P.logN - HM_ROW_LOG) {
    case 5: launch_ntt<5>(c, a, form, f.firstPassOnly, f.secondPassOnly); break;
    case 6: launch_ntt<6>(c, a, form, f.firstPassOnly, f.secondPassOnly); break;
    case 7: launch_ntt<7>(c, a, form, f.firstPassOnly, f.secondPassOnly); break;
    case 8: launch_ntt<8>(c, a, form, f.firstPassOnly, f.secondPassOnly); break;
    case 9: launch_ntt<9>(c, a, form, f.firstPassOnly, f.secondPassOnly); break;
    default: return fail(c, HM_ERR_UNSUPPORTED, "%s: logN %u", what, c->P.logN);
    }
    HM_HIP(c, hipGetLastError());
  }
  return HM_OK;
}

extern "C" hm_status hm_ntt(hm_ctx *c, const uint64_t *in, const uint32_t *in_limbs, uint64_t *out,
                            const uint32_t *out_limbs, const uint32_t *mod_ids, uint32_t n, int inverse,
                            const uint64_t *scale) {
  if (!c) return HM_ERR_ARG;
  if (!in || !out) return fail(c, HM_ERR_ARG, "hm_ntt: null buffer");
  if (scale && !inverse) return fail(c, HM_ERR_ARG, "hm_ntt: scale is only defined for the inverse transform");
  return ntt_common(c, "hm_ntt", in, in_limbs, out, out_limbs, mod_ids, n, inverse, scale, NttFused{});
}

extern "C" hm_status hm_ntt_second_pass(hm_ctx *c, uint64_t *buf, const uint32_t *limbs, const uint32_t *mod_ids, uint32_t n, int inverse,
                                        const uint64_t *scale) {
  if (!c) return HM_ERR_ARG;
  if (!buf) return fail(c, HM_ERR_ARG, "hm_ntt_second_pass: null buffer");
  if (scale && !inverse) return fail(c, HM_ERR_ARG, "hm_ntt_second_pass: scale is only defined for the inverse transform");
  NttFused f;
  f.secondPassOnly = true;
  return ntt_common(c, "hm_ntt_second_pass", buf, limbs, buf, limbs, mod_ids, n, inverse, scale, f);
}

extern "C" hm_status hm_ntt_ex(hm_ctx *c, const hm_ntt_desc *d) {
  if (!c) return HM_ERR_ARG;
  if (!d || !d->out || (!d->in && !d->second_pass_only)) return fail(c, HM_ERR_ARG, "hm_ntt_ex: null buffer");
  if (d->scale && !d->inverse) return fail(c, HM_ERR_ARG, "hm_ntt_ex: scale is only defined for the inverse transform");
  if (d->out_packed && !d->inverse) return fail(c, HM_ERR_ARG, "hm_ntt_ex: out_packed is only defined for the inverse transform");
  NttFused f;
  f.secondPassOnly = d->second_pass_only != 0;
  f.outPacked = d->out_packed;
  f.inGalois = d->in_galois;
  if (d->in_galois && !d->inverse) return fail(c, HM_ERR_ARG, "hm_ntt_ex: in_galois is only defined for the inverse transform");
  if (f.secondPassOnly) return ntt_common(c, "hm_ntt_ex", d->out, d->out_limbs, d->out, d->out_limbs, d->mod_ids, d->n, d->inverse, d->scale, f);
  return ntt_common(c, "hm_ntt_ex", d->in, d->in_limbs, d->out, d->out_limbs, d->mod_ids, d->n, d->inverse, d->scale, f);
}

extern "C" hm_status hm_ntt_sub_scale(hm_ctx *c, const uint64_t *in, const uint32_t *in_limbs, const uint64_t *minuend,
                                      const uint32_t *minuend_limbs, const uint64_t *addend, const uint32_t *addend_limbs,
                                      uint64_t *out, const uint32_t *out_limbs, const uint32_t *mod_ids, uint32_t n,
                                      const uint64_t *k) {
  if (!c) return HM_ERR_ARG;
  if (!in || !out || !minuend || !k) return fail(c, HM_ERR_ARG, "hm_ntt_sub_scale: null argument");
  NttFused f;
  f.minuend = minuend; f.minuend_limbs = minuend_limbs; f.addend = addend; f.addend_limbs = addend_limbs;
  return ntt_common(c, "hm_ntt_sub_scale", in, in_limbs, out, out_limbs, mod_ids, n, 0, k, f);
}

static hm_status bconv_col_launch(hm_ctx *c, const hm_bconv_desc *descs, uint32_t n_desc, const struct BcolMix *mix, uint32_t tile0 = 0, uint32_t n_tiles = 0);
struct BcolMix {   // the MODE 4 prologue of a fused conversion (x = conv + k * mix): per conversion, per output, the operand's limb and the constant
  const uint64_t *mix;
  const uint32_t *const *mix_limbs;
  const uint64_t *const *mix_k;
};
extern "C" hm_status hm_ntt_mix_sub_scale(hm_ctx *c, const hm_ntt_fused_desc *d) {
  if (!c) return HM_ERR_ARG;
  if (!d || (!d->in && !d->n_conv) || !d->out || !d->minuend || !d->k) return fail(c, HM_ERR_ARG, "hm_ntt_mix_sub_scale: null argument");
  if ((d->mix != nullptr) != (d->mix_k != nullptr)) return fail(c, HM_ERR_ARG, "hm_ntt_mix_sub_scale: mix and mix_k go together");
  if (d->addend_k && !d->addend) return fail(c, HM_ERR_ARG, "hm_ntt_mix_sub_scale: addend_k without addend");
  NttFused f;
  f.minuend = d->minuend; f.minuend_limbs = d->minuend_limbs; f.addend = d->addend; f.addend_limbs = d->addend_limbs;
  f.addend_k = d->addend_k; f.mix = d->mix; f.mix_limbs = d->mix_limbs; f.mix_k = d->mix_k;
  f.addGalois = d->addend_galois;
  if (d->addend_galois && (d->n_conv || d->mix || !d->addend)) return fail(c, HM_ERR_UNSUPPORTED,
      "hm_ntt_mix_sub_scale: addend_galois needs an addend and combines with neither the mix prologue nor conversions inside the transform");
  if (!d->n_conv) return ntt_common(c, "hm_ntt_mix_sub_scale", d->in, d->in_limbs, d->out, d->out_limbs, d->mod_ids, d->n, 0, d->k, f);
  // ---- some or all inputs are conversions that run inside their transform's first pass (round 4: the ModDown side)
  if (!d->conv) return fail(c, HM_ERR_ARG, "hm_ntt_mix_sub_scale: n_conv without descriptors");
  const uint32_t n = d->n;
  std::map<uint32_t, uint32_t> byOut;   // output limb -> limb-poly of the call
  for (uint32_t i = 0; i < n; ++i)
    if (!byOut.emplace(limb_at(d->out_limbs, i), i).second) return fail(c, HM_ERR_ARG, "hm_ntt_mix_sub_scale: output limb %u appears twice",
        limb_at(d->out_limbs, i));
  std::vector<char> covered(n, 0);
  std::vector<std::vector<uint32_t>> mixLimbs(d->n_conv);
  std::vector<std::vector<uint64_t>> mixK(d->n_conv);
  std::vector<const uint32_t *> mlp(d->n_conv);
  std::vector<const uint64_t *> mkp(d->n_conv);
  for (uint32_t j = 0; j < d->n_conv; ++j) {
    const hm_bconv_desc &cv = d->conv[j];
    if (cv.out != d->out) return fail(c, HM_ERR_ARG, "hm_ntt_mix_sub_scale: conv[%u].out must be the output buffer", j);
    if (!cv.out_ids || cv.n_out == 0 || cv.n_out > HM_BCONV_MAX_OUT) return fail(c, HM_ERR_ARG, "hm_ntt_mix_sub_scale: conv[%u]: bad output basis", j);
    for (uint32_t t = 0; t < cv.n_out; ++t) {
      auto it = byOut.find(limb_at(cv.out_limbs, t));
      if (it == byOut.end()) return fail(c, HM_ERR_ARG, "hm_ntt_mix_sub_scale: conv[%u] output %u feeds no limb-poly of the call", j, t);
      const uint32_t i = it->second;
      if (covered[i]) return fail(c, HM_ERR_ARG, "hm_ntt_mix_sub_scale: limb-poly %u is fed by two conversions", i);
      if (cv.out_ids[t] != d->mod_ids[i]) return fail(c, HM_ERR_ARG,
          "hm_ntt_mix_sub_scale: conv[%u] output %u has another modulus than the limb-poly it feeds", j, t);
      covered[i] = 1;
      mixLimbs[j].push_back(d->mix ? limb_at(d->mix_limbs, i) : 0);
      mixK[j].push_back(d->mix ? d->mix_k[i] : 0);
    }
    mlp[j] = mixLimbs[j].data(); mkp[j] = mixK[j].data();
  }
  const BcolMix bm = {d->mix, mlp.data(), mkp.data()};
  hm_status st;
  if ((st = bconv_col_launch(c, d->conv, d->n_conv, d->mix ? &bm : nullptr))) return st;
  // the covered limb-polys: last pass only (in place on the hand-off the conversions wrote); the others: the whole transform
  auto sub = [&](char want, bool second) -> hm_status {
    std::vector<uint32_t> il, ml, mnl, al, ol, mods;
    std::vector<uint64_t> mk, ak, kk;
    for (uint32_t i = 0; i < n; ++i) {
      if (covered[i] != want) continue;
      il.push_back(limb_at(d->in_limbs, i)); ol.push_back(limb_at(d->out_limbs, i)); mnl.push_back(limb_at(d->minuend_limbs, i));
      mods.push_back(d->mod_ids[i]); kk.push_back(d->k[i]);
      if (d->mix) { ml.push_back(limb_at(d->mix_limbs, i)); mk.push_back(d->mix_k[i]); }
      if (d->addend) al.push_back(limb_at(d->addend_limbs, i));
      if (d->addend_k) ak.push_back(d->addend_k[i]);
    }
    if (mods.empty()) return HM_OK;
    if (!second && !d->in) return fail(c, HM_ERR_ARG, "hm_ntt_mix_sub_scale: `in` is null but not every limb-poly is fed by a conversion");
    NttFused g = f;
    g.minuend_limbs = mnl.data(); g.addend_limbs = d->addend ? al.data() : nullptr; g.addend_k = d->addend_k ? ak.data() : nullptr;
    g.mix_limbs = d->mix ? ml.data() : nullptr; g.mix_k = d->mix ? mk.data() : nullptr;
    g.secondPassOnly = second;
    if (second) { g.mix = nullptr; g.mix_limbs = nullptr; g.mix_k = nullptr; }   // the prologue ran inside the conversion kernel
    return ntt_common(c, "hm_ntt_mix_sub_scale", second ? d->out : d->in, second ? ol.data() : il.data(), d->out, ol.data(), mods.data(),
        (uint32_t)mods.size(), 0, kk.data(), g);
  };
  if ((st = sub(1, true))) return st;
  return sub(0, false);
}

extern "C" hm_status hm_tensor(hm_ctx *c, const uint64_t *pa, const uint32_t *la, const uint64_t *pb, const uint32_t *lb,
                               const uint64_t *pc, const uint32_t *lc, const uint64_t *pd, const uint32_t *ld, uint64_t *o0,
                               const uint32_t *l0, uint64_t *o1, const uint32_t *l1, uint64_t *o2, const uint32_t *l2,
                               const uint32_t *mod_ids, uint32_t n) {
  if (!c) return HM_ERR_ARG;
  if (!pa || !pb || !pc || !pd || !o0 || !o1 || !o2) return fail(c, HM_ERR_ARG, "hm_tensor: null buffer");
  hm_status st;
  const uint32_t *lists[7] = {la, lb, lc, ld, l0, l1, l2};
  for (auto l : lists)
    if ((st = check_limbs(c, "hm_tensor", l, n))) return st;
  if ((st = check_mods(c, "hm_tensor", mod_ids, n))) return st;
  HM_HIP(c, hipSetDevice(c->device));
  for (uint32_t base = 0; base < n; base += HM_MAX_LIMBS) {
    const uint32_t cnt = std::min<uint32_t>(HM_MAX_LIMBS, n - base);
    HmTensorArgs a;
    a.a = pa; a.b = pb; a.c = pc; a.d = pd; a.o0 = o0; a.o1 = o1; a.o2 = o2;
    a.mods = c->d_mods; a.logN = c->P.logN; a.n_limbs = cnt;
    for (uint32_t i = 0; i < cnt; ++i) {
      const uint32_t g = base + i;
      a.limb[i] = HmTensorLimb{(uint16_t)limb_at(la, g), (uint16_t)limb_at(lb, g), (uint16_t)limb_at(lc, g), (uint16_t)limb_at(ld, g),
                               (uint16_t)limb_at(l0, g), (uint16_t)limb_at(l1, g), (uint16_t)limb_at(l2, g), (uint16_t)mod_ids[g]};
    }
    hipLaunchKernelGGL(k_tensor, dim3(cnt * (c->P.N / 512)), dim3(256), 0, c->stream, a);
    HM_HIP(c, hipGetLastError());
  }
  return HM_OK;
}

extern "C" hm_status hm_automorph(hm_ctx *c, const uint64_t *in, const uint32_t *in_limbs, uint64_t *out,
                                  const uint32_t *out_limbs, uint32_t n, uint32_t galois) {
  if (!c) return HM_ERR_ARG;
  if (!in || !out) return fail(c, HM_ERR_ARG, "hm_automorph: null buffer");
  if (!(galois & 1) || galois >= 2 * c->P.N) return fail(c, HM_ERR_ARG, "hm_automorph: galois element must be odd and < 2N");
  hm_status st;
  if ((st = check_limbs(c, "hm_automorph", in_limbs, n)) || (st = check_limbs(c, "hm_automorph", out_limbs, n))) return st;
  HM_HIP(c, hipSetDevice(c->device));
  for (uint32_t base = 0; base < n; base += HM_MAX_LIMBS) {
    const uint32_t cnt = std::min<uint32_t>(HM_MAX_LIMBS, n - base);
    HmAutoArgs a;
    a.in = in; a.out = out; a.logN = c->P.logN; a.n_limbs = cnt; a.g = galois;
    for (uint32_t i = 0; i < cnt; ++i)
      a.limb[i] = HmLimb{(uint16_t)limb_at(in_limbs, base + i), (uint16_t)limb_at(out_limbs, base + i), 0, 0};
    hipLaunchKernelGGL(k_automorph, dim3(cnt * (c->P.N / 1024)), dim3(256), 0, c->stream, a);
    HM_HIP(c, hipGetLastError());
  }
  return HM_OK;
}

template <int OP>
static void launch_ewe(hm_ctx *c, const HmEweArgs &a) {
  hipLaunchKernelGGL((k_ewe<OP>), dim3(a.n_limbs * (c->P.N / (512 * HM_EWE_UNITS))), dim3(256), 0, c->stream, a);
}

extern "C" hm_status hm_ewe(hm_ctx *c, int op, const uint64_t *pa, const uint32_t *la, const uint64_t *pb,
                            const uint32_t *lb, const uint64_t *pc, const uint32_t *lc, const uint64_t *pd,
                            const uint32_t *ld, uint64_t *out, const uint32_t *lo, const uint32_t *mod_ids,
                            uint32_t n, const uint64_t *k) {
  if (!c) return HM_ERR_ARG;
  if (op < 0 || op >= HM_EWE_NOPS) return fail(c, HM_ERR_ARG, "hm_ewe: bad opcode %d", op);
  const int uses = hm_ewe_uses(op);
  if (((uses & 1) && !pa) || ((uses & 2) && !pb) || ((uses & 4) && !pc) || ((uses & 8) && !pd) || !out)
    return fail(c, HM_ERR_ARG, "hm_ewe: opcode %d is missing an operand", op);
  const bool needs_k = op == HM_EWE_MUL_CONST || op == HM_EWE_SUB_SCALE || op == HM_EWE_SUB_SCALE_ADD;
  if (needs_k && !k) return fail(c, HM_ERR_ARG, "hm_ewe: opcode %d needs constants", op);
  hm_status st;
  if ((st = check_limbs(c, "hm_ewe", la, n)) || (st = check_limbs(c, "hm_ewe", lb, n)) ||
      (st = check_limbs(c, "hm_ewe", lc, n)) || (st = check_limbs(c, "hm_ewe", ld, n)) ||
      (st = check_limbs(c, "hm_ewe", lo, n)) || (st = check_mods(c, "hm_ewe", mod_ids, n)))
    return st;
  HM_HIP(c, hipSetDevice(c->device));
  for (uint32_t base = 0; base < n; base += HM_MAX_LIMBS) {
    const uint32_t cnt = std::min<uint32_t>(HM_MAX_LIMBS, n - base);
    HmEweArgs a;
    a.a = pa; a.b = pb; a.c = pc; a.d = pd; a.out = out;
    a.mods = c->d_mods; a.logN = c->P.logN; a.n_limbs = cnt; a.op = (uint32_t)op;
    for (uint32_t i = 0; i < cnt; ++i) {
      const uint32_t g = base + i, m = mod_ids[g];
      a.limb[i] = HmEweLimb{(uint16_t)limb_at(la, g), (uint16_t)limb_at(lb, g), (uint16_t)limb_at(lc, g),
                            (uint16_t)limb_at(ld, g), (uint16_t)limb_at(lo, g), (uint16_t)m};
      a.k[i] = HmTw{0, 0};
      if (needs_k) {
        if (k[g] >= c->P.mod[m]) return fail(c, HM_ERR_ARG, "hm_ewe: k[%u] is not reduced", g);
        a.k[i] = HmTw{k[g], hm::shoup(k[g], c->P.mod[m])};
      }
    }
    switch (op) {
    case HM_EWE_MUL: launch_ewe<HM_EWE_MUL>(c, a); break;
    case HM_EWE_MAC2: launch_ewe<HM_EWE_MAC2>(c, a); break;
    case HM_EWE_MAC_ADD: launch_ewe<HM_EWE_MAC_ADD>(c, a); break;
    case HM_EWE_ADD: launch_ewe<HM_EWE_ADD>(c, a); break;
    case HM_EWE_SUB: launch_ewe<HM_EWE_SUB>(c, a); break;
    case HM_EWE_MUL_CONST: launch_ewe<HM_EWE_MUL_CONST>(c, a); break;
    case HM_EWE_SUB_SCALE: launch_ewe<HM_EWE_SUB_SCALE>(c, a); break;
    case HM_EWE_COPY: launch_ewe<HM_EWE_COPY>(c, a); break;
    case HM_EWE_SUB_SCALE_ADD: launch_ewe<HM_EWE_SUB_SCALE_ADD>(c, a); break;
    }
    HM_HIP(c, hipGetLastError());
  }
  return HM_OK;
}

template <int T, int O>
static void launch_ip(hm_ctx *c, const HmIpArgs &a) {
  hipLaunchKernelGGL((k_inner_product<T, O>), dim3(a.n_limbs * (c->P.N / 512)), dim3(256), 0, c->stream, a);
}
extern "C" hm_status hm_inner_product(hm_ctx *c, const uint64_t *x, const uint32_t *x_limbs, const uint64_t *y,
                                      const uint32_t *y_limbs, uint64_t *out, const uint32_t *out_limbs,
                                      const uint32_t *mod_ids, uint32_t n, uint32_t n_terms, uint32_t n_out) {
  const hm_ip_desc d = {x, x_limbs, y, y_limbs, out, out_limbs, mod_ids, n, n_terms, n_out, 0};
  return hm_inner_product_ex(c, &d);
}
extern "C" hm_status hm_inner_product_ex(hm_ctx *c, const hm_ip_desc *d) {
  if (!c) return HM_ERR_ARG;
  if (!d) return fail(c, HM_ERR_ARG, "hm_inner_product: null argument");
  const uint64_t *x = d->x, *y = d->y;
  uint64_t *out = d->out;
  const uint32_t *x_limbs = d->x_limbs, *y_limbs = d->y_limbs, *out_limbs = d->out_limbs, *mod_ids = d->mod_ids;
  const uint32_t n = d->n, n_terms = d->n_terms, n_out = d->n_out;
  if (d->x_galois && (!(d->x_galois & 1) || d->x_galois >= 2 * c->P.N)) return fail(c, HM_ERR_ARG, "hm_inner_product: x_galois is not an odd number below 2N");
  if (!x || !y || !out || !x_limbs || !y_limbs || !out_limbs) return fail(c, HM_ERR_ARG, "hm_inner_product: null argument");
  if (d->x_galois > 1 && x == out) {   // gathered operands come from other positions than the ones a workgroup writes
    for (uint32_t i = 0; i < n * n_terms; ++i)
      for (uint32_t k = 0; k < n * n_out; ++k)
        if (x_limbs[i] == out_limbs[k]) return fail(c, HM_ERR_ARG, "hm_inner_product: an operand read through the automorphism is a limb the call writes");
  }
  if (n_terms == 0 || n_terms > HM_IP_MAX_TERMS || n_out == 0 || n_out > HM_IP_MAX_OUT)
    return fail(c, HM_ERR_ARG, "hm_inner_product: n_terms in [1,%d], n_out in [1,%d]", HM_IP_MAX_TERMS, HM_IP_MAX_OUT);
  hm_status st;
  if ((st = check_limbs(c, "hm_inner_product", x_limbs, n * n_terms)) || (st = check_limbs(c, "hm_inner_product", y_limbs, n * n_terms * n_out)) ||
      (st = check_limbs(c, "hm_inner_product", out_limbs, n * n_out)) || (st = check_mods(c, "hm_inner_product", mod_ids, n)))
    return st;
  HM_HIP(c, hipSetDevice(c->device));
  for (uint32_t base = 0; base < n; base += HM_IP_MAX_LIMBS) {
    const uint32_t cnt = std::min<uint32_t>(HM_IP_MAX_LIMBS, n - base);
    HmIpArgs a;
    a.x = x; a.y = y; a.out = out; a.mods = c->d_mods; a.logN = c->P.logN; a.n_limbs = cnt; a.n_terms = n_terms; a.n_out = n_out;
    a.x_galois = d->x_galois;
    for (uint32_t i = 0; i < cnt; ++i) {
      const uint32_t g = base + i;
      HmIpLimb &l = a.limb[i];
      l.mod = (uint16_t)mod_ids[g]; l.pad = 0;
      for (uint32_t j = 0; j < n_terms; ++j) l.x[j] = (uint16_t)x_limbs[g * n_terms + j];
      for (uint32_t k = 0; k < n_out; ++k) {
        l.out[k] = (uint16_t)out_limbs[g * n_out + k];
        for (uint32_t j = 0; j < n_terms; ++j) l.y[k][j] = (uint16_t)y_limbs[(g * n_out + k) * n_terms + j];
      }
    }
    switch (n_terms * 10 + n_out) {
    case 11: launch_ip<1, 1>(c, a); break; case 12: launch_ip<1, 2>(c, a); break;
    case 21: launch_ip<2, 1>(c, a); break; case 22: launch_ip<2, 2>(c, a); break;
    case 31: launch_ip<3, 1>(c, a); break; case 32: launch_ip<3, 2>(c, a); break;
    case 41: launch_ip<4, 1>(c, a); break; case 42: launch_ip<4, 2>(c, a); break;
    }
    HM_HIP(c, hipGetLastError());
  }
  return HM_OK;
}


// K1 x K5: out[i][k] = sum_j X_j[i] * y[i][k][j] with X_j[i] = NTT(x[i][j]) for the digits that go through
// the transform (x_is_coeff) and x[i][j] itself for a digit's own limbs.  Two launches: the COL pass of every transformed
// (limb, digit) into `hand`, then k_ntt_row_ip: ROW pass, product with both keys, accumulation over the digits in registers.
extern "C" hm_status hm_ntt_inner_product(hm_ctx *c, const hm_ntt_ip_desc *d) {
  if (!c) return HM_ERR_ARG;
  if (!d || !d->x || !d->x_limbs || !d->x_is_coeff || !d->y || !d->y_limbs || !d->out || !d->out_limbs || !d->mod_ids)
    return fail(c, HM_ERR_ARG, "hm_ntt_inner_product: null argument");
  const uint32_t n = d->n, T = d->n_terms, K = d->n_out;
  if (T == 0 || T > HM_NIP_MAX_TERMS || K == 0 || K > HM_NIP_MAX_OUT)
    return fail(c, HM_ERR_ARG, "hm_ntt_inner_product: n_terms in [1,%d], n_out in [1,%d]", HM_NIP_MAX_TERMS, HM_NIP_MAX_OUT);
  // the epilogue is the inverse ROW pass over the workgroup's own rows: tiles of whole rows at any N; tested at 2^15 and 2^16
  if (d->out_inverse && !hm_caps(c->P.logN).ip_inverse_out) {
    for (uint32_t i = 0; i < n; ++i)
      if (d->out_inverse[i]) return fail(c, HM_ERR_UNSUPPORTED, "hm_ntt_inner_product: out_inverse needs N = 2^15 or 2^16");
  }
  if (d->x_galois && (!(d->x_galois & 1) || d->x_galois >= 2 * c->P.N)) return fail(c, HM_ERR_ARG, "hm_ntt_inner_product: x_galois is not an odd number below 2N");
  if (d->x_galois > 1 && d->x == d->out) {   // gathered operands come from other positions than the ones a workgroup writes
    std::vector<char> written;
    for (uint32_t i = 0; i < n * K; ++i) { const uint32_t l = d->out_limbs[i]; if (l >= written.size()) written.resize((size_t)l + 1, 0); written[l] = 1; }
    for (uint32_t i = 0; i < n * T; ++i)
      if (!d->x_is_coeff[i] && d->x_limbs[i] < written.size() && written[d->x_limbs[i]]) return fail(c, HM_ERR_ARG,
          "hm_ntt_inner_product: an operand read through the automorphism is a limb the call writes");
  }
  hm_status st;
  if ((st = check_limbs(c, "hm_ntt_inner_product", d->x_limbs, n * T)) || (st = check_limbs(c, "hm_ntt_inner_product", d->y_limbs, n * T * K)) ||
      (st = check_limbs(c, "hm_ntt_inner_product", d->out_limbs, n * K)) || (st = check_mods(c, "hm_ntt_inner_product", d->mod_ids, n)))
    return st;
  // 1. first pass of every transformed (limb, digit): x -> hand; or conversion + first pass in one kernel: conv -> hand
  for (uint32_t i = 0; i < n * T; ++i)
    if (d->x_is_coeff[i]) {
      if (!d->hand || !d->hand_limbs) return fail(c, HM_ERR_ARG, "hm_ntt_inner_product: transformed digits need the hand-off buffer and its limb list");
      if (d->hand_limbs[i] > 0xFFFFu) return fail(c, HM_ERR_ARG, "hm_ntt_inner_product: limb index exceeds 65535");
    }
  if (d->n_conv) {
    if (!d->conv || !d->hand) return fail(c, HM_ERR_ARG, "hm_ntt_inner_product: conversions need their descriptors and the hand-off buffer");
    for (uint32_t k = 0; k < d->n_conv; ++k)
      if (d->conv[k].out != d->hand) return fail(c, HM_ERR_ARG, "hm_ntt_inner_product: conv[%u].out must be the hand-off buffer", k);
    if ((st = bconv_col_launch(c, d->conv, d->n_conv, nullptr))) return st;
  }
  // a launch may mix digits whose conversion runs inside their first pass with digits that arrive converted (Arch decides per
  // inner-product record: a digit of more than HM_BCOL_MAX_IN limbs keeps its own conversion): every transformed (limb, digit) whose
  // hand-off limb no conversion writes gets its first pass here
  std::vector<char> covered;
  for (uint32_t k = 0; k < d->n_conv; ++k)
    for (uint32_t t = 0; t < d->conv[k].n_out; ++t) {
      const uint32_t hl = limb_at(d->conv[k].out_limbs, t);
      if (hl >= covered.size()) covered.resize((size_t)hl + 1, 0);
      covered[hl] = 1;
    }
  std::vector<uint32_t> cin, chand, cmod;
  for (uint32_t i = 0; i < n; ++i)
    for (uint32_t j = 0; j < T; ++j)
      if (d->x_is_coeff[i * T + j]) {
        const uint32_t hl = d->hand_limbs[i * T + j];
        if (d->x_is_coeff[i * T + j] == 2) continue;   // the caller has run the first pass already (hm_bconv_col + the exchange back)
        if (hl < covered.size() && covered[hl]) continue;
        cin.push_back(d->x_limbs[i * T + j]); chand.push_back(hl); cmod.push_back(d->mod_ids[i]);
      }
  if (!cin.empty()) {
    NttFused f;
    f.firstPassOnly = true;
    if ((st = ntt_common(c, "hm_ntt_inner_product", d->x, cin.data(), d->hand, chand.data(), cmod.data(), (uint32_t)cin.size(), 0, nullptr, f))) return st;
  }
  HM_HIP(c, hipSetDevice(c->device));
  // 2. same-modulus groups (the ops of a batch bring one limb-poly per modulus each: they share the row twiddles AND the key).  Limbs whose
  // outputs leave as the first pass of their inverse transform (out_inverse): in the same launch (mont32) or in one of their own (generic)
  bool anyInv = false;
  for (uint32_t i = 0; d->out_inverse && i < n; ++i) anyInv |= d->out_inverse[i] != 0;
  const bool split = anyInv && HM_GENERIC != 0;
  for (int set = 0; set < (split ? 2 : 1); ++set) {
  auto inSet = [&](uint32_t i) { return !split || (d->out_inverse[i] ? 1 : 0) == set; };
  const int invForm = !anyInv ? 0 : split ? set : 2;
  std::map<uint32_t, std::vector<uint32_t>> byMod;
  uint32_t nSet = 0;
  for (uint32_t i = 0; i < n; ++i)
    if (inSet(i)) { byMod[d->mod_ids[i]].push_back(i); ++nSet; }
  if (!nSet) continue;
  uint32_t logG = 0;
  for (uint32_t lg = 3; lg >= 1; --lg) {
    size_t full = 0;
    for (auto &kv : byMod) full += kv.second.size() >> lg << lg;
    if (full * 8 >= (size_t)nSet * 7 && nSet >= (8u << lg)) { logG = lg; break; }
  }
  const uint32_t G = 1u << logG;
  std::vector<std::vector<int>> groups;
  {
    std::vector<int> rest;
    for (auto &kv : byMod) {
      auto &v = kv.second;
      size_t i = 0;
      for (; i + G <= v.size(); i += G) groups.emplace_back(v.begin() + i, v.begin() + i + G);
      rest.insert(rest.end(), v.begin() + i, v.end());
    }
    for (size_t i = 0; i < rest.size(); i += G) {
      std::vector<int> g(rest.begin() + i, rest.begin() + std::min(rest.size(), i + G));
      g.resize(G, -1);
      groups.push_back(g);
    }
  }
  // longest first: a limb whose digits all go through the transform (the special limbs of a ModUp: beta transforms) costs more than one
  // with a digit of its own; workgroups are dispatched in entry order, so the heavy ones start first and the partly filled last round of a
  // small launch holds light ones
  {
    auto weight = [&](const std::vector<int> &g) {
      uint32_t w = 0;
      for (int gi : g)
        if (gi >= 0)
          {
            for (uint32_t j = 0; j < T; ++j) w += d->x_is_coeff[(uint32_t)gi * T + j] ? 3 : 1;
            if (d->out_inverse && d->out_inverse[gi]) w += 2 * K;   // ... and an inverse first pass per output on top
          }
      return w;
    };
    std::stable_sort(groups.begin(), groups.end(), [&](const std::vector<int> &a, const std::vector<int> &b) { return weight(a) > weight(b); });
  }
  const uint32_t maxGroups = HM_NIP_MAX_LIMBS / G / 8 * 8;
  const uint32_t nLaunch = ((uint32_t)groups.size() + maxGroups - 1) / maxGroups;
  const uint32_t perLaunch = nLaunch ? (((uint32_t)groups.size() + nLaunch - 1) / nLaunch + 7) / 8 * 8 : 0;
  for (uint32_t base = 0; base < groups.size(); base += perLaunch) {
    const uint32_t ng = std::min<uint32_t>(perLaunch, (uint32_t)groups.size() - base);
    const uint32_t cnt = ((ng + 7) / 8) * 8 * G;
    HmNipArgs a;
    std::vector<HmNipLimb> recs(cnt);
    memset(recs.data(), 0, sizeof(HmNipLimb) * cnt);
    for (uint32_t e = 0; e < cnt; ++e) recs[e].mod = (uint16_t)HM_NTT_NONE;
    for (uint32_t kk = 0; kk < ng; ++kk)
      for (uint32_t which = 0; which < G; ++which) {
        const int gi = groups[base + kk][which];
        if (gi < 0) continue;
        const uint32_t i = (uint32_t)gi, e = (kk / 8) * 8 * G + which * 8 + (kk % 8);
        HmNipLimb &l = recs[e];
        l.mod = (uint16_t)d->mod_ids[i];
        for (uint32_t j = 0; j < T; ++j) {
          const bool tr = d->x_is_coeff[i * T + j] != 0;
          l.x[j] = (uint16_t)(tr ? d->hand_limbs[i * T + j] : d->x_limbs[i * T + j]);
          if (tr) l.coeff_mask |= (uint16_t)(1u << j);
          if (d->out_inverse && d->out_inverse[i]) l.coeff_mask |= (uint16_t)HM_NIP_INV_OUT;
          for (uint32_t k = 0; k < K; ++k) l.y[k][j] = (uint16_t)d->y_limbs[(i * K + k) * T + j];
        }
        for (uint32_t k = 0; k < K; ++k) l.out[k] = (uint16_t)d->out_limbs[i * K + k];
      }
    const void *dtab = nullptr;
    if ((st = device_table(c, recs.data(), sizeof(HmNipLimb) * cnt, &dtab))) return st;
    a.limb = static_cast<const HmNipLimb *>(dtab);
    a.hand = d->hand; a.x = d->x; a.y = d->y; a.out = d->out;
    a.tw = c->d_tw_fwd; a.twist = c->d_twist_fwd; a.tw_inv = c->d_tw_inv; a.twist_inv = c->d_twist_inv; a.mods = c->d_mods;
    a.logN = c->P.logN; a.n_limbs = cnt; a.logG = logG; a.n_terms = T;
    const dim3 grid(cnt * (c->P.N >> HM_TL_ROW)), block((1 << HM_TL_ROW) / HM_EPT);
    typedef void (*nip_kernel)(HmNipArgs);
#if HM_GENERIC
    static const nip_kernel kern[2][3] = {{k_ntt_row_ip<1, 0>, k_ntt_row_ip<1, 1>, nullptr}, {k_ntt_row_ip<2, 0>, k_ntt_row_ip<2, 1>, nullptr}};
    static const nip_kernel kern8[2][3] = {{k_ntt_row_ip8<1, 0>, k_ntt_row_ip8<1, 1>, nullptr}, {k_ntt_row_ip8<2, 0>, k_ntt_row_ip8<2, 1>, nullptr}};
#else
    static const nip_kernel kern[2][3] = {{k_ntt_row_ip<1, 0>, nullptr, k_ntt_row_ip<1, 2>}, {k_ntt_row_ip<2, 0>, nullptr, k_ntt_row_ip<2, 2>}};
    static const nip_kernel kern8[2][3] = {{k_ntt_row_ip8<1, 0>, nullptr, k_ntt_row_ip8<1, 2>}, {k_ntt_row_ip8<2, 0>, nullptr, k_ntt_row_ip8<2, 2>}};
#endif
    // (round 6) the same with the evaluation-form operands read through an automorphism (hm_ntt_ip_desc.x_galois)
#if HM_GENERIC
    static const nip_kernel kernG[2][3] = {{k_ntt_row_ip<1, 0, true>, k_ntt_row_ip<1, 1, true>, nullptr}, {k_ntt_row_ip<2, 0, true>, k_ntt_row_ip<2, 1, true>, nullptr}};
    static const nip_kernel kern8G[2][3] = {{k_ntt_row_ip8<1, 0, true>, k_ntt_row_ip8<1, 1, true>, nullptr}, {k_ntt_row_ip8<2, 0, true>, k_ntt_row_ip8<2, 1, true>, nullptr}};
#else
    static const nip_kernel kernG[2][3] = {{k_ntt_row_ip<1, 0, true>, nullptr, k_ntt_row_ip<1, 2, true>}, {k_ntt_row_ip<2, 0, true>, nullptr, k_ntt_row_ip<2, 2, true>}};
    static const nip_kernel kern8G[2][3] = {{k_ntt_row_ip8<1, 0, true>, nullptr, k_ntt_row_ip8<1, 2, true>}, {k_ntt_row_ip8<2, 0, true>, nullptr, k_ntt_row_ip8<2, 2, true>}};
#endif
    const bool xg = d->x_galois > 1;
    a.x_galois = d->x_galois;
    // small launches (one op at a time: 50 limb records = 800 workgroups on 768 slots of the wide form) take the small-launch geometry
    if (cnt <= small_entries(c, c->nip_small)) hipLaunchKernelGGL((xg ? kern8G : kern8)[K - 1][invForm], grid, dim3((1 << HM_TL_ROW) / 8), 0, c->stream, a);
    else hipLaunchKernelGGL((xg ? kernG : kern)[K - 1][invForm], grid, block, 0, c->stream, a);
    HM_HIP(c, hipGetLastError());
  }
  }   // sets
  return HM_OK;
}

// conversion + first transform pass as a call of its own, on a range of column tiles: what a rank runs on its column slice between the two
// transposed-domain exchanges (hm_limbs_to_colslices -> hm_bconv_col -> hm_colslices_to_limbs -> hm_ntt_inner_product with x_is_coeff = 2)
extern "C" hm_status hm_bconv_col(hm_ctx *c, const hm_bconv_desc *descs, uint32_t n_desc, uint32_t tile0, uint32_t n_tiles) {
  if (!c) return HM_ERR_ARG;
  if (!descs || n_desc == 0) return fail(c, HM_ERR_ARG, "hm_bconv_col: no problems");
  for (uint32_t k = 0; k < n_desc; ++k)
    if (descs[k].sub_from) return fail(c, HM_ERR_UNSUPPORTED, "hm_bconv_col: no epilogue on a fused conversion");
  return bconv_col_launch(c, descs, n_desc, nullptr, tile0, n_tiles);
}

extern "C" hm_status hm_bconv_consts(hm_ctx *c, const uint32_t *in_ids, uint32_t n_in, const uint32_t *out_ids,
                                     uint32_t n_out, uint64_t *qhat_inv, uint64_t *table) {
  if (!c) return HM_ERR_ARG;
  if (!in_ids || n_in == 0 || (n_out && !out_ids)) return fail(c, HM_ERR_ARG, "hm_bconv_consts: bad basis");
  hm_status st;
  if ((st = check_mods(c, "hm_bconv_consts", in_ids, n_in)) || (n_out && (st = check_mods(c, "hm_bconv_consts", out_ids, n_out))))
    return st;
  std::vector<uint64_t> qh(n_in), tb((size_t)n_in * std::max<uint32_t>(n_out, 1));
  c->P.bconv_consts(in_ids, n_in, out_ids, n_out, qh.data(), tb.data());
  if (qhat_inv) memcpy(qhat_inv, qh.data(), 8ull * n_in);
  if (table && n_out) memcpy(table, tb.data(), 8ull * n_in * n_out);
  return HM_OK;
}

extern "C" hm_status hm_bconv_batch(hm_ctx *c, const hm_bconv_desc *descs, uint32_t n_desc) {
  if (!c) return HM_ERR_ARG;
  if (!descs || n_desc == 0) return fail(c, HM_ERR_ARG, "hm_bconv_batch: no problems");
  HM_HIP(c, hipSetDevice(c->device));
  const uint32_t logN = descs[0].log_len ? descs[0].log_len : c->P.logN;
  if (logN < 8 || logN > c->P.logN) return fail(c, HM_ERR_ARG, "hm_bconv: log_len %u", logN);
  std::vector<HmBconvProb> probs(n_desc);
  for (uint32_t pi = 0; pi < n_desc; ++pi) {
    const hm_bconv_desc &d = descs[pi];
    if (!d.in || !d.out) return fail(c, HM_ERR_ARG, "hm_bconv: null buffer");
    if ((d.log_len ? d.log_len : c->P.logN) != logN) return fail(c, HM_ERR_ARG, "hm_bconv_batch: mixed log_len");
    if (d.n_in == 0 || d.n_in > HM_BCONV_MAX_IN) return fail(c, HM_ERR_ARG, "hm_bconv: n_in %u not in [1,%d]", d.n_in, HM_BCONV_MAX_IN);
    if (d.n_out == 0 || d.n_out > HM_BCONV_MAX_OUT) return fail(c, HM_ERR_ARG, "hm_bconv: n_out %u not in [1,%d]", d.n_out, HM_BCONV_MAX_OUT);
    hm_status st;
    if ((st = check_limbs(c, "hm_bconv", d.in_limbs, d.n_in)) || (st = check_limbs(c, "hm_bconv", d.out_limbs, d.n_out)) ||
        (st = check_mods(c, "hm_bconv", d.in_ids, d.n_in)) || (st = check_mods(c, "hm_bconv", d.out_ids, d.n_out)))
      return st;
    for (uint32_t i = 0; i < d.n_in; ++i)
      for (uint32_t t = 0; t < d.n_out; ++t)
        if (d.in_ids[i] == d.out_ids[t]) return fail(c, HM_ERR_ARG, "hm_bconv: modulus %u is in both bases", d.in_ids[i]);
    // conversion tables are cached per (input basis, output basis); built and uploaded on first use
    std::vector<uint32_t> key;
    key.push_back(d.n_in);
    key.insert(key.end(), d.in_ids, d.in_ids + d.n_in);
    key.insert(key.end(), d.out_ids, d.out_ids + d.n_out);
    auto it = c->bconv_tables.find(key);
    if (it == c->bconv_tables.end()) {
      std::vector<uint64_t> qh(d.n_in), tb((size_t)d.n_in * d.n_out);
      c->P.bconv_consts(d.in_ids, d.n_in, d.out_ids, d.n_out, qh.data(), tb.data());
      const uint32_t row = HM_BCONV_ROW(d.n_in);
      {  // device format: [n_out][row] (one output's factors contiguous and padded: wide scalar loads), Montgomery form, split-30 packed
        std::vector<uint64_t> tt((size_t)row * d.n_out, 0);
        for (uint32_t i = 0; i < d.n_in; ++i)
          for (uint32_t t = 0; t < d.n_out; ++t) tt[(size_t)t * row + i] = hm_bconv_entry(tb[(size_t)i * d.n_out + t], c->P.modc[d.out_ids[t]]);
        tb.swap(tt);
      }
      for (uint32_t t = 0; t < d.n_out; ++t) {  // behind the rows: {q, -q^-1} per output (HmQn)
        tb.push_back(c->P.modc[d.out_ids[t]].q);
        tb.push_back(c->P.modc[d.out_ids[t]].nqinv);
      }
      uint64_t *dev = nullptr;
      HM_HIP(c, hipMalloc(&dev, 8ull * tb.size()));
      HM_HIP(c, hipMemcpy(dev, tb.data(), 8ull * tb.size(), hipMemcpyHostToDevice));
      it = c->bconv_tables.emplace(key, dev).first;
    }
    HmBconvProb &p = probs[pi];
    memset(&p, 0, sizeof p);
    p.in = d.in; p.out = d.out; p.table = it->second; p.n_in = d.n_in; p.n_out = d.n_out;
    p.in_packed = d.in_packed ? 1u : 0u;
    p.qn = it->second + (size_t)HM_BCONV_ROW(d.n_in) * d.n_out;
    for (uint32_t i = 0; i < d.n_in; ++i) p.in_limb[i] = limb_at(d.in_limbs, i);
    for (uint32_t t = 0; t < d.n_out; ++t) {
      p.out_limb[t] = limb_at(d.out_limbs, t);
    }
    if (d.sub_from) {   // epilogue out = (sub_from - conv) * k [+ add]
      if (!d.sub_k) return fail(c, HM_ERR_ARG, "hm_bconv: the epilogue needs its constants (sub_k)");
      if (d.log_len && d.log_len != c->P.logN) return fail(c, HM_ERR_UNSUPPORTED, "hm_bconv: the epilogue works on whole limb-polys");
      hm_status est;
      if ((est = check_limbs(c, "hm_bconv", d.sub_from_limbs, d.n_out)) || (est = check_limbs(c, "hm_bconv", d.add_limbs, d.n_out))) return est;
      std::vector<HmTw> ek(d.n_out);
      for (uint32_t t = 0; t < d.n_out; ++t) {
        const uint64_t q = c->P.mod[d.out_ids[t]];
        if (d.sub_k[t] >= q) return fail(c, HM_ERR_ARG, "hm_bconv: sub_k[%u] is not reduced", t);
        ek[t] = HmTw{d.sub_k[t], hm::shoup(d.sub_k[t], q)};
        p.ep_a_limb[t] = limb_at(d.sub_from_limbs, t);
        p.ep_b_limb[t] = d.add ? limb_at(d.add_limbs, t) : 0;
      }
      const void *dk = nullptr;
      if ((est = device_table(c, ek.data(), sizeof(HmTw) * ek.size(), &dk))) return est;
      p.ep_a = d.sub_from; p.ep_b = d.add; p.ep_k = static_cast<const HmTw *>(dk);
    }
  }
  // one launch per distinct input-basis size (the digits of a ModUp differ only in the last, shorter digit), up to
  // HM_BCONV_MAX_PROB problems each; the problem records go into a device table cached by content (plans repeat)
  std::vector<char> done(n_desc, 0);
  for (uint32_t first = 0; first < n_desc; ++first) {
    if (done[first]) continue;
    const uint32_t n_in = probs[first].n_in;
    std::vector<HmBconvProb> grp;
    uint32_t max_out = 0;
    auto launch = [&]() -> hm_status {
      const void *dtab = nullptr;
      hm_status st = device_table(c, grp.data(), sizeof(HmBconvProb) * grp.size(), &dtab);
      if (st) return st;
      HmBconvArgs a;
      a.prob = static_cast<const HmBconvProb *>(dtab); a.logN = logN; a.n_prob = (uint32_t)grp.size();
      // output limbs per block: a block re-reads its N_IN input limbs for every chunk, so the chunk should be as large
      // as the launch allows while leaving >= ~4 rounds of blocks for the chip (3 blocks of 256 threads per CU)
      const uint32_t xb = std::max(1u, (1u << logN) / (HM_BCONV_THREADS * HM_BCONV_CPT));
      const uint32_t want = c->bconv_blocks;
      uint32_t nchunk = std::max<uint32_t>(1, (want + xb * a.n_prob - 1) / (xb * a.n_prob));
      nchunk = std::min(nchunk, (max_out + HM_BCONV_CHUNK / 2 - 1) / std::max(1, HM_BCONV_CHUNK / 2));  // chunks of >= 4 outputs
      nchunk = std::max<uint32_t>(1, nchunk);
      a.chunk = (max_out + nchunk - 1) / nchunk;
      dim3 grid(xb, (max_out + a.chunk - 1) / a.chunk, a.n_prob);
      hipLaunchKernelGGL(k_bconv_by_n_in[n_in], grid, dim3(HM_BCONV_THREADS), 0, c->stream, a);
      grp.clear(); max_out = 0;
      return HM_OK;
    };
    for (uint32_t pi = first; pi < n_desc; ++pi) {
      if (done[pi] || probs[pi].n_in != n_in) continue;
      done[pi] = 1;
      grp.push_back(probs[pi]);
      max_out = std::max(max_out, probs[pi].n_out);
      if (grp.size() == HM_BCONV_MAX_PROB) { hm_status st = launch(); if (st) return st; }
    }
    if (!grp.empty()) { hm_status st = launch(); if (st) return st; }
    HM_HIP(c, hipGetLastError());
  }
  return HM_OK;
}


// conversion + first transform pass in one kernel (see k_bconv_col).  Same descriptors as hm_bconv_batch; `out` receives the COL pass's
// hand-off of NTT(conversion), the form k_ntt_row_ip reads.  N = 2^16, n_in <= HM_BCOL_MAX_IN.
// tile0 / n_tiles: the column tiles (16 columns each) to work on — all of them (n_tiles = 0) or a rank's column slice (hm_bconv_col)
static hm_status bconv_col_launch(hm_ctx *c, const hm_bconv_desc *descs, uint32_t n_desc, const BcolMix *mix, uint32_t tile0, uint32_t n_tiles) {
  if (!c || !descs || n_desc == 0) return HM_ERR_ARG;
  const uint32_t allTiles = c->P.N >> HM_TL_COL;
  if (!n_tiles) { tile0 = 0; n_tiles = allTiles; }
  if ((n_tiles & (n_tiles - 1)) || tile0 % n_tiles || tile0 + n_tiles > allTiles) return fail(c, HM_ERR_ARG, "fused conversion: tile range [%u, %u) of %u",
      tile0, tile0 + n_tiles, allTiles);
  if (!hm_caps(c->P.logN).bcol_max_in) return fail(c, HM_ERR_UNSUPPORTED, "fused conversion: N = 2^15 or 2^16 only");
  HM_HIP(c, hipSetDevice(c->device));
  // output limbs per workgroup: two share the loaded and split inputs (+2 % hmult/s at batch 10), but halve the workgroups of a launch that
  // fills the chip only once or twice (one op at a time: -2 %): by launch size unless the option says otherwise
  // Small calls whose digits differ in width run as ONE launch of the widest digit's kernel (below) — and then with two outputs per workgroup: 920
  // workgroups on the chip's 1 024 slots (one round) where one output per workgroup made 1 840 (1.8 rounds): one op at a time +1.3 % on top of
  // the merge (gpurun_out: tools/r06_nout_ab.sh).  A small call of ONE width keeps one output per workgroup (level since the four-wave kernels).
  uint32_t NOUT = c->bcol_outs;
  size_t wgsAll = 0;
  bool widths[2][HM_BCONV_MAX_IN + 1] = {};
  uint32_t nWidths = 0;
  for (uint32_t pi = 0; pi < n_desc; ++pi) {
    wgsAll += (size_t)descs[pi].n_out * n_tiles;
    bool &w = widths[descs[pi].in_packed ? 1 : 0][std::min<uint32_t>(descs[pi].n_in, HM_BCONV_MAX_IN)];
    nWidths += !w;
    w = true;
  }
  const bool mayMerge = c->bcol_merge && !mix && wgsAll <= 4096 && nWidths > 1;
  if (!NOUT) NOUT = wgsAll > 4096 || mayMerge ? 2 : 1;
  std::map<uint32_t, std::vector<HmBcolProb>> byIn;   // key: n_in, + 256 for conversions whose inputs are stored packed (kernels of their own)
  bool farApart = false;
  // Round 6, small launches (one op at a time, a rank's share of a sharded op): digits of different width are launches of different kernels, one
  // behind the other, and each leaves the chip part empty — hmult 45/35/15: 1 120 workgroups of <15> on 1 024 slots (a second, nearly empty
  // round: 40.7 us) and then 720 of <5> (19.5 us).  When the whole call is small, the narrower digits run the WIDEST digit's kernel with zero
  // table columns for the inputs they do not have (the padded inputs re-read the digit's first limb: exact zeros are added): ONE launch of
  // 1 840 workgroups (920 with two outputs each).  More multiply-adds for the narrow digit, fewer rounds for the launch; option "bconv_col_merge"
  // (default 1; 0 = off).
  std::vector<uint32_t> kernelNin(n_desc);
  {
    uint32_t widest[2] = {0, 0};
    for (uint32_t pi = 0; pi < n_desc; ++pi) widest[descs[pi].in_packed ? 1 : 0] = std::max(widest[descs[pi].in_packed ? 1 : 0], descs[pi].n_in);
    const bool merge = mayMerge;
    for (uint32_t pi = 0; pi < n_desc; ++pi) {
      const uint32_t w = widest[descs[pi].in_packed ? 1 : 0];
      // (a digit runs the widest digit's kernel only inside one family: up to 15 limbs, or two input groups; and not for more than four times its own work)
      kernelNin[pi] = merge && (w <= HM_BCOL_ONE_GROUP || descs[pi].n_in > HM_BCOL_ONE_GROUP) && w <= 4 * descs[pi].n_in ? w : descs[pi].n_in;
    }
  }
  for (uint32_t pi = 0; pi < n_desc; ++pi) {
    const hm_bconv_desc &d = descs[pi];
    const uint32_t kn = kernelNin[pi];   // the input-basis size of the kernel this conversion runs (>= d.n_in)
    if (!d.in || !d.out || !d.in_ids || !d.out_ids) return fail(c, HM_ERR_ARG, "fused conversion: null argument");
    const uint32_t maxIn = mix ? hm_caps(c->P.logN).bcol_max_in_mix : hm_caps(c->P.logN).bcol_max_in;
    if (d.n_in == 0 || d.n_in > maxIn) return fail(c, HM_ERR_UNSUPPORTED, "fused conversion: n_in %u not in [1,%u]%s", d.n_in, maxIn, mix ?
        " (with the mix prologue)" : "");
    if (d.n_out == 0 || d.n_out > HM_BCONV_MAX_OUT) return fail(c, HM_ERR_ARG, "fused conversion: n_out %u not in [1,%d]", d.n_out, HM_BCONV_MAX_OUT);
    if (d.log_len && d.log_len != c->P.logN) return fail(c, HM_ERR_UNSUPPORTED, "fused conversion: whole limb-polys only");
    if (d.out != descs[0].out) return fail(c, HM_ERR_ARG, "fused conversion: one hand-off buffer per call");
    hm_status cst;
    if ((cst = check_limbs(c, "fused conversion", d.in_limbs, d.n_in)) || (cst = check_limbs(c, "fused conversion", d.out_limbs, d.n_out)) ||
        (cst = check_mods(c, "fused conversion", d.in_ids, d.n_in)) || (cst = check_mods(c, "fused conversion", d.out_ids, d.n_out)))
      return cst;
    for (uint32_t i = 0; i < d.n_in; ++i)
      for (uint32_t t = 0; t < d.n_out; ++t)
        if (d.in_ids[i] == d.out_ids[t]) return fail(c, HM_ERR_ARG, "fused conversion: modulus %u is in both bases", d.in_ids[i]);
    std::vector<uint32_t> key;
    key.push_back(d.n_in | (kn != d.n_in ? kn << 16 : 0u));   // (a table padded to a wider kernel's rows is a table of its own)
    key.insert(key.end(), d.in_ids, d.in_ids + d.n_in);
    key.insert(key.end(), d.out_ids, d.out_ids + d.n_out);
    auto it = c->bconv_tables.find(key);
    if (it == c->bconv_tables.end()) {
      std::vector<uint64_t> qh(d.n_in), tb((size_t)d.n_in * d.n_out);
      c->P.bconv_consts(d.in_ids, d.n_in, d.out_ids, d.n_out, qh.data(), tb.data());
      const uint32_t row = HM_BCONV_ROW(kn);
      std::vector<uint64_t> tt((size_t)row * d.n_out, 0);
      for (uint32_t i = 0; i < d.n_in; ++i)
        for (uint32_t t = 0; t < d.n_out; ++t) tt[(size_t)t * row + i] = hm_bconv_entry(tb[(size_t)i * d.n_out + t], c->P.modc[d.out_ids[t]]);
      for (uint32_t t = 0; t < d.n_out; ++t) { tt.push_back(c->P.modc[d.out_ids[t]].q); tt.push_back(c->P.modc[d.out_ids[t]].nqinv); }
      uint64_t *dev = nullptr;
      HM_HIP(c, hipMalloc(&dev, 8ull * tt.size()));
      HM_HIP(c, hipMemcpy(dev, tt.data(), 8ull * tt.size(), hipMemcpyHostToDevice));
      it = c->bconv_tables.emplace(key, dev).first;
    }
    HmBcolProb p;
    memset(&p, 0, sizeof p);
    p.in = d.in; p.table = it->second; p.qn = it->second + (size_t)HM_BCONV_ROW(kn) * d.n_out; p.n_in = d.n_in; p.n_out = d.n_out;
    p.in_packed = d.in_packed ? 1u : 0u;
    for (uint32_t i = 0; i < d.n_in; ++i) p.in_limb[i] = limb_at(d.in_limbs, i);
    {   // ONE buffer descriptor per conversion: the lowest input limb-poly is the base, the others are byte offsets from it (HmBcolProb::in_off)
      uint32_t lo = p.in_limb[0], hi = p.in_limb[0];
      for (uint32_t i = 1; i < d.n_in; ++i) { lo = std::min(lo, p.in_limb[i]); hi = std::max(hi, p.in_limb[i]); }
      if (((uint64_t)(hi - lo) + 1) << (c->P.logN + 3) > (1ull << 32)) { farApart = true; break; }   // inputs more than 4 GiB apart: the fallback below
      p.in_base = d.in + (size_t)lo * c->P.N;
      for (uint32_t i = 0; i < d.n_in; ++i) p.in_off[i] = (p.in_limb[i] - lo) << (c->P.logN + 3);
      // padded inputs: a valid limb-poly, zero table columns
      for (uint32_t i = d.n_in; i < kn; ++i) { p.in_limb[i] = p.in_limb[0]; p.in_off[i] = p.in_off[0]; }
    }
    for (uint32_t t = 0; t < d.n_out; ++t) { p.out_limb[t] = limb_at(d.out_limbs, t); p.out_mod[t] = d.out_ids[t]; }
    if (mix) {   // x = conv + k * mix before the first butterfly: constants in Shoup form, a device table cached by content
      std::vector<HmTw> mk(d.n_out);
      for (uint32_t t = 0; t < d.n_out; ++t) {
        const uint64_t q = c->P.mod[d.out_ids[t]], k = mix->mix_k[pi][t];
        if (k >= q) return fail(c, HM_ERR_ARG, "fused conversion: mix constant [%u][%u] is not reduced", pi, t);
        if (mix->mix_limbs[pi][t] > 0xFFFFu) return fail(c, HM_ERR_ARG, "fused conversion: limb index exceeds 65535");
        mk[t] = hm_kconst(k, q);
        p.mix_limb[t] = mix->mix_limbs[pi][t];
      }
      const void *dk = nullptr;
      hm_status st = device_table(c, mk.data(), sizeof(HmTw) * mk.size(), &dk);
      if (st) return st;
      p.mixk = static_cast<const HmTw *>(dk);
    }
    // (measured with the combination built: the fused ModDown conversion stays 2.5 % behind the separate one with packed inputs too —
    // profiles/r05_late_ab.txt — so the 120 instantiations it needs are not shipped)
    if (mix && d.in_packed) return fail(c, HM_ERR_UNSUPPORTED,
        "fused conversion: packed inputs and the mix prologue do not combine (convert from plain inputs)");
    byIn[kn + (d.in_packed ? 256u : 0u)].push_back(p);
  }
  if (farApart) {
    // A conversion whose input limb-polys are spread over more than 4 GiB of the buffer cannot be addressed from one descriptor with 32-bit
    // offsets.  The plans of the host layer never produce one (a digit's limbs are neighbours in the pool); a caller's list that does is
    // served by the two steps the fused kernel stands for: the conversion into the hand-off limbs, then the first pass in place on them.
    if (n_tiles != allTiles) return fail(c, HM_ERR_UNSUPPORTED,
        "fused conversion on a column slice: the inputs of a conversion must lie within 4 GiB of each other");
    hm_status st = hm_bconv_batch(c, descs, n_desc);
    if (st) return st;
    std::vector<uint32_t> limbs, mods, ml;
    std::vector<uint64_t> mk;
    for (uint32_t pi = 0; pi < n_desc; ++pi)
      for (uint32_t t = 0; t < descs[pi].n_out; ++t) {
        limbs.push_back(limb_at(descs[pi].out_limbs, t)); mods.push_back(descs[pi].out_ids[t]);
        if (mix) { ml.push_back(mix->mix_limbs[pi][t]); mk.push_back(mix->mix_k[pi][t]); }
      }
    NttFused f;
    f.firstPassOnly = true;
    if (mix) { f.mix = mix->mix; f.mix_limbs = ml.data(); f.mix_k = mk.data(); f.minuend = descs[0].out;
        /* (marks the fused form: the first pass only reads the mix operand) */ }
    std::vector<uint64_t> ones(limbs.size(), 1);
    return ntt_common(c, "fused conversion", descs[0].out, limbs.data(), descs[0].out, limbs.data(), mods.data(), (uint32_t)limbs.size(), 0, mix ?
        ones.data() : nullptr, f);
  }
  struct Lnch { uint32_t n_in; dim3 grid; HmBcolArgs a; };
  std::vector<Lnch> ls;
  for (auto &kv : byIn) {
    auto &grp = kv.second;
    uint32_t max_out = 0;
    uint64_t *out = descs[0].out;   // one hand-off buffer for the call (checked above)
    for (auto &p : grp) max_out = std::max(max_out, p.n_out);
    const uint32_t groups = (max_out + NOUT - 1) / NOUT;   // output groups per (conversion, tile)
    const void *dtab = nullptr;
    hm_status st = device_table(c, grp.data(), sizeof(HmBcolProb) * grp.size(), &dtab);
    if (st) return st;
    uint32_t logTiles = 0;
    while ((1u << logTiles) < n_tiles) ++logTiles;
    HmBcolArgs a = {static_cast<const HmBcolProb *>(dtab), out, c->d_tw_fwd, c->P.logN, (uint32_t)grp.size(), groups, mix ? mix->mix : nullptr, tile0,
        logTiles};
    const uint32_t pairs = ((uint32_t)grp.size() * n_tiles + 7) / 8 * 8;
    ls.push_back(Lnch{kv.first, dim3(pairs * groups), a});
  }
  // Digits of different size are launches of different kernels (N_IN is a template parameter).  (Running them side by side on a second
  // stream between a fork and a join event was measured slower: +21 us per op, profiles/README.md "side launches"; removed in round 5.)
  const dim3 block((1 << HM_TL_COL) / HM_EPT);
  for (size_t i = 0; i < ls.size(); ++i) {
    const hm_bcol_kernel kern = hm_bcol_kernel_for(ls[i].n_in & 255u, c->P.logN, NOUT, mix != nullptr, ls[i].n_in >= 256u);
    if (!kern) return fail(c, HM_ERR_UNSUPPORTED, "fused conversion: no kernel for n_in %u at N = 2^%u", ls[i].n_in, c->P.logN);
    hipLaunchKernelGGL(kern, ls[i].grid, block, 0, c->stream, ls[i].a);
    HM_HIP(c, hipGetLastError());
  }
  return HM_OK;
}

extern "C" hm_status hm_bconv(hm_ctx *c, const uint64_t *in, const uint32_t *in_limbs, const uint32_t *in_ids,
                              uint32_t n_in, uint64_t *out, const uint32_t *out_limbs, const uint32_t *out_ids,
                              uint32_t n_out) {
  hm_bconv_desc d = {in, in_limbs, in_ids, n_in, out, out_limbs, out_ids, n_out, 0};
  return hm_bconv_batch(c, &d, 1);
}

#include "hm_exchange.inl"   // multi-GPU exchange: communicators, limb <-> slice exchanges, replicate

extern "C" hm_status hm_fill_uniform(hm_ctx *c, uint64_t *out, const uint32_t *out_limbs, const uint32_t *mod_ids,
                                     uint32_t n, uint64_t seed) {
  if (!c) return HM_ERR_ARG;
  if (!out) return fail(c, HM_ERR_ARG, "hm_fill_uniform: null buffer");
  hm_status st;
  if ((st = check_limbs(c, "hm_fill_uniform", out_limbs, n)) || (st = check_mods(c, "hm_fill_uniform", mod_ids, n))) return st;
  HM_HIP(c, hipSetDevice(c->device));
  for (uint32_t base = 0; base < n; base += HM_MAX_LIMBS) {
    const uint32_t cnt = std::min<uint32_t>(HM_MAX_LIMBS, n - base);
    HmFillArgs a;
    a.out = out; a.mods = c->d_mods; a.seed = seed + base; a.logN = c->P.logN; a.n_limbs = cnt;
    for (uint32_t i = 0; i < cnt; ++i)
      a.limb[i] = HmLimb{0, (uint16_t)limb_at(out_limbs, base + i), (uint16_t)mod_ids[base + i], (uint16_t)i};
    hipLaunchKernelGGL(k_fill, dim3(cnt * (c->P.N / 256)), dim3(256), 0, c->stream, a);
    HM_HIP(c, hipGetLastError());
  }
  return HM_OK;
}
