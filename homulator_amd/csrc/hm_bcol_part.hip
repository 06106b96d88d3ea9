// hm_bcol_part.hip — instantiations of the fused conversion + first-pass kernels (hm_bcol.h), one slice per translation unit:
// HM_BCOL_PART = 0 .. 5: digits of 1 .. 15 limbs; bit 0 = ring (2^16 | 2^15); parts 0 / 1 plain, 2 / 3 mix prologue, 4 / 5 split-30 packed inputs;
// HM_BCOL_PART = 6 .. 9 (round 6): digits of 16 .. 32 limbs (two input groups); bit 0 = ring; 6 / 7 plain, 8 / 9 packed.  Part 0 also holds the lookup.
#include <hip/hip_runtime.h>
#include "hm_bcol.h"
#ifndef HM_BCOL_PART
#error "compile with -DHM_BCOL_PART=0..9"
#endif
#define HM_PART_LOG1 ((HM_BCOL_PART & 1) ? 7 : 8)
#define HM_PART_WIDE (HM_BCOL_PART >= 6)
#define HM_PART_MIX (!HM_PART_WIDE && (HM_BCOL_PART >> 1) == 1)
#define HM_PART_PACKED (HM_PART_WIDE ? ((HM_BCOL_PART - 6) >> 1) == 1 : (HM_BCOL_PART >> 1) == 2)
#define HM_PASTE2(a, b) a##b
#define HM_PASTE(a, b) HM_PASTE2(a, b)
#define HM_K(n) k_bconv_col<n, HM_PART_LOG1, HM_PART_MIX, HM_PART_PACKED>,
#if HM_PART_WIDE
#define HM_K2(n) k_bconv_col2w<n, HM_PART_LOG1, HM_PART_MIX, HM_PART_PACKED>,
#else
#define HM_K2(n) k_bconv_col2<n, HM_PART_LOG1, HM_PART_MIX, HM_PART_PACKED>,
#endif
#if HM_PART_WIDE
#define HM_ALL(K) K(16) K(17) K(18) K(19) K(20) K(21) K(22) K(23) K(24) K(25) K(26) K(27) K(28) K(29) K(30) K(31) K(32)
#define HM_FIRST 16
#define HM_LAST 32
#else
#define HM_ALL(K) K(1) K(2) K(3) K(4) K(5) K(6) K(7) K(8) K(9) K(10) K(11) K(12) K(13) K(14) K(15)
#define HM_FIRST 1
#define HM_LAST 15
#endif
static_assert(HM_LAST <= HM_BCOL_MAX_IN && (!HM_PART_MIX || HM_LAST <= HM_BCOL_MAX_IN_MIX), "kernel table");
hm_bcol_kernel HM_PASTE(hm_bcol_part_, HM_BCOL_PART)(uint32_t n_in, uint32_t nout) {
  static const hm_bcol_kernel one[HM_LAST - HM_FIRST + 1] = {HM_ALL(HM_K)};
  static const hm_bcol_kernel two[HM_LAST - HM_FIRST + 1] = {HM_ALL(HM_K2)};
  if (n_in < HM_FIRST || n_in > HM_LAST) return nullptr;
  return nout == 2 ? two[n_in - HM_FIRST] : one[n_in - HM_FIRST];
}
#if HM_BCOL_PART == 0
hm_bcol_kernel hm_bcol_part_1(uint32_t, uint32_t);
hm_bcol_kernel hm_bcol_part_2(uint32_t, uint32_t);
hm_bcol_kernel hm_bcol_part_3(uint32_t, uint32_t);
hm_bcol_kernel hm_bcol_part_4(uint32_t, uint32_t);
hm_bcol_kernel hm_bcol_part_5(uint32_t, uint32_t);
hm_bcol_kernel hm_bcol_part_6(uint32_t, uint32_t);
hm_bcol_kernel hm_bcol_part_7(uint32_t, uint32_t);
hm_bcol_kernel hm_bcol_part_8(uint32_t, uint32_t);
hm_bcol_kernel hm_bcol_part_9(uint32_t, uint32_t);
hm_bcol_kernel hm_bcol_kernel_for(uint32_t n_in, uint32_t logN, uint32_t nout, bool mix, bool packed) {
  if ((logN != 16 && logN != 15) || (mix && packed)) return nullptr;
  const int ring = logN == 15 ? 1 : 0;
  if (n_in > 15) {   // two input groups: no mix prologue
    if (mix) return nullptr;
    switch (6 + ring + (packed ? 2 : 0)) {
    case 6: return hm_bcol_part_6(n_in, nout);
    case 7: return hm_bcol_part_7(n_in, nout);
    case 8: return hm_bcol_part_8(n_in, nout);
    default: return hm_bcol_part_9(n_in, nout);
    }
  }
  switch (ring | (mix ? 2 : packed ? 4 : 0)) {
  case 0: return hm_bcol_part_0(n_in, nout);
  case 1: return hm_bcol_part_1(n_in, nout);
  case 2: return hm_bcol_part_2(n_in, nout);
  case 3: return hm_bcol_part_3(n_in, nout);
  case 4: return hm_bcol_part_4(n_in, nout);
  default: return hm_bcol_part_5(n_in, nout);
  }
}
#endif
