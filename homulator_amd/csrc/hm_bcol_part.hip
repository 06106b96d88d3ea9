// hm_bcol_part.hip — instantiations of the fused conversion + first-pass kernels (hm_bcol.h), one slice per translation unit:
// HM_BCOL_PART = 0 .. 5: bit 0 = ring (2^16 | 2^15); parts 0 / 1 plain, 2 / 3 mix prologue, 4 / 5 split-30 packed inputs; part 0 also holds the lookup.
#include <hip/hip_runtime.h>
#include "hm_bcol.h"
#ifndef HM_BCOL_PART
#error "compile with -DHM_BCOL_PART=0..5"
#endif
#define HM_PART_LOG1 ((HM_BCOL_PART & 1) ? 7 : 8)
#define HM_PART_MIX ((HM_BCOL_PART >> 1) == 1)
#define HM_PART_PACKED ((HM_BCOL_PART >> 1) == 2)
#define HM_PASTE2(a, b) a##b
#define HM_PASTE(a, b) HM_PASTE2(a, b)
#define HM_K(n) k_bconv_col<n, HM_PART_LOG1, HM_PART_MIX, HM_PART_PACKED>,
#define HM_K2(n) k_bconv_col2<n, HM_PART_LOG1, HM_PART_MIX, HM_PART_PACKED>,
hm_bcol_kernel HM_PASTE(hm_bcol_part_, HM_BCOL_PART)(uint32_t n_in, uint32_t nout) {
  static const hm_bcol_kernel one[HM_BCOL_MAX_IN + 1] = {nullptr, HM_K(1) HM_K(2) HM_K(3) HM_K(4) HM_K(5) HM_K(6) HM_K(7) HM_K(8) HM_K(9) HM_K(10) HM_K(11) HM_K(12) HM_K(13) HM_K(14) HM_K(15)};
  static const hm_bcol_kernel two[HM_BCOL_MAX_IN + 1] = {nullptr, HM_K2(1) HM_K2(2) HM_K2(3) HM_K2(4) HM_K2(5) HM_K2(6) HM_K2(7) HM_K2(8) HM_K2(9) HM_K2(10) HM_K2(11) HM_K2(12) HM_K2(13) HM_K2(14) HM_K2(15)};
  if (n_in == 0 || n_in > HM_BCOL_MAX_IN) return nullptr;
  return nout == 2 ? two[n_in] : one[n_in];
}
#if HM_BCOL_PART == 0
hm_bcol_kernel hm_bcol_part_1(uint32_t, uint32_t);
hm_bcol_kernel hm_bcol_part_2(uint32_t, uint32_t);
hm_bcol_kernel hm_bcol_part_3(uint32_t, uint32_t);
hm_bcol_kernel hm_bcol_part_4(uint32_t, uint32_t);
hm_bcol_kernel hm_bcol_part_5(uint32_t, uint32_t);
hm_bcol_kernel hm_bcol_kernel_for(uint32_t n_in, uint32_t logN, uint32_t nout, bool mix, bool packed) {
  if ((logN != 16 && logN != 15) || (mix && packed)) return nullptr;
  const int part = (logN == 15 ? 1 : 0) | (mix ? 2 : packed ? 4 : 0);
  switch (part) {
  case 0: return hm_bcol_part_0(n_in, nout);
  case 1: return hm_bcol_part_1(n_in, nout);
  case 2: return hm_bcol_part_2(n_in, nout);
  case 3: return hm_bcol_part_3(n_in, nout);
  case 4: return hm_bcol_part_4(n_in, nout);
  default: return hm_bcol_part_5(n_in, nout);
  }
}
#endif
