// hm_caps.h — what the back-end has kernels for, per ring size: ONE table (round 6) instead of `logN == 16` / `<= 15` tests spread over the
// launch code and the host layer.  Plain C++ with no HIP in it: the kernels' launch code (hm_backend.hip), the dispatcher (hm_dispatch.cpp:
// hm_capability needs no context and no GPU) and the host layer (host/src/Arch.cpp plans its fusion passes from these rows and never names
// a ring size) all read the same function.  A new shape is a new row here plus its kernel instantiations.
// Reference: the shapes are the reference's configurations and parameter sets (config/config_4.cfg: N = 2^16; config/config_4_N15.cfg:
// N = 2^15; script/README.md:17-22: digits of up to alpha = 28 limbs).
#pragma once
#include <stdint.h>
#include <string.h>

#define HM_BCOL_MAX_IN 32       // fused conversion + first pass: digits of up to 15 limbs hold every input of an access unit at once; 16 .. 32 (round 6) take two groups
#define HM_BCOL_MAX_IN_MIX 15   // ... with the mix prologue (the opt-in fused ModDown conversion): one group only

struct HmCaps {
  uint32_t small_geometry;    // the 8-coefficient passes (k_ntt_col8 / k_ntt_row8), the one-launch transform (k_ntt_fused8), the small-launch transform x key kernel (k_ntt_row_ip8)
  uint32_t bcol_max_in;       // widest digit the fused conversion + first pass takes (k_bconv_col / k_bconv_col2); 0: no such kernel at this ring size
  uint32_t bcol_pref_in;      // widest digit for which the fused form is the FASTER plan (measured: gpurun_out/r06_sweep, profiles/r06_sweep_sets.txt); wider digits keep a conversion launch of their own unless the caller says otherwise (config key fuse_bconv_max_in)
  uint32_t bcol_max_in_mix;   // ... with the mix prologue
  uint32_t ip_inverse_out;    // hm_ntt_ip_desc.out_inverse: the transform x key kernel hands its outputs over as the first pass of their inverse transform (pass 7b)
  uint32_t col_slices;        // the transposed-domain exchange + conversion on a rank's column slice (hm_bconv_col with a tile range): ranks a limb-poly's 4096-coefficient column tiles can be dealt to
};
static inline HmCaps hm_caps(uint32_t logN) {
  // rows: ring size -> capabilities.  The two ring sizes of the reference's configuration files carry every fused form; the other sizes the
  // transforms support (2^13, 2^14, 2^17: tests and small-ring sweeps) run the plain plan.
  switch (logN) {
  // N = 2^16: the two-group conversion inside the first pass is 1-2 % SLOWER per op than k_bconv<n> + first pass at 16-20 limbs and up to 5 % at 28
  // (every pair of outputs re-reads the digit's tiles from L2: 14 readers per tile at 28 limbs; profiles/r06_sweep_sets.txt); N = 2^15: 1-5 % faster
  case 16: return HmCaps{1, HM_BCOL_MAX_IN, 15, HM_BCOL_MAX_IN_MIX, 1, 16};
  case 15: return HmCaps{1, HM_BCOL_MAX_IN, HM_BCOL_MAX_IN, HM_BCOL_MAX_IN_MIX, 1, 8};
  default: return HmCaps{0, 0, 0, 0, 0, 0};
  }
}
// 0 = found
static inline int hm_cap_by_name(uint32_t logN, const char *name, uint64_t *value) {
  const HmCaps c = hm_caps(logN);
  if (!strcmp(name, "cap_small_geometry")) { *value = c.small_geometry; return 0; }
  if (!strcmp(name, "cap_bconv_col_max_in")) { *value = c.bcol_max_in; return 0; }
  if (!strcmp(name, "cap_bconv_col_max_in_mix")) { *value = c.bcol_max_in_mix; return 0; }
  if (!strcmp(name, "cap_bconv_col_pref_in")) { *value = c.bcol_pref_in; return 0; }
  if (!strcmp(name, "cap_ip_inverse_out")) { *value = c.ip_inverse_out; return 0; }
  if (!strcmp(name, "cap_col_slices")) { *value = c.col_slices; return 0; }
  return 1;
}
