"""Launch plans of the host layer on the count backend (no GPU): which fusion passes apply where.  The instruction total of every plan
must stay upstream's getTotalIns() whatever is fused (tests/test_host_structural.py pins the totals against the compiled reference)."""
import re

import pytest

from homulator_amd import host


def plan(cfg, op, L, ell, alpha, **ov):
    o = host.Op(cfg, op, L, ell, alpha, backend=host.BACKEND_COUNT, overrides=ov or None)
    try:
        return o.plan(), o.total_instructions(), o.launch_count()
    finally:
        o.close()


def kinds(p):
    return [ln.split()[0] for ln in p]


def test_headline_plan_is_six_launches():
    p, total, n = plan("config_4.cfg", "hmult", 45, 35, 15)
    assert n == 6 and kinds(p) == ["TENSOR", "INTT", "NTT_IP", "INTT", "BCONV", "NTT_SUBSCALE"]
    assert total == 7381760                                   # upstream's total for this command line
    sizes = [int(re.search(r" n=(\d+)", ln).group(1)) for ln in p]
    assert sizes[4] == 70 and sizes[5] == 68                  # round 4: the rescale residue is formed in the ModDown conversion's epilogue (no EWE launch)
    # the ModUp conversion, the ModUp transforms and the inner product are ONE launch; its instruction share = all three stages'
    ref = {ln.split()[0]: int(re.search(r"ref=(\d+)", ln).group(1)) for ln in p}
    p8, total8, n8 = plan("config_4.cfg", "hmult", 45, 35, 15, fuse_bconv=0)
    assert n8 == 8 and kinds(p8) == ["TENSOR", "INTT", "BCONV", "NTT_IP", "INTT", "BCONV", "EWE", "NTT_SUBSCALE"] and total8 == total
    r8 = [int(re.search(r"ref=(\d+)", ln).group(1)) for ln in p8]
    assert ref["NTT_IP"] == r8[2] + r8[3]
    assert ref["BCONV"] == r8[5] + r8[6]                      # conversion + residue: the same instructions in one launch
    # opt-in (pass 9): the ModDown conversion of 68 limb-polys inside the merged transform's first pass; the conversion launch that is left
    # converts the two last limbs (the residue comes from them)
    pm, totalm, nm = plan("config_4.cfg", "hmult", 45, 35, 15, fuse_moddown=1)
    sm = [int(re.search(r" n=(\d+)", ln).group(1)) for ln in pm]
    rm = {ln.split()[0]: int(re.search(r"ref=(\d+)", ln).group(1)) for ln in pm}
    assert nm == 6 and kinds(pm) == kinds(p) and sm[4] == 2 and sm[5] == 68 and totalm == total
    assert rm["BCONV"] + rm["NTT_SUBSCALE"] == r8[5] + r8[6] + r8[7]
    p9, total9, n9 = plan("config_4.cfg", "hmult", 45, 35, 15, fuse_hpip=0)
    assert kinds(p9)[2:5] == ["BCONV", "NTT", "IP"] and total9 == total
    ph, totalh, nh = plan("config_4.cfg", "hrotate", 45, 35, 15)
    assert nh == 5 and kinds(ph) == ["INTT", "NTT_IP", "INTT", "BCONV", "NTT_SUBSCALE"] and totalh == 7328000   # (pass 12: no automorphism launch)
    pha, totalha, nha = plan("config_4.cfg", "hrotate", 45, 35, 15, fuse_auto=0)
    assert nha == 6 and kinds(pha) == ["AUTO"] + kinds(ph) and totalha == totalh
    phm, _, nhm = plan("config_4.cfg", "hrotate", 45, 35, 15, fuse_moddown=1)
    assert nhm == 5 and kinds(phm) == ["AUTO", "INTT", "NTT_IP", "INTT", "NTT_SUBSCALE"]   # no conversion launch at all


@pytest.mark.parametrize("cfg,L,ell,alpha,conv_inside", [
    ("config_4_N15.cfg", 16, 10, 4, True),     # N = 2^15 (round 4: the fused conversion exists for N = 2^15 and 2^16)
    ("config_4.cfg", 28, 28, 28, False),       # the `motivation` sweep at N = 2^16: 28 input limbs per digit — the two-group fused form exists (round 6) but measured
                                               # 1-5 % slower than conversion + first pass there (cap_bconv_col_pref_in = 15): the planner keeps the conversion launch
    ("config_4.cfg", 28, 15, 28, True),        # ... and fuses up to 15
    ("config_4_N15.cfg", 28, 28, 28, True),    # parameter set A at its top level (N = 2^15, beta = 1)
    ("config_4_N15.cfg", 28, 17, 28, True),    # set A, a 17-limb digit
    ("config_4.cfg", 24, 24, 6, True),         # set C: beta = 4
    ("config_4.cfg", 26, 20, 9, True),         # set D: uneven last digit (9, 9, 2)
    ("config_4.cfg", 8, 8, 8, True),           # beta = 1: the digit's own limbs need no transform
])
def test_where_the_conversion_moves_into_the_transform(cfg, L, ell, alpha, conv_inside):
    p, total, n = plan(cfg, "hmult", L, ell, alpha)
    k = kinds(p)
    assert "NTT_IP" in k                                      # the transform x key fusion applies at every shape
    modup_bconv = [ln for ln in p if ln.startswith("BCONV") and "ModUp_BCONV" in ln]
    assert (len(modup_bconv) == 0) == conv_inside, p
    p0, total0, _ = plan(cfg, "hmult", L, ell, alpha, fuse_hpip=0, fuse_bconv=0)
    assert total0 == total


def test_mixed_launch_keeps_the_wide_digits_conversion():
    """config_4.cfg hmult 45 20 16 with the fused conversion capped at 15 input limbs (config key fuse_bconv_max_in: the plan of rounds 3-5,
    kept for A/B runs): digits of 16 and 4 limbs.  The 4-limb digit's conversion moves into the first pass of its transforms (16 limbs: the
    ones whose only transformed digit it is); the 16-limb digit is wider than the cap and keeps its own BCONV launch.  ONE NTT_IP launch then
    mixes both kinds: the backend runs the first pass of every transformed (limb, digit) that no conversion of the call covers (round 3
    skipped it as soon as any conversion was fused: ADVICE round 3, tests/test_gpu_ops.py has the parity case).  Without the cap (round 6)
    both digits convert inside their first pass and no ModUp conversion launch is left once the cap is lifted (fuse_bconv_max_in = 32; the
    planner's own default at N = 2^16 is the measured crossover, cap_bconv_col_pref_in = 15)."""
    p, total, n = plan("config_4.cfg", "hmult", 45, 20, 16, fuse_bconv_max_in=15)
    modup_bconv = [ln for ln in p if ln.startswith("BCONV") and "ModUp_BCONV" in ln]
    nip = [ln for ln in p if ln.startswith("NTT_IP")]
    assert len(modup_bconv) == 1 and len(nip) == 1, p
    # 20 outputs of the wide digit + the 16 special limbs of the narrow one (both of their digits are transformed, one of them not
    # fusable, so the record keeps both conversions); the other 16 outputs of the narrow digit are converted inside their first pass
    assert int(re.search(r"n=(\d+)", modup_bconv[0]).group(1)) == 20 + 16
    assert int(re.search(r"n=(\d+)", nip[0]).group(1)) == 36
    p0, total0, _ = plan("config_4.cfg", "hmult", 45, 20, 16, fuse_hpip=0, fuse_bconv=0)
    assert total0 == total
    pw, totalw, nw = plan("config_4.cfg", "hmult", 45, 20, 16, fuse_bconv_max_in=32)   # every width the kernels take
    assert not [ln for ln in pw if ln.startswith("BCONV") and "ModUp_BCONV" in ln] and totalw == total and nw == n - 1


def test_no_ring_size_is_named_in_the_planner():
    """round 6: the fusion passes ask the back-end's capability table (hm_capability, homulator_amd/csrc/hm_caps.h) — no ring size or digit
    width is spelled out in host/src/Arch.cpp"""
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "host", "src", "Arch.cpp")).read()
    assert not re.search(r"logN\s*[=!<>]=\s*1[3-7]", src) and "<= 15" not in src and "> 15" not in src and "n >> 12" not in src


def test_packed_conversion_inputs_in_the_default_plan():
    """pass 11 (split-30 packed conversion inputs) in the default plan of the headline op: BOTH inverse-transform launches store packed
    limb-polys and BOTH conversions read them — the ModUp conversions inside NTT_IP and the ModDown conversions (whose inputs' inverse
    transform is the in-place second pass of pass 7b: its read of its own output is no other reader)"""
    p, _, _ = plan("config_4.cfg", "hmult", 45, 35, 15)
    by = {}
    for ln in p:
        by.setdefault(ln.split()[0], []).append(ln)
    assert " packed_out=35" in by["INTT"][0], by["INTT"][0]               # ModUp_DecompOut: the 35 scaled input limbs
    assert " packed_in=3/3" in by["NTT_IP"][0], by["NTT_IP"][0]           # the three digits' conversions
    assert " packed_out=30" in by["INTT"][1] and " second_pass_only" in by["INTT"][1], by["INTT"][1]   # ModDownBConvStep1_Key(k): 2 x 15 special limbs
    assert " packed_in=" in by["BCONV"][0], by["BCONV"][0]
    pn, _, _ = plan("config_4.cfg", "hmult", 45, 35, 15, pack_bconv_in=0)
    assert not any("packed" in ln for ln in pn)


def test_the_automorphisms_of_hrotate_fold_into_their_readers():
    """pass 12 (round 6): an automorphism whose output only feeds inverse transforms' inputs, fused forward transforms' addends and the
    evaluation-form digits of transform x key records is read THROUGH by those kernels.  hrotate: AUTO_Key(1) -> ModUp_INTT + the key product's
    own digits, AUTO_Key(0) -> the final add inside ModDowNTT's epilogue; no automorphism launch is left (6 -> 5), the instruction total stays
    upstream's.  Where a reader cannot gather (the plain inner-product kernel of fuse_hpip = 0; the final transform with the conversion inside,
    fuse_moddown = 1) that automorphism stays a launch of its 35 limb-polys; fuse_auto = 0 keeps both"""
    p, total, n = plan("config_4.cfg", "hrotate", 45, 35, 15)
    assert n == 5 and "AUTO" not in kinds(p)
    assert " auto_in=35/g5" in p[0] and " auto_x=g5" in p[1] and p[-1].startswith("NTT_SUBSCALE") and " auto_addend=35/g5" in p[-1]
    p0, total0, n0 = plan("config_4.cfg", "hrotate", 45, 35, 15, fuse_auto=0)
    assert n0 == 6 and total0 == total and " n=70 " in p0[0] + " " and not any("auto_" in ln for ln in p0)
    ph, totalh, _ = plan("config_4.cfg", "hrotate", 45, 35, 15, fuse_hpip=0)       # the plain inner product reads the rotated c1 from memory
    assert kinds(ph)[0] == "AUTO" and "AUTO_Key(0)" not in ph[0] and " n=35 " in ph[0] + " " and " auto_addend=35/g5" in ph[-1] and totalh == total
    assert not any("auto_in" in ln or "auto_x" in ln for ln in ph)
    pm, totalm, _ = plan("config_4.cfg", "hrotate", 45, 35, 15, fuse_moddown=1)    # the final transform converts inside: it takes a plain addend
    assert kinds(pm)[0] == "AUTO" and "AUTO_Key(1)" not in pm[0] and " auto_in=35/g5" in pm[1] and " auto_x=g5" in pm[2] and totalm == total
    pg, _, _ = plan("config_4_N15.cfg", "hrotate", 16, 10, 4, galois=25)
    assert " auto_in=10/g25" in pg[0] and " auto_x=g25" in pg[1] and " auto_addend=10/g25" in pg[-1]
    # a one-digit key switch (parameter sets A and `motivation`): the Q limbs multiply the rotated c1 itself with the key in a plain inner product,
    # which gathers too (hm_inner_product_ex)
    p1, _, _ = plan("config_4.cfg", "hrotate", 28, 28, 28)
    assert "AUTO" not in kinds(p1) and p1[1].startswith("IP ") and " auto_x=g5" in p1[1] and " auto_in=28/g5" in p1[0] and " auto_addend=28/g5" in p1[-1]
