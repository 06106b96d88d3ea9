"""Parameter sets A, C, D of the reference's sweep scripts (script/README.md:17-22) on the GPU,
bit-exact against the oracle: A = N 2^15, L 28, alpha 28 (beta = 1, 28-limb base conversions), C = N 2^16, L 24,
alpha 6 (beta up to 4: four-term inner product), D = N 2^16, L 26, alpha 9.  Plus the sweep runner's layout."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle.homoracle import Oracle

pytestmark = pytest.mark.gpu
SEED = 0x484F4D55
BATCH_SEED_STRIDE = 100000   # host/src/Arch.cpp kBatchSeedStride
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_or = {}


def oracle(logN, L, K):
    if (logN, L, K) not in _or:
        _or[(logN, L, K)] = Oracle(logN, L, K)
        _or[(logN, L, K)].set_threads(8)
    return _or[(logN, L, K)]


@pytest.mark.parametrize("cfg,logN,L,alpha,ell,op", [
    ("config_4_N15.cfg", 15, 28, 28, 28, "hmult"),     # set A, top level: beta = 1, ModUp 28 -> 28 inside the first pass (two input groups, round 6), ModDown 28 -> 28
    ("config_4_N15.cfg", 15, 28, 28, 28, "hrotate"),
    ("config_4_N15.cfg", 15, 28, 28, 20, "hmult"),     # set A, a 20-limb digit
    ("config_4_N15.cfg", 15, 28, 28, 17, "hmult"),     # set A, 17 > 16 inputs: two column groups in k_bconv
    ("config_4_N15.cfg", 15, 28, 28, 16, "hrotate"),   # set A, exactly one full input group
    ("config_4_N15.cfg", 15, 28, 28, 1, "hrotate"),    # set A, lowest rotate level
    ("config_4.cfg", 16, 28, 28, 28, "hmult"),         # the `motivation` sweep (script/motivation/micro24_motivation.sh): N = 2^16, alpha = 28
    ("config_4.cfg", 16, 28, 28, 19, "hrotate"),
    ("config_4.cfg", 16, 24, 6, 24, "hmult"),          # set C, beta = 4
    ("config_4.cfg", 16, 24, 6, 19, "hrotate"),        # set C, beta = 4 with a 1-limb last digit
    ("config_4.cfg", 16, 26, 9, 26, "hmult"),          # set D, beta = 3, last digit 8
    ("config_4.cfg", 16, 26, 9, 10, "hrotate"),        # set D, beta = 2, last digit 1
])
def test_parameter_set_op_bit_exact(cfg, logN, L, alpha, ell, op):
    from homulator_amd import host
    o = oracle(logN, L, alpha)
    ct1, ct2, evk = o.synth_ct(ell, SEED), o.synth_ct(ell, SEED + 2000), o.synth_evk(ell, SEED + 10000)
    h = host.Op(cfg, op, L, ell, alpha)
    wide16 = logN == 16 and min(ell, alpha) > 15   # N = 2^16: digits above cap_bconv_col_pref_in keep their conversion launch (measured faster)
    assert any(ln.startswith("BCONV") and "ModUp_BCONV" in ln for ln in h.plan()) == wide16
    h.execute(1)
    exp = o.hmult(ell, ct1, ct2, evk) if op == "hmult" else o.hrotate(ell, ct1, 5, evk)
    assert np.array_equal(h.read("out.c0"), exp[0]) and np.array_equal(h.read("out.c1"), exp[1])
    h.close()


@pytest.mark.parametrize("chain_bits", [60, 36])
@pytest.mark.parametrize("cfg,logN,L,alpha,ell,op,mode", [
    ("config_4_N15.cfg", 15, 28, 28, 28, "hmult", "batch"),      # set A on the generic back-end, three ops per launch (the two-output conversion kernels)
    ("config_4_N15.cfg", 15, 28, 28, 23, "hrotate", "no_pack"),  # plain (not split-30 packed) inputs into the two-group conversion
    ("config_4.cfg", 16, 28, 28, 28, "hmult", "no_ip_inv"),      # motivation without pass 7b
    ("config_4.cfg", 16, 28, 28, 28, "hrotate", "wide32"),       # motivation with the 28-limb digit forced inside the first pass (fuse_bconv_max_in = 32)
    ("config_4_N15.cfg", 15, 28, 28, 28, "hmult", "cap15"),      # the plan of rounds 3-5: a conversion launch of its own for the wide digit
])
def test_wide_digit_sets_in_the_other_modes(cfg, logN, L, alpha, ell, op, mode, chain_bits):
    """the parameter sets with digits of more than 15 limbs (A, motivation) in the other fusion modes and on the generic arithmetic back-end
    (SURVEY.md 8(d)'s chain as written; 36-bit words as config_4.cfg:9 models, N = 2^15 only: a 2^16 ring has too few 36-bit primes for
    L + alpha = 56), bit-exact"""
    from homulator_amd import host
    if chain_bits == 36 and logN != 15:
        pytest.skip("36-bit chain: N = 2^15 cases")
    key = (logN, L, alpha, chain_bits)
    if key not in _or:
        _or[key] = Oracle(logN, L, alpha, chain="survey" if chain_bits == 60 else chain_bits)
        _or[key].set_threads(8)
    o = _or[key]
    ov = {"chain_bits": chain_bits}
    nb = 1
    if mode == "batch":
        ov["batch"] = nb = 3
    elif mode == "no_pack":
        ov["pack_bconv_in"] = 0
    elif mode == "no_ip_inv":
        ov["fuse_ip_inv"] = 0
    elif mode == "cap15":
        ov["fuse_bconv_max_in"] = 15
    elif mode == "wide32":
        ov["fuse_bconv_max_in"] = 32
    h = host.Op(cfg, op, L, ell, alpha, overrides=ov)
    assert h.backend_counter("arith") == 1
    h.execute(1)
    for b in range(nb):
        s = SEED + b * BATCH_SEED_STRIDE   # op b of a batch: own inputs, ONE evaluation key
        ct1, ct2, evk = o.synth_ct(ell, s), o.synth_ct(ell, s + 2000), o.synth_evk(ell, SEED + 10000)
        exp = o.hmult(ell, ct1, ct2, evk) if op == "hmult" else o.hrotate(ell, ct1, 5, evk)
        assert np.array_equal(h.read("out.c0", copy=b), exp[0]) and np.array_equal(h.read("out.c1", copy=b), exp[1]), (mode, b)
    h.close()


def test_sweep_runner_layout(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "sweep.py"), "--set", "D", "--ops", "hadd,hrotate", "--levels", "26,1",
                        "--out", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    for op in ("hadd", "hrotate"):
        for lv in (26, 1):
            log = tmp_path / "paraD" / "4" / op / "26_9" / f"{op}_26_9_{lv}.log"
            assert log.exists()
            txt = log.read_text()
            assert "Completed Simulate!" in txt and "Remaining 0 instructions!" in txt


@pytest.mark.parametrize("alpha", [1, 2, 3, 5, 13])
def test_every_level_small_ring(alpha):
    """hmult and hrotate at EVERY level of a 13-limb chain on a small ring (N = 2^13 through the `N` config override):
    all digit shapes (beta = 1 .. 13, short last digits, a one-limb last digit) go through the fused plan, including
    the merged ModDown + rescale transform, and must equal the oracle bit for bit"""
    from homulator_amd import host
    L, logN = 13, 13
    o = Oracle(logN, L, alpha)
    o.set_threads(4)
    for ell in range(1, L + 1):
        ct1, ct2, evk = o.synth_ct(ell, SEED), o.synth_ct(ell, SEED + 2000), o.synth_evk(ell, SEED + 10000)
        for name in ("hmult", "hrotate"):
            if name == "hmult" and ell < 2:
                continue
            op = host.Op("config_4_N15.cfg", name, L, ell, alpha, overrides={"N": 1 << logN})
            op.execute(1)
            exp = o.hmult(ell, ct1, ct2, evk) if name == "hmult" else o.hrotate(ell, ct1, 5, evk)
            assert np.array_equal(op.read("out.c0"), exp[0]), (name, ell, alpha)
            assert np.array_equal(op.read("out.c1"), exp[1]), (name, ell, alpha)
            op.close()
