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
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_or = {}


def oracle(logN, L, K):
    if (logN, L, K) not in _or:
        _or[(logN, L, K)] = Oracle(logN, L, K)
        _or[(logN, L, K)].set_threads(8)
    return _or[(logN, L, K)]


@pytest.mark.parametrize("cfg,logN,L,alpha,ell,op", [
    ("config_4_N15.cfg", 15, 28, 28, 28, "hmult"),     # set A, top level: beta = 1, ModUp 28 -> 28, ModDown 28 -> 28
    ("config_4_N15.cfg", 15, 28, 28, 17, "hmult"),     # set A, 17 > 16 inputs: two column groups in k_bconv
    ("config_4_N15.cfg", 15, 28, 28, 1, "hrotate"),    # set A, lowest rotate level
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
    h.execute(1)
    exp = o.hmult(ell, ct1, ct2, evk) if op == "hmult" else o.hrotate(ell, ct1, 5, evk)
    assert np.array_equal(h.read("out.c0"), exp[0]) and np.array_equal(h.read("out.c1"), exp[1])
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
