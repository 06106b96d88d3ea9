"""INTEGRATION.md §B compiled and run: tests/shim/arch_shim.cpp is an upstream-shaped Arch (issueIns / update / simulateComplete /
getCycle) whose execution side is ONLY include/homulator_hip.h — nothing of host/ is linked — driving one hybrid key switch stage by
stage in upstream's order; its output must equal the oracle's key switch bit for bit (beta = 3 with an uneven last digit, beta = 1,
beta = 2)."""
import os
import subprocess

import numpy as np
import pytest

from oracle.homoracle import Oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("logN,L,ell,alpha", [(13, 6, 5, 2), (14, 4, 3, 3), (13, 5, 4, 2)])
def test_key_switch_through_the_c_abi_only(tmp_path, logN, L, ell, alpha):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "shim")], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "tests", "shim", "arch_shim")
    # only the C-ABI library may be linked: not the host layer, not the oracle
    needed = subprocess.check_output(["readelf", "-d", exe], text=True)
    assert "libhomulator_hip.so" in needed and "libhomulator_host" not in needed and "homoracle" not in needed
    o = Oracle(logN, L, alpha)
    d = o.fill_uniform(list(range(ell)), 4711)
    evk = o.synth_evk(ell, 991)
    (tmp_path / "d.bin").write_bytes(np.ascontiguousarray(d, dtype=np.uint64).tobytes())
    (tmp_path / "evk.bin").write_bytes(np.ascontiguousarray(evk, dtype=np.uint64).tobytes())
    r = subprocess.run([exe, str(logN), str(L), str(ell), str(alpha), str(tmp_path / "d.bin"), str(tmp_path / "evk.bin"), str(tmp_path / "out.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(tmp_path / "out.bin", dtype=np.uint64).reshape(2, ell, 1 << logN)
    k0, k1 = o.keyswitch(ell, d, evk)
    assert np.array_equal(got[0], k0) and np.array_equal(got[1], k1)
