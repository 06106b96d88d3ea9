"""AddressSanitizer + UBSan over the host layer and the oracle on the CPU (the reference's UB at
include/Instruction.h:57,166 is why).  The count backend runs the whole graph construction — constructors, stage emission,
fusion passes, launch coalescing, sharded plans for 2/4/8 ranks, batch replication — with no GPU: 530 lines of index
bookkeeping under `-fsanitize=address,undefined` (`make -C host asan`).  The sanitized oracle (`make -C oracle
libhomoracle_asan.so`) runs one small hmult + hrotate.  Both in a child process: an ASan runtime has to be the first
library of the process (LD_PRELOAD), which this interpreter was not started with."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_env(**extra):
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=67", **extra)
    env.pop("HOMULATOR_HIP_LIB", None)
    return env


HOST_SCRIPT = r"""
import re, sys
sys.path.insert(0, %r)
from homulator_amd import host
n = 0
for cfg, L, ell, alpha in (("config_4_N15.cfg", 16, 10, 4), ("config_4_N15.cfg", 4, 2, 2), ("config_4_N15.cfg", 4, 3, 2), ("config_4.cfg", 45, 35, 15),
                           ("config_4.cfg", 45, 45, 15), ("config_4_N15.cfg", 28, 28, 28), ("config_4.cfg", 24, 7, 6)):
    for opn in ("hmult", "hrotate", "hadd", "pmult", "padd"):
        for fuse in (True, False):
            op = host.Op(cfg, opn, L, ell, alpha, backend=host.BACKEND_COUNT, fuse=fuse)
            assert op.total_instructions() > 0 and len(op.plan()) > 0
            op.close(); n += 1
for world in (2, 4, 8):
    for batch in (1, 3):
        for plan in (1, 2):      # all-to-all on column slices | gather of the conversions' inputs
            for r in range(world):
                op = host.Op("config_4.cfg", "hmult", 45, 35, 15, backend=host.BACKEND_COUNT, rank=r, world=world, overrides={"batch": batch, "shard_plan": plan})
                assert any(l.startswith("EXCH_IN" if plan == 1 else "REPLICATE") for l in op.plan())
                op.close(); n += 1
ch = host.Chain("config_4_N15.cfg", "hmult,hrotate,hadd,hmult,padd", 6, 5, 2, overrides={"backend": host.BACKEND_COUNT})
assert len(ch) == 5
ch.close()
print("sanitized host layer: %%d plans built" %% n)
""" % ROOT


def test_host_layer_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "asan"], stdout=subprocess.DEVNULL)
    env = _asan_env(HOMULATOR_HOST_LIB=os.path.join(ROOT, "host", "lib", "libhomulator_host_asan.so"), HOMULATOR_BACKEND="count")
    out = subprocess.run([sys.executable, "-c", HOST_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-6000:]
    assert "plans built" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]


SIM_SCRIPT = r"""
import sys
sys.path.insert(0, %r)
from homulator_amd import host
want = {("hmult", 4, 3, 2, None): 4775, ("hrotate", 6, 4, 2, None): 3455, ("hadd", 16, 10, 4, None): 1296, ("hmult", 6, 4, 2, 8): 4487, ("pmult", 4, 3, 2, 3): 400}
for (opn, L, ell, alpha, cluster), cycles in want.items():
    op = host.Op("config_4_N15.cfg", opn, L, ell, alpha, backend=host.BACKEND_SIM, overrides={"cluster": cluster} if cluster else None)
    r = op.sim_run()
    assert r["drained"] and r["cycles"] == cycles, (opn, L, ell, alpha, cluster, r["cycles"])
    op.close()
print("sanitized cycle model: %%d points" %% len(want))
""" % ROOT


def test_cycle_model_under_asan_ubsan():
    """backend = sim (host/src/SimModel.cpp, SimProgram.cpp): dense address-indexed arrays, ring-buffer pipelines and queue cursors,
    run under the sanitizers on points whose cycle counts the compiled reference fixes"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "asan"], stdout=subprocess.DEVNULL)
    env = _asan_env(HOMULATOR_HOST_LIB=os.path.join(ROOT, "host", "lib", "libhomulator_host_asan.so"))
    env.pop("HOMULATOR_BACKEND", None)
    out = subprocess.run([sys.executable, "-c", SIM_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-6000:]
    assert "5 points" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]


ORACLE_SCRIPT = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from oracle.homoracle import Oracle
o = Oracle(13, 6, 2)
o.set_threads(2)
ct1, ct2, evk = o.synth_ct(5, 1), o.synth_ct(5, 2001), o.synth_evk(5, 10001)
a = o.hmult(5, ct1, ct2, evk)
b = o.hrotate(5, ct1, 5, evk)
x = o.fill_uniform([0, 6], 3)
assert np.array_equal(o.ntt([0, 6], o.ntt([0, 6], x), inverse=True), x)
print("sanitized oracle ok", int(a[0][0][0] & 0xffff), int(b[1][0][0] & 0xffff))
""" % ROOT


def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libhomoracle_asan.so"], stdout=subprocess.DEVNULL)
    env = _asan_env(HOMORACLE_LIB=os.path.join(ROOT, "oracle", "libhomoracle_asan.so"))
    out = subprocess.run([sys.executable, "-c", ORACLE_SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-6000:]
    assert "sanitized oracle ok" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]
