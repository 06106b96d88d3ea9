"""Limb-sharded execution on the GPU: 2 and 4 ranks share the one GPU of the test box and exchange
through the gloo transport (RCCL refuses two ranks per device); the assembled outputs must equal the oracle bit for
bit.  Exercises: ownership rule, limbs->slices / slices->limbs packing, coefficient-sliced base conversion, the
replicate of the rescale's r, and the fused plan under sharding.  Both sharded plans at every rank count: plan 1 = all-to-all on
column slices, plan 2 = gather of the conversions' inputs (the automatic choice up to 4 ranks; HOMULATOR_SHARD_PLAN is read by every rank)."""
import re

import pytest

from test_dist_cpu import launch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,cfg,opname,L,ell,alpha,logN", [
    (2, "config_4_N15.cfg", "hmult", 16, 10, 4, 15),
    (4, "config_4_N15.cfg", "hmult", 6, 5, 2, 15),
    (2, "config_4_N15.cfg", "hrotate", 16, 10, 4, 15),
    (4, "config_4.cfg", "hmult", 45, 35, 15, 16),
])
@pytest.mark.parametrize("plan", [1, 2])
def test_sharded_op_matches_oracle(world, cfg, opname, L, ell, alpha, logN, plan, monkeypatch):
    monkeypatch.setenv("HOMULATOR_SHARD_PLAN", str(plan))
    rcs, outs = launch(world, ["gpu", cfg, opname, str(L), str(ell), str(alpha), str(logN)], timeout=900)
    assert all(rc == 0 for rc in rcs), "\n".join(outs)
    assert "OK" in outs[0]
    n_coll = int(re.search(r"exchanges=(\d+)", outs[0]).group(1)) // 2     # two executions
    beta = -(-ell // alpha)
    assert n_coll == ((3 if opname == "hmult" else 2) if plan == 2 else 2 * beta + 2 + (opname == "hmult")), outs[0]


@pytest.mark.parametrize("world,cfg,opname,L,ell,alpha,logN,batch", [
    (2, "config_4_N15.cfg", "hmult", 16, 10, 4, 15, 3),
    (4, "config_4_N15.cfg", "hrotate", 6, 5, 2, 15, 2),
])
@pytest.mark.parametrize("plan", [1, 2])
def test_sharded_batch_matches_oracle(world, cfg, opname, L, ell, alpha, logN, batch, plan, monkeypatch):
    """batch > 1 under sharding: the ops of a batch share the exchanges around each base conversion and the replicate"""
    monkeypatch.setenv("HOMULATOR_SHARD_PLAN", str(plan))
    rcs, outs = launch(world, ["gpu", cfg, opname, str(L), str(ell), str(alpha), str(logN), str(batch)], timeout=900)
    assert all(rc == 0 for rc in rcs), "\n".join(outs)
    assert "OK" in outs[0]


def test_rccl_loads_and_world1_communicator():
    """the RCCL transport itself cannot run two ranks on one GPU; check that librccl loads, a unique id can be drawn,
    a 1-rank communicator is created inside the HIP library and the op still runs bit-exact with it"""
    import numpy as np
    import torch  # noqa: F401  (maps PyTorch's own librccl.so, which the HIP library then reuses instead of loading another copy)
    from homulator_amd import host
    from oracle.homoracle import Oracle
    uid = host.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    op = host.Op("config_4_N15.cfg", "hmult", 6, 5, 2)
    op.comm_init_rccl(uid)
    op.execute(1)
    o = Oracle(15, 6, 2)
    exp = o.hmult(5, o.synth_ct(5, host.SEED), o.synth_ct(5, host.SEED + 2000), o.synth_evk(5, host.SEED + 10000))
    assert np.array_equal(op.read("out.c0"), exp[0]) and np.array_equal(op.read("out.c1"), exp[1])
    op.close()
