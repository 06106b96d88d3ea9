"""Continuous multi-op execution: a chain of operations with the ciphertext resident in HBM must
equal the same sequence of oracle calls, bit-exact, at every link; the levels follow the data (hmult's rescale drops a
limb).  Upstream cannot chain operations at all (src/Operation.cpp:636)."""
import numpy as np
import pytest

from oracle.homoracle import Oracle

pytestmark = pytest.mark.gpu
SEED = 0x484F4D55
BATCH_SEED_STRIDE = 100000


def expected_chain(o, ops, ell, base_seed):
    """oracle restatement of OpChain: op k uses seed + 31 k for its own operands; its ct1 is the previous output"""
    ct = o.synth_ct(ell, base_seed)
    outs = []
    for k, name in enumerate(ops):
        s = base_seed + 31 * k
        if name == "hmult":
            ct = o.hmult(ell, ct, o.synth_ct(ell, s + 2000), o.synth_evk(ell, SEED + 31 * k + 10000), rescale=True)
            ell -= 1
        elif name == "hrotate":
            ct = o.hrotate(ell, ct, 5, o.synth_evk(ell, SEED + 31 * k + 10000))
        elif name == "hadd":
            ct = o.hadd(ell, ct, o.synth_ct(ell, s + 2000))
        elif name == "pmult":
            ct = o.pmult(ell, ct, o.fill_uniform(list(range(ell)), s + 4000))
        elif name == "padd":
            ct = o.padd(ell, ct, o.fill_uniform(list(range(ell)), s + 4000))
        outs.append(ct)
    return outs


@pytest.mark.parametrize("cfg,logN,L,ell,alpha,ops", [
    ("config_4_N15.cfg", 15, 6, 5, 2, "hmult,hrotate,hadd,hmult,padd"),
    ("config_4_N15.cfg", 15, 16, 10, 4, "hrotate,pmult,hmult,hmult"),
    ("config_4.cfg", 16, 45, 35, 15, "hmult,hrotate,hmult"),
])
def test_chain_bit_exact(cfg, logN, L, ell, alpha, ops):
    from homulator_amd import host
    o = Oracle(logN, L, alpha)
    o.set_threads(8)
    names = ops.split(",")
    chain = host.Chain(cfg, ops, L, ell, alpha)
    assert len(chain) == len(names)
    chain.execute(2)
    exp = expected_chain(o, names, ell, SEED)
    for k in range(len(names)):
        assert np.array_equal(chain[k].read("out.c0"), exp[k][0]), f"link {k} ({names[k]}) c0"
        assert np.array_equal(chain[k].read("out.c1"), exp[k][1]), f"link {k} ({names[k]}) c1"
        if k:   # the bound input really is the previous output
            assert np.array_equal(chain[k].read("ct1.c0"), exp[k - 1][0])
    chain.close()


def test_chain_batched():
    """batch = 2: both ciphertexts of the batch travel down the chain"""
    from homulator_amd import host
    o = Oracle(15, 6, 2)
    names = ["hmult", "hrotate", "hmult"]
    chain = host.Chain("config_4_N15.cfg", ",".join(names), 6, 5, 2, overrides={"batch": 2})
    chain.execute(1)
    for c in range(2):
        exp = expected_chain(o, names, 5, SEED + c * BATCH_SEED_STRIDE)
        assert np.array_equal(chain[2].read("out.c0", copy=c), exp[2][0]), f"copy {c}"
        assert np.array_equal(chain[2].read("out.c1", copy=c), exp[2][1]), f"copy {c}"
    chain.close()


def test_bind_input_between_two_ops_and_level_mismatch():
    from homulator_amd import host
    o = Oracle(15, 6, 2)
    a = host.Op("config_4_N15.cfg", "hmult", 6, 5, 2)
    a.execute(1)
    b = host.Op("config_4_N15.cfg", "hadd", 6, 4, 2, overrides={"seed": SEED + 31})
    b.bind_input("ct2", a)          # second operand this time
    b.execute(1)
    exp_a = o.hmult(5, o.synth_ct(5, SEED), o.synth_ct(5, SEED + 2000), o.synth_evk(5, SEED + 10000))
    exp_b = o.hadd(4, o.synth_ct(4, SEED + 31), exp_a)
    assert np.array_equal(b.read("out.c0"), exp_b[0]) and np.array_equal(b.read("out.c1"), exp_b[1])
    wrong = host.Op("config_4_N15.cfg", "hadd", 6, 5, 2)
    with pytest.raises(host.HostError, match="levels differ"):
        wrong.bind_input("ct1", a)
    for x in (b, wrong, a):
        x.close()


def test_chain_passes_with_changing_inputs_are_ordered_both_ways():
    """A fast producer (hadd) in front of a slow consumer (hmult), several passes enqueued back to back with NEW producer
    input before every pass.  The consumer's pass i must see the producer's pass i: producer -> consumer order AND the
    back edge (the producer's pass i+1 may not overwrite its output before the consumer's copy of pass i has read it).
    Every pass's consumer output is kept by a stream-ordered device snapshot and compared with the oracle afterwards."""
    from homulator_amd import host
    # N = 2^16, 45/35/15: one consumer pass is ~350 us of GPU work against ~100 us of host enqueue time, so the producer's
    # stream really runs passes ahead of the consumer's (with the small configurations the host is the slower side and
    # the hazard never opens: the test was checked to FAIL at this size with the back edge removed)
    logN, L, ell, alpha = 16, 45, 35, 15
    o = Oracle(logN, L, alpha)
    o.set_threads(8)
    chain = host.Chain("config_4.cfg", "hadd,hmult", L, ell, alpha)
    passes, seeds = 4, [SEED + 7777 * i for i in range(4)]
    chain.execute(1)                                   # prepared, tables warm
    for i in range(passes):
        chain[0].refill("ct1", seeds[i])               # producer input of pass i (stream-ordered after its pass i-1)
        chain.enqueue(1)
        chain[1].snapshot("out.c0", slot=i)
    chain.sync()
    evk = o.synth_evk(ell, SEED + 31 + 10000)
    for i in range(passes):
        mid = o.hadd(ell, o.synth_ct(ell, seeds[i]), o.synth_ct(ell, SEED + 2000))
        exp = o.hmult(ell, mid, o.synth_ct(ell, SEED + 31 + 2000), evk, rescale=True)
        assert np.array_equal(chain[1].snapshot_read("out.c0", slot=i), exp[0]), f"pass {i}"
    chain.close()
