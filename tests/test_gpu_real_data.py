"""Real data through the product path: ciphertexts and evaluation keys of a toy RLWE scheme (tests/toy_ckks.py: keygen, encrypt,
decrypt with Python integers) are UPLOADED into the op's input and key buffers (`Op.write` -> hh_op_write_buffer), the MI355X
executes hmult / hrotate through the C++ host layer, and the downloaded result must (1) equal the oracle's on the same data bit
for bit and (2) DECRYPT to the product / the rotated message within the analytic noise bound.  (2) does not depend on the
oracle's key switch at all: it pins the GPU path by the mathematics (SURVEY.md §8c KATs v-vi, which round 1 only ran on the CPU)."""
import numpy as np
import pytest

from homulator_amd import host

pytestmark = pytest.mark.gpu
LOGN, L, ELL, ALPHA = 13, 6, 5, 2
OV = {"N": 1 << LOGN}


def negacyclic_small(a, b):
    """exact product mod X^N + 1 of small integer polynomials (int64 convolution)"""
    n = len(a)
    full = np.convolve(np.asarray(a, dtype=np.int64), np.asarray(b, dtype=np.int64))
    res = full[:n].copy()
    res[: n - 1] -= full[n:]
    return res


@pytest.fixture(scope="module")
def toy():
    from oracle.homoracle import Oracle
    from toy_ckks import Toy
    o = Oracle(LOGN, L, ALPHA)
    o.set_threads(8)
    return Toy(o, seed=4242)


def upload_keys(op, evk):
    for j in range(evk.shape[0]):
        for k in range(2):
            op.write(f"IP_Key{k}_{j}", evk[j][k])


def test_key_buffer_layout_is_the_oracles(toy):
    """the synthetic key the op generates on the device is the oracle's synth_evk in the same [digit][component][limb] order the
    upload uses"""
    op = host.Op("config_4_N15.cfg", "hmult", L, ELL, ALPHA, overrides=OV)
    op.execute(1)
    ref = toy.o.synth_evk(ELL, host.SEED + 10000)
    for j in range(ref.shape[0]):
        for k in range(2):
            assert np.array_equal(op.read(f"IP_Key{k}_{j}"), ref[j][k]), (j, k)
    op.close()


def test_hmult_on_uploaded_ciphertexts_decrypts_to_the_product(toy):
    o = toy.o
    n = o.N
    s2 = negacyclic_small(np.array([int(x) for x in toy.s]), np.array([int(x) for x in toy.s])).astype(object)
    evk = toy.evk_at_level(toy.gen_evk(s2), ELL)
    a1, a2 = toy.rng.integers(-50, 50, n), toy.rng.integers(-50, 50, n)
    m1, m2 = (a1.astype(object) * (1 << 40)), (a2.astype(object) * (1 << 40))
    ct1, ct2 = toy.encrypt(m1, ELL), toy.encrypt(m2, ELL)
    op = host.Op("config_4_N15.cfg", "hmult", L, ELL, ALPHA, overrides=OV)
    for name, data in (("ct1.c0", ct1[0]), ("ct1.c1", ct1[1]), ("ct2.c0", ct2[0]), ("ct2.c1", ct2[1])):
        op.write(name, data)
    upload_keys(op, evk)
    op.execute(1)
    out = np.stack([op.read("out.c0"), op.read("out.c1")])
    op.close()
    exp_ct = o.hmult(ELL, ct1, ct2, evk, rescale=True)
    assert np.array_equal(out[0], exp_ct[0]) and np.array_equal(out[1], exp_ct[1])
    got, _ = toy.decrypt(out, ELL - 1)
    ql = o.moduli[ELL - 1]
    exp = negacyclic_small(a1, a2).astype(object) * (1 << 80)
    err = max(abs(int(g) * ql - int(e)) for g, e in zip(got, exp))
    assert err < ql << 12, err.bit_length()  # |Dec - m1 m2 / q_last| < 2^12 against a signal of ~2^39 (measured: 2^7)
    assert max(abs(int(g)) for g in got) > 1 << 30  # and it is a signal, not zeros


def test_hrotate_on_uploaded_ciphertext_decrypts_to_the_rotation(toy):
    o = toy.o
    g = 5
    evk = toy.evk_at_level(toy.gen_evk(toy.automorph(toy.s, g)), ELL)
    m = (toy.rng.integers(-1000, 1000, o.N).astype(object) * (1 << 30))
    ct = toy.encrypt(m, ELL)
    op = host.Op("config_4_N15.cfg", "hrotate", L, ELL, ALPHA, overrides=dict(OV, galois=g))
    op.write("ct1.c0", ct[0])
    op.write("ct1.c1", ct[1])
    upload_keys(op, evk)
    op.execute(1)
    out = np.stack([op.read("out.c0"), op.read("out.c1")])
    op.close()
    exp_ct = o.hrotate(ELL, ct, g, evk)
    assert np.array_equal(out[0], exp_ct[0]) and np.array_equal(out[1], exp_ct[1])
    got, _ = toy.decrypt(out, ELL)
    exp = toy.automorph(m, g)
    assert max(abs(int(a) - int(b)) for a, b in zip(got, exp)) < 1 << 16  # measured: 2^11 against a signal of 2^40


def test_chain_on_real_data_multiplies_then_rotates(toy):
    """continuous execution on real data: hmult (relinearisation key) -> hrotate (rotation key), the ciphertext never leaves HBM;
    the result decrypts to sigma_5(m1 m2 / q_last)"""
    o = toy.o
    n, g = o.N, 5
    s2 = negacyclic_small(np.array([int(x) for x in toy.s]), np.array([int(x) for x in toy.s])).astype(object)
    relin = toy.evk_at_level(toy.gen_evk(s2), ELL)
    rot = toy.evk_at_level(toy.gen_evk(toy.automorph(toy.s, g)), ELL - 1)
    a1, a2 = toy.rng.integers(-50, 50, n), toy.rng.integers(-50, 50, n)
    ct1 = toy.encrypt(a1.astype(object) * (1 << 40), ELL)
    ct2 = toy.encrypt(a2.astype(object) * (1 << 40), ELL)
    chain = host.Chain("config_4_N15.cfg", "hmult,hrotate", L, ELL, ALPHA, overrides=dict(OV, galois=g))
    for name, data in (("ct1.c0", ct1[0]), ("ct1.c1", ct1[1]), ("ct2.c0", ct2[0]), ("ct2.c1", ct2[1])):
        chain[0].write(name, data)
    upload_keys(chain[0], relin)
    upload_keys(chain[1], rot)
    chain.execute(1)
    out = np.stack([chain[1].read("out.c0"), chain[1].read("out.c1")])
    chain.close()
    exp_ct = o.hrotate(ELL - 1, o.hmult(ELL, ct1, ct2, relin, rescale=True), g, rot)
    assert np.array_equal(out[0], exp_ct[0]) and np.array_equal(out[1], exp_ct[1])
    got, _ = toy.decrypt(out, ELL - 1)
    ql = o.moduli[ELL - 1]
    exp = toy.automorph(negacyclic_small(a1, a2).astype(object) * (1 << 80), g)
    err = max(abs(int(x) * ql - int(e)) for x, e in zip(got, exp))
    assert err < ql << 14, err.bit_length()


def test_uploads_address_the_ops_of_a_batch(toy):
    """batch = 2: copy 0 and copy 1 carry different uploaded ciphertexts, share the uploaded key (copy 0), and each equals the oracle"""
    o = toy.o
    s2 = negacyclic_small(np.array([int(x) for x in toy.s]), np.array([int(x) for x in toy.s])).astype(object)
    evk = toy.evk_at_level(toy.gen_evk(s2), ELL)
    cts = [(toy.encrypt((toy.rng.integers(-9, 9, o.N).astype(object) * (1 << 40)), ELL), toy.encrypt((toy.rng.integers(-9, 9, o.N).astype(object) * (1 << 40)), ELL))
           for _ in range(2)]
    op = host.Op("config_4_N15.cfg", "hmult", L, ELL, ALPHA, overrides=dict(OV, batch=2))
    for c, (a, b) in enumerate(cts):
        for name, data in (("ct1.c0", a[0]), ("ct1.c1", a[1]), ("ct2.c0", b[0]), ("ct2.c1", b[1])):
            op.write(name, data, copy=c)
    upload_keys(op, evk)
    op.execute(1)
    for c, (a, b) in enumerate(cts):
        exp = o.hmult(ELL, a, b, evk, rescale=True)
        assert np.array_equal(op.read("out.c0", copy=c), exp[0]) and np.array_equal(op.read("out.c1", copy=c), exp[1]), c
    op.close()
