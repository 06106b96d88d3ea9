"""Size-independent properties of the path at BASELINE.json's FULL sizes (N = 2^16, L = 45, l = 35, alpha = 15), on the MI355X, without the oracle:
what the mathematics of the reference's stage graph (src/Operation.cpp:592-739 TensorCompute, :9-590 KeySwitch, :741-911 Rescale) implies for ANY
correct implementation, bit for bit because everything is exact modular arithmetic:
  * hmult is symmetric in its two ciphertexts (d0 = c00 c10, d1 = c00 c11 + c01 c10, d2 = c01 c11 are symmetric; the rest is a function of them);
  * INTT(NTT(x)) = x and NTT(INTT(x)) = x for every limb of the extended basis, through every launch form (one launch, two kernels, both geometries);
  * the evaluation-form automorphism by g followed by the one by g^-1 mod 2N is the identity, and it commutes with the element-wise product;
  * base conversion is linear on inputs whose sum does not wrap: conv(a) + conv(b) = conv(a + b) mod q_t when a_i + b_i < q_i for every limb;
  * the ops of a batch are independent: op c of a batch equals the same op run alone.
  * two launch plans of one op are one function: hrotate with its automorphisms read through by their consumers (pass 12) = hrotate with the
    automorphism as a launch, at the sizes of parameter sets B and `motivation`.
These complement the oracle comparisons (tests/test_gpu_ops.py, test_gpu_kernels.py), which pin the values; here nothing but the product path runs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CFG, L, ELL, ALPHA, LOGN = "config_4.cfg", 45, 35, 15, 16
N = 1 << LOGN


def test_hmult_is_symmetric_in_its_inputs_at_full_size():
    from homulator_amd import host
    a = host.Op(CFG, "hmult", L, ELL, ALPHA)
    a.execute(1)
    ct1 = [a.read("ct1.c0"), a.read("ct1.c1")]
    ct2 = [a.read("ct2.c0"), a.read("ct2.c1")]
    out = [a.read("out.c0"), a.read("out.c1")]
    a.close()
    b = host.Op(CFG, "hmult", L, ELL, ALPHA)
    b.write("ct1.c0", ct2[0]); b.write("ct1.c1", ct2[1]); b.write("ct2.c0", ct1[0]); b.write("ct2.c1", ct1[1])
    b.execute(1)
    assert np.array_equal(b.read("out.c0"), out[0]) and np.array_equal(b.read("out.c1"), out[1])
    assert not np.array_equal(out[0], ct1[0][:ELL - 1])
    b.close()


def test_batched_op_equals_the_op_alone_at_full_size():
    from homulator_amd import host
    B = 3
    batched = host.Op(CFG, "hmult", L, ELL, ALPHA, overrides={"batch": B})
    batched.execute(1)
    for c in (0, B - 1):
        alone = host.Op(CFG, "hmult", L, ELL, ALPHA)
        for name in ("ct1.c0", "ct1.c1", "ct2.c0", "ct2.c1"):
            alone.write(name, batched.read(name, copy=c))
        alone.execute(1)
        assert np.array_equal(alone.read("out.c0"), batched.read("out.c0", copy=c)), c
        assert np.array_equal(alone.read("out.c1"), batched.read("out.c1", copy=c)), c
        alone.close()
    batched.close()


@pytest.mark.parametrize("n_copies", [1, 3, 10])   # 50 / 150 / 500 limb-polys: the one-launch form, two kernels, both geometries
def test_transform_round_trips_at_full_size(n_copies):
    from homulator_amd import hip
    ctx = hip.Context(LOGN, L, ALPHA)
    try:
        ids = ctx.ext_ids(ELL) * n_copies
        n = len(ids)
        x, y, z = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
        ctx.fill_uniform(x, ids, 11)
        ctx.ntt(x, y, ids)
        ctx.ntt(y, z, ids, inverse=True)
        want = x.download()
        assert np.array_equal(z.download(), want)
        ctx.ntt(x, y, ids, inverse=True)
        ctx.ntt(y, y, ids)                      # in place
        assert np.array_equal(y.download(), want)
        assert ctx.counter("ntt_cross_xcd") == 0
    finally:
        ctx.close()


def test_automorphism_inverse_and_multiplicativity_at_full_size():
    from homulator_amd import hip
    ctx = hip.Context(LOGN, L, ALPHA)
    try:
        ids = list(range(ELL))
        a, b, t, u, v = (ctx.alloc(ELL) for _ in range(5))
        ctx.fill_uniform(a, ids, 21)
        ctx.fill_uniform(b, ids, 22)
        g = 5
        ginv = pow(g, -1, 2 * N)
        ctx.automorph(a, t, ELL, g)
        ctx.automorph(t, u, ELL, ginv)
        assert np.array_equal(u.download(), a.download())
        # sigma_g(a * b) = sigma_g(a) * sigma_g(b) in evaluation form
        ctx.ewe(0, v, ids, a=a, b=b)            # v = a * b
        ctx.automorph(v, u, ELL, g)             # u = sigma(a b)
        ctx.automorph(b, v, ELL, g)             # v = sigma(b), t = sigma(a)
        ctx.ewe(0, t, ids, a=t, b=v)            # t = sigma(a) sigma(b)
        assert np.array_equal(t.download(), u.download())
    finally:
        ctx.close()


def test_base_conversion_is_linear_without_wrap_at_full_size():
    from homulator_amd import hip
    ctx = hip.Context(LOGN, L, ALPHA)
    try:
        ps, qs = [L + i for i in range(ALPHA)], list(range(ELL))     # the ModDown conversion's shape: 15 -> 35
        a, b, s = ctx.alloc(ALPHA), ctx.alloc(ALPHA), ctx.alloc(ALPHA)
        ctx.fill_uniform(a, ps, 31)
        ctx.fill_uniform(b, ps, 32)
        ha, hb = a.download() >> np.uint64(1), b.download() >> np.uint64(1)   # halves: a_i + b_i < q_i, no wrap
        a.upload(ha); b.upload(hb); s.upload(ha + hb)
        ca, cb, cs = ctx.alloc(ELL), ctx.alloc(ELL), ctx.alloc(ELL)
        ctx.bconv_batch([(a, None, ps, ca, None, qs), (b, None, ps, cb, None, qs), (s, None, ps, cs, None, qs)])
        ctx.ewe(3, ca, qs, a=ca, c=cb)           # ca = conv(a) + conv(b) mod q_t
        assert np.array_equal(ca.download(), cs.download())
    finally:
        ctx.close()


@pytest.mark.parametrize("shape", [(45, 35, 15), (28, 28, 28)], ids=["three-digits", "one-digit"])
@pytest.mark.parametrize("batch", [1, 3])
def test_hrotate_reads_through_the_automorphism_or_launches_it_at_full_size(shape, batch):
    """hrotate at the full sizes of parameter sets B (three digits) and `motivation` (one digit) with its automorphisms folded into their readers
    (planner pass 12: the ModUp INTT, the key product — transform x key kernel or plain inner product — and the final add gather through X -> X^g)
    against the same op with the automorphism as a launch of its own: two launch plans, one function — identical outputs, every limb, every op of a
    batch.  And rotating by g, then by g^-1 with the SAME key material is not the identity (a key switch happens each time): the outputs differ
    from the input, i.e. the comparison is not between two untouched buffers"""
    from homulator_amd import host
    Lq, ell, alpha = shape
    ov = {"batch": batch} if batch > 1 else {}
    folded = host.Op(CFG, "hrotate", Lq, ell, alpha, overrides=ov or None)
    assert not any(ln.startswith("AUTO") for ln in folded.plan()) and any(" auto_in=" in ln for ln in folded.plan())
    launched = host.Op(CFG, "hrotate", Lq, ell, alpha, overrides={**ov, "fuse_auto": 0})
    assert sum(ln.startswith("AUTO") for ln in launched.plan()) == 1
    folded.execute(1)
    launched.execute(1)
    for c in range(batch):
        for name in ("out.c0", "out.c1"):
            a, b = folded.read(name, copy=c), launched.read(name, copy=c)
            assert np.array_equal(a, b), (name, c)
        assert not np.array_equal(folded.read("out.c0", copy=c), folded.read("ct1.c0", copy=c))
    folded.close(); launched.close()
