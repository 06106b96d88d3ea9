"""GPU parity: every HIP kernel family through the C ABI vs the CPU oracle, bit-exact (integer work).
Sizes: N = 2^15 and 2^16 (the oracle finishes each case in well under a second)."""
import os

import numpy as np
import pytest

from oracle.homoracle import Oracle

pytestmark = pytest.mark.gpu
# HOMULATOR_ARITH=generic runs the whole suite on the generic arithmetic back-end (the default chain is a valid chain for it too)
FORCED_GENERIC = os.environ.get("HOMULATOR_ARITH") == "generic"


# Every kernel family runs on BOTH arithmetic back-ends (hm_create picks one from the chain it is given): "mont32" = the default chain
# (primes h 2^32 + 1: word-wise Montgomery), "survey" = SURVEY.md 8(d)'s chain as written (the largest primes = 1 mod 2N below 2^60) and
# 36 = the largest such primes below 2^36 (36-bit words as the reference's configuration models): the generic back-end.
CHAINS = [(16, "mont32"), (15, "mont32"), (16, "survey"), (15, 36)]


def make_env(logN, L, K, chain):
    from homulator_amd import hip
    o = Oracle(logN, L, K, chain=chain)
    ctx = hip.Context(logN, L, K) if chain == "mont32" else hip.Context(logN, L, K, q=o.moduli[:L], p=o.moduli[L:])
    assert ctx.counter("arith") == (1 if FORCED_GENERIC else 0 if chain == "mont32" else 1)
    return ctx, o, hip


@pytest.fixture(scope="module", params=CHAINS, ids=lambda p: f"N{p[0]}-{p[1]}")
def env(request):
    logN, chain = request.param
    ctx, o, hip = make_env(logN, 6, 3, chain)
    yield ctx, o, hip
    ctx.close()


def test_params_match_oracle(env):
    ctx, o, _ = env
    assert ctx.moduli == o.moduli


def test_ntt_forward_inverse_bit_exact(env):
    """BASELINE config #2: N=2^16 forward+inverse NTT, bit-exact vs CPU (also N=2^15, several moduli, edge values)."""
    ctx, o, _ = env
    ids = list(range(o.L + o.K))
    x = o.fill_uniform(ids, 123)
    for r, m in enumerate(ids):
        x[r, 0], x[r, 1], x[r, -1] = 0, o.moduli[m] - 1, o.moduli[m] - 1
    x[0, :] = o.moduli[0] - 1   # worst case for the lazy ranges
    x[1, :] = 0
    d = ctx.from_host(x)
    out = ctx.alloc(len(ids))
    ctx.ntt(d, out, ids)
    X = out.download()
    assert np.array_equal(X, o.ntt(ids, x))
    ctx.ntt(out, out, ids, inverse=True)  # in place
    assert np.array_equal(out.download(), x)
    # single limb, permuted limb lists, fused scale
    scale = [o.moduli[m] - 7 - m for m in ids]
    perm = ids[::-1]
    ctx.ntt(d, out, [ids[i] for i in perm], inverse=True, in_limbs=perm, out_limbs=perm, scale=[scale[i] for i in perm])
    exp = o.ewe(5, ids, o.ntt(ids, x, inverse=True), k=scale)
    assert np.array_equal(out.download(), exp)
    d.free(); out.free()


@pytest.mark.parametrize("chain", ["mont32", "survey", 45])
@pytest.mark.parametrize("logN", [13, 14, 17])
def test_ntt_other_ring_sizes(logN, chain):
    """every COL-pass instantiation (N / 256 = 32, 64, 512 rows: two rounds, three rounds, 8-column tiles) on the GPU: forward,
    inverse, fused epilogue with mix prologue, with the all-(q-1) worst case of the lazy ranges; 2^15 and 2^16 run above"""
    L, K = 3, 2
    ctx, o, _ = make_env(logN, L, K, chain)
    try:
        ids = [0, 1, 2, 3, 4, 0, 4]
        x, mn, ad, mx = (o.fill_uniform(ids, s) for s in (123, 124, 125, 126))
        x[0, :] = o.moduli[ids[0]] - 1
        x[1, :4] = [0, o.moduli[ids[1]] - 1, 1, 2]
        d, out = ctx.from_host(x), ctx.alloc(len(ids))
        ctx.ntt(d, out, ids)
        assert np.array_equal(out.download(), o.ntt(ids, x))
        ctx.ntt(out, out, ids, inverse=True)
        assert np.array_equal(out.download(), x)
        k = [o.moduli[m] - 2 - r for r, m in enumerate(ids)]
        mk = [(kk * 3 + 1) % o.moduli[m] for kk, m in zip(k, ids)]
        ak = [(kk * 5 + 1) % o.moduli[m] for kk, m in zip(k, ids)]
        dmn, dad, dmx = ctx.from_host(mn), ctx.from_host(ad), ctx.from_host(mx)
        ctx.ntt_mix_sub_scale(d, dmn, out, ids, k, addend=dad, addend_k=ak, mix=dmx, mix_k=mk)
        xin = o.ewe(3, ids, x, None, o.ewe(5, ids, mx, k=mk))
        exp = o.ewe(3, ids, o.ewe(6, ids, mn, None, o.ntt(ids, xin), k=k), None, o.ewe(5, ids, ad, k=ak))
        assert np.array_equal(out.download(), exp)
    finally:
        ctx.close()


def test_ntt_many_limbs_over_one_launch(env):
    ctx, o, _ = env
    ids = [i % (o.L + o.K) for i in range(140)]   # > HM_MAX_LIMBS: exercises the chunked launch
    d = ctx.alloc(140)
    ctx.fill_uniform(d, ids, 77)
    x = d.download()
    assert np.array_equal(x[[0, 5, 139]], np.stack([o.fill_uniform(ids, 77)[i] for i in (0, 5, 139)]))
    ctx.ntt(d, d, ids)
    got = d.download()
    sel = [0, 1, 127, 128, 139]
    assert np.array_equal(got[sel], o.ntt([ids[i] for i in sel], x[sel]))
    d.free()


def test_ewe_all_opcodes(env):
    ctx, o, hip = env
    ids = [0, 3, o.L, o.L + o.K - 1]
    n = len(ids)
    A, B, Cc, D = (o.fill_uniform(ids, s) for s in (1, 2, 3, 4))
    for r, m in enumerate(ids):
        q = o.moduli[m]
        A[r, :4] = [0, q - 1, q - 1, 1]
        B[r, :4] = [q - 1, q - 1, 0, q - 1]
        Cc[r, :4] = [q - 1, 0, q - 1, q - 1]
        D[r, :4] = [q - 1, q - 1, q - 1, 0]
    k = [o.moduli[m] - 2 - r for r, m in enumerate(ids)]
    da, db, dc, dd = (ctx.from_host(v) for v in (A, B, Cc, D))
    out = ctx.alloc(n)
    for op in range(8):
        ctx.ewe(op, out, ids, a=da, b=db, c=dc, d=dd, k=k)
        assert np.array_equal(out.download(), o.ewe(op, ids, A, B, Cc, D, k=k)), f"opcode {op}"
    ctx.ewe(hip.OP_SUB_SCALE_ADD, out, ids, a=da, c=dc, d=dd, k=k)
    exp = o.ewe(3, ids, o.ewe(6, ids, A, None, Cc, k=k), None, D)
    assert np.array_equal(out.download(), exp)
    # limb lists
    ctx.ewe(hip.OP_MUL, out, [ids[2], ids[0]], a=da, b=db, la=[2, 0], lb=[2, 0], lo=[1, 3])
    got = out.download()
    assert np.array_equal(got[1], o.ewe(0, [ids[2]], A[2:3], B[2:3])[0])
    assert np.array_equal(got[3], o.ewe(0, [ids[0]], A[0:1], B[0:1])[0])
    for b_ in (da, db, dc, dd, out):
        b_.free()


def test_bconv_modup_moddown_shapes(env):
    ctx, o, _ = env
    L, K = o.L, o.K
    cases = [([0, 1, 2], [3, 4, 5, L, L + 1, L + 2]),       # digit 0 of a ModUp at ell = 6
             ([3, 4, 5], [0, 1, 2, L, L + 1, L + 2]),       # digit 1
             ([L, L + 1, L + 2], [0, 1, 2, 3, 4, 5]),       # ModDown P -> Q
             ([4], [0, L + 2])]                               # partial digit of one limb
    for in_ids, out_ids in cases:
        x = o.fill_uniform(in_ids, 31)
        for r, m in enumerate(in_ids):
            x[r, :3] = [o.moduli[m] - 1, 0, 1]
        d = ctx.from_host(x)
        out = ctx.alloc(len(out_ids))
        ctx.bconv(d, in_ids, out, out_ids)
        assert np.array_equal(out.download(), o.bconv_matmul(in_ids, out_ids, x))
        qh, tb = ctx.bconv_consts(in_ids, out_ids)
        eq, et = o.bconv_consts(in_ids, out_ids)
        assert np.array_equal(qh, eq) and np.array_equal(tb, et)
        d.free(); out.free()


def test_bconv_batch_many_problems_one_launch(env):
    """hm_bconv_batch with more problems than the kernel-argument segment used to hold (4): the records travel in a device
    table, the output chunk is sized by the launch; mixed input-basis sizes split into one launch per size; ragged output
    counts (not a multiple of any chunk); inputs at q - 1, 0, 1"""
    ctx, o, _ = env
    L, K = o.L, o.K
    shapes = [([0, 1, 2], [3, 4, 5, L, L + 1, L + 2, L + 3][:6 + (i % 2)]) if i % 3 else ([4, 5], [0, 1, 2, 3, L]) for i in range(23)]
    srcs, dsts, probs, exps = [], [], [], []
    for i, (in_ids, out_ids) in enumerate(shapes):
        out_ids = [t for t in out_ids if t < L + K]
        x = o.fill_uniform(in_ids, 100 + i)
        for r, m in enumerate(in_ids):
            x[r, :3] = [o.moduli[m] - 1, 0, 1]
        d, out = ctx.from_host(x), ctx.alloc(len(out_ids))
        srcs.append(d); dsts.append(out)
        probs.append((d, None, in_ids, out, None, out_ids))
        exps.append(o.bconv_matmul(in_ids, out_ids, x))
    ctx.bconv_batch(probs)
    for out, e in zip(dsts, exps):
        assert np.array_equal(out.download(), e)
    for b in srcs + dsts:
        b.free()


@pytest.mark.parametrize("rot", [1, 2, 7])
def test_automorph_eval(env, rot):
    ctx, o, _ = env
    g = pow(5, rot, 2 * o.N)
    ids = [0, 2, o.L]
    x = o.fill_uniform(ids, 55)
    d = ctx.from_host(x)
    out = ctx.alloc(3)
    ctx.automorph(d, out, 3, g)
    assert np.array_equal(out.download(), o.automorph_eval(x, g))
    d.free(); out.free()


@pytest.mark.parametrize("n,opts", [(7, {}), (7, {"ntt_fused_small": 0}), (130, {})], ids=["one-launch", "small-two-kernels", "wide"])
def test_transforms_that_read_through_an_automorphism(env, n, opts):
    """round 6: hrotate's AUTO launch folded into its consumers (hm_ntt_desc.in_galois, hm_ntt_fused_desc.addend_galois).  INTT(automorph_g(x)) and
    (minuend - NTT(x)) * k + automorph_g(addend) [* addend_k] in one call each, against the oracle's automorphism + transform: per-limb Galois
    elements (rotations, the conjugation, 1 / 0 = as stored), limb-polys without addend beside gathered ones, in every kernel form a launch
    size selects (both passes in one launch, the small-launch pair of kernels, the wide geometry); and what the call refuses"""
    ctx, o, hip = env
    before = {name: ctx.counter(name) for name in opts}
    for name, v in opts.items():
        ctx.set_option(name, v)
    try:
        N2 = 2 * o.N
        ids = [i % (o.L + o.K) for i in range(n)]
        gs = [[5, N2 - 1, 25, 1, 0, pow(5, 7, N2), 3][i % 7] for i in range(n)]
        x, mn, ad = (o.fill_uniform(ids, s) for s in (11, 12, 13))
        dx, dmn, dad = (ctx.from_host(v) for v in (x, mn, ad))
        out = ctx.alloc(n)
        auto = lambda v: np.stack([o.automorph_eval(v[r][None], g if g else 1)[0] for r, g in enumerate(gs)])
        sel = list(range(n)) if n <= 16 else [0, 1, 2, 3, 4, 5, 6, 63, 64, 127, 128, 129]
        pick = lambda a: a[sel]
        sids = [ids[i] for i in sel]
        ctx.ntt(dx, out, ids, inverse=True, in_galois=gs)
        assert np.array_equal(pick(out.download()), o.ntt(sids, pick(auto(x)), inverse=True))
        scale = [o.moduli[m] - 7 - (r % 5) for r, m in enumerate(ids)]
        ctx.ntt(dx, out, ids, inverse=True, in_galois=gs, scale=scale, out_packed=[r % 2 for r in range(n)])
        got, exp = pick(out.download()), o.ewe(5, sids, o.ntt(sids, pick(auto(x)), inverse=True), k=[scale[i] for i in sel])
        for j, r in enumerate(sel):
            e = exp[j]
            assert np.array_equal(got[j], (e & np.uint64(0x3FFFFFFF)) | ((e >> np.uint64(30)) << np.uint64(32)) if r % 2 else e), r
        k = [o.moduli[m] - 3 - (r % 9) for r, m in enumerate(ids)]
        ak = [(kk * 11 + 5) % o.moduli[m] for kk, m in zip(k, ids)]
        NO = 0xFFFFFFFF
        al = [NO if r % 3 == 2 else r for r in range(n)]
        base = o.ewe(6, sids, pick(mn), None, o.ntt(sids, pick(x)), k=[k[i] for i in sel])
        ctx.ntt_mix_sub_scale(dx, dmn, out, ids, k, addend=dad, addend_limbs=al, addend_galois=gs)
        got, with_add = pick(out.download()), o.ewe(3, sids, base, None, pick(auto(ad)))
        for j, r in enumerate(sel):
            assert np.array_equal(got[j], base[j] if r % 3 == 2 else with_add[j]), r
        ctx.ntt_mix_sub_scale(dx, dmn, out, ids, k, addend=dad, addend_k=ak, addend_galois=gs)
        assert np.array_equal(pick(out.download()), o.ewe(3, sids, base, None, o.ewe(5, sids, pick(auto(ad)), k=[ak[i] for i in sel])))
        with pytest.raises(hip.HmError):
            ctx.ntt(dx, dx, ids, inverse=True, in_galois=gs)                                  # gathered input, in place
        with pytest.raises(hip.HmError):                                                      # ... or onto another entry's source
            ctx.ntt(dx, dx, ids, inverse=True, in_galois=gs, out_limbs=list(range(1, n)) + [0])
        with pytest.raises(hip.HmError):                                                      # a gathered addend that the call writes
            ctx.ntt_mix_sub_scale(dx, dmn, dad, ids, k, addend=dad, addend_galois=gs)
        with pytest.raises(hip.HmError):
            ctx.ntt(dx, out, ids, inverse=False, in_galois=gs)                                # forward transform
        with pytest.raises(hip.HmError):
            ctx.ntt(dx, out, ids, inverse=True, in_galois=[4] * n)                            # even Galois element
        with pytest.raises(hip.HmError):
            ctx.ntt_mix_sub_scale(dx, dmn, out, ids, k, addend_galois=gs)                     # no addend
        with pytest.raises(hip.HmError):
            ctx.ntt_mix_sub_scale(dx, dmn, out, ids, k, mix=dad, mix_k=ak, addend=dad, addend_galois=gs)   # with the mix prologue
        for b_ in (dx, dmn, dad, out):
            b_.free()
    finally:
        for name, v in before.items():
            ctx.set_option(name, v)


def test_errors_are_loud(env):
    ctx, o, hip = env
    d = ctx.alloc(2)
    with pytest.raises(hip.HmError):
        ctx.ntt(d, d, [999, 0])                       # bad modulus id
    with pytest.raises(hip.HmError):
        ctx.automorph(d, d, 2, 4)                     # even galois element
    with pytest.raises(hip.HmError):
        ctx.bconv(d, [0, 1], d, [1, 2])               # modulus in both bases
    with pytest.raises(hip.HmError):
        ctx.ewe(hip.OP_MUL_CONST, d, [0, 1], a=d)     # missing constants
    d.free()


def test_fused_ntt_sub_scale_and_tensor(env):
    """fused forward NTT epilogue out = (minuend - NTT(in)) * k [+ addend] and the one-pass tensor product"""
    ctx, o, hip = env
    ids = [0, 0, 3, o.L, o.L, 5, 2]            # pairs of equal moduli + singles: exercises the XCD pairing
    n = len(ids)
    x, mn, ad, dd = (o.fill_uniform(ids, s) for s in (1, 2, 3, 4))
    for r, m in enumerate(ids):
        q = o.moduli[m]
        x[r, :3] = [q - 1, 0, q - 1]
        mn[r, :3] = [0, q - 1, q - 1]
        ad[r, :3] = [q - 1, q - 1, 0]
    k = [o.moduli[m] - 3 - r for r, m in enumerate(ids)]
    dx, dmn, dad, ddd = (ctx.from_host(v) for v in (x, mn, ad, dd))
    out = ctx.alloc(n)
    exp = o.ewe(6, ids, mn, None, o.ntt(ids, x), k=k)
    ctx.ntt_sub_scale(dx, dmn, out, ids, k)
    assert np.array_equal(out.download(), exp)
    ctx.ntt_sub_scale(dx, dmn, out, ids, k, addend=dad)
    assert np.array_equal(out.download(), o.ewe(3, ids, exp, None, ad))
    # merged ModDown + rescale form: prologue x + mk * dd, constant on the addend, mixed limb lists
    mk = [(kk * 7 + 3) % o.moduli[m] for kk, m in zip(k, ids)]
    ak = [(kk * 11 + 5) % o.moduli[m] for kk, m in zip(k, ids)]
    xin = o.ewe(3, ids, x, None, o.ewe(5, ids, dd, k=mk))
    exp2 = o.ewe(3, ids, o.ewe(6, ids, mn, None, o.ntt(ids, xin), k=k), None, o.ewe(5, ids, ad, k=ak))
    ctx.ntt_mix_sub_scale(dx, dmn, out, ids, k, mix=ddd, mix_k=mk, addend=dad, addend_k=ak)
    assert np.array_equal(out.download(), exp2)
    ctx.ntt_mix_sub_scale(dx, dmn, out, ids, k, mix=ddd, mix_k=mk)                      # no addend
    assert np.array_equal(out.download(), o.ewe(6, ids, mn, None, o.ntt(ids, xin), k=k))
    ctx.ntt_mix_sub_scale(dx, dmn, out, ids, k, addend=dad, addend_k=ak)                # no prologue
    assert np.array_equal(out.download(), o.ewe(3, ids, exp, None, o.ewe(5, ids, ad, k=ak)))
    # per-limb optional addend: HM_NO_LIMB drops it for that limb-poly only
    NO = 0xFFFFFFFF
    al = [NO if r % 2 else r for r in range(n)]
    ctx.ntt_sub_scale(dx, dmn, out, ids, k, addend=dad, addend_limbs=al)
    got, with_add = out.download(), o.ewe(3, ids, exp, None, ad)
    for r in range(n):
        assert np.array_equal(got[r], exp[r] if r % 2 else with_add[r]), r
    with pytest.raises(hip.HmError):
        ctx.ntt_mix_sub_scale(dx, dmn, out, ids, k, mix=ddd)                            # mix without mix_k
    with pytest.raises(hip.HmError):
        ctx.ntt_mix_sub_scale(dx, dmn, out, ids, k, addend=dad, addend_k=[0] * n)       # zero constant
    o0, o1, o2 = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
    ctx.tensor(dx, dmn, dad, ddd, o0, o1, o2, ids)
    assert np.array_equal(o0.download(), o.ewe(0, ids, x, mn))
    assert np.array_equal(o1.download(), o.ewe(1, ids, x, dd, ad, mn))
    assert np.array_equal(o2.download(), o.ewe(0, ids, ad, dd))
    for b_ in (dx, dmn, dad, ddd, out, o0, o1, o2):
        b_.free()


def test_fused_ntt_epilogue_all_distinct_constants(env):
    """every limb-poly with its own (k, addend_k): the per-limb constants come from the launch's device table"""
    ctx, o, hip = env
    n = 70
    ids = [r % (o.L + o.K) for r in range(n)]
    x, mn, ad = (o.fill_uniform(ids, s) for s in (11, 12, 13))
    k = [o.moduli[m] - 2 - r for r, m in enumerate(ids)]          # all different
    ak = [(kk * 5 + 1) % o.moduli[m] for kk, m in zip(k, ids)]
    dx, dmn, dad, out = ctx.from_host(x), ctx.from_host(mn), ctx.from_host(ad), ctx.alloc(n)
    ctx.ntt_mix_sub_scale(dx, dmn, out, ids, k, addend=dad, addend_k=ak)
    exp = o.ewe(3, ids, o.ewe(6, ids, mn, None, o.ntt(ids, x), k=k), None, o.ewe(5, ids, ad, k=ak))
    assert np.array_equal(out.download(), exp)
    for b_ in (dx, dmn, dad, out):
        b_.free()


@pytest.mark.parametrize("terms,outs", [(1, 2), (2, 1), (3, 2), (4, 2)])
def test_inner_product_one_pass(env, terms, outs):
    """K5: out[i][k] = sum_j x[i][j] * y[i][k][j], worst-case operands (q-1) included"""
    ctx, o, _ = env
    ids = [0, 2, o.L, o.L + 1, 5]
    n = len(ids)
    X = [o.fill_uniform(ids, 10 + j) for j in range(terms)]
    Y = [[o.fill_uniform(ids, 100 + 10 * k + j) for j in range(terms)] for k in range(outs)]
    for r, m in enumerate(ids):
        for j in range(terms):
            X[j][r, :2] = o.moduli[m] - 1
            for k in range(outs):
                Y[k][j][r, :2] = o.moduli[m] - 1
    xb = ctx.from_host(np.concatenate(X))                      # limb of x_j[i] = j*n + i
    yb = ctx.from_host(np.concatenate([Y[k][j] for k in range(outs) for j in range(terms)]))
    out = ctx.alloc(n * outs)
    xl = [j * n + i for i in range(n) for j in range(terms)]
    yl = [(k * terms + j) * n + i for i in range(n) for k in range(outs) for j in range(terms)]
    ol = [k * n + i for i in range(n) for k in range(outs)]
    ctx.inner_product(xb, xl, yb, yl, out, ol, ids, terms, outs)
    got = out.download()
    for k in range(outs):
        exp = o.ewe(0, ids, X[0], Y[k][0])
        for j in range(1, terms):
            exp = o.ewe(2, ids, X[j], Y[k][j], exp)
        assert np.array_equal(got[k * n:(k + 1) * n], exp), (k, terms)
    for b_ in (xb, yb, out):
        b_.free()


@pytest.mark.parametrize("terms,outs", [(1, 1), (1, 2), (2, 2), (3, 2), (4, 2), (3, 1)])
def test_ntt_inner_product_fused(env, terms, outs):
    """K1 x K5 (the HPIP unit): out[i][k] = sum_j X_j[i] * y[i][k][j] with X_j = NTT(x_j) for the digits that are
    transformed and x_j itself for a digit's own (evaluation-form) limbs, against the oracle's NTT + MAC chain.  Limbs with a
    mix of transformed / own digits, worst-case operands (q - 1), duplicate moduli (same-modulus groups), more limbs than one
    launch carries."""
    ctx, o, _ = env
    M = o.L + o.K
    ids = [(i * 5 + 1) % M for i in range(37)] + [0, 0, 0, 0, M - 1, M - 1]
    n = len(ids)
    X = [o.fill_uniform(ids, 10 + j) for j in range(terms)]
    Y = [[o.fill_uniform(ids, 100 + 10 * k + j) for j in range(terms)] for k in range(outs)]
    for r, m in enumerate(ids[:6]):
        for j in range(terms):
            X[j][r, :] = o.moduli[m] - 1          # worst case of the lazy ranges through the transform and the MAC
            for k in range(outs):
                Y[k][j][r, :3] = o.moduli[m] - 1
    coeff = [[(i + j) % 3 != 0 for j in range(terms)] for i in range(n)]   # which digits go through the transform
    coeff[1] = [True] * terms
    coeff[2] = [False] * terms
    xb = ctx.from_host(np.concatenate(X))
    yb = ctx.from_host(np.concatenate([Y[k][j] for k in range(outs) for j in range(terms)]))
    hand, out = ctx.alloc(n * terms), ctx.alloc(n * outs)
    xl = [j * n + i for i in range(n) for j in range(terms)]
    yl = [(k * terms + j) * n + i for i in range(n) for k in range(outs) for j in range(terms)]
    ol = [k * n + i for i in range(n) for k in range(outs)]
    ctx.ntt_inner_product(xb, xl, [c for row in coeff for c in row], hand, xl, yb, yl, out, ol, ids, terms, outs)
    got = out.download()
    XE = []
    for j in range(terms):
        t = o.ntt(ids, X[j])
        for i in range(n):
            if not coeff[i][j]:
                t[i] = X[j][i]
        XE.append(t)
    for k in range(outs):
        exp = o.ewe(0, ids, XE[0], Y[k][0])
        for j in range(1, terms):
            exp = o.ewe(2, ids, XE[j], Y[k][j], exp)
        assert np.array_equal(got[k * n:(k + 1) * n], exp), (k, terms)
    for b_ in (xb, yb, out, hand):
        b_.free()


@pytest.mark.parametrize("terms,outs", [(1, 2), (3, 2), (2, 1)])
def test_inner_product_reads_x_through_an_automorphism(env, terms, outs):
    """round 6 (hm_inner_product_ex, x_galois): the plain inner product with its x operands read through X -> X^g, against the call on operands the
    oracle has rotated: a one-digit key switch of hrotate multiplies the rotated c1 itself with the key"""
    ctx, o, hip = env
    M = o.L + o.K
    ids = [(i * 5 + 1) % M for i in range(21)]
    n = len(ids)
    X = [o.fill_uniform(ids, 10 + j) for j in range(terms)]
    Y = [[o.fill_uniform(ids, 100 + 10 * k + j) for j in range(terms)] for k in range(outs)]
    xb = ctx.from_host(np.concatenate(X))
    yb = ctx.from_host(np.concatenate([Y[k][j] for k in range(outs) for j in range(terms)]))
    out = ctx.alloc(n * outs)
    xl = [j * n + i for i in range(n) for j in range(terms)]
    yl = [(k * terms + j) * n + i for i in range(n) for k in range(outs) for j in range(terms)]
    ol = [k * n + i for i in range(n) for k in range(outs)]
    for g in (5, 2 * o.N - 1, 25):
        ctx.inner_product(xb, xl, yb, yl, out, ol, ids, terms, outs, x_galois=g)
        got = out.download()
        XR = [o.automorph_eval(X[j], g) for j in range(terms)]
        for k in range(outs):
            exp = o.ewe(0, ids, XR[0], Y[k][0])
            for j in range(1, terms):
                exp = o.ewe(2, ids, XR[j], Y[k][j], exp)
            assert np.array_equal(got[k * n:(k + 1) * n], exp), (g, k)
    with pytest.raises(hip.HmError):
        ctx.inner_product(xb, xl, yb, yl, out, ol, ids, terms, outs, x_galois=2)
    with pytest.raises(hip.HmError):                                                          # gathered operands the call writes
        ctx.inner_product(xb, xl, yb, yl, xb, ol, ids, terms, outs, x_galois=5)
    for b_ in (xb, yb, out):
        b_.free()


@pytest.mark.parametrize("n,inv", [(70, False), (70, True), (9, True), (9, False)], ids=["wide", "wide-inverse-out", "small-inverse-out", "small"])
@pytest.mark.parametrize("outs", [2, 1])
def test_ntt_inner_product_reads_own_digits_through_an_automorphism(env, n, inv, outs):
    """round 6 (hm_ntt_ip_desc.x_galois): the digits that arrive in evaluation form are read through X -> X^g — the same result as the call on
    operands hm_automorph has rotated first, bit for bit, in the wide and the small-launch geometry, with and without the inverse first pass on
    the outputs, for a rotation that swaps the words of a 16-byte unit (5) and the conjugation that does not; transformed digits are untouched"""
    ctx, o, hip = env
    M, terms = o.L + o.K, 3
    ids = [(i * 5 + 1) % M for i in range(n)]
    X = [o.fill_uniform(ids, 10 + j) for j in range(terms)]
    Y = [[o.fill_uniform(ids, 100 + 10 * k + j) for j in range(terms)] for k in range(outs)]
    coeff = [[(i + j) % 3 != 0 for j in range(terms)] for i in range(n)]
    xb = ctx.from_host(np.concatenate(X))
    yb = ctx.from_host(np.concatenate([Y[k][j] for k in range(outs) for j in range(terms)]))
    hand, out, ref, rot = ctx.alloc(n * terms), ctx.alloc(n * outs), ctx.alloc(n * outs), ctx.alloc(n * terms)
    xl = [j * n + i for i in range(n) for j in range(terms)]
    yl = [(k * terms + j) * n + i for i in range(n) for k in range(outs) for j in range(terms)]
    ol = [k * n + i for i in range(n) for k in range(outs)]
    flags = [c for row in coeff for c in row]
    oi = [1 if inv and i % 2 else 0 for i in range(n)] if inv else None
    for g in (5, 2 * o.N - 1):
        # reference: rotate the evaluation-form operands first (the transformed digits stay as they are), then the plain call
        Xr = [X[j].copy() for j in range(terms)]
        for j in range(terms):
            for i in range(n):
                if not coeff[i][j]:
                    Xr[j][i] = o.automorph_eval(X[j][i][None], g)[0]
        rot.upload(np.concatenate(Xr))
        ctx.ntt_inner_product(rot, xl, flags, hand, xl, yb, yl, ref, ol, ids, terms, outs, out_inverse=oi)
        ctx.ntt_inner_product(xb, xl, flags, hand, xl, yb, yl, out, ol, ids, terms, outs, out_inverse=oi, x_galois=g)
        assert np.array_equal(out.download(), ref.download()), g
    with pytest.raises(hip.HmError):
        ctx.ntt_inner_product(xb, xl, flags, hand, xl, yb, yl, out, ol, ids, terms, outs, x_galois=6)
    for b_ in (xb, yb, out, ref, hand, rot):
        b_.free()


@pytest.mark.parametrize("chain", ["mont32", "survey"])
@pytest.mark.parametrize("n_in", [1, 2, 5, 9, 15])
def test_ntt_inner_product_with_conversion_inside(n_in, chain):
    """ModUp_BCONV + ModUp_NTT + inner product in one call (hm_ntt_ip_desc.conv): the conversion runs inside the first pass of the
    transform that consumes it (k_bconv_col), the converted limb-polys never exist.  Against the oracle's conversion, transform and
    MAC chain; two digits of n_in limbs each, two "ops" sharing the key, output limbs = everything outside the digit."""
    L, K = 2 * n_in, 3
    ctx, o, _ = make_env(16, L, K, chain)
    try:
        ell, beta, nops = 2 * n_in, 2, 2
        ext = o.ext_ids(ell); E = len(ext)
        digits = [list(range(0, n_in)), list(range(n_in, 2 * n_in))]
        own = o.fill_uniform(list(range(ell)) * nops, 5).reshape(nops, ell, -1)          # evaluation-form limbs (a digit's own)
        scaled = o.fill_uniform(list(range(ell)) * nops, 6).reshape(nops, ell, -1)       # coefficient-form, already x q_hat^-1
        scaled[0, 0, :] = o.moduli[0] - 1
        evk = np.stack([o.fill_uniform(ext, 100 + 10 * k + j) for k in range(2) for j in range(beta)])   # limb (k*beta + j)*E + t
        src, ownb, evkb = ctx.from_host(scaled.reshape(-1, 1 << 16)), ctx.from_host(own.reshape(-1, 1 << 16)), ctx.from_host(evk.reshape(-1, 1 << 16))
        hand, out = ctx.alloc(nops * beta * E), ctx.alloc(nops * 2 * E)
        conv, xl, flags, hl, yl, ol, mods = [], [], [], [], [], [], []
        for b in range(nops):
            for j, dj in enumerate(digits):
                outs = [t for t in range(E) if t not in dj]
                conv.append((src, [b * ell + i for i in dj], dj, [(b * beta + j) * E + t for t in outs], [ext[t] for t in outs]))
            for t in range(E):
                for j, dj in enumerate(digits):
                    isown = t in dj
                    xl.append(b * ell + t if isown else 0); hl.append((b * beta + j) * E + t); flags.append(0 if isown else 1)
                for k in range(2):
                    yl += [(k * beta + j) * E + t for j in range(beta)]
                    ol.append((b * 2 + k) * E + t)
                mods.append(ext[t])
        # both geometries of the transform x key kernel: the small-launch one (512-thread workgroups, 8 coefficients per thread: launches of
        # up to 64 limb records, round 5) and the wide one must agree bit for bit
        ctx.set_option("nip_small_limbs", 0)
        ctx.ntt_inner_product(ownb, xl, flags, hand, hl, evkb, yl, out, ol, mods, beta, 2, conv=conv)
        wide = out.download()
        ctx.set_option("nip_small_limbs", 4096)
        ctx.fill_uniform(out, [0] * (nops * 2 * E), 3)
        ctx.ntt_inner_product(ownb, xl, flags, hand, hl, evkb, yl, out, ol, mods, beta, 2, conv=conv)
        assert np.array_equal(out.download(), wide), "k_ntt_row_ip8 differs from k_ntt_row_ip"
        ctx.set_option("nip_small_limbs", 64)
        got = out.download().reshape(nops, 2, E, -1)
        for b in range(nops):
            X = []
            for j, dj in enumerate(digits):
                outs = [t for t in range(E) if t not in dj]
                z = o.ntt([ext[t] for t in outs], o.bconv_matmul(dj, [ext[t] for t in outs], scaled[b, dj]))
                full = np.zeros((E, 1 << 16), dtype=np.uint64)
                full[outs] = z
                full[dj] = own[b, dj]
                X.append(full)
            for k in range(2):
                exp = o.ewe(1, ext, X[0], evk[k * beta + 0], X[1], evk[k * beta + 1])
                assert np.array_equal(got[b, k], exp), (n_in, b, k)
    finally:
        ctx.close()


@pytest.mark.parametrize("logN,chain", [(16, "mont32"), (15, "mont32"), (16, 50), (15, "survey")])
@pytest.mark.parametrize("n_in,outs_per_wg", [(1, 0), (2, 2), (5, 1), (9, 2), (15, 2), (15, 1)])
def test_mix_sub_scale_with_conversion_inside(logN, chain, n_in, outs_per_wg):
    """ModDown_BCONV + ModDowNTT + ModDownSub + rescale in one call (hm_ntt_fused_desc.conv, round 4): the P -> Q conversion runs inside the
    first pass of the merged transform (k_bconv_col with the mix prologue), ModdownBConvOut never exists.  Two "keys" of n_in special limbs
    each converted to an ODD number of Q limbs (the last output group of a workgroup pair is half empty), with and without the mix operand,
    plus limb-polys that are NOT fed by a conversion in the same call; one or two outputs per workgroup; N = 2^16 and 2^15.  Against the
    oracle's conversion, transform and element-wise chain."""
    ell, K, N = 7, n_in, 1 << logN
    ctx, o, _ = make_env(logN, ell, K, chain)
    try:
        ctx.set_option("bconv_col_outs", outs_per_wg)
        qs, ps = list(range(ell)), list(range(ell, ell + K))
        nkeys = 2
        y = o.fill_uniform(ps * nkeys, 21).reshape(nkeys, K, N)              # coefficient form, already x p_hat^-1
        y[0, 0, :] = o.moduli[ps[0]] - 1
        extra_ids = [1, 4]                                                    # two limb-polys with an ordinary input (not converted)
        extra = o.fill_uniform(extra_ids, 22)
        nq = nkeys * ell + len(extra_ids)
        ids = qs * nkeys + extra_ids
        mn, ad, mx = (o.fill_uniform(ids, s) for s in (23, 24, 25))
        k = [o.moduli[m] - 2 - r for r, m in enumerate(ids)]
        mk = [(kk * 3 + 1) % o.moduli[m] for kk, m in zip(k, ids)]
        ak = [(kk * 5 + 1) % o.moduli[m] for kk, m in zip(k, ids)]
        ysrc, xsrc = ctx.from_host(y.reshape(-1, N)), ctx.from_host(extra)
        dmn, dad, dmx, out = ctx.from_host(mn), ctx.from_host(ad), ctx.from_host(mx), ctx.alloc(nq)
        conv = [(ysrc, [kk_ * K + i for i in range(K)], ps, [kk_ * ell + t for t in range(ell)], qs) for kk_ in range(nkeys)]
        in_limbs = [0] * (nkeys * ell) + [0, 1]
        # expected: the input of every limb-poly, then the fused chain
        xin = np.concatenate([o.bconv_matmul(ps, qs, y[kk_]) for kk_ in range(nkeys)] + [extra])
        for with_mix in (True, False):
            x = o.ewe(3, ids, xin, None, o.ewe(5, ids, mx, k=mk)) if with_mix else xin
            exp = o.ewe(3, ids, o.ewe(6, ids, mn, None, o.ntt(ids, x), k=k), None, o.ewe(5, ids, ad, k=ak))
            ctx.fill_uniform(out, ids, 99)
            ctx.ntt_mix_sub_scale(xsrc, dmn, out, ids, k, addend=dad, addend_k=ak, mix=dmx if with_mix else None, mix_k=mk if with_mix else None,
                                  in_limbs=in_limbs, conv=conv)
            assert np.array_equal(out.download(), exp), (logN, n_in, with_mix)
        # every limb-poly converted, no `in` at all
        sel = list(range(nkeys * ell))
        sub = lambda a: a[sel]
        ids2 = [ids[i] for i in sel]
        out2 = ctx.alloc(len(sel))
        d2 = [ctx.from_host(sub(a)) for a in (mn, ad, mx)]
        x = o.ewe(3, ids2, sub(xin), None, o.ewe(5, ids2, sub(mx), k=[mk[i] for i in sel]))
        exp = o.ewe(3, ids2, o.ewe(6, ids2, sub(mn), None, o.ntt(ids2, x), k=[k[i] for i in sel]), None, o.ewe(5, ids2, sub(ad), k=[ak[i] for i in sel]))
        ctx.ntt_mix_sub_scale(None, d2[0], out2, ids2, [k[i] for i in sel], addend=d2[1], addend_k=[ak[i] for i in sel], mix=d2[2], mix_k=[mk[i] for i in sel],
                              conv=[(ysrc, c_[1], c_[2], c_[3], c_[4]) for c_ in conv])
        assert np.array_equal(out2.download(), exp)
    finally:
        ctx.close()


@pytest.mark.parametrize("chain", ["mont32", "survey"])
def test_inner_product_hands_over_the_first_pass_of_the_inverse_transform(chain):
    """hm_ntt_ip_desc.out_inverse + hm_ntt_second_pass (round 5; InnerProOut -> ModDownINTTOut, src/Operation.cpp:294-445): the flagged limbs'
    outputs leave the kernel as the first pass (the ROW pass over the workgroup's own rows) of their inverse transform and the second
    call finishes it in place, with a scale; together bit-identical to the plain call followed by hm_ntt(inverse).  Mixed launch
    (flagged and unflagged limbs), transformed and evaluation-form digits, worst-case operands, both keys and one key, repeated"""
    L, K, N = 4, 3, 1 << 16
    ctx, o, _ = make_env(16, L, K, chain)
    try:
        ids = [0, 1, 2, 3, 4, 5, 6, 4, 5]
        n, T = len(ids), 2
        flags = [0, 0, 1, 0, 1, 1, 1, 0, 1]
        xe = [o.fill_uniform(ids, 7 + j) for j in range(T)]            # digit 0: coefficient form (transformed inside), digit 1: evaluation form
        xe[0][2, :] = o.moduli[ids[2]] - 1
        xe[1][4, :] = o.moduli[ids[4]] - 1
        for outs in (2, 1):
            y = [[o.fill_uniform(ids, 30 + 10 * k + j) for j in range(T)] for k in range(outs)]
            y[0][0][4, :] = o.moduli[ids[4]] - 1
            xb = ctx.from_host(np.concatenate(xe))
            yb = ctx.from_host(np.concatenate([y[k][j] for k in range(outs) for j in range(T)]))
            hand, out = ctx.alloc(n), ctx.alloc(n * outs)
            xl = [j * n + i for i in range(n) for j in range(T)]
            coeff = [1 if j == 0 else 0 for i in range(n) for j in range(T)]
            hl = [i for i in range(n) for j in range(T)]
            yl = [(k * T + j) * n + i for i in range(n) for k in range(outs) for j in range(T)]
            ol = [k * n + i for i in range(n) for k in range(outs)]
            scale = [o.moduli[m] - 3 - i for i, m in enumerate(ids)]
            X0 = o.ntt(ids, xe[0])
            for rep in range(2):
                ctx.set_option("nip_small_limbs", 64 if rep == 0 else 0)   # the small-launch geometry, then the wide one
                ctx.fill_uniform(out, ids * outs, 99)
                ctx.ntt_inner_product(xb, xl, coeff, hand, hl, yb, yl, out, ol, ids, T, outs, out_inverse=flags)
                fl = [k * n + i for k in range(outs) for i in range(n) if flags[i]]
                ctx.ntt_second_pass(out, [ids[l % n] for l in fl], inverse=True, limbs=fl, scale=[scale[l % n] for l in fl])
                got = out.download().reshape(outs, n, N)
                for k in range(outs):
                    acc = o.ewe(1, ids, X0, y[k][0], xe[1], y[k][1])
                    inv = o.ewe(5, ids, o.ntt(ids, acc, inverse=True), k=scale)
                    for i in range(n):
                        assert np.array_equal(got[k, i], inv[i] if flags[i] else acc[i]), (chain, outs, rep, k, i)
            for b_ in (xb, yb, hand, out):
                b_.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("form", ["generic", "mont32"])
def test_inner_product_with_narrow_moduli(form):
    """the key multiply-accumulate of hm_ntt_inner_product over a caller-chosen chain of mixed widths (the lazy product's operand shift
    and quotient constant depend on the modulus width): 59-, 45- and 31-bit primes = 1 mod 2N on the generic back-end (a chain an FHE
    library would hand over), 59-, 45- and 40-bit primes h 2^32 + 1 on the word-wise Montgomery one.  Evaluation-form operands only (no
    transform, so no oracle is needed), 4 terms, both keys, extreme operands; expected values from Python integers"""
    from sympy import isprime
    from homulator_amd import hip
    logN, N = 13, 1 << 13
    chain = []
    step = 1 << 32 if form == "mont32" else 2 * N
    for bits in ((59, 45, 40, 59, 45, 40) if form == "mont32" else (59, 45, 31, 59, 45, 31)):
        c = (1 << bits) + 1 - step
        while not isprime(c) or c in chain:
            c -= step
            assert c > 1 << (bits - 1)
        chain.append(c)
    ctx = hip.Context(logN, 4, 2, q=chain[:4], p=chain[4:])
    try:
        assert ctx.moduli == chain and ctx.counter("arith") == (1 if FORCED_GENERIC else 0 if form == "mont32" else 1)
        ids = [0, 1, 2, 3, 4, 5, 2, 2]
        n, terms, outs = len(ids), 4, 2
        rng = np.random.default_rng(11)
        def rnd(m):
            v = rng.integers(0, 1 << 62, N, dtype=np.uint64) % np.uint64(chain[m])
            v[:4] = [chain[m] - 1, 0, 1, chain[m] - 2]
            return v
        X = np.stack([np.stack([rnd(m) for m in ids]) for _ in range(terms)])                      # [term][limb]
        Y = np.stack([np.stack([np.stack([rnd(m) for m in ids]) for _ in range(terms)]) for _ in range(outs)])   # [key][term][limb]
        X[:, 0, :] = chain[ids[0]] - 1
        Y[:, :, 0, :64] = chain[ids[0]] - 1
        xb, yb = ctx.from_host(X.reshape(-1, N)), ctx.from_host(Y.reshape(-1, N))
        out = ctx.alloc(n * outs)
        xl = [j * n + i for i in range(n) for j in range(terms)]
        yl = [(k * terms + j) * n + i for i in range(n) for k in range(outs) for j in range(terms)]
        ol = [k * n + i for i in range(n) for k in range(outs)]
        ctx.ntt_inner_product(xb, xl, [0] * (n * terms), None, xl, yb, yl, out, ol, ids, terms, outs)
        got = out.download()
        for k in range(outs):
            for i, m in enumerate(ids):
                q = chain[m]
                acc = np.zeros(N, dtype=object)
                for j in range(terms):
                    acc = (acc + X[j, i].astype(object) * Y[k, j, i].astype(object)) % q
                assert np.array_equal(got[k * n + i].astype(object), acc), (k, i, m)
    finally:
        ctx.close()


@pytest.mark.parametrize("chain", ["mont32", "survey"])
def test_packed_conversion_inputs(chain):
    """round 5: hm_ntt_ex(out_packed) stores the inverse transform's output in the split-30 packed form (x mod 2^30) | ((x >> 30) << 32), and
    the three conversion entry points take it with hm_bconv_desc.in_packed: k_bconv (hm_bconv_batch), k_bconv_col inside the merged
    transform (hm_ntt_fused_desc.conv) and inside the transform x key call (hm_ntt_ip_desc.conv).  Same results as the plain form, bit for
    bit, with a launch that mixes packed and plain limbs; worst-case operands"""
    L, K, N = 7, 5, 1 << 16
    ctx, o, _ = make_env(16, L, K, chain)
    try:
        qs, ps = list(range(L)), list(range(L, L + K))
        y = o.fill_uniform(ps, 21)
        y[0, :] = o.moduli[ps[0]] - 1
        y[1, :4] = [0, 1, o.moduli[ps[1]] - 1, 12345]
        Y = o.ntt(ps, y)                                     # evaluation form: the inverse transform brings y back, packed
        src, plain, packed = ctx.from_host(Y), ctx.alloc(K), ctx.alloc(K)
        scale = [o.moduli[m] - 3 - i for i, m in enumerate(ps)]
        ctx.ntt(src, plain, ps, inverse=True, scale=scale)
        flags = [1, 1, 0, 1, 1]
        ctx.ntt(src, packed, ps, inverse=True, scale=scale, out_packed=flags)
        P, Q = plain.download(), packed.download()
        assert np.array_equal(P, o.ewe(5, ps, y, k=scale))
        pk = lambda a: (a & np.uint64(0x3FFFFFFF)) | ((a >> np.uint64(30)) << np.uint64(32))
        for i in range(K):
            assert np.array_equal(Q[i], pk(P[i]) if flags[i] else P[i]), i
        ctx.ntt(src, packed, ps, inverse=True, scale=scale, out_packed=[1] * K)
        # 1. k_bconv
        a, b = ctx.alloc(L), ctx.alloc(L)
        ctx.bconv_batch([(plain, None, ps, a, None, qs)])
        ctx.bconv_batch([(packed, None, ps, b, None, qs, 1)])
        exp = o.bconv_matmul(ps, qs, P)
        assert np.array_equal(a.download(), exp) and np.array_equal(b.download(), exp)
        # 2. inside a fused transform's first pass (ModDown finish without the rescale's mix operand: hrotate's form); with the mix prologue
        #    the fused conversion takes plain inputs only (the opt-in fused ModDown conversion of an hmult): refused, not silently wrong
        ids = qs
        mn, ad, mx = (o.fill_uniform(ids, s) for s in (23, 24, 25))
        k = [o.moduli[m] - 2 - r for r, m in enumerate(ids)]
        mk = [(kk * 3 + 1) % o.moduli[m] for kk, m in zip(k, ids)]
        ak = [(kk * 5 + 1) % o.moduli[m] for kk, m in zip(k, ids)]
        dmn, dad, dmx = ctx.from_host(mn), ctx.from_host(ad), ctx.from_host(mx)
        want = o.ewe(3, ids, o.ewe(6, ids, mn, None, o.ntt(ids, exp), k=k), None, o.ewe(5, ids, ad, k=ak))
        for srcbuf, flag in ((plain, 0), (packed, 1)):
            ctx.fill_uniform(a, ids, 99)
            ctx.ntt_mix_sub_scale(None, dmn, a, ids, k, addend=dad, addend_k=ak, conv=[(srcbuf, None, ps, list(range(L)), qs, flag)])
            assert np.array_equal(a.download(), want), flag
        x = o.ewe(3, ids, exp, None, o.ewe(5, ids, mx, k=mk))
        want_mix = o.ewe(3, ids, o.ewe(6, ids, mn, None, o.ntt(ids, x), k=k), None, o.ewe(5, ids, ad, k=ak))
        ctx.ntt_mix_sub_scale(None, dmn, a, ids, k, addend=dad, addend_k=ak, mix=dmx, mix_k=mk, conv=[(plain, None, ps, list(range(L)), qs, 0)])
        assert np.array_equal(a.download(), want_mix)
        with pytest.raises(_.HmError, match="packed"):
            ctx.ntt_mix_sub_scale(None, dmn, a, ids, k, addend=dad, addend_k=ak, mix=dmx, mix_k=mk, conv=[(packed, None, ps, list(range(L)), qs, 1)])
        # 3. inside the transform x key call: one digit (the special limbs) converted to the Q limbs, one key
        evk = o.fill_uniform(qs, 77)
        evkb, hand, out = ctx.from_host(evk), ctx.alloc(L), ctx.alloc(L)
        want3 = o.ewe(0, qs, o.ntt(qs, exp), evk)
        for srcbuf, flag in ((plain, 0), (packed, 1)):
            ctx.fill_uniform(out, qs, 98)
            ctx.ntt_inner_product(srcbuf, [0] * L, [1] * L, hand, list(range(L)), evkb, list(range(L)), out, list(range(L)), qs, 1, 1,
                                  conv=[(srcbuf, None, ps, list(range(L)), qs, flag)])
            assert np.array_equal(out.download(), want3), flag
    finally:
        ctx.close()


@pytest.mark.parametrize("logN,chain", [(16, "mont32"), (15, "mont32"), (16, "survey"), (15, 36)])
@pytest.mark.parametrize("n_in,outs_per_wg", [(16, 1), (16, 2), (17, 2), (20, 0), (24, 2), (28, 1), (28, 2), (31, 2), (32, 1), (32, 2)])
def test_wide_digit_conversion_inside_the_first_pass(logN, chain, n_in, outs_per_wg):
    """round 6 (parameter set A: N = 2^15, alpha = 28, beta = 1; `motivation`: N = 2^16, alpha = 28 — script/README.md:17-22,
    src/Operation.cpp:137-188, 314-352): digits of 16 .. 32 limbs convert inside the first pass of the transform that consumes them, in
    two input groups (k_bconv_col / k_bconv_col2 with N_IN > 15).  One digit of n_in limbs converted to 9 (an odd count: the last
    workgroup of the two-output form has one output) other limbs, transformed and multiplied with a key limb in one hm_ntt_inner_product
    call, from plain and from split-30 packed inputs, one and two outputs per workgroup; worst-case operands (every input at q_i - 1: the
    128-bit sums of both groups at their largest) on the first coefficients.  Against the oracle's conversion, transform and product."""
    n_out, N = 9, 1 << logN
    L, K = n_in, n_out
    ctx, o, _ = make_env(logN, L, K, chain)
    try:
        assert ctx.counter("cap_bconv_col_max_in") == 32
        ins, outs = list(range(n_in)), list(range(n_in, n_in + n_out))
        y = o.fill_uniform(ins, 31)
        for r, m in enumerate(ins):
            y[r, :6] = o.moduli[m] - 1
            y[r, 6:8] = [0, 1]
        evk = o.fill_uniform(outs, 77)
        want = o.ewe(0, outs, o.ntt(outs, o.bconv_matmul(ins, outs, y)), evk)
        pk = lambda a: (a & np.uint64(0x3FFFFFFF)) | ((a >> np.uint64(30)) << np.uint64(32))
        plain, packed = ctx.from_host(y), ctx.from_host(pk(y))
        evkb, hand, out = ctx.from_host(evk), ctx.alloc(n_out), ctx.alloc(n_out)
        ctx.set_option("bconv_col_outs", outs_per_wg)
        for srcbuf, flag in ((plain, 0), (packed, 1)):
            for small in (0, 4096):   # both geometries of the transform x key kernel
                ctx.set_option("nip_small_limbs", small)
                ctx.fill_uniform(out, outs, 98)
                ctx.fill_uniform(hand, outs, 97)
                ctx.ntt_inner_product(srcbuf, [0] * n_out, [1] * n_out, hand, list(range(n_out)), evkb, list(range(n_out)), out, list(range(n_out)), outs, 1, 1,
                                      conv=[(srcbuf, None, ins, list(range(n_out)), outs, flag)])
                assert np.array_equal(out.download(), want), (flag, small)
    finally:
        ctx.close()


def test_conversion_inputs_more_than_4_gib_apart():
    """round 6: the fused conversion + first pass addresses all inputs of a conversion from ONE buffer descriptor with 32-bit offsets (the
    scalar-register diet of k_bconv_col).  A caller whose input limb-polys lie more than 4 GiB apart (never the host layer's plans: a
    digit's limbs are neighbours in the pool) is served by conversion + first pass as two steps inside the same C-ABI call: same results,
    through hm_ntt_inner_product's conv list and through hm_ntt_mix_sub_scale's (with and without the mix prologue)."""
    logN, N = 16, 1 << 16
    L, K = 4, 3
    ctx, o, _ = make_env(logN, L, K, "mont32")
    try:
        far = (1 << 32) // (8 * N) + 5                       # 8 197 limb-polys = 4 GiB + 2.5 MiB
        big = ctx.alloc(far + 1)                              # 4.3 GB of HBM
        ps, qs = [L, L + 1, L + 2], list(range(L))
        y = o.fill_uniform(ps, 41)
        y[:, :4] = [[o.moduli[m] - 1] * 4 for m in ps]
        spread = [0, far, 7]                                  # input limbs of the one conversion: 4 GiB apart
        for r, limb in enumerate(spread):
            big.upload(y[r:r + 1], limb0=limb)
        conv_exp = o.bconv_matmul(ps, qs, y)
        # 1. inside the transform x key call
        evk = o.fill_uniform(qs, 77)
        evkb, hand, out = ctx.from_host(evk), ctx.alloc(L), ctx.alloc(L)
        ctx.ntt_inner_product(big, [0] * L, [1] * L, hand, list(range(L)), evkb, list(range(L)), out, list(range(L)), qs, 1, 1,
                              conv=[(big, spread, ps, list(range(L)), qs, 0)])
        assert np.array_equal(out.download(), o.ewe(0, qs, o.ntt(qs, conv_exp), evk))
        # 2. inside the merged transform: without and with the mix prologue
        mn, ad, mx = (o.fill_uniform(qs, s) for s in (23, 24, 25))
        k = [o.moduli[m] - 2 - r for r, m in enumerate(qs)]
        mk = [(kk * 3 + 1) % o.moduli[m] for kk, m in zip(k, qs)]
        ak = [(kk * 5 + 1) % o.moduli[m] for kk, m in zip(k, qs)]
        dmn, dad, dmx, a = ctx.from_host(mn), ctx.from_host(ad), ctx.from_host(mx), ctx.alloc(L)
        want = o.ewe(3, qs, o.ewe(6, qs, mn, None, o.ntt(qs, conv_exp), k=k), None, o.ewe(5, qs, ad, k=ak))
        ctx.ntt_mix_sub_scale(None, dmn, a, qs, k, addend=dad, addend_k=ak, conv=[(big, spread, ps, list(range(L)), qs, 0)])
        assert np.array_equal(a.download(), want)
        x = o.ewe(3, qs, conv_exp, None, o.ewe(5, qs, mx, k=mk))
        want_mix = o.ewe(3, qs, o.ewe(6, qs, mn, None, o.ntt(qs, x), k=k), None, o.ewe(5, qs, ad, k=ak))
        ctx.fill_uniform(a, qs, 9)
        ctx.ntt_mix_sub_scale(None, dmn, a, qs, k, addend=dad, addend_k=ak, mix=dmx, mix_k=mk, conv=[(big, spread, ps, list(range(L)), qs, 0)])
        assert np.array_equal(a.download(), want_mix)
        big.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("logN,chain", [(16, "mont32"), (15, "survey")])
@pytest.mark.parametrize("widths", [(15, 5), (9, 3), (4, 4, 2), (28, 17), (20, 6)])
def test_digits_of_a_small_call_merged_into_one_launch(logN, chain, widths):
    """round 6: a SMALL fused-conversion call whose digits differ in width runs ONE launch of the widest digit's kernel (the narrower digits
    with zero table columns for the inputs they lack; two outputs per workgroup) instead of one launch per width (option bconv_col_merge).
    Both forms, packed and plain inputs mixed, against each other and against the oracle: hm_ntt_inner_product with one conversion per
    digit, every digit converted to the limbs outside itself.  (28, 17): the two-group family merges among itself; (20, 6): a wide and a
    narrow digit stay two launches (one family per kernel)."""
    ell = sum(widths)
    L, K, N = ell, 3, 1 << logN
    ctx, o, _ = make_env(logN, L, K, chain)
    try:
        ext = o.ext_ids(ell); E = len(ext); T = len(widths)
        digits, lo = [], 0
        for w in widths:
            digits.append(list(range(lo, lo + w))); lo += w
        own = o.fill_uniform(list(range(ell)), 5)
        scaled = o.fill_uniform(list(range(ell)), 6)
        scaled[0, :4] = o.moduli[0] - 1
        pk = (scaled & np.uint64(0x3FFFFFFF)) | ((scaled >> np.uint64(30)) << np.uint64(32))
        evk = np.stack([o.fill_uniform(ext, 100 + j) for j in range(T)])
        plain, packed, ownb, evkb = ctx.from_host(scaled), ctx.from_host(pk), ctx.from_host(own), ctx.from_host(evk.reshape(-1, N))
        hand, out = ctx.alloc(T * E), ctx.alloc(E)
        xl, flags, hl, yl, mods = [], [], [], [], []
        for t in range(E):
            for j, dj in enumerate(digits):
                isown = t in dj
                xl.append(t if isown else 0); hl.append(j * E + t); flags.append(0 if isown else 1)
            yl += [j * E + t for j in range(T)]
            mods.append(ext[t])
        X = []
        for j, dj in enumerate(digits):
            outs = [t for t in range(E) if t not in dj]
            full = np.zeros((E, N), dtype=np.uint64)
            full[outs] = o.ntt([ext[t] for t in outs], o.bconv_matmul(dj, [ext[t] for t in outs], scaled[dj]))
            full[dj] = own[dj]
            X.append(full)
        want = np.zeros((E, N), dtype=np.uint64)
        for j in range(T):
            want = o.ewe(3, ext, want, None, o.ewe(0, ext, X[j], evk[j]))
        got = {}
        for merge in (0, 1):
            for use_packed in (0, 1):
                src = packed if use_packed else plain
                conv = [(src, dj, dj, [j * E + t for t in range(E) if t not in dj], [ext[t] for t in range(E) if t not in dj], use_packed) for j, dj in enumerate(digits)]
                ctx.set_option("bconv_col_merge", merge)
                ctx.fill_uniform(out, ext, 3)
                ctx.fill_uniform(hand, ext * T, 4)
                ctx.ntt_inner_product(ownb, xl, flags, hand, hl, evkb, yl, out, list(range(E)), mods, T, 1, conv=conv)
                got[merge, use_packed] = out.download()
                assert np.array_equal(got[merge, use_packed], want), (merge, use_packed)
    finally:
        ctx.close()
