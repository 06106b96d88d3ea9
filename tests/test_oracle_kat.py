"""Known-answer tests that pin the CPU oracle independently of its own code:
Python big integers, sympy primality, O(N^2) schoolbook products, exact CRT lifts, decryption with a toy keygen.
The reference defines no arithmetic, so these — not reference vectors — are what pins the oracle's maths."""
import numpy as np
import pytest
import sympy

from oracle.homoracle import Oracle
from toy_ckks import Toy


def brev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2)


def check_minimal_primitive_root(psi, q, N):
    assert pow(psi, N, q) == q - 1  # primitive 2N-th root
    r, best = psi, psi              # minimal among all primitive 2N-th roots (the odd powers of any one of them)
    r2 = psi * psi % q
    for _ in range(N - 1):
        r = r * r2 % q
        best = min(best, r)
    assert best == psi


def test_prime_chain_and_roots():
    o = Oracle(10, 6, 2)
    N = o.N
    mods = o.moduli
    assert len(set(mods)) == 8 and mods == sorted(mods, reverse=True)
    # exactly the largest primes = 1 mod 2^32 below 2^60 (q = h 2^32 + 1 makes a Montgomery step one multiply)
    cand, found = (1 << 60) + 1, []
    while len(found) < 8:
        cand -= 1 << 32
        if sympy.isprime(cand):
            found.append(cand)
    assert found == mods
    for q, psi in zip(mods, o.psis):
        check_minimal_primitive_root(psi, q, N)


@pytest.mark.parametrize("logN,bits,count", [(10, 60, 8), (15, 60, 14), (16, 60, 60), (10, 36, 8), (15, 36, 56), (16, 36, 60), (13, 45, 20), (12, 31, 9)])
def test_generated_chains_are_exactly_the_largest_primes_below_the_bound(logN, bits, count):
    """the chains the GPU suite and bench.py's generic_chain leg stand on ("survey" = SURVEY.md 8(d) as written: bits = 60; 36-bit words;
    the 45- and 31-bit chains of the kernel tests): the oracle's generator (ho_chain_below) against an independent sympy enumeration of the
    candidates c = 1 mod 2N below 2^bits, descending — every prime, no composite, nothing skipped, in order"""
    from oracle.homoracle import chain_below
    got = chain_below(logN, bits, count)
    step = 2 << logN
    c, want = ((1 << bits) - 2) // step * step + 1, []
    while len(want) < count:
        if sympy.isprime(c):
            want.append(c)
        c -= step
    assert got == want
    assert all(q % step == 1 and q < (1 << bits) for q in got)


@pytest.mark.parametrize("chain", ["survey", 36, "caller"])
@pytest.mark.parametrize("logN", [10, 13])
def test_roots_of_the_other_chains_are_minimal(chain, logN):
    from conftest import make_oracle
    o = make_oracle(logN, 4, 2, chain)
    assert len(set(o.moduli)) == 6
    if chain != "caller":
        assert o.moduli == sorted(o.moduli, reverse=True)
    for q, psi in zip(o.moduli, o.psis):
        assert sympy.isprime(q) and q % (2 * o.N) == 1
        check_minimal_primitive_root(psi, q, o.N)


def test_scalar_mulmod_powmod():
    from oracle.homoracle import lib
    L = lib()
    rng = np.random.default_rng(0)
    q = Oracle(10, 1, 0).moduli[0]
    for _ in range(200):
        a, b = int(rng.integers(0, q, dtype=np.uint64)), int(rng.integers(0, q, dtype=np.uint64))
        assert L.ho_mulmod(a, b, q) == a * b % q
        assert L.ho_powmod(a, b, q) == pow(a, b, q)
    assert L.ho_invmod(12345, q) == pow(12345, -1, q)


def test_barrett_ewe_edges(oracle_small):
    o = oracle_small
    N = o.N
    for m in (0, 5, 6, 7):
        q = o.moduli[m]
        top = 1 << (q.bit_length() - 1)   # the highest power of two below q (2^59 on the 60-bit chains)
        edge = [0, 1, 2, q - 1, q - 2, q // 2, q // 2 + 1, top, top + 1]
        rng = np.random.default_rng(m)
        a = np.array((edge * (N // len(edge) + 1))[:N], dtype=np.uint64)
        b = np.array([int(x) for x in rng.integers(0, q, N, dtype=np.uint64)], dtype=np.uint64)
        b[:9] = np.array(edge[::-1], dtype=np.uint64)
        c = np.roll(a, 3)
        d = np.roll(b, 5)
        ai, bi, ci, di = ([int(x) for x in v] for v in (a, b, c, d))
        k = q - 3
        exp = {
            0: [x * y % q for x, y in zip(ai, bi)],
            1: [(x * y + z * w) % q for x, y, z, w in zip(ai, bi, ci, di)],
            2: [(x * y + z) % q for x, y, z in zip(ai, bi, ci)],
            3: [(x + z) % q for x, z in zip(ai, ci)],
            4: [(x - z) % q for x, z in zip(ai, ci)],
            5: [x * k % q for x in ai],
            6: [(x - z) * k % q for x, z in zip(ai, ci)],
            7: ai,
        }
        for op, e in exp.items():
            got = o.ewe(op, [m], a[None], b[None], c[None], d[None], k=[k])[0]
            assert [int(x) for x in got] == e, f"ewe op {op} mod {m}"


@pytest.mark.parametrize("chain", ["mont32", "survey", 36, "caller"])
@pytest.mark.parametrize("logN", [3, 4, 6])
def test_ntt_is_evaluation_at_odd_powers(logN, chain):
    """forward NTT out[i] = a(psi^(2*brev(i)+1)) — fixes the ordering convention (Appendix A (1))."""
    from conftest import make_oracle
    o = make_oracle(logN, 2, 1, chain)
    N = o.N
    rng = np.random.default_rng(logN)
    for m in range(3):
        q, psi = o.moduli[m], o.psis[m]
        a = [int(x) for x in rng.integers(0, q, N, dtype=np.uint64)]
        got = o.ntt([m], np.array(a, dtype=np.uint64)[None])[0]
        for i in range(N):
            x = pow(psi, 2 * brev(i, logN) + 1, q)
            assert int(got[i]) == sum(c * pow(x, k, q) for k, c in enumerate(a)) % q


@pytest.mark.parametrize("chain", ["mont32", "survey", 36, "caller"])
@pytest.mark.parametrize("logN", [3, 5, 8, 10])
def test_ntt_roundtrip_and_convolution(logN, chain):
    from conftest import make_oracle
    o = make_oracle(logN, 3, 2, chain)
    N = o.N
    ids = list(range(5))
    rng = np.random.default_rng(logN)
    a = np.stack([np.array([int(x) for x in rng.integers(0, o.moduli[m], N, dtype=np.uint64)], dtype=np.uint64) for m in ids])
    b = np.stack([np.array([int(x) for x in rng.integers(0, o.moduli[m], N, dtype=np.uint64)], dtype=np.uint64) for m in ids])
    A, B = o.ntt(ids, a), o.ntt(ids, b)
    assert np.array_equal(o.ntt(ids, A, inverse=True), a)  # (i)
    prod = o.ntt(ids, o.ewe(0, ids, A, B), inverse=True)   # (ii)
    for r, m in enumerate(ids):
        q = o.moduli[m]
        ai, bi = [int(x) for x in a[r]], [int(x) for x in b[r]]
        res = [0] * (2 * N)
        for i in range(N):
            for j in range(N):
                res[i + j] += ai[i] * bi[j]
        exp = [(res[i] - res[i + N]) % q for i in range(N)]
        assert [int(x) for x in prod[r]] == exp
        if logN >= 8:
            break  # one limb is enough at O(N^2)


def test_ntt_roundtrip_n16_single_limb():
    """BASELINE config #2 shape on the oracle."""
    o = Oracle(16, 2, 1)
    x = o.fill_uniform([0, 1, 2], 7)
    X = o.ntt([0, 1, 2], x)
    assert not np.array_equal(X, x)
    assert np.array_equal(o.ntt([0, 1, 2], X, inverse=True), x)


def test_bconv_vs_exact_crt(oracle_small):
    """(iii) fast base conversion = x + u*Q_D with 0 <= u < d."""
    o = oracle_small
    N = o.N
    rng = np.random.default_rng(3)
    for in_ids, out_ids in (([0, 1], [2, 3, 4, 5, 6, 7]), ([2, 3], [0, 1, 4, 5, 6, 7]), ([6, 7], [0, 1, 2, 3, 4, 5]), ([4], [0, 7])):
        QD = 1
        for m in in_ids:
            QD *= o.moduli[m]
        xs = [int(rng.integers(0, 1 << 62)) * int(rng.integers(0, 1 << 62)) % QD for _ in range(N)]
        xs[0], xs[1], xs[2] = 0, 1, QD - 1
        limbs = np.array([[x % o.moduli[m] for x in xs] for m in in_ids], dtype=np.uint64)
        out = o.bconv_matmul(in_ids, out_ids, o.bconv_scale(in_ids, limbs))
        for r, t in enumerate(out_ids):
            qt = o.moduli[t]
            for c in range(N):
                ok = any((xs[c] + u * QD) % qt == int(out[r, c]) for u in range(len(in_ids)))
                assert ok, (in_ids, t, c)
        # the constants themselves
        qhi, tab = o.bconv_consts(in_ids, out_ids)
        for i, m in enumerate(in_ids):
            qh = QD // o.moduli[m]
            assert int(qhi[i]) == pow(qh % o.moduli[m], -1, o.moduli[m])
            for r, t in enumerate(out_ids):
                assert int(tab[i, r]) == qh % o.moduli[t]


@pytest.mark.parametrize("g", [5, 25, 3, 2047, 1025])
def test_automorphism_eval_equals_coef(oracle_small, g):
    """(iv) eval-form permutation = NTT(sigma_g(INTT(x)))."""
    o = oracle_small
    ids = [0, 3, 7]
    x = o.fill_uniform(ids, 11)
    via_coef = o.ntt(ids, o.automorph_coef(ids, o.ntt(ids, x, inverse=True), g))
    assert np.array_equal(o.automorph_eval(x, g), via_coef)
    # coefficient-form map against a direct big-int definition
    toy = Toy(o)
    a = np.array([int(v) for v in x[0]], dtype=object)
    exp = np.array([int(v) % o.moduli[0] for v in toy.automorph(a, g)], dtype=np.uint64)
    assert np.array_equal(o.automorph_coef([0], x[:1], g)[0], exp)


@pytest.mark.parametrize("ell", [6, 5, 4, 2, 1])
def test_keyswitch_decrypts(oracle_small, ell):
    """(v) Dec_s(KS_{s'->s}(d)) - d*s' is small, for full and partial last digits (beta = 3,3,2,1,1)."""
    o = oracle_small
    toy = Toy(o, seed=ell)
    s_from = toy.rng.integers(-1, 2, o.N).astype(object)
    evk = toy.evk_at_level(toy.gen_evk(s_from), ell)
    ids = list(range(ell))
    d = o.fill_uniform(ids, 99)
    k0, k1 = o.keyswitch(ell, d, evk)
    got, Q = toy.decrypt(np.stack([k0, k1]), ell)
    d_int, _ = toy.crt_center(o.ntt(ids, d, inverse=True), ids)
    exp = toy.negacyclic_mul(d_int, s_from)
    diff = np.array([int(x) % Q for x in (got - exp)], dtype=object)
    diff = np.array([x - Q if x > Q // 2 else x for x in diff], dtype=object)
    assert max(abs(int(x)) for x in diff) < 1 << 32


def test_keyswitch_dump_is_consistent(oracle_small):
    o = oracle_small
    ell = 5
    d = o.fill_uniform(list(range(ell)), 5)
    evk = o.synth_evk(ell, 77)
    k0, k1, dd = o.keyswitch(ell, d, evk, dump=True)
    assert np.array_equal(dd["modup_intt"], o.ntt(list(range(ell)), d, inverse=True))
    # digit limbs of ext are the original eval-form limbs
    for j in range(o.beta(ell)):
        for t in range(j * o.K, min(ell, (j + 1) * o.K)):
            assert np.array_equal(dd["ext"][j, t], d[t])
    # final output from the dumped pieces
    for k, out in ((0, k0), (1, k1)):
        for i in range(ell):
            q = o.moduli[i]
            P = 1
            for p in range(o.K):
                P = P * o.moduli[o.L + p] % q
            pinv = pow(P, -1, q)
            exp = [(int(a) - int(b)) * pinv % q for a, b in zip(dd["ip"][k, i], dd["moddown_ntt"][k, i])]
            assert [int(x) for x in out[i]] == exp


@pytest.mark.parametrize("ell", [6, 4])
def test_hmult_decrypts_to_product(oracle_small, ell):
    """(vi) Dec(hmult(ct1, ct2)) ~ m1*m2 / q_last."""
    o = oracle_small
    toy = Toy(o, seed=10 + ell)
    s2 = toy.negacyclic_mul(toy.s, toy.s)
    evk = toy.evk_at_level(toy.gen_evk(s2), ell)
    m1 = (toy.rng.integers(-50, 50, o.N) * (1 << 40)).astype(object)
    m2 = (toy.rng.integers(-50, 50, o.N) * (1 << 40)).astype(object)
    ct1, ct2 = toy.encrypt(m1, ell), toy.encrypt(m2, ell)
    out = o.hmult(ell, ct1, ct2, evk, rescale=True)
    got, Q = toy.decrypt(out, ell - 1)
    ql = o.moduli[ell - 1]
    exp = toy.negacyclic_mul(m1, m2)
    err = max(abs(int(g) * ql - int(e)) for g, e in zip(got, exp))
    assert err < ql << 26  # |got - m1*m2/q_last| < 2^26 against a signal of ~2^37
    # and without rescale the product is exact up to key-switch noise
    out_nr = o.hmult(ell, ct1, ct2, evk, rescale=False)
    got_nr, _ = toy.decrypt(out_nr, ell)
    assert max(abs(int(g) - int(e)) for g, e in zip(got_nr, exp)) < 1 << 70


@pytest.mark.parametrize("g", [5, 25])
def test_hrotate_decrypts_to_rotation(oracle_small, g):
    o = oracle_small
    ell = 5
    toy = Toy(o, seed=20 + g)
    evk = toy.evk_at_level(toy.gen_evk(toy.automorph(toy.s, g)), ell)
    m = (toy.rng.integers(-1000, 1000, o.N) * (1 << 30)).astype(object)
    out = o.hrotate(ell, toy.encrypt(m, ell), g, evk)
    got, _ = toy.decrypt(out, ell)
    exp = toy.automorph(m, g)
    assert max(abs(int(a) - int(b)) for a, b in zip(got, exp)) < 1 << 32


def test_hadd_pmult_padd(oracle_small):
    o = oracle_small
    ell = 3
    ids = list(range(ell))
    a, b = o.synth_ct(ell, 1), o.synth_ct(ell, 2)
    pt = o.fill_uniform(ids, 3)
    s = o.hadd(ell, a, b)
    pm = o.pmult(ell, a, pt)
    pa = o.padd(ell, a, pt)
    for i in ids:
        q = o.moduli[i]
        for k in range(2):
            assert [int(x) for x in s[k, i][:64]] == [(int(x) + int(y)) % q for x, y in zip(a[k, i][:64], b[k, i][:64])]
            assert [int(x) for x in pm[k, i][:64]] == [int(x) * int(y) % q for x, y in zip(a[k, i][:64], pt[i][:64])]
        assert [int(x) for x in pa[0, i][:64]] == [(int(x) + int(y)) % q for x, y in zip(a[0, i][:64], pt[i][:64])]
    assert np.array_equal(pa[1], a[1])


def test_fill_uniform_definition(oracle_small):
    o = oracle_small
    M = (1 << 64) - 1

    def mix(z):
        z ^= z >> 30; z = z * 0xBF58476D1CE4E5B9 & M
        z ^= z >> 27; z = z * 0x94D049BB133111EB & M
        return z ^ (z >> 31)
    out = o.fill_uniform([0, 7], 0x484F4D55)
    for i, m in enumerate([0, 7]):
        q = o.moduli[m]
        for x in (0, 1, 17, o.N - 1):
            z = mix(((0x484F4D55 + i) * 0xD1342543DE82EF95 + x * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & M)
            assert int(out[i, x]) == (z * q) >> 64
    assert int(out.max()) < max(o.moduli)


@pytest.mark.parametrize("chain", ["mont32", "survey", 36])
@pytest.mark.parametrize("logN,L,ell,alpha", [(16, 45, 35, 15), (15, 16, 10, 4)])
def test_full_size_known_answers(logN, L, ell, alpha, chain):
    """one full-size known answer per chain (tests/golden/oracle_hashes.json, written by tests/golden/make_oracle_hashes.py): hmult and
    hrotate at BASELINE configs[2] / [3] and configs[0] — an edit of the oracle cannot move a full-size result silently"""
    import json
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_oracle_hashes as g
    want = json.load(open(os.path.join(here, "golden", "oracle_hashes.json")))[f"N{logN}_L{L}_l{ell}_a{alpha}_{chain}"]
    assert g.compute(logN, L, ell, alpha, chain) == want
