"""CPU emulation of the HIP kernels' per-thread phase functions vs the oracle (no GPU needed).
The emulator (tests/emu/hm_emu.cpp) compiles the SAME device headers with g++ and runs every workgroup
phase by phase, so tile/LDS indexing, twiddle indexing, lazy-reduction ranges, Barrett/Shoup forms and the
host-side tables of hm_params.cpp are all bit-checked here before the GPU run."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle.homoracle import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    subprocess.check_call(["make", "-C", os.path.join(HERE, "emu")], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(HERE, "emu", name))
    L.emu_create.restype = C.c_void_p
    L.emu_create.argtypes = [C.c_uint32] * 3
    L.emu_create_chain.restype = C.c_void_p
    L.emu_create_chain.argtypes = [C.c_uint32] * 3 + [C.c_void_p] * 2
    L.emu_create_mods.restype = C.c_void_p
    L.emu_create_mods.argtypes = [C.c_uint32] * 3 + [C.c_void_p] * 2
    L.emu_destroy.argtypes = [C.c_void_p]
    L.emu_modulus.restype = C.c_uint64
    L.emu_modulus.argtypes = [C.c_void_p, C.c_uint32]
    L.emu_psi.restype = C.c_uint64
    L.emu_psi.argtypes = [C.c_void_p, C.c_uint32]
    L.emu_ntt.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_int]
    L.emu_ntt_sub_scale.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64]
    L.emu_tensor.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 7
    L.emu_ewe.argtypes = [C.c_void_p, C.c_int, C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p]
    L.emu_bconv.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.emu_bconv_form.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
    L.emu_bconv_consts.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.emu_automorph.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.emu_fill.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
    L.emu_mac.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    L.emu_mac_final.restype = C.c_uint64
    L.emu_mac_final.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
    L.emu_redc_wide.restype = C.c_uint64
    L.emu_redc_wide.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int]
    L.emu_bfly.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
    return L


class Emu:
    """one build of the emulator (the kernels' headers compiled for one arithmetic back-end) + the chain the tests give it:
    "mont32" = libhm_emu.so on the default chain; "survey" / an int b = libhm_emu_gen.so (HM_GENERIC) on the largest primes = 1 mod 2N
    below 2^60 / 2^b; "generic-on-mont32" = the generic arithmetic on the default chain"""

    def __init__(self, chain):
        self.chain = chain
        self.lib = _load("libhm_emu.so" if chain == "mont32" else "libhm_emu_gen.so")
        assert self.lib.emu_generic() == (0 if chain == "mont32" else 1)

    def oracle(self, logN, L, K):
        return Oracle(logN, L, K, chain="mont32" if self.chain in ("mont32", "generic-on-mont32") else self.chain)

    def create(self, o):
        """an emulator context on the oracle's chain"""
        mods = np.array(o.moduli, dtype=np.uint64)
        h = self.lib.emu_create_chain(o.logN, o.L, o.K, p(mods[:o.L]), p(mods[o.L:]))
        assert h, "the emulator refused the chain"
        return h

    def __getattr__(self, name):
        return getattr(self.lib, name)


@pytest.fixture(scope="module", params=["mont32", "survey", 36, "generic-on-mont32"], ids=str)
def emu(request):
    return Emu(request.param)


@pytest.fixture(scope="module")
def emu_m32():
    return Emu("mont32")


@pytest.fixture(scope="module")
def emu_gen():
    return Emu("survey")


def p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("logN", [13, 14, 15, 16, 17])
def test_emu_params_and_ntt(emu, logN):
    L, K = 3, 2
    o = emu.oracle(logN, L, K)
    h = emu.create(o)
    try:
        assert [emu.emu_modulus(h, m) for m in range(L + K)] == o.moduli
        assert [emu.emu_psi(h, m) for m in range(L + K)] == o.psis
        for m in (0, L + K - 1):
            x = o.fill_uniform([m], 42 + m)[0]
            # edge values: 0, q-1
            x[0], x[1], x[-1] = 0, o.moduli[m] - 1, o.moduli[m] - 1
            out = np.empty_like(x)
            assert emu.emu_ntt(h, m, p(x), p(out), 0, 0, 0) == 0
            exp = o.ntt([m], x[None])[0]
            assert np.array_equal(out, exp), f"forward logN={logN} mod={m}"
            back = np.empty_like(x)
            assert emu.emu_ntt(h, m, p(out), p(back), 1, 0, 0) == 0
            assert np.array_equal(back, x), f"inverse logN={logN} mod={m}"
            # in-place + fused scale
            k = o.moduli[m] - 12345
            buf = out.copy()
            emu.emu_ntt(h, m, p(buf), p(buf), 1, k, 1)
            assert np.array_equal(buf, o.ewe(5, [m], x[None], k=[k])[0])
    finally:
        emu.emu_destroy(h)


@pytest.mark.parametrize("logN", [13, 15, 16])
def test_emu_fused_ntt_sub_scale_and_tensor(emu, logN):
    L, K = 3, 1
    o = emu.oracle(logN, L, K)
    h = emu.create(o)
    try:
        for m in (0, L):
            q = o.moduli[m]
            x, mn, ad, d = (o.fill_uniform([m], s)[0] for s in (1, 2, 3, 4))
            x[:3] = [q - 1, 0, q - 1]
            mn[:3] = [0, q - 1, q - 1]
            ad[:3] = [q - 1, q - 1, 0]
            k = q - 5
            ntt = o.ntt([m], x[None])
            exp = o.ewe(6, [m], mn[None], None, ntt, k=[k])
            out = np.empty_like(x)
            assert emu.emu_ntt_sub_scale(h, m, p(x), p(mn), None, p(out), k, None, 0, 0) == 0
            assert np.array_equal(out, exp[0])
            assert emu.emu_ntt_sub_scale(h, m, p(x), p(mn), p(ad), p(out), k, None, 0, 0) == 0
            assert np.array_equal(out, o.ewe(3, [m], exp, None, ad[None])[0])
            # merged ModDown + rescale form: prologue x + mk * d, constant on the addend
            mk, ak = (k * 7 + 3) % q, (k * 11 + 5) % q
            xin = o.ewe(3, [m], x[None], None, o.ewe(5, [m], d[None], k=[mk]))            # x + mk * d
            body = o.ewe(6, [m], mn[None], None, o.ntt([m], xin), k=[k])                   # (mn - NTT(.)) * k
            exp2 = o.ewe(3, [m], body, None, o.ewe(5, [m], ad[None], k=[ak]))              # + ak * ad
            assert emu.emu_ntt_sub_scale(h, m, p(x), p(mn), p(ad), p(out), k, p(d), mk, ak) == 0
            assert np.array_equal(out, exp2[0])
            o0, o1, o2 = (np.empty_like(x) for _ in range(3))
            emu.emu_tensor(h, m, p(x), p(mn), p(ad), p(d), p(o0), p(o1), p(o2))
            assert np.array_equal(o0, o.ewe(0, [m], x[None], mn[None])[0])
            assert np.array_equal(o1, o.ewe(1, [m], x[None], d[None], ad[None], mn[None])[0])
            assert np.array_equal(o2, o.ewe(0, [m], ad[None], d[None])[0])
    finally:
        emu.emu_destroy(h)


def test_emu_ewe_bconv_auto_fill(emu):
    logN, L, K = 13, 5, 2
    o = emu.oracle(logN, L, K)
    h = emu.create(o)
    try:
        N = o.N
        for m in (0, 6):
            q = o.moduli[m]
            a, b, c, d = (o.fill_uniform([m], s)[0] for s in (1, 2, 3, 4))
            a[:4] = [0, q - 1, q - 1, 1]
            b[:4] = [q - 1, q - 1, 0, q - 1]
            c[:4] = [q - 1, 0, q - 1, q - 1]
            d[:4] = [q - 1, q - 1, q - 1, 0]
            k = q - 2
            for op in range(8):
                out = np.empty(N, dtype=np.uint64)
                emu.emu_ewe(h, op, m, p(a), p(b), p(c), p(d), k, p(out))
                exp = o.ewe(op, [m], a[None], b[None], c[None], d[None], k=[k])[0]
                assert np.array_equal(out, exp), f"op {op}"
            out = np.empty(N, dtype=np.uint64)
            emu.emu_ewe(h, 8, m, p(a), p(b), p(c), p(d), k, p(out))
            exp = o.ewe(3, [m], o.ewe(6, [m], a[None], None, c[None], k=[k]), None, d[None])[0]
            assert np.array_equal(out, exp)
        # base conversion: ModUp-like (2 -> 5) and ModDown-like (2 -> 5), edge: all inputs q-1
        for in_ids, out_ids in (([0, 1], [2, 3, 4, 5, 6]), ([5, 6], [0, 1, 2, 3, 4]), ([2], [0, 6])):
            ii, oi = np.array(in_ids, dtype=np.uint32), np.array(out_ids, dtype=np.uint32)
            x = o.fill_uniform(in_ids, 9)
            for r, m in enumerate(in_ids):
                x[r, :2] = [o.moduli[m] - 1, 0]
            out = np.empty((len(out_ids), N), dtype=np.uint64)
            emu.emu_bconv(h, p(ii), len(ii), p(oi), len(oi), p(x), p(out))
            assert np.array_equal(out, o.bconv_matmul(in_ids, out_ids, x))
            out[:] = 0   # the same from inputs in the split-30 packed form (round 5)
            emu.emu_bconv_form(h, p(ii), len(ii), p(oi), len(oi), p(x), p(out), 1)
            assert np.array_equal(out, o.bconv_matmul(in_ids, out_ids, x))
            qh = np.empty(len(ii), dtype=np.uint64)
            tb = np.empty((len(ii), len(oi)), dtype=np.uint64)
            emu.emu_bconv_consts(h, p(ii), len(ii), p(oi), len(oi), p(qh), p(tb))
            eq, et = o.bconv_consts(in_ids, out_ids)
            assert np.array_equal(qh, eq) and np.array_equal(tb, et)
        # 16 inputs of q-1 against table entries: the widest accumulator (n_in = 16 needs K+L >= 17)
        for g in (5, 25, 2 * N - 1, 3):
            x = o.fill_uniform([0], g)[0]
            out = np.empty_like(x)
            emu.emu_automorph(h, p(x), p(out), g)
            assert np.array_equal(out, o.automorph_eval(x[None], g)[0])
        out = np.empty(N, dtype=np.uint64)
        emu.emu_fill(h, 3, 1234 + 2, p(out))
        assert np.array_equal(out, o.fill_uniform([1, 2, 3], 1234)[2])
    finally:
        emu.emu_destroy(h)


def test_emu_bconv_28_inputs_parameter_set_A(emu):
    """alpha = 28 (parameter set A): two carry-free column groups per output, all inputs at q_i - 1"""
    logN, L, K = 13, 30, 28
    o = emu.oracle(logN, L, K)
    h = emu.create(o)
    try:
        in_ids, out_ids = [L + i for i in range(28)], [0, 1, 2, 29]
        x = o.fill_uniform(in_ids, 5)
        for r, m in enumerate(in_ids):
            x[r, :8] = o.moduli[m] - 1
        ii, oi = np.array(in_ids, dtype=np.uint32), np.array(out_ids, dtype=np.uint32)
        out = np.empty((len(out_ids), o.N), dtype=np.uint64)
        emu.emu_bconv(h, p(ii), 28, p(oi), len(oi), p(x), p(out))
        assert np.array_equal(out, o.bconv_matmul(in_ids, out_ids, x))
    finally:
        emu.emu_destroy(h)


def test_emu_bconv_widest_accumulator(emu):
    """16 input limbs all at q_i - 1: the 128-bit accumulator reaches ~2^124 (fold path of hm_barrett_wide)."""
    logN, L, K = 13, 20, 2
    o = emu.oracle(logN, L, K)
    h = emu.create(o)
    try:
        in_ids, out_ids = list(range(16)), [16, 17, 18, 19, 20, 21]
        x = o.fill_uniform(in_ids, 5)
        for r, m in enumerate(in_ids):
            x[r, :8] = o.moduli[m] - 1
        ii, oi = np.array(in_ids, dtype=np.uint32), np.array(out_ids, dtype=np.uint32)
        out = np.empty((len(out_ids), o.N), dtype=np.uint64)
        emu.emu_bconv(h, p(ii), 16, p(oi), len(oi), p(x), p(out))
        assert np.array_equal(out, o.bconv_matmul(in_ids, out_ids, x))
    finally:
        emu.emu_destroy(h)


@pytest.mark.parametrize("logN", [16, 15])
def test_emu_ntt_eight_coefficients_per_thread(emu, logN):
    """the small-launch geometry (hm8: 512-thread workgroups, 8 coefficients per thread, four radix-4 rounds per pass) on the
    CPU emulator at N = 2^16 and N = 2^15 (round 6: the 128-point COL pass as three radix-4 rounds and a radix-2 round): forward,
    inverse (in place, fused scale), the all-(q-1) worst case of the lazy ranges, and the merged ModDown + rescale form (mix
    prologue + sub-scale-add epilogue)"""
    emu.emu_ntt8.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_int]
    emu.emu_ntt_sub_scale8.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64]
    L, K = 3, 2
    o = emu.oracle(logN, L, K)
    h = emu.create(o)
    try:
        for m in (0, L + K - 1):
            x = o.fill_uniform([m], 42 + m)[0]
            x[0], x[1], x[-1] = 0, o.moduli[m] - 1, o.moduli[m] - 1
            out = np.empty_like(x)
            assert emu.emu_ntt8(h, m, p(x), p(out), 0, 0, 0) == 0
            exp = o.ntt([m], x[None])[0]
            assert np.array_equal(out, exp), f"forward mod={m}"
            back = np.empty_like(x)
            assert emu.emu_ntt8(h, m, p(out), p(back), 1, 0, 0) == 0
            assert np.array_equal(back, x), f"inverse mod={m}"
            k = o.moduli[m] - 12345
            buf = out.copy()
            emu.emu_ntt8(h, m, p(buf), p(buf), 1, k, 1)
            assert np.array_equal(buf, o.ewe(5, [m], x[None], k=[k])[0])
        m = 1
        worst = np.full(1 << logN, o.moduli[m] - 1, dtype=np.uint64)
        out = np.empty_like(worst)
        emu.emu_ntt8(h, m, p(worst), p(out), 0, 0, 0)
        assert np.array_equal(out, o.ntt([m], worst[None])[0])
        emu.emu_ntt8(h, m, p(out), p(out), 1, 0, 0)
        assert np.array_equal(out, worst)
        x, mn, ad, mx = (o.fill_uniform([m], s)[0] for s in (1, 2, 3, 4))
        q = o.moduli[m]
        k, mk, ak = q - 2, q - 77, q - 5
        out = np.empty_like(x)
        assert emu.emu_ntt_sub_scale8(h, m, p(x), p(mn), p(ad), p(out), k, p(mx), mk, ak) == 0
        xin = o.ewe(3, [m], x[None], None, o.ewe(5, [m], mx[None], k=[mk]))
        exp = o.ewe(3, [m], o.ewe(6, [m], mn[None], None, o.ntt([m], xin), k=[k]), None, o.ewe(5, [m], ad[None], k=[ak]))[0]
        assert np.array_equal(out, exp)
        assert emu.emu_ntt_sub_scale8(h, m, p(x), p(mn), None, p(out), k, None, 0, 0) == 0
        assert np.array_equal(out, o.ewe(6, [m], mn[None], None, o.ntt([m], x[None]), k=[k])[0])
    finally:
        emu.emu_destroy(h)


def test_emu_key_mac_lazy_ranges(emu_m32):
    emu = emu_m32
    """the fused transform x key kernel's multiply-accumulate (hm_mac_add): the word-wise Montgomery product with the key word as the
    constant.  For x anywhere below 8q (twice the transform's lazy output range), y below q: every product adds x*y*2^-64 mod q plus at
    most q, less than 1.5q + 2^28, so five terms stay below 8q <= 2^63 (the kernel takes at most four); the final product with 2^128 mod q
    and one subtraction give x.y mod q — with the extreme operands (0, 1, q-1, 8q-1, values next to powers of two) and random ones, for
    the default chain (60-bit moduli) and for 59-, 45- and 40-bit moduli"""
    from sympy import isprime
    small = []
    for bits in (59, 45, 40, 59, 45, 40):   # (every modulus of a context is 1 mod 2^32)
        c = (1 << bits) + 1 - (1 << 32)
        while not isprime(c) or c in small:
            c -= 1 << 32
            assert c > 1 << (bits - 1)
        small.append(c)
    chain = np.array(small, dtype=np.uint64)
    rng = np.random.default_rng(5)
    for h, mod in [(emu.emu_create(13, 4, 2), m) for m in range(6)] + [(emu.emu_create_mods(13, 4, 2, p(chain[:4]), p(chain[4:])), m) for m in range(6)]:
        q = emu.emu_modulus(h, mod)
        xs = [0, 1, q - 1, q, 2 * q, 4 * q - 1, 8 * q - 1, 8 * q - 2, (1 << 32) - 1, 1 << 32, (1 << 62) + 1, (1 << 32) + 1]
        ys = [0, 1, q - 1, q - 2, (1 << 32) - 1, 1 << 32, (1 << 59) + 12345, q >> 1]
        ex = [(x, y) for x in xs if x < 8 * q for y in ys if y < q]
        n = 4096
        X = np.concatenate([np.array([e[0] for e in ex], dtype=np.uint64), (rng.integers(0, 1 << 63, n, dtype=np.uint64) % np.uint64(8 * q))])
        Y = np.concatenate([np.array([e[1] for e in ex], dtype=np.uint64), (rng.integers(0, 1 << 63, n, dtype=np.uint64) % np.uint64(q))])
        n = len(X)
        acc = np.zeros(n, dtype=np.uint64)
        want = [0] * n
        for term in range(5):   # one more term than the kernel takes
            Xt, Yt = np.roll(X, term * 7), np.roll(Y, term * 3)
            before = acc.copy()
            emu.emu_mac(h, mod, p(acc), p(Xt), p(Yt), n, 1 if term >= 2 else 0)
            for i in range(n):
                want[i] = (want[i] + int(Xt[i]) * int(Yt[i])) % q
                a = int(acc[i])
                assert (a << 64) % q == want[i], (mod, term, i)
                assert a - int(before[i]) < (3 * q) // 2 + (1 << 28) and a < 8 * q, (mod, term, i, a // q)
        for i in range(0, n, 97):
            assert emu.emu_mac_final(h, mod, int(acc[i])) == want[i]
        emu.emu_destroy(h)


def test_emu_montgomery_products_and_butterfly_ranges(emu_m32):
    emu = emu_m32
    """the word-wise Montgomery arithmetic on moduli q = h 2^32 + 1 (hm_modarith.h) against Python integers, at the edges of every stated range:
    hm_mont_acc (c + x wt 2^-64 mod q + {0, q}, at most floor(x wt / 2^64) + q + h + 1 < 1.5q + 2^28 for x up to 2^63 - 1, wt up to q - 1, no 64-bit overflow with c up
    to 8q), the per-launch constant product, the two-step reduction of the 128-bit conversion accumulator (16 and 32 terms at their bounds),
    and the butterflies of each kind at the bounds hm_fwd_bound assumes (inputs below 6q / 8q, outputs below 8q / 6q / 4q; inverse 4q -> 4q);
    60-, 59-, 45- and 40-bit moduli"""
    from sympy import isprime
    emu.emu_mont_acc.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint32]
    emu.emu_mont_const_mul.restype = C.c_uint64
    emu.emu_mont_const_mul.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]
    small = []
    for bits in (59, 45, 40, 59, 45, 40):
        c = (1 << bits) + 1 - (1 << 32)
        while not isprime(c) or c in small:
            c -= 1 << 32
            assert c > 1 << (bits - 1)
        small.append(c)
    chain = np.array(small, dtype=np.uint64)
    rng = np.random.default_rng(17)
    for h, mod in [(emu.emu_create(13, 4, 2), m) for m in (0, 5)] + [(emu.emu_create_mods(13, 4, 2, p(chain[:4]), p(chain[4:])), m) for m in range(3)]:
        q = emu.emu_modulus(h, mod)
        r64inv = pow(1 << 64, -1, q)
        xs = [0, 1, q - 1, q, 4 * q - 1, 8 * q - 1, (1 << 63) - 1, (1 << 32) - 1, 1 << 32, (1 << 62) + 12345]
        ws = [0, 1, q - 1, q - 2, (1 << 32) - 1, 1 << 32, q >> 1]
        cs = [0, q, 6 * q - 1, 8 * q - 1]
        ex = [(c, x, w) for c in cs for x in xs if x < (1 << 63) for w in ws if w < q]
        n = 4096
        Cc = np.concatenate([np.array([e[0] for e in ex], dtype=np.uint64), rng.integers(0, 1 << 62, n, dtype=np.uint64) % np.uint64(8 * q)])
        X = np.concatenate([np.array([e[1] for e in ex], dtype=np.uint64), rng.integers(0, 1 << 63, n, dtype=np.uint64)])
        W = np.concatenate([np.array([e[2] for e in ex], dtype=np.uint64), rng.integers(0, 1 << 62, n, dtype=np.uint64) % np.uint64(q)])
        out = np.zeros(len(X), dtype=np.uint64)
        emu.emu_mont_acc(h, mod, p(Cc), p(X), p(W), p(out), len(X))
        for c, x, w, o in zip(Cc.tolist(), X.tolist(), W.tolist(), out.tolist()):
            v = o - c
            assert 0 <= v <= (x * w >> 64) + q + (q >> 32) + 1 and v < (3 * q) // 2 + (1 << 28) and o < (1 << 64), (mod, c, x, w)
            assert v % q == x * w * r64inv % q, (mod, c, x, w)
        for x in xs:
            if x >= (1 << 63):
                continue
            for k in ws:
                assert emu.emu_mont_const_mul(h, mod, x, k) == x * k % q, (mod, x, k)
        # the conversion accumulator: TERMS products y w with y below 2^60 - 2^32 (any input modulus) and w below q, at the bound and random
        for terms in (16, 32):
            zmax = terms * ((1 << 60) - (1 << 32)) * (q - 1)
            for z in [0, 1, q, zmax, zmax - 1, (1 << 64) - 1, 1 << 64, (zmax >> 1) + 977] + [int(rng.integers(0, 1 << 62)) * int(rng.integers(0, 1 << 62)) % (zmax + 1) for _ in range(2000)]:
                assert emu.emu_redc_wide(h, mod, z & ((1 << 64) - 1), z >> 64, terms) == z * r64inv % q, (mod, terms, z)
        # butterflies at the bounds of the stage schedule
        Xb, Yb = (C.c_uint64 * 1)(), (C.c_uint64 * 1)()
        for kind, xmax, ymax, omax in ((0, 6, 8, 8), (1, 8, 8, 6), (2, 8, 8, 4), (3, 4, 4, 4)):
            vals = lambda m: [0, 1, q - 1, m * q - 1, m * q - 2, (m * q) >> 1] + [int(v) % (m * q) for v in rng.integers(0, 1 << 62, 200)]
            for x in vals(xmax):
                for y in vals(ymax)[::7] + [ymax * q - 1]:
                    for w in (1, q - 1, 3 + (q >> 3)):
                        Xb[0], Yb[0] = x, y
                        emu.emu_bfly(h, mod, kind, Xb, Yb, w)
                        a, b = int(Xb[0]), int(Yb[0])
                        assert a < omax * q and b < omax * q, (mod, kind, x, y, w, a // q, b // q)
                        if kind < 3:
                            assert a % q == (x + w * y) % q and b % q == (x - w * y) % q, (mod, kind, x, y, w)
                        else:
                            assert a % q == (x + y) % q and b % q == (x - y) * w % q, (mod, kind, x, y, w)
        emu.emu_destroy(h)


def _mixed_chain(step_log):
    """59-, 45- and 31-bit primes = 1 mod 2^step_log (twice): a chain an FHE library might hand over"""
    from sympy import isprime
    small = []
    for bits in (59, 45, 31, 59, 45, 31):
        c = (1 << bits) - (1 << step_log) + 1 - (len(small) << 20)
        while not isprime(c) or c in small:
            c -= 1 << step_log
        small.append(c)
    return small


def test_emu_generic_key_mac_lazy_ranges(emu_gen):
    """the generic back-end's key multiply-accumulate (hm_mac_add under HM_GENERIC): Barrett's quotient from two approximate high
    products.  For x anywhere below 8q (the generic transform's lazy output), y below q: every product adds x*y mod q plus at most 6q,
    two terms stay below 14q, with the fold from the third term on any number of terms stays below 15q, and the final reduction is
    x.y mod q — extreme operands (0, 1, q-1, 8q-1, values next to powers of two) and random ones, for SURVEY.md 8d's 60-bit chain and
    for 59-, 45- and 31-bit moduli (the operand shift 64 - k and the quotient constant depend on the width k)"""
    emu = emu_gen
    o = emu.oracle(13, 4, 2)
    chain = np.array(_mixed_chain(14), dtype=np.uint64)
    rng = np.random.default_rng(5)
    for h, mod in [(emu.create(o), m) for m in range(6)] + [(emu.emu_create_mods(13, 4, 2, p(chain[:4]), p(chain[4:])), m) for m in range(6)]:
        assert h
        q = emu.emu_modulus(h, mod)
        xs = [0, 1, q - 1, q, 2 * q, 4 * q - 1, 8 * q - 1, 8 * q - 2, (1 << 32) - 1, 1 << 32, (1 << 62) + 1, (1 << 32) + 1]
        ys = [0, 1, q - 1, q - 2, (1 << 32) - 1, 1 << 32, (1 << 59) + 12345, q >> 1]
        ex = [(x, y) for x in xs if x < 8 * q for y in ys if y < q]
        n = 4096
        X = np.concatenate([np.array([e[0] for e in ex], dtype=np.uint64), (rng.integers(0, 1 << 63, n, dtype=np.uint64) % np.uint64(8 * q))])
        Y = np.concatenate([np.array([e[1] for e in ex], dtype=np.uint64), (rng.integers(0, 1 << 63, n, dtype=np.uint64) % np.uint64(q))])
        n = len(X)
        acc = np.zeros(n, dtype=np.uint64)
        want = [0] * n
        for term in range(6):   # more terms than the kernel takes: the fold keeps the sum in range for any number
            Xt, Yt = np.roll(X, term * 7), np.roll(Y, term * 3)
            before = acc.copy()
            emu.emu_mac(h, mod, p(acc), p(Xt), p(Yt), n, 1 if term >= 2 else 0)
            for i in range(n):
                want[i] = (want[i] + int(Xt[i]) * int(Yt[i])) % q
                a = int(acc[i])
                assert a % q == want[i], (mod, term, i)
                assert a < (14 * q if term < 2 else 15 * q), (mod, term, i, a // q)
                if term < 2:
                    assert a - int(before[i]) < 7 * q
        for i in range(0, n, 97):
            assert emu.emu_mac_final(h, mod, int(acc[i])) == want[i]
        emu.emu_destroy(h)


def test_emu_generic_butterfly_and_reduction_ranges(emu_gen):
    """the generic back-end's butterflies (Shoup product with the approximate quotient: [0, 4q) for ANY 64-bit operand) at the bounds its
    stage schedule assumes — twice the word-wise Montgomery ones: inputs below 12q / 16q, outputs below 16q / 12q / 8q; inverse 4q -> 4q —
    and the Montgomery reduction of the 128-bit conversion accumulator for any odd q (16 and 32 terms at their bounds), for SURVEY.md
    8d's 60-bit chain and 59-, 45- and 31-bit moduli"""
    emu = emu_gen
    o = emu.oracle(13, 4, 2)
    chain = np.array(_mixed_chain(14), dtype=np.uint64)
    rng = np.random.default_rng(23)
    for h, mod in [(emu.create(o), m) for m in (0, 5)] + [(emu.emu_create_mods(13, 4, 2, p(chain[:4]), p(chain[4:])), m) for m in range(3)]:
        assert h
        q = emu.emu_modulus(h, mod)
        r64inv = pow(1 << 64, -1, q)
        for terms in (16, 32):
            zmax = terms * ((1 << 60) - 1) * (q - 1)
            for z in [0, 1, q, zmax, zmax - 1, (1 << 64) - 1, 1 << 64, (zmax >> 1) + 977] + [int(rng.integers(0, 1 << 62)) * int(rng.integers(0, 1 << 62)) % (zmax + 1) for _ in range(2000)]:
                assert emu.emu_redc_wide(h, mod, z & ((1 << 64) - 1), z >> 64, terms) == z * r64inv % q, (mod, terms, z)
        Xb, Yb = (C.c_uint64 * 1)(), (C.c_uint64 * 1)()
        for kind, xmax, ymax, omax in ((0, 12, 16, 16), (1, 16, 16, 12), (2, 16, 16, 8), (3, 4, 4, 4)):
            vals = lambda m: [0, 1, q - 1, m * q - 1, m * q - 2, (m * q) >> 1] + [int(v) % (m * q) for v in rng.integers(0, 1 << 62, 200)]
            for x in vals(xmax):
                for y in vals(ymax)[::7] + [ymax * q - 1, (1 << 64) - 1 if kind < 3 else 4 * q - 1]:
                    for w in (1, q - 1, 3 + (q >> 3)):
                        Xb[0], Yb[0] = x, y
                        emu.emu_bfly(h, mod, kind, Xb, Yb, w)
                        a, b = int(Xb[0]), int(Yb[0])
                        assert a < omax * q and b < omax * q, (mod, kind, x, y, w, a // q, b // q)
                        if kind < 3:
                            assert a % q == (x + w * y) % q and b % q == (x - w * y) % q, (mod, kind, x, y, w)
                        else:
                            assert a % q == (x + y) % q and b % q == (x - y) * w % q, (mod, kind, x, y, w)
        emu.emu_destroy(h)


@pytest.mark.parametrize("n_in", [16, 17, 24, 28, 32])
def test_emu_two_group_conversion_of_wide_digits(emu, n_in):
    """round 6: the arithmetic of the conversion inside the first pass for digits of 16 .. 32 limbs (k_bconv_col's two input groups: hm_bconv_cols
    per group, one hm_redc_wide over all terms) on the CPU emulator, on both arithmetic builds: random inputs, every input at q_i - 1 (the 128-bit
    sums at their largest: two conditional subtractions above 16 terms), zeros and ones; plain and split-30 packed inputs; against the oracle's
    conversion"""
    emu.emu_bconv_two_groups.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
    emu.emu_bconv_two_groups.restype = C.c_int
    logN, n_out = 13, 5
    o = emu.oracle(logN, n_in, n_out)
    h = emu.create(o)
    try:
        in_ids, out_ids = list(range(n_in)), list(range(n_in, n_in + n_out))
        x = o.fill_uniform(in_ids, 9)
        for r, m in enumerate(in_ids):
            x[r, :8] = o.moduli[m] - 1
            x[r, 8:10] = [0, 1]
        x[n_in - 1, 10] = o.moduli[in_ids[-1]] - 1
        want = o.bconv_matmul(in_ids, out_ids, x)
        ii, oi = np.array(in_ids, dtype=np.uint32), np.array(out_ids, dtype=np.uint32)
        pk = (x & np.uint64(0x3FFFFFFF)) | ((x >> np.uint64(30)) << np.uint64(32))
        for src, packed in ((x, 0), (pk, 1)):
            out = np.zeros((n_out, o.N), dtype=np.uint64)
            assert emu.emu_bconv_two_groups(h, p(ii), n_in, p(oi), n_out, p(np.ascontiguousarray(src)), p(out), packed) == 0
            assert np.array_equal(out, want), packed
    finally:
        emu.emu_destroy(h)


@pytest.mark.parametrize("ept", [16, 8])
def test_lds_bank_model_follows_the_kernels_and_finds_no_conflict(ept):
    """tools/lds_banks.py restates the ROW pass's geometry and swizzles to count LDS bank conflicts by the chip's banking rules.  Pinned here to the
    headers the kernels compile (every LDS word of every thread, round and access unit, both directions, through the emulator build), and the
    claim itself: under the sets the passes use, no read or write of any round shares a bank inside a lane group — and the image is a bijection"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lds_banks", os.path.join(HERE, "..", "tools", "lds_banks.py"))
    lb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lb)
    emu = _load("libhm_emu.so")
    emu.emu_row_lds_word.argtypes = [C.c_int] * 5
    X, Cc = lb.coords(ept)
    n, threads, units = X.shape
    for inverse in (0, 1):
        k = 2 if ept == 8 else inverse
        swz = lb.xor_swizzle(lb.SETS[k])
        for R in range(n):
            W = swz((Cc[R] << lb.LOGR) | X[R], X[R])
            got = np.array([[emu.emu_row_lds_word(ept, R, t, a, inverse) for a in range(units)] for t in range(threads)])
            assert np.array_equal(W, got), (ept, R, inverse)
        fwd, inv = lb.evaluate(ept, X, Cc, swz)
        assert (inv if inverse else fwd)[0] == 0
    # and the set the passes used up to round 5 costs what the counters of profiles/r06_pmc_kernels_batch10.txt say (inverse hm16: 25 %)
    fwd, inv = lb.evaluate(ept, X, Cc, lb.xor_swizzle(lb.SETS[0]))
    assert (fwd, inv) == (((0, 768), (256, 1024)) if ept == 16 else ((384, 1536), (640, 1792)))


@pytest.mark.parametrize("logN,ept", [(13, 16), (15, 16), (16, 16), (15, 8), (16, 8)])
def test_emu_transforms_that_read_through_an_automorphism(emu, logN, ept):
    """round 6: hrotate's automorphism launch folded into its consumers.  MODE 6: the inverse transform reads its input through X -> X^g
    (INTT(automorph_g(x)) with automorph_g(x) never in memory); MODE 7: the fused forward transform's epilogue reads the ADDEND through it
    (out = (minuend - NTT(x)) * k + automorph_g(addend) [* addend_k]).  The index map takes aligned pairs to aligned pairs (in order or swapped),
    so the passes' own 16-byte units serve; here every unit of every thread of both geometries against the oracle's automorphism + transform,
    with Galois elements that swap pairs (5: rotation by one slot), that do not (2N - 1: conjugation), 25, 3 and the identity"""
    emu.emu_intt_auto.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    emu.emu_ntt_sub_scale_auto.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint64, C.c_uint64, C.c_uint32, C.c_int]
    L, K = 2, 1
    o = emu.oracle(logN, L, K)
    h = emu.create(o)
    N = 1 << logN
    try:
        for m in (0, L):
            q = o.moduli[m]
            x, mn, ad = (o.fill_uniform([m], s + 10 * m)[0] for s in (1, 2, 3))
            x[:2] = [q - 1, 0]
            ad[:2] = [0, q - 1]
            for g in (5, 2 * N - 1, 25, 3, 1):
                out = np.empty_like(x)
                assert emu.emu_intt_auto(h, m, p(x), p(out), g, ept) == 0
                assert np.array_equal(out, o.ntt([m], o.automorph_eval(x[None], g), inverse=True)[0]), (m, g)
                k, ak = q - 2, q - 5
                assert emu.emu_ntt_sub_scale_auto(h, m, p(x), p(mn), p(ad), p(out), k, 0, g, ept) == 0
                base = o.ewe(6, [m], mn[None], None, o.ntt([m], x[None]), k=[k])
                assert np.array_equal(out, o.ewe(3, [m], base, None, o.automorph_eval(ad[None], g))[0]), (m, g)
                assert emu.emu_ntt_sub_scale_auto(h, m, p(x), p(mn), p(ad), p(out), k, ak, g, ept) == 0
                assert np.array_equal(out, o.ewe(3, [m], base, None, o.ewe(5, [m], o.automorph_eval(ad[None], g), k=[ak]))[0]), (m, g)
    finally:
        emu.emu_destroy(h)
