"""ctypes wrapper around oracle/libhomoracle.so.

TEST INFRASTRUCTURE ONLY (see oracle/homoracle.h): imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg — never by the product package homulator_amd/.
Arithmetic parity vs the reference is "parity unpinned" (the reference has no arithmetic); the
oracle is pinned by big-integer KATs in tests/test_oracle_kat.py.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EWE_MUL, EWE_MAC2, EWE_MAC_ADD, EWE_ADD, EWE_SUB, EWE_MUL_CONST, EWE_SUB_SCALE, EWE_COPY = range(8)


def build(force=False):
    so = os.path.join(_HERE, "libhomoracle.so")
    src = os.path.join(_HERE, "homoracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "libhomoracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        name = os.environ.get("HOMORACLE_LIB") or build()
        L = C.CDLL(name)
        u32, u64, p = C.c_uint32, C.c_uint64, C.c_void_p
        L.ho_create.restype = p
        L.ho_create.argtypes = [u32, u32, u32]
        L.ho_create_chain.restype = p
        L.ho_create_chain.argtypes = [u32, u32, u32, p]
        L.ho_chain_below.restype = C.c_int
        L.ho_chain_below.argtypes = [u32, u32, u32, p]
        L.ho_destroy.argtypes = [p]
        for f in ("ho_N", "ho_L", "ho_K"):
            getattr(L, f).restype = u32
            getattr(L, f).argtypes = [p]
        L.ho_modulus.restype = u64
        L.ho_modulus.argtypes = [p, u32]
        L.ho_psi.restype = u64
        L.ho_psi.argtypes = [p, u32]
        L.ho_set_threads.argtypes = [C.c_int]
        for f in ("ho_mulmod", "ho_powmod"):
            getattr(L, f).restype = u64
            getattr(L, f).argtypes = [u64, u64, u64]
        L.ho_invmod.restype = u64
        L.ho_invmod.argtypes = [u64, u64]
        L.ho_ntt.argtypes = [p, u32, p, C.c_int]
        L.ho_ntt_limbs.argtypes = [p, p, u32, p, C.c_int]
        L.ho_automorph_eval.argtypes = [p, p, p, u32]
        L.ho_automorph_coef.argtypes = [p, u32, p, p, u32]
        L.ho_ewe.argtypes = [p, C.c_int, u32, p, p, p, p, u64, p]
        L.ho_bconv_consts.argtypes = [p, p, u32, p, u32, p, p]
        L.ho_bconv_scale.argtypes = [p, p, u32, p, u32, p, p]
        L.ho_bconv_matmul.argtypes = [p, p, u32, p, u32, p, p]
        L.ho_keyswitch.argtypes = [p, u32, p, p, p, p, p]
        L.ho_rescale.argtypes = [p, u32, p, p]
        L.ho_hmult.argtypes = [p, u32, p, p, p, C.c_int, p]
        L.ho_hrotate.argtypes = [p, u32, p, u32, p, p]
        L.ho_hadd.argtypes = [p, u32, p, p, p]
        L.ho_pmult.argtypes = [p, u32, p, p, p]
        L.ho_padd.argtypes = [p, u32, p, p, p]
        L.ho_fill_uniform.argtypes = [p, p, u32, u64, p]
        _LIB = L
    return _LIB


def _ptr(a):
    if a is None:
        return None
    assert a.dtype == np.uint64 or a.dtype == np.uint32
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def _ids(ids):
    return np.ascontiguousarray(np.asarray(ids, dtype=np.uint32))


class KsDump(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("modup_intt", "modup_decomp", "ext", "ip", "moddown_intt", "moddown_bconv", "moddown_ntt")]


def chain_below(logN, bits, count):
    """the `count` largest primes below 2^bits that are 1 mod 2N, descending (bits = 60: SURVEY.md 8(d)'s chain as written)"""
    out = np.zeros(count, dtype=np.uint64)
    if not lib().ho_chain_below(logN, bits, count, _ptr(out)):
        raise ValueError(f"no {count} primes = 1 mod 2^{logN + 1} below 2^{bits}")
    return [int(x) for x in out]


class Oracle:
    """One parameter set (N = 2^logN, L Q-primes, K = alpha special primes).
    chain: "mont32" (default: the L+K largest primes h 2^32 + 1 below 2^60), "survey" (SURVEY.md 8(d) as written: the
    largest primes = 1 mod 2N below 2^60), an int b (the largest such primes below 2^b, e.g. 36 for the reference's 36-bit words), or an
    explicit list of L + K moduli (first L = Q, next K = P)."""

    def __init__(self, logN, L, K, chain="mont32"):
        self.l = lib()
        if chain == "mont32" or chain is None:
            self.h = self.l.ho_create(logN, L, K)
        else:
            mods = chain_below(logN, 60, L + K) if chain == "survey" else chain_below(logN, int(chain), L + K) if isinstance(chain, int) else [int(x) for x in chain]
            if len(mods) != L + K:
                raise ValueError("the chain needs L + K moduli")
            arr = np.ascontiguousarray(np.asarray(mods, dtype=np.uint64))
            self.h = self.l.ho_create_chain(logN, L, K, _ptr(arr))
        if not self.h:
            raise ValueError("ho_create failed (moduli must be distinct primes = 1 mod 2N below 2^60)")
        self.logN, self.N, self.L, self.K = logN, 1 << logN, L, K
        self.moduli = [int(self.l.ho_modulus(self.h, i)) for i in range(L + K)]
        self.psis = [int(self.l.ho_psi(self.h, i)) for i in range(L + K)]

    def __del__(self):
        try:
            self.l.ho_destroy(self.h)
        except Exception:
            pass

    def set_threads(self, n):
        self.l.ho_set_threads(int(n))

    # ids of the extended basis at level ell: Q limbs then P limbs
    def ext_ids(self, ell):
        return list(range(ell)) + [self.L + i for i in range(self.K)]

    def beta(self, ell):
        return (ell + self.K - 1) // self.K

    def ntt(self, mod_ids, a, inverse=False):
        a = np.array(a, dtype=np.uint64, copy=True).reshape(-1, self.N)
        ids = _ids(mod_ids)
        assert len(ids) == a.shape[0]
        self.l.ho_ntt_limbs(self.h, _ptr(ids), len(ids), _ptr(a), 1 if inverse else 0)
        return a

    def automorph_eval(self, a, g):
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, self.N)
        out = np.empty_like(a)
        for i in range(a.shape[0]):
            self.l.ho_automorph_eval(self.h, _ptr(a[i]), _ptr(out[i]), g)
        return out

    def automorph_coef(self, mod_ids, a, g):
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, self.N)
        out = np.empty_like(a)
        for i, m in enumerate(mod_ids):
            self.l.ho_automorph_coef(self.h, int(m), _ptr(a[i]), _ptr(out[i]), g)
        return out

    def ewe(self, op, mod_ids, a=None, b=None, c=None, d=None, k=None):
        """Per-limb element-wise op; operands are [n][N]; k is a per-limb list of constants."""
        n = len(mod_ids)
        ops = [None if x is None else np.ascontiguousarray(x, dtype=np.uint64).reshape(n, self.N) for x in (a, b, c, d)]
        out = np.empty((n, self.N), dtype=np.uint64)
        for i, m in enumerate(mod_ids):
            ptrs = [None if x is None else _ptr(x[i]) for x in ops]
            self.l.ho_ewe(self.h, op, int(m), ptrs[0], ptrs[1], ptrs[2], ptrs[3], int(k[i]) if k is not None else 0,
                          _ptr(out[i]))
        return out

    def bconv_consts(self, in_ids, out_ids):
        i, o = _ids(in_ids), _ids(out_ids)
        qhi = np.empty(len(i), dtype=np.uint64)
        tab = np.empty((len(i), max(len(o), 1)), dtype=np.uint64)
        self.l.ho_bconv_consts(self.h, _ptr(i), len(i), _ptr(o), len(o), _ptr(qhi), _ptr(tab))
        return qhi, tab[:, :len(o)]

    def bconv_scale(self, in_ids, a):
        i = _ids(in_ids)
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(len(i), self.N)
        out = np.empty_like(a)
        self.l.ho_bconv_scale(self.h, _ptr(i), len(i), _ptr(i), 0, _ptr(a), _ptr(out))
        return out

    def bconv_matmul(self, in_ids, out_ids, scaled):
        i, o = _ids(in_ids), _ids(out_ids)
        s = np.ascontiguousarray(scaled, dtype=np.uint64).reshape(len(i), self.N)
        out = np.empty((len(o), self.N), dtype=np.uint64)
        self.l.ho_bconv_matmul(self.h, _ptr(i), len(i), _ptr(o), len(o), _ptr(s), _ptr(out))
        return out

    def keyswitch(self, ell, d, evk, dump=False):
        E, beta, N = ell + self.K, self.beta(ell), self.N
        d = np.ascontiguousarray(d, dtype=np.uint64).reshape(ell, N)
        evk = np.ascontiguousarray(evk, dtype=np.uint64).reshape(beta, 2, E, N)
        o0 = np.empty((ell, N), dtype=np.uint64)
        o1 = np.empty((ell, N), dtype=np.uint64)
        dd, st = None, None
        if dump:
            dd = dict(modup_intt=np.empty((ell, N), np.uint64), modup_decomp=np.empty((ell, N), np.uint64),
                      ext=np.empty((beta, E, N), np.uint64), ip=np.empty((2, E, N), np.uint64),
                      moddown_intt=np.empty((2, self.K, N), np.uint64),
                      moddown_bconv=np.empty((2, ell, N), np.uint64), moddown_ntt=np.empty((2, ell, N), np.uint64))
            st = KsDump(**{k: v.ctypes.data for k, v in dd.items()})
        self.l.ho_keyswitch(self.h, ell, _ptr(d), _ptr(evk), _ptr(o0), _ptr(o1), C.byref(st) if st else None)
        return (o0, o1, dd) if dump else (o0, o1)

    def rescale(self, ell, x):
        x = np.ascontiguousarray(x, dtype=np.uint64).reshape(ell, self.N)
        out = np.empty((ell - 1, self.N), dtype=np.uint64)
        self.l.ho_rescale(self.h, ell, _ptr(x), _ptr(out))
        return out

    def hmult(self, ell, ct1, ct2, evk, rescale=True):
        N = self.N
        ct1 = np.ascontiguousarray(ct1, dtype=np.uint64).reshape(2, ell, N)
        ct2 = np.ascontiguousarray(ct2, dtype=np.uint64).reshape(2, ell, N)
        evk = np.ascontiguousarray(evk, dtype=np.uint64).reshape(self.beta(ell), 2, ell + self.K, N)
        out = np.empty((2, ell - 1 if rescale else ell, N), dtype=np.uint64)
        self.l.ho_hmult(self.h, ell, _ptr(ct1), _ptr(ct2), _ptr(evk), 1 if rescale else 0, _ptr(out))
        return out

    def hrotate(self, ell, ct, galois, evk):
        N = self.N
        ct = np.ascontiguousarray(ct, dtype=np.uint64).reshape(2, ell, N)
        evk = np.ascontiguousarray(evk, dtype=np.uint64).reshape(self.beta(ell), 2, ell + self.K, N)
        out = np.empty((2, ell, N), dtype=np.uint64)
        self.l.ho_hrotate(self.h, ell, _ptr(ct), int(galois), _ptr(evk), _ptr(out))
        return out

    def hadd(self, ell, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(2, ell, self.N)
        b = np.ascontiguousarray(b, dtype=np.uint64).reshape(2, ell, self.N)
        out = np.empty_like(a)
        self.l.ho_hadd(self.h, ell, _ptr(a), _ptr(b), _ptr(out))
        return out

    def pmult(self, ell, ct, pt):
        ct = np.ascontiguousarray(ct, dtype=np.uint64).reshape(2, ell, self.N)
        pt = np.ascontiguousarray(pt, dtype=np.uint64).reshape(ell, self.N)
        out = np.empty_like(ct)
        self.l.ho_pmult(self.h, ell, _ptr(ct), _ptr(pt), _ptr(out))
        return out

    def padd(self, ell, ct, pt):
        ct = np.ascontiguousarray(ct, dtype=np.uint64).reshape(2, ell, self.N)
        pt = np.ascontiguousarray(pt, dtype=np.uint64).reshape(ell, self.N)
        out = np.empty_like(ct)
        self.l.ho_padd(self.h, ell, _ptr(ct), _ptr(pt), _ptr(out))
        return out

    def fill_uniform(self, mod_ids, seed):
        ids = _ids(mod_ids)
        out = np.empty((len(ids), self.N), dtype=np.uint64)
        self.l.ho_fill_uniform(self.h, _ptr(ids), len(ids), int(seed) & (2**64 - 1), _ptr(out))
        return out

    # synthetic inputs of SURVEY §8d: ciphertexts and evk from fixed seeds
    def synth_ct(self, ell, seed):
        ids = list(range(ell))
        return np.stack([self.fill_uniform(ids, seed), self.fill_uniform(ids, seed + 1000)])

    def synth_evk(self, ell, seed):
        ids = self.ext_ids(ell)
        E = len(ids)
        return np.stack([np.stack([self.fill_uniform(ids, seed + (j * 2 + k) * 1000) for k in range(2)])
                         for j in range(self.beta(ell))]).reshape(self.beta(ell), 2, E, self.N)
