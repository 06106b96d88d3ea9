"""Toy RLWE scheme on top of the ORACLE (test helper only): keygen / encrypt / decrypt / evk generation with
Python big integers, used to pin the oracle's key-switch, hmult and hrotate by decryption (SURVEY §8c KATs v, vi)."""
import numpy as np


class Toy:
    def __init__(self, orc, seed=1):
        self.o = orc
        self.N, self.L, self.K = orc.N, orc.L, orc.K
        self.rng = np.random.default_rng(seed)
        self.all_ids = list(range(self.L + self.K))
        self.s = self.rng.integers(-1, 2, self.N).astype(object)  # ternary secret (coefficient form)

    # ---- signed integer polynomial (object array) -> RNS eval form over mod ids
    def to_rns_eval(self, poly, ids):
        a = np.empty((len(ids), self.N), dtype=np.uint64)
        for r, m in enumerate(ids):
            q = self.o.moduli[m]
            a[r] = np.array([int(x) % q for x in poly], dtype=np.uint64)
        return self.o.ntt(ids, a)

    def small_err(self):
        return np.rint(self.rng.normal(0, 3.2, self.N)).astype(np.int64).astype(object)

    def uniform(self, ids):
        return np.stack([np.array([int(x) for x in self.rng.integers(0, self.o.moduli[m], self.N, dtype=np.uint64)],
                                  dtype=np.uint64) for m in ids])

    def mul(self, ids, a, b):
        return self.o.ewe(0, ids, a, b)

    def add(self, ids, a, c):
        return self.o.ewe(3, ids, a, None, c)

    def sub(self, ids, a, c):
        return self.o.ewe(4, ids, a, None, c)

    def encrypt(self, msg, ell, secret=None):
        """ct = (c0, c1) eval form at level ell with c0 + c1*s = msg + e."""
        s = self.s if secret is None else secret
        ids = list(range(ell))
        a = self.uniform(ids)
        s_e = self.to_rns_eval(s, ids)
        me = self.to_rns_eval(np.array(msg, dtype=object) + self.small_err(), ids)
        c0 = self.sub(ids, me, self.mul(ids, a, s_e))
        return np.stack([c0, a])

    def decrypt(self, ct, ell, secret=None):
        """returns centered big-int coefficients of c0 + c1*s mod Q_ell."""
        s = self.s if secret is None else secret
        ids = list(range(ell))
        s_e = self.to_rns_eval(s, ids)
        v = self.add(ids, ct[0], self.mul(ids, ct[1], s_e))
        c = self.o.ntt(ids, v, inverse=True)
        return self.crt_center(c, ids)

    def crt_center(self, limbs, ids):
        mods = [self.o.moduli[m] for m in ids]
        Q = 1
        for q in mods:
            Q *= q
        out = np.zeros(self.N, dtype=object)
        for r, q in enumerate(mods):
            Qh = Q // q
            coef = Qh * pow(Qh % q, -1, q)
            out = out + np.array([int(x) for x in limbs[r]], dtype=object) * coef
        out = np.array([int(x) % Q for x in out], dtype=object)
        return np.array([x - Q if x > Q // 2 else x for x in out], dtype=object), Q

    def negacyclic_mul(self, a, b):
        """schoolbook product of integer polys mod X^N+1 (object arrays)."""
        N = self.N
        res = np.zeros(2 * N, dtype=object)
        for i in range(N):
            if a[i] != 0:
                res[i:i + N] += a[i] * b
        return res[:N] - res[N:]

    def automorph(self, a, g):
        N = self.N
        out = np.zeros(N, dtype=object)
        for i in range(N):
            e = (i * g) % (2 * N)
            if e < N:
                out[e] = a[i]
            else:
                out[e - N] = -a[i]
        return out

    def gen_evk(self, s_from):
        """evk for s_from -> s over the FULL extended basis: returns [beta_L][2][L+K][N] (k=0: b, k=1: a)."""
        L, K = self.L, self.K
        ids = self.all_ids
        beta_L = (L + K - 1) // K
        P = 1
        for p in range(K):
            P *= self.o.moduli[L + p]
        s_e = self.to_rns_eval(self.s, ids)
        sf_e = self.to_rns_eval(s_from, ids)
        evk = np.empty((beta_L, 2, L + K, self.N), dtype=np.uint64)
        for j in range(beta_L):
            a = self.uniform(ids)
            e = self.to_rns_eval(self.small_err(), ids)
            b = self.sub(ids, e, self.mul(ids, a, s_e))
            # + P * g_j * s_from: g_j = 1 on the limbs of digit j, 0 elsewhere (and 0 on P limbs)
            fac = [(P % self.o.moduli[m]) if (m < L and j * K <= m < (j + 1) * K) else 0 for m in ids]
            b = self.add(ids, b, self.o.ewe(5, ids, sf_e, k=fac))
            evk[j, 0], evk[j, 1] = b, a
        return evk

    def evk_at_level(self, evk_full, ell):
        """slice [beta][2][E][N] for level ell: Q limbs [0,ell) then the P limbs."""
        beta = (ell + self.K - 1) // self.K
        sel = list(range(ell)) + [self.L + p for p in range(self.K)]
        return np.ascontiguousarray(evk_full[:beta][:, :, sel, :])
