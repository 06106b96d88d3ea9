#!/usr/bin/env python3
"""Full-size known answers of the CPU oracle, one per prime chain it serves (round 6): SHA-256 of the output limbs of hmult (with rescale) and
hrotate at BASELINE configs[2] / [3] (N = 2^16, L = 45, l = 35, alpha = 15; SURVEY.md 8(d)'s synthetic inputs: seeds 0x484F4D55, + 2000,
+ 10000; Galois element 5), and of hmult at configs[0] (N = 2^15, 16 / 10 / 4).  The reference holds no expected value for this path
(SURVEY.md 8c), so these do not pin the arithmetic — tests/test_oracle_kat.py does, by independent mathematics on every chain — they make
sure an edit of oracle/homoracle.c cannot move a full-size result silently: the GPU suite compares the HIP path with the oracle, and the
oracle with these.

    python3 tests/golden/make_oracle_hashes.py        # rewrites tests/golden/oracle_hashes.json (a few seconds per point)
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SEED = 0x484F4D55
POINTS = [(16, 45, 35, 15), (15, 16, 10, 4)]
CHAINS = ["mont32", "survey", 36]


def digest(arr):
    import numpy as np
    return hashlib.sha256(np.ascontiguousarray(arr, dtype=np.uint64).tobytes()).hexdigest()


def compute(logN, L, ell, alpha, chain, threads=8):
    from oracle.homoracle import Oracle
    o = Oracle(logN, L, alpha, chain=chain)
    o.set_threads(threads)
    ct1, ct2, evk = o.synth_ct(ell, SEED), o.synth_ct(ell, SEED + 2000), o.synth_evk(ell, SEED + 10000)
    hm = o.hmult(ell, ct1, ct2, evk, rescale=True)
    hr = o.hrotate(ell, ct1, 5, evk)
    return {"moduli_sha256": hashlib.sha256(",".join(str(q) for q in o.moduli).encode()).hexdigest(), "q0": o.moduli[0], "p_last": o.moduli[-1],
            "inputs_sha256": digest(ct1), "hmult_c0": digest(hm[0]), "hmult_c1": digest(hm[1]), "hrotate_c0": digest(hr[0]), "hrotate_c1": digest(hr[1])}


def main():
    out = {}
    for logN, L, ell, alpha in POINTS:
        for chain in CHAINS:
            key = f"N{logN}_L{L}_l{ell}_a{alpha}_{chain}"
            out[key] = compute(logN, L, ell, alpha, chain)
            print(key, out[key]["hmult_c0"][:16], flush=True)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_hashes.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
