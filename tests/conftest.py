import os
import sys

import pytest
import torch  # noqa: F401  (first: see homulator_amd/__init__.py on HIP-runtime load order)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The chains the oracle serves (round 6: every known-answer test runs on each of them, not only on the default one):
#   mont32  the default chain: the largest primes h 2^32 + 1 below 2^60
#   survey  SURVEY.md 8(d) as written: the largest primes = 1 mod 2N below 2^60
#   36      the largest primes = 1 mod 2N below 2^36 (36-bit words as the reference's configuration models, config/config_4.cfg:9)
#   caller  a chain an FHE library might hand over: 31-bit Q primes and 45-bit special primes = 1 mod 2N, found here with sympy (not with
#           the oracle's own generator), NOT in descending order
ORACLE_CHAINS = ["mont32", "survey", 36, "caller"]


def caller_chain(logN, L, K):
    """L primes just below 2^31 and K just below 2^45, all = 1 mod 2N, by sympy; the Q part ascending (a caller owes no order)"""
    import sympy
    step = 2 << logN

    def below(bits, count):
        out, c = [], ((1 << bits) - 1) // step * step + 1
        while len(out) < count:
            if c < (1 << bits) and sympy.isprime(c):
                out.append(c)
            c -= step
        return out
    return below(31, L)[::-1] + below(45, K)


def make_oracle(logN, L, K, chain):
    from oracle.homoracle import Oracle
    return Oracle(logN, L, K, chain=caller_chain(logN, L, K) if chain == "caller" else chain)


@pytest.fixture(scope="session", params=ORACLE_CHAINS, ids=lambda c: f"chain-{c}")
def oracle_small(request):
    o = make_oracle(10, 6, 2, request.param)
    o.chain_name = request.param
    return o


@pytest.fixture(scope="session")
def oracle_mid():
    from oracle.homoracle import Oracle
    return Oracle(12, 8, 3)
