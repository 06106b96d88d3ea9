"""GPU parity of the one-launch transform (k_ntt_fused8: both passes in one kernel of the small-launch geometry, the hand-off between
them through the XCD's L2 behind a per-limb rendezvous on XCD-local atomics; the default for launches of up to 96 limb-polys at
N = 2^16) against the two-kernel transform and the CPU oracle, bit for bit.  Covers what the rendezvous can get wrong: more limb-polys than
the chip holds at once (workgroups of later limbs start while earlier ones wait), repeated launches (the rendezvous words reset
themselves), in-place transforms, two contexts sharing the chip (uneven load), the fused prologue / epilogue variants, the agent-scope
path, a timed-out rendezvous, and both arithmetic back-ends.  (The wide-geometry form of rounds 3 / 4, any ring size, lost every A/B
and left the tree in round 5.)"""
import numpy as np
import pytest

from oracle.homoracle import Oracle

pytestmark = pytest.mark.gpu


def _ctx(logN, L, K, chain="mont32"):
    from homulator_amd import hip
    o = Oracle(logN, L, K, chain=chain)
    return (hip.Context(logN, L, K) if chain == "mont32" else hip.Context(logN, L, K, q=o.moduli[:L], p=o.moduli[L:])), o


def test_more_limbs_than_the_chip_holds_repeated_and_in_place():
    """448 limb-polys of N = 2^16 = 7 168 workgroups in ONE one-launch transform (threshold raised to the launch-table size): later
    limbs start while earlier ones sit in their rendezvous; three launches back to back reuse the same rendezvous words; then the inverse
    in place brings the input back."""
    ctx, o = _ctx(16, 6, 3)
    try:
        n = 448
        ids = [(i * 7) % 9 for i in range(n)]
        src = ctx.alloc(n)
        ctx.fill_uniform(src, ids, 4242)
        x = src.download()
        out, ref = ctx.alloc(n), ctx.alloc(n)
        ctx.set_option("ntt_fused_small", 0)
        ctx.ntt(src, ref, ids)
        R = ref.download()
        ctx.set_option("ntt_fused_small", 448)
        for _ in range(3):
            ctx.ntt(src, out, ids)
        assert np.array_equal(out.download(), R)
        ctx.ntt(out, out, ids, inverse=True)
        assert np.array_equal(out.download(), x)
        # a sample of limbs against the oracle (the two-kernel path is compared with it in test_gpu_kernels.py)
        pick = [0, 1, 224, 447]
        assert np.array_equal(R[pick], o.ntt([ids[i] for i in pick], x[pick]))
        assert ctx.counter("ntt_cross_xcd") == 0
    finally:
        ctx.close()


def test_two_contexts_share_the_chip():
    """uneven load: two contexts (own stream each) enqueue one-launch transforms of different sizes alternately (threshold raised so that
    every size takes the form); every output is checked against the two-kernel result, and no limb-poly may have been spread over XCDs"""
    from homulator_amd import hip
    c1, c2 = hip.Context(16, 6, 3), hip.Context(16, 6, 3)
    try:
        jobs = []
        for c, n, seed in ((c1, 50, 1), (c2, 130, 2), (c1, 9, 3), (c2, 50, 4), (c1, 260, 5), (c2, 3, 6)):
            ids = [(i * 5 + seed) % 9 for i in range(n)]
            s, f, r = c.alloc(n), c.alloc(n), c.alloc(n)
            c.fill_uniform(s, ids, 900 + seed)
            jobs.append((c, ids, s, f, r))
        for c in (c1, c2):
            c.sync()
            c.set_option("ntt_fused_small", 448)
        for rep in range(4):
            for c, ids, s, f, r in jobs:
                c.ntt(s, f, ids)
        for c in (c1, c2):
            c.sync()
            c.set_option("ntt_fused_small", 0)
        for c, ids, s, f, r in jobs:
            c.ntt(s, r, ids)
        for c, ids, s, f, r in jobs:
            assert np.array_equal(f.download(), r.download())
        # the shipped threshold under the same concurrency: alternating launches of <= 96 limb-polys
        for c in (c1, c2):
            c.set_option("ntt_fused_small", 96)
        small = [j for j in jobs if len(j[1]) <= 96]
        for rep in range(6):
            for c, ids, s, f, r in small:
                c.ntt(s, f, ids)
        for c, ids, s, f, r in small:
            assert np.array_equal(f.download(), r.download())
        for c in (c1, c2):
            assert c.counter("ntt_cross_xcd") == 0, "a limb-poly was spread over XCDs under concurrency: the slow path ran"
    finally:
        c1.close(); c2.close()


def test_small_launch_geometry_equals_wide_and_oracle():
    """launches of up to 128 limb-polys at N = 2^16 run in the 8-coefficient geometry (k_ntt_col8 / k_ntt_row8: 512-thread
    workgroups, four radix-4 rounds per pass); the same calls with the geometry switched off (hm_set_option ntt_small_limbs 0)
    and the oracle must agree bit for bit: forward, inverse in place with a scale, fused epilogue with and without the mix
    prologue, worst-case operands, 1 .. 128 limb-polys"""
    ctx, o = _ctx(16, 6, 3)
    try:
        for n in (1, 9, 50, 128):
            ids = [(i * 5 + 2) % 9 for i in range(n)]
            x = o.fill_uniform(ids, 31 + n)
            x[0, :] = o.moduli[ids[0]] - 1
            d, a, b = ctx.from_host(x), ctx.alloc(n), ctx.alloc(n)
            mn, ad, mx = (ctx.alloc(n) for _ in range(3))
            for buf, s in ((mn, 5), (ad, 6), (mx, 7)):
                ctx.fill_uniform(buf, ids, s)
            k = [o.moduli[m] - 2 - r for r, m in enumerate(ids)]
            mk = [(kk * 3 + 1) % o.moduli[m] for kk, m in zip(k, ids)]
            ak = [(kk * 5 + 1) % o.moduli[m] for kk, m in zip(k, ids)]
            res = {}
            for small in (128, 0):
                ctx.set_option("ntt_small_limbs", small)
                ctx.ntt(d, a, ids)
                fwd = a.download()
                ctx.ntt(a, a, ids, inverse=True, scale=k)
                inv = a.download()
                ctx.ntt_sub_scale(d, mn, b, ids, k)
                f3 = b.download()
                ctx.ntt_mix_sub_scale(d, mn, b, ids, k, addend=ad, addend_k=ak, mix=mx, mix_k=mk)
                f43 = b.download()
                res[small] = (fwd, inv, f3, f43)
            for u, v in zip(res[128], res[0]):
                assert np.array_equal(u, v)
            if n <= 50:
                assert np.array_equal(res[128][0], o.ntt(ids, x))
                assert np.array_equal(res[128][1], o.ewe(5, ids, x, k=k))
            for buf in (d, a, b, mn, ad, mx):
                buf.free()
        ctx.set_option("ntt_small_limbs", 128)
    finally:
        ctx.close()


@pytest.mark.parametrize("chain", ["mont32", "survey"])
def test_one_launch_small_geometry_equals_two_kernels_and_oracle(chain):
    """launches of up to `ntt_fused_small` limb-polys at N = 2^16 run both passes in ONE launch of the 8-coefficient geometry
    (k_ntt_fused8: the hand-off through the XCD's L2 behind a rendezvous on XCD-local atomics).  The same calls as two kernels
    (option 0) and the oracle must agree bit for bit: forward, inverse in place with a scale, fused epilogue with and without the
    mix prologue, worst-case operands, 1 .. 96 limb-polys (96 = the shipped threshold: 1 536 workgroups, past one round of the chip),
    launches repeated back to back (the rendezvous words return to zero), out of place and in place; no limb-poly may have taken
    the agent-scope path; on both arithmetic back-ends"""
    ctx, o = _ctx(16, 6, 3, chain)
    try:
        assert ctx.counter("ntt_fused_small") == 96, "the shipped default"
        for n in (1, 9, 35, 50, 64, 72, 96):
            ids = [(i * 5 + 2) % 9 for i in range(n)]
            x = o.fill_uniform(ids, 131 + n)
            x[0, :] = o.moduli[ids[0]] - 1
            d, a, b = ctx.from_host(x), ctx.alloc(n), ctx.alloc(n)
            mn, ad, mx = (ctx.alloc(n) for _ in range(3))
            for buf, s in ((mn, 5), (ad, 6), (mx, 7)):
                ctx.fill_uniform(buf, ids, s)
            k = [o.moduli[m] - 2 - r for r, m in enumerate(ids)]
            mk = [(kk * 3 + 1) % o.moduli[m] for kk, m in zip(k, ids)]
            ak = [(kk * 5 + 1) % o.moduli[m] for kk, m in zip(k, ids)]
            res = {}
            for one in (96, 0):
                ctx.set_option("ntt_fused_small", one)
                for _ in range(3):
                    ctx.ntt(d, a, ids)
                fwd = a.download()
                ctx.ntt(a, a, ids, inverse=True, scale=k)
                inv = a.download()
                ctx.ntt_sub_scale(d, mn, b, ids, k)
                ss = b.download()
                ctx.ntt_mix_sub_scale(d, mn, b, ids, k, addend=ad, addend_k=ak, mix=mx, mix_k=mk)
                mss = b.download()
                res[one] = (fwd, inv, ss, mss)
            for u, v in zip(res[96], res[0]):
                assert np.array_equal(u, v), n
            assert np.array_equal(res[96][0], o.ntt(ids, x)), n
            assert np.array_equal(res[96][1], o.ewe(5, ids, x, k=k)), n     # INTT(NTT(x)) * k
            assert ctx.counter("ntt_cross_xcd") == 0, n
            for buf in (d, a, b, mn, ad, mx):
                buf.free()
        assert ctx.counter("ntt_cross_xcd") == 0
    finally:
        ctx.close()


def test_agent_scope_path_of_the_rendezvous():
    """the dispatcher has never spread a limb-poly's workgroups over XCDs, so the agent-scope path of the rendezvous (found through the
    XCC-id mask: L2 write-back, second rendezvous on an agent-scope counter, acquire) would never run: the test hook makes the workgroups
    of odd tiles publish another XCC id and nobody accept the XCD-local count.  Results must not change, every limb-poly must be counted
    as spread, and the words must be back at rest for the next (ordinary) launch"""
    ctx, o = _ctx(16, 6, 3)
    try:
        n = 21
        ids = [(i * 5 + 1) % 9 for i in range(n)]
        x = o.fill_uniform(ids, 991)
        d, a = ctx.from_host(x), ctx.alloc(n)
        exp = o.ntt(ids, x)
        for rep in (0, 1):
            before = ctx.counter("ntt_cross_xcd")
            ctx.set_option("ntt_fused_test_spread", 1)
            for _ in range(2):
                ctx.ntt(d, a, ids)
            assert np.array_equal(a.download(), exp)
            assert ctx.counter("ntt_cross_xcd") - before == 2 * n
            ctx.set_option("ntt_fused_test_spread", 0)
            ctx.ntt(d, a, ids)
            ctx.ntt(a, a, ids, inverse=True)
            assert np.array_equal(a.download(), x)
            assert ctx.counter("ntt_cross_xcd") - before == 2 * n
    finally:
        ctx.close()


def test_rendezvous_timeout_is_reported_everywhere_and_switches_the_form_off():
    """a rendezvous that times out (test hook: tile 0 of every limb-poly withholds its arrival, the spins are short) leaves invalid
    output.  It must be reported by the next synchronising call whatever that is (a download, not only hm_sync), switch off EVERY
    one-launch form of the context (the default k_ntt_fused8 included), leave the rendezvous words at rest, and make graphs that
    hold one-launch transforms refuse to replay; later launches run as two kernels and are right"""
    import ctypes as C
    from homulator_amd import hip
    ctx, o = _ctx(16, 4, 2)
    try:
        L = ctx.L
        n = 9
        ids = [(i * 5 + 1) % 6 for i in range(n)]
        x = o.fill_uniform(ids, 17)
        d, a = ctx.from_host(x), ctx.alloc(n)
        exp = o.ntt(ids, x)
        assert ctx.counter("ntt_fused_slots_per_xcd") >= 16
        ctx.ntt(d, a, ids)                       # warm: the launch tables exist before the capture
        assert np.array_equal(a.download(), exp)
        g = C.c_void_p()
        L.hm_capture_begin.argtypes = [C.c_void_p]
        L.hm_capture_end.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.hm_graph_launch.argtypes = [C.c_void_p, C.c_void_p]
        L.hm_graph_destroy.argtypes = [C.c_void_p]
        ctx._ck(L.hm_capture_begin(ctx.h))
        ctx.ntt(d, a, ids)
        ctx._ck(L.hm_capture_end(ctx.h, C.byref(g)))
        ctx._ck(L.hm_graph_launch(ctx.h, g))
        assert np.array_equal(a.download(), exp)
        ctx.set_option("ntt_fused_test_timeout", 1)
        ctx.ntt(d, a, ids)
        with pytest.raises(hip.HmError, match="timed out"):
            a.download()
        ctx.set_option("ntt_fused_test_timeout", 0)
        assert ctx.counter("ntt_fused_small") == 0
        assert L.hm_graph_launch(ctx.h, g) != 0 and b"one-launch" in L.hm_last_error(ctx.h)
        with pytest.raises(hip.HmError):
            ctx.set_option("ntt_fused_small", 64)
        for _ in range(2):
            ctx.ntt(d, a, ids)                   # two kernels now
            assert np.array_equal(a.download(), exp)
        ctx.ntt(a, a, ids, inverse=True)
        assert np.array_equal(a.download(), x)
        L.hm_graph_destroy(g)
    finally:
        ctx.close()
