"""GPU parity of the persistent two-pass transform (k_ntt_queue: workgroups pull (pass, tile) items from the queue of the XCD they run
on; the hand-off between the passes stays in that XCD's L2) against the two-kernel transform and the CPU oracle, bit for bit.  Covers
what the queues can get wrong: every grid size from fewer workgroups than XCDs' worth of items to more workgroups than items, every
look-ahead and group size (incl. groups that straddle the end of the list), more limb-polys than one launch table holds, repeated
launches (the host zeroes the queue words in front of every launch), in-place transforms, two contexts sharing the chip and the
fused prologue / epilogue forms (N = 2^16, the 8-coefficient geometry: the only form left in round 5, on XCD-local atomics)."""
import numpy as np
import pytest

from oracle.homoracle import Oracle

pytestmark = pytest.mark.gpu


def _ctx(logN, L, K):
    from homulator_amd import hip
    return hip.Context(logN, L, K), Oracle(logN, L, K)


@pytest.mark.parametrize("logN", [16])
@pytest.mark.parametrize("geo", [1])
def test_queue_equals_two_kernel_and_oracle(logN, geo):
    ctx, o = _ctx(logN, 4, 2)
    try:
        ids = [0, 1, 2, 3, 4, 5, 0, 5, 3]
        x = o.fill_uniform(ids, 77)
        x[0, :] = o.moduli[ids[0]] - 1          # worst case of the lazy ranges
        x[1, :3] = [0, 1, o.moduli[ids[1]] - 1]
        d, a, b = ctx.from_host(x), ctx.alloc(len(ids)), ctx.alloc(len(ids))
        exp = o.ntt(ids, x)
        for inverse in (False, True):
            src = x if not inverse else exp
            dsrc = ctx.from_host(src)
            ctx.set_option("ntt_queue", geo)
            ctx.ntt(dsrc, a, ids, inverse=inverse)
            ctx.set_option("ntt_queue", 0)
            ctx.ntt(dsrc, b, ids, inverse=inverse)
            A, B = a.download(), b.download()
            assert np.array_equal(A, B)
            assert np.array_equal(A, o.ntt(ids, src, inverse=inverse))
            dsrc.free()
        mn, ad, mx = (o.fill_uniform(ids, s) for s in (124, 125, 126))
        k = [o.moduli[m] - 2 - r for r, m in enumerate(ids)]
        mk = [(kk * 3 + 1) % o.moduli[m] for kk, m in zip(k, ids)]
        ak = [(kk * 5 + 1) % o.moduli[m] for kk, m in zip(k, ids)]
        dmn, dad, dmx = ctx.from_host(mn), ctx.from_host(ad), ctx.from_host(mx)
        xin = o.ewe(3, ids, x, None, o.ewe(5, ids, mx, k=mk))
        exp3 = o.ewe(3, ids, o.ewe(6, ids, mn, None, o.ntt(ids, xin), k=k), None, o.ewe(5, ids, ad, k=ak))
        for q in (geo, 0):
            ctx.set_option("ntt_queue", q)
            ctx.ntt_mix_sub_scale(d, dmn, a, ids, k, addend=dad, addend_k=ak, mix=dmx, mix_k=mk)
            assert np.array_equal(a.download(), exp3)
            ctx.ntt_sub_scale(d, dmn, a, ids, k)
            assert np.array_equal(a.download(), o.ewe(6, ids, mn, None, o.ntt(ids, x), k=k))
    finally:
        ctx.close()


@pytest.mark.parametrize("geo", [1])
def test_grid_sizes_lookahead_and_groups(geo):
    """50 limb-polys (the sweep of the extended basis: does not divide by the 8 queues) and 13, through grids of 8 .. 2048 workgroups,
    look-aheads 1 .. 4 and groups of 1, 2, 3 limb-polys (3 does not divide 50: the last group is padded)"""
    ctx, o = _ctx(16, 6, 3)
    try:
        for n in (50, 13):
            ids = [(i * 7 + 1) % 9 for i in range(n)]
            src, out, ref = ctx.alloc(n), ctx.alloc(n), ctx.alloc(n)
            ctx.fill_uniform(src, ids, 99 + n)
            ctx.set_option("ntt_queue", 0)
            ctx.ntt(src, ref, ids)
            R = ref.download()
            ctx.set_option("ntt_queue", geo)
            for wgs, la, gc in ((8, 1, 1), (24, 2, 1), (64, 1, 2), (512, 2, 2), (512, 3, 3), (1024, 4, 1), (2048, 2, 2), (0, 2, 0)):
                ctx.set_option("ntt_queue_wgs", wgs)
                ctx.set_option("ntt_queue_lookahead", la)
                ctx.set_option("ntt_queue_group", gc)
                ctx.fill_uniform(out, ids, 5)     # a stale result must not pass
                ctx.ntt(src, out, ids)
                assert np.array_equal(out.download(), R), (n, wgs, la, gc)
            for b in (src, out, ref):
                b.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("geo", [1])
def test_more_limbs_than_a_launch_table_repeated_and_in_place(geo):
    """700 limb-polys of N = 2^16 (two launches of the 448-entry table); three runs back to back reuse the queue words; then the inverse
    in place (the hand-off lands on the input's own lines) brings the input back"""
    ctx, o = _ctx(16, 6, 3)
    try:
        n = 700
        ids = [(i * 7) % 9 for i in range(n)]
        src = ctx.alloc(n)
        ctx.fill_uniform(src, ids, 4242)
        x = src.download()
        out, ref = ctx.alloc(n), ctx.alloc(n)
        ctx.set_option("ntt_queue", 0)
        ctx.ntt(src, ref, ids)
        R = ref.download()
        ctx.set_option("ntt_queue", geo)
        for _ in range(3):
            ctx.ntt(src, out, ids)
        assert np.array_equal(out.download(), R)
        ctx.ntt(out, out, ids, inverse=True)
        assert np.array_equal(out.download(), x)
        ctx.ntt(out, out, ids)                      # forward in place
        assert np.array_equal(out.download(), R)
        pick = [0, 1, 350, 699]
        assert np.array_equal(R[pick], o.ntt([ids[i] for i in pick], x[pick]))
    finally:
        ctx.close()


def test_two_contexts_share_the_chip():
    """uneven load: two contexts (own stream and queue words each) enqueue persistent transforms of different sizes alternately"""
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
        c1.set_option("ntt_queue", 1)
        c2.set_option("ntt_queue", 1)
        for rep in range(4):
            for c, ids, s, f, r in jobs:
                c.ntt(s, f, ids)
        for c in (c1, c2):
            c.sync()
            c.set_option("ntt_queue", 0)
        for c, ids, s, f, r in jobs:
            c.ntt(s, r, ids)
        for c, ids, s, f, r in jobs:
            assert np.array_equal(f.download(), r.download())
    finally:
        c1.close(); c2.close()
