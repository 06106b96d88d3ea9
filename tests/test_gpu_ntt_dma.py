"""GPU parity of the persistent double-buffered passes (k_ntt_*_dma: a workgroup walks a run of tiles, the next tile arrives in LDS by
LDS-DMA while the current one is transformed) against the one-tile-per-workgroup passes and the CPU oracle, bit for bit.  What they can get
wrong: the tile image built through the DMA's source addresses, the counted vmcnt wait at the top of an iteration, runs of 1 .. many tiles
per workgroup (grids of 8 .. more workgroups than tiles), launches with padding entries, the fused prologue / epilogue forms, every ring
size, in-place transforms, both geometries."""
import numpy as np
import pytest

from oracle.homoracle import Oracle

pytestmark = pytest.mark.gpu


def _ctx(logN, L, K):
    from homulator_amd import hip
    return hip.Context(logN, L, K), Oracle(logN, L, K)


@pytest.mark.parametrize("logN", [13, 14, 15, 16, 17])
@pytest.mark.parametrize("geo", [1, 2, 3])
def test_dma_equals_one_tile_kernels_and_oracle(logN, geo):
    ctx, o = _ctx(logN, 4, 2)
    try:
        ids = [0, 1, 2, 3, 4, 5, 0, 5, 3]
        x = o.fill_uniform(ids, 77)
        x[0, :] = o.moduli[ids[0]] - 1          # worst case of the lazy ranges
        x[1, :3] = [0, 1, o.moduli[ids[1]] - 1]
        d, a, b = ctx.from_host(x), ctx.alloc(len(ids)), ctx.alloc(len(ids))
        exp = o.ntt(ids, x)
        for wgs in (0, 8, 16):
            ctx.set_option("ntt_dma_wgs", wgs)
            for inverse in (False, True):
                src = x if not inverse else exp
                dsrc = ctx.from_host(src)
                ctx.set_option("ntt_dma", geo)
                ctx.ntt(dsrc, a, ids, inverse=inverse)
                ctx.set_option("ntt_dma", 0)
                ctx.ntt(dsrc, b, ids, inverse=inverse)
                A, B = a.download(), b.download()
                assert np.array_equal(A, B), (wgs, inverse)
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
                ctx.set_option("ntt_dma", q)
                ctx.ntt_mix_sub_scale(d, dmn, a, ids, k, addend=dad, addend_k=ak, mix=dmx, mix_k=mk)
                assert np.array_equal(a.download(), exp3)
                ctx.ntt_sub_scale(d, dmn, a, ids, k)
                assert np.array_equal(a.download(), o.ewe(6, ids, mn, None, o.ntt(ids, x), k=k))
            for t in (dmn, dad, dmx):
                t.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("geo", [1, 2, 3])
def test_runs_of_tiles_many_limbs_repeated_and_in_place(geo):
    """700 limb-polys of N = 2^16 = 11 200 tiles per pass on 512 (and 64, and 2048) persistent workgroups: runs of 22 (175, 5) tiles;
    three launches back to back; then the inverse in place brings the input back; 50 and 13 limb-polys: runs of one or two tiles"""
    ctx, o = _ctx(16, 6, 3)
    try:
        for n, grids in ((700, (0, 64, 2048)), (50, (0, 8, 200)), (13, (0, 24))):
            ids = [(i * 7) % 9 for i in range(n)]
            src = ctx.alloc(n)
            ctx.fill_uniform(src, ids, 4242)
            x = src.download()
            out, ref = ctx.alloc(n), ctx.alloc(n)
            ctx.set_option("ntt_dma", 0)
            ctx.ntt(src, ref, ids)
            R = ref.download()
            ctx.set_option("ntt_dma", geo)
            for wgs in grids:
                ctx.set_option("ntt_dma_wgs", wgs)
                ctx.fill_uniform(out, ids, 5)
                for _ in range(3):
                    ctx.ntt(src, out, ids)
                assert np.array_equal(out.download(), R), (n, wgs)
                ctx.ntt(out, out, ids, inverse=True)
                assert np.array_equal(out.download(), x), (n, wgs)
            pick = [0, 1, n // 2, n - 1]
            assert np.array_equal(R[pick], o.ntt([ids[i] for i in pick], x[pick]))
            for b in (src, out, ref):
                b.free()
    finally:
        ctx.close()


def test_two_contexts_share_the_chip():
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
        c1.set_option("ntt_dma", 1)
        c2.set_option("ntt_dma", 2)
        for rep in range(4):
            for c, ids, s, f, r in jobs:
                c.ntt(s, f, ids)
        for c in (c1, c2):
            c.sync()
            c.set_option("ntt_dma", 0)
        for c, ids, s, f, r in jobs:
            c.ntt(s, r, ids)
        for c, ids, s, f, r in jobs:
            assert np.array_equal(f.download(), r.download())
    finally:
        c1.close(); c2.close()
