"""how well does a VALU-bound kernel stream overlap a memory-bound one on this chip?  Two contexts (own stream each): A = base conversions
(k_bconv<15>, batch-10 ModUp shape), B = forward transforms of 1150 limb-polys (k_ntt_col + k_ntt_row), M = tensor products; each alone,
then pairs enqueued alternately.  Prints wall time per round and the overlap achieved: 1 = the pair takes max(a, b), 0 = a + b."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from homulator_amd import hip
Lq, ell, K = 45, 35, 15
def make_bconv():
    ctx = hip.Context(16, Lq, K)
    B = 10; ext = ctx.ext_ids(ell); E = len(ext)
    src, dst = ctx.alloc(ell * B), ctx.alloc(3 * E * B)
    ctx.fill_uniform(src, [i % ell for i in range(ell * B)], 3)
    probs = []
    for b in range(B):
        for j in range(3):
            lo, hi = j * K, min(ell, (j + 1) * K)
            outs = [t for t in range(E) if not lo <= t < hi]
            probs.append((src, [b * ell + i for i in range(lo, hi)], list(range(lo, hi)), dst, [(b * 3 + j) * E + t for t in outs], [ext[t] for t in outs]))
    import ctypes as C   # descriptors built once: building them costs more host time than the kernel takes
    keep, descs = [], (hip.hm_bconv_desc * len(probs))()
    for d, (s_, il, ii, d_, ol, oi) in zip(descs, probs):
        arrs = [hip._u32(il), hip._u32(ii), hip._u32(ol), hip._u32(oi)]
        keep.append(arrs)
        d.in_, d.in_limbs, d.in_ids, d.n_in = s_.ptr, arrs[0][1], arrs[1][1], len(ii)
        d.out, d.out_limbs, d.out_ids, d.n_out, d.log_len = d_.ptr, arrs[2][1], arrs[3][1], len(oi), 0
    ctx._keep = (keep, descs)
    return ctx, (lambda: ctx._ck(ctx.L.hm_bconv_batch(ctx.h, descs, len(probs))))
def make_ntt():
    ctx = hip.Context(16, Lq, K)
    n = 1150; ids = [(i // 2) % 50 for i in range(n)]
    a, b = ctx.alloc(n), ctx.alloc(n)
    ctx.fill_uniform(a, ids, 5)
    return ctx, (lambda: ctx.ntt(a, b, ids))
def make_tensor():
    ctx = hip.Context(16, Lq, K)
    n = 350; ids = [i % ell for i in range(n)]
    bufs = [ctx.alloc(n) for _ in range(7)]
    for i in range(4): ctx.fill_uniform(bufs[i], ids, 7 + i)
    return ctx, (lambda: ctx.tensor(*bufs, ids))
def wall(jobs, rounds=30):
    for c, f in jobs:
        for _ in range(5): f()
    for c, f in jobs: c.sync()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(rounds):
        for c, f in jobs: f()
    for c, f in jobs: c.sync()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / rounds * 1e6
V, N1, M = make_bconv(), make_ntt(), make_tensor()
V2, N2 = make_bconv(), make_ntt()
tv, tn, tm = wall([V]), wall([N1]), wall([M])
print(f"alone: conversion {tv:.0f} us, transforms {tn:.0f} us, tensor {tm:.0f} us per round")
for name, jobs, a, b in (("conversion || transforms", [V, N1], tv, tn), ("conversion || tensor", [V, M], tv, tm), ("transforms || tensor", [N1, M], tn, tm),
                         ("conversion || conversion", [V, V2], tv, tv), ("transforms || transforms", [N1, N2], tn, tn)):
    t = wall(jobs)
    print(f"{name:28s} {t:7.0f} us  (sum {a + b:.0f}, max {max(a, b):.0f})  overlap {(a + b - t) / min(a, b):.2f}")
