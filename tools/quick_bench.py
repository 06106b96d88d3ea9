"""Quick per-kernel timing on the GPU (development tool): NTT sweep, EWE, BConv at N=2^16."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip

def timeit(ctx, fn, iters=20, warm=3):
    for _ in range(warm): fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(iters): fn()
    return ctx.timer_stop() / iters * 1e-3  # us

def main():
    logN, L, K = 16, 45, 15
    t0 = time.time()
    ctx = hip.Context(logN, L, K)
    print(f"context create {time.time()-t0:.2f}s")
    N = 1 << logN
    LP = N * 8
    for n in (1, 8, 35, 50, 115):
        ids = [i % (L + K) for i in range(n)]
        a = ctx.alloc(n); b = ctx.alloc(n)
        ctx.fill_uniform(a, ids, 1)
        us = timeit(ctx, lambda: ctx.ntt(a, b, ids))
        usi = timeit(ctx, lambda: ctx.ntt(a, b, ids, inverse=True))
        print(f"NTT  n={n:4d}: fwd {us:8.1f} us ({us/n:6.2f} us/limb, alg {2*LP*n/us*1e-6:7.3f} TB/s)   inv {usi:8.1f} us ({usi/n:6.2f} us/limb)")
        a.free(); b.free()
    n = 35
    ids = list(range(n))
    a = ctx.alloc(n); b = ctx.alloc(n); c = ctx.alloc(n); d = ctx.alloc(n); o = ctx.alloc(n)
    for x, s in ((a, 1), (b, 2), (c, 3), (d, 4)): ctx.fill_uniform(x, ids, s)
    for op, name, nops in ((0, "MUL", 3), (1, "MAC2", 5), (3, "ADD", 3), (6, "SUB_SCALE", 3)):
        us = timeit(ctx, lambda: ctx.ewe(op, o, ids, a=a, b=b, c=c, d=d, k=[5] * n))
        print(f"EWE {name:10s} n={n}: {us:8.1f} us  ({nops*LP*n/us*1e-6:6.3f} TB/s)")
    # ModUp digit: 15 -> 35 ; ModDown 15 -> 35
    in_ids = list(range(15)); out_ids = list(range(15, 35)) + [L + i for i in range(K)]
    src = ctx.alloc(15); dst = ctx.alloc(len(out_ids))
    ctx.fill_uniform(src, in_ids, 9)
    us = timeit(ctx, lambda: ctx.bconv(src, in_ids, dst, out_ids))
    print(f"BCONV 15->{len(out_ids)}: {us:8.1f} us  ({15*len(out_ids)*N/us*1e-3:6.2f} GMAC/s, {(15+len(out_ids))*LP/us*1e-6:6.3f} TB/s)")
    us = timeit(ctx, lambda: ctx.automorph(a, o, n, 5))
    print(f"AUTO n={n}: {us:8.1f} us ({2*LP*n/us*1e-6:6.3f} TB/s)")

if __name__ == "__main__":
    main()
