"""Per-workgroup timeline of the one-launch transform (k_ntt_fused8) for the contract's 50-limb sweep, from a development build with
-DHM_FUSED_TRACE (tools/ablate.sh trace "-DHM_FUSED_TRACE"; run with HOMULATOR_HIP_LIB=ab_builds/trace/libhomulator_hip.so).
Stamps (100 MHz wall clock, 10 ns): 0 start, 1 first pass's stores issued, 2 stores in L2 + arrival published, 3 second pass's twiddle requests
issued (starts to wait), 4 every sibling has arrived, 5 second pass stored.  Prints where the workgroups' time goes and how they line up."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from homulator_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
inplace = len(sys.argv) > 2 and sys.argv[2] == "inplace"
ctx = hip.Context(16, 45, 15)
ids = ctx.ext_ids(35)[:n]
sets = 6
bufs = [(ctx.alloc(n), ctx.alloc(n)) for _ in range(sets)]
for i, (a, _) in enumerate(bufs):
    ctx.fill_uniform(a, ids, 1 + i)
nwg = ((n + 7) // 8 * 8) * 16
trace = ctx.from_host(np.zeros(((nwg * 8 + (1 << 16) - 1) >> 16, 1 << 16), dtype=np.uint64))     # nwg x 8 words, in limb-polys of 2^16 words
for i in range(24):
    a, b = bufs[i % sets]
    ctx.ntt(a, a if inplace else b, ids)
ctx.sync()
rows = []
for rep in range(6):
    ctx.set_option("ntt_fused_trace", trace.ptr)
    a, b = bufs[rep % sets]
    ctx.ntt(a, a if inplace else b, ids)
    ctx.sync()
    ctx.set_option("ntt_fused_trace", 0)
    t = trace.download().reshape(-1)[: nwg * 8].reshape(nwg, 8).astype(np.int64)
    t = t[t[:, 0] != 0]
    rows.append(t)
t = rows[-1]
t0 = t[:, 0].min()
us = lambda x: x * 0.01
print(f"{len(t)} workgroups of {n} limb-polys ({'in place' if inplace else 'out of place'}); last repetition; times in us")
print(f"  kernel span (first start .. last end):            {us(t[:, 5].max() - t0):7.2f}")
print(f"  start skew (last workgroup's start - first):      {us(t[:, 0].max() - t0):7.2f}   median start {us(np.median(t[:, 0]) - t0):.2f}")
for name, a, b in (("first pass (start -> stores issued)", 0, 1), ("stores reach L2 + arrive", 1, 2), ("twiddle requests before the wait", 2, 3),
                   ("WAIT for the siblings", 3, 4), ("second pass (after the wait -> stored)", 4, 5), ("whole workgroup", 0, 5)):
    d = us(t[:, b] - t[:, a])
    print(f"  {name:48s} median {np.median(d):6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f}")
ent = (t[:, 7] >> 32)
arr = {}
for e in np.unique(ent):
    m = ent == e
    arr[e] = (us(t[m, 2].max() - t[m, 2].min()), us(t[m, 0].max() - t[m, 0].min()))
sk = np.array(list(arr.values()))
print(f"  per limb-poly: spread of its 16 arrivals median {np.median(sk[:, 0]):.2f} max {sk[:, 0].max():.2f}; spread of its 16 starts median {np.median(sk[:, 1]):.2f} max {sk[:, 1].max():.2f}")
xcc = (t[:, 6] >> 32) & 7
cu = ((t[:, 6] & 0xFFFFFFFF) >> 8) & 0xF
se = ((t[:, 6] & 0xFFFFFFFF) >> 13) & 0x7
occ = {}
for x, s_, c in zip(xcc, se, cu):
    occ[(x, s_, c)] = occ.get((x, s_, c), 0) + 1
v = np.array(list(occ.values()))
print(f"  placement: {len(occ)} (xcc, se, cu) slots used; workgroups per CU min {v.min()} median {int(np.median(v))} max {v.max()}")
for r, tt in enumerate(rows):
    print(f"  rep {r}: span {us(tt[:, 5].max() - tt[:, 0].min()):6.2f}  wait median {np.median(us(tt[:, 4] - tt[:, 3])):5.2f}  first pass median {np.median(us(tt[:, 1] - tt[:, 0])):5.2f}  second {np.median(us(tt[:, 5] - tt[:, 4])):5.2f}")
# where the slow first passes are: by XCD, by the workgroups sharing the CU, by tile (column slice) and by dispatch order
fp = us(t[:, 1] - t[:, 0])
tile = t[:, 7] & 0xFFFFFFFF
print("  first pass by XCD (workgroups, median, p90, last end):  " + "  ".join(
    f"x{x}: {int((xcc == x).sum())} {np.median(fp[xcc == x]):.1f} {np.percentile(fp[xcc == x], 90):.1f} {us(t[xcc == x, 5].max() - t0):.1f}" for x in range(8) if (xcc == x).any()))
per_cu = np.array([occ[(x, s_, c)] for x, s_, c in zip(xcc, se, cu)])
print("  first pass by workgroups on the CU:  " + "  ".join(f"{k}: n={int((per_cu == k).sum())} median {np.median(fp[per_cu == k]):.1f} p90 {np.percentile(fp[per_cu == k], 90):.1f}"
      for k in np.unique(per_cu)))
print("  first pass by tile (median):  " + " ".join(f"{np.median(fp[tile == k]):.1f}" for k in range(16)))
order = np.argsort(t[:, 0])
q = np.array_split(order, 8)
print("  first pass by start order (eighths, median / p90):  " + "  ".join(f"{np.median(fp[i]):.1f}/{np.percentile(fp[i], 90):.1f}" for i in q))
ents = np.unique(ent)
lim = np.array([np.median(fp[ent == e]) for e in ents])
print("  first pass by limb-poly (median of its 16 workgroups), in entry order:  " + " ".join(f"{x:.0f}" for x in lim))
if os.environ.get("FUSED_TRACE_DUMP"):
    np.save(os.environ["FUSED_TRACE_DUMP"], t)
