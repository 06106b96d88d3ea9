"""ModUp of config_4.cfg hmult 45 35 15 at batch B through the C ABI: conversion inside the transform's first pass (k_bconv_col +
k_ntt_row_ip: hm_ntt_ip_desc.conv) against a separate conversion (k_bconv, then k_ntt_col + k_ntt_row_ip).  Checks hand-off and outputs
bit for bit, then times both forms (or runs them a few times for rocprofv3).
usage: python3 tools/bcol_ab.py [batch] [rounds]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from homulator_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
Lq, ell, K = 45, 35, 15
ctx = hip.Context(16, Lq, K)
ext = ctx.ext_ids(ell); E = len(ext); beta = 3
src = ctx.alloc(ell * B)                    # scaled digits (ModUpDecompOut), per op
bc = ctx.alloc(beta * E * B)                # BConvOut
hand_a, hand_b = ctx.alloc(beta * E * B), ctx.alloc(beta * E * B)
evk = ctx.alloc(2 * beta * E); outb = ctx.alloc(2 * E * B); outb2 = ctx.alloc(2 * E * B)
ctx.fill_uniform(src, [i % ell for i in range(ell * B)], 7)
ctx.fill_uniform(evk, (ext * (2 * beta)), 9)
probs = []
for b in range(B):
    for j in range(beta):
        lo, hi = j * K, min(ell, (j + 1) * K)
        outs = [t for t in range(E) if not (lo <= t < hi)]
        probs.append((b, j, list(range(lo, hi)), outs))
def descs(dst):
    keep, arr = [], (hip.hm_bconv_desc * len(probs))()
    for d, (b, j, ins, outs) in zip(arr, probs):
        a = [hip._u32([b * ell + i for i in ins]), hip._u32(ins), hip._u32([(b * beta + j) * E + t for t in outs]), hip._u32([ext[t] for t in outs])]
        keep.append(a)
        d.in_, d.in_limbs, d.in_ids, d.n_in = src.ptr, a[0][1], a[1][1], len(ins)
        d.out, d.out_limbs, d.out_ids, d.n_out, d.log_len = dst.ptr, a[2][1], a[3][1], len(outs), 0
    return keep, arr
ka, da = descs(bc)
conv_b = [(src, [b * ell + i for i in ins], ins, [(b * beta + j) * E + t for t in outs], [ext[t] for t in outs]) for (b, j, ins, outs) in probs]
# path A: conversion, then the fused transform x key product whose first kernel is the COL pass into hand_a
xl, flags, hl, yl, ol, mods = [], [], [], [], [], []
for b in range(B):
    for t in range(E):
        for j in range(beta):
            own = j * K <= t < min(ell, (j + 1) * K)
            xl.append((b * beta + j) * E + t); hl.append((b * beta + j) * E + t); flags.append(0 if own else 1)
        for k in range(2):
            for j in range(beta): yl.append((j * 2 + k) * E + t)
            ol.append((b * 2 + k) * E + t)
        mods.append(ext[t])
def path_a():
    ctx._ck(ctx.L.hm_bconv_batch(ctx.h, da, len(probs)))
    ctx.ntt_inner_product(bc, xl, flags, hand_a, hl, evk, yl, outb, ol, mods, beta, 2)
def path_b():
    ctx.ntt_inner_product(bc, xl, flags, hand_b, hl, evk, yl, outb2, ol, mods, beta, 2, conv=conv_b)
path_a(); path_b(); ctx.sync()
if B <= 2:
    A, Bh = hand_a.download(), hand_b.download()
    sel = [(b * beta + j) * E + t for (b, j, ins, outs) in probs for t in outs]
    print("hand-off identical:", np.array_equal(A[sel], Bh[sel]), len(sel), "limb-polys; outputs identical:", np.array_equal(outb.download(), outb2.download()))
for _ in range(R):
    path_a()
for _ in range(R):
    path_b()
ctx.sync()
def t(fn, n=10):
    ctx.sync(); ctx.timer_start()
    for _ in range(n): fn()
    return ctx.timer_stop() / n * 1e-3
print(f"batch {B}: separate conversion (k_bconv, k_ntt_col, k_ntt_row_ip) {t(path_a)/B:.1f} us/op; conversion inside the first pass (k_bconv_col, k_ntt_row_ip) {t(path_b)/B:.1f} us/op; k_bconv alone {t(lambda: ctx._ck(ctx.L.hm_bconv_batch(ctx.h, da, len(probs))))/B:.1f} us/op")
