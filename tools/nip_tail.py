"""How much does the last, partly filled round of workgroups cost the fused transform x key kernel at latency-mode launch sizes?
hm_ntt_inner_product (beta = 3 digits, 2 keys, N = 2^16) over E output limbs, E swept around the 768-workgroup capacity of the chip at
3 waves per SIMD (E = 48).  Under rocprofv3 --kernel-trace the per-launch k_ntt_row_ip durations come out in sweep order.
usage: python3 tools/nip_tail.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip
R = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Lq, K, beta = 45, 15, 3
ctx = hip.Context(16, Lq, K)
EMAX = 60
ids = list(range(EMAX))
x = ctx.alloc(beta * EMAX); hand = ctx.alloc(beta * EMAX); evk = ctx.alloc(2 * beta * EMAX); out = ctx.alloc(2 * EMAX)
ctx.fill_uniform(x, ids * beta, 3); ctx.fill_uniform(evk, ids * (2 * beta), 5)
def args(E):
    xl, fl, hl, yl, ol = [], [], [], [], []
    for t in range(E):
        for j in range(beta):
            xl.append(j * EMAX + t); hl.append(j * EMAX + t); fl.append(1)
        for k in range(2):
            for j in range(beta): yl.append((j * 2 + k) * EMAX + t)
            ol.append(k * EMAX + t)
    return xl, fl, hl, yl, ol, ids[:E]
for E in (32, 40, 44, 46, 47, 48, 49, 50, 52, 56, 60):
    xl, fl, hl, yl, ol, mods = args(E)
    call = lambda: ctx.ntt_inner_product(x, xl, fl, hand, hl, evk, yl, out, ol, mods, beta, 2)
    call(); ctx.sync(); ctx.timer_start()
    for _ in range(R): call()
    us = ctx.timer_stop() / R * 1e-3
    print(f"E {E:3d}  workgroups {E * 16:5d}  both kernels {us:7.2f} us  {us / E:6.3f} us per output limb", flush=True)
