"""clock and package power while ONE kernel family runs in a loop (2.5 s each), sampled with rocm-smi: which stages of the op hold the MI355X at its power limit
    python3 tools/r06_power_by_kernel.py"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import hip

LOGN, L, K, ELL = 16, 45, 15, 35
ctx = hip.Context(LOGN, L, K)
ext = ctx.ext_ids(ELL)
ids = ext * 10                                   # 500 limb-polys: the batched launches' size
a, b, c, d, o0, o1, o2 = (ctx.alloc(len(ids)) for _ in range(7))
for i, x in enumerate((a, b, c, d)):
    ctx.fill_uniform(x, ids, 5 + i)
ps, qs = [L + i for i in range(K)], list(range(ELL))
pin, qout = ctx.alloc(K * 20), ctx.alloc(ELL * 20)
ctx.fill_uniform(pin, ps * 20, 9)
probs = [(pin, [k * K + i for i in range(K)], ps, qout, [k * ELL + t for t in range(ELL)], qs) for k in range(20)]


def sample():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    pw = re.search(r"Power \(W\): ([0-9.]+)", out)
    return (int(sclk.group(1)) if sclk else 0, float(pw.group(1)) if pw else 0.0)


def run(name, fn, seconds=2.5):
    stop = [False]
    n = [0]

    def loop():
        while not stop[0]:
            for _ in range(8):
                fn()
            ctx.sync()
            n[0] += 8
    t = threading.Thread(target=loop)
    t0 = time.perf_counter()
    t.start()
    time.sleep(0.9)
    s = [sample() for _ in range(3)]
    while time.perf_counter() - t0 < seconds:
        time.sleep(0.05)
    stop[0] = True
    t.join()
    dt = time.perf_counter() - t0
    print(f"{name:58s} {dt / n[0] * 1e6:8.1f} us per launch   sclk {min(x[0] for x in s)}-{max(x[0] for x in s)} MHz   power {min(x[1] for x in s):.0f}-{max(x[1] for x in s):.0f} W", flush=True)


print("idle:", sample())
run("k_tensor, 500 limb-polys (memory-bound)", lambda: ctx.tensor(a, b, c, d, o0, o1, o2, ids))
run("forward transform, 500 limb-polys (k_ntt_col + k_ntt_row)", lambda: ctx.ntt(a, o0, ids))
run("inverse transform, 500 limb-polys", lambda: ctx.ntt(a, o0, ids, inverse=True))
run("k_bconv<15>, 20 conversions 15 -> 35 (multiply-adds)", lambda: ctx.bconv_batch(probs))
run("element-wise add, 500 limb-polys (pure streaming)", lambda: ctx.ewe(3, o0, ids, a=a, c=b))
ctx.close()
from homulator_amd import host
op = host.Op("config_4.cfg", "hmult", 45, 35, 15, overrides={"batch": 10, "graph": 1})
op.execute(2)
stop = [False]
def whole():
    while not stop[0]:
        op.enqueue(4); op.sync()
t = threading.Thread(target=whole); t.start(); time.sleep(1.0)
s = [sample() for _ in range(3)]
stop[0] = True; t.join()
print(f"{'whole hmult, one instance x batch 10 (graph)':58s}            sclk {min(x[0] for x in s)}-{max(x[0] for x in s)} MHz   power {min(x[1] for x in s):.0f}-{max(x[1] for x in s):.0f} W")
op.close()
