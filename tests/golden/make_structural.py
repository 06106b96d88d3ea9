#!/usr/bin/env python3
"""Generate tests/golden/structural.json from the compiled, unmodified reference simulator.

Run in the build container (needs /root/reference and `make -C oracle ref`):
    python tests/golden/make_structural.py
The fixture is DATA: per (cfg, op, L, l, alpha) the `Malloc <name> from <a> to <b>` lines, the total
instruction count (executed + remaining of the first progress block, = Driver::getTotalIns(),
include/Driver.h:370), the simulated cycle count (src/Operation.cpp:1095) and the stat block keys.
Also the usage / unknown-op messages and exit codes of bench_test/bench_micro24.cpp:5-52.
"""
import json, os, re, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "ForgeHomulator.run")
REF_CFG = "/root/reference/config"

POINTS = [  # (cfg, op, L, l, alpha, run_to_completion)
    ("config_4_N15.cfg", "hmult", 4, 2, 2, True), ("config_4_N15.cfg", "hrotate", 4, 2, 2, True),
    ("config_4_N15.cfg", "hmult", 4, 3, 2, True), ("config_4_N15.cfg", "hrotate", 4, 3, 2, True),
    ("config_4_N15.cfg", "hmult", 6, 4, 2, True), ("config_4_N15.cfg", "hrotate", 6, 4, 2, True),
    ("config_4_N15.cfg", "hmult", 8, 8, 8, True), ("config_4_N15.cfg", "hrotate", 8, 8, 8, True),
    ("config_4_N15.cfg", "hmult", 16, 10, 4, True), ("config_4_N15.cfg", "hrotate", 16, 10, 4, True),
    ("config_4_N15.cfg", "hadd", 16, 10, 4, True), ("config_4_N15.cfg", "padd", 16, 10, 4, True),
    ("config_4_N15.cfg", "pmult", 16, 10, 4, True),
    # N=2^16: construction phase + first progress block only (a full run takes hours)
    ("config_4.cfg", "hmult", 45, 35, 15, False), ("config_4.cfg", "hrotate", 45, 35, 15, False),
]


def run_point(cfg, op, L, l, a, complete):
    cmd = ["stdbuf", "-oL", REF_BIN, os.path.join(REF_CFG, cfg), op, str(L), str(l), str(a)]
    t0 = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    lines = []
    for line in p.stdout:
        lines.append(line.rstrip("\n"))
        if not complete and line.startswith("Remaining"):
            p.kill()
            break
    p.wait()
    out = "\n".join(lines)
    rec = {"cfg": cfg, "op": op, "L": L, "l": l, "alpha": a, "complete": complete}
    rec["malloc"] = [ln for ln in lines if ln.startswith("Malloc ")]
    ex = re.search(r"We have executed (\d+) instructions!\nRemaining (\d+) instructions!", out)
    rec["total_instructions"] = int(ex.group(1)) + int(ex.group(2)) if ex else None
    cyc = re.search(r"FHE-Sim Total simulated\t(\d+) cycles!", out)
    rec["cycles"] = int(cyc.group(1)) if cyc else None
    wl = re.search(r"Welcome! Start simulating (\w+)!", out)
    rec["welcome"] = wl.group(1) if wl else None
    if complete:
        i = lines.index("Start outPut statistic informations:")
        rec["stat_keys"] = [ln.split(" :\t")[0] for ln in lines[i + 2:] if " :\t" in ln]
        rec["stats"] = {ln.split(" :\t")[0]: int(ln.split(" :\t")[1]) for ln in lines[i + 2:] if " :\t" in ln}
    # config echo (Config.cpp:39-51)
    j = lines.index("Configuration details are as follow:")
    k = [n for n, ln in enumerate(lines) if ln.startswith("*****")]
    rec["config_echo"] = lines[j:k[1] + 1]
    rec["wall_s"] = round(time.time() - t0, 1)
    return rec


def main():
    if not os.path.exists(REF_BIN):
        sys.exit("build the reference first: make -C oracle ref")
    out = {"generated_by": "tests/golden/make_structural.py", "reference_build": "g++ -std=c++17 -O2 (oracle/Makefile: ref)",
           "points": []}
    for pt in POINTS:
        rec = run_point(*pt)
        print(pt, "->", rec["total_instructions"], rec["cycles"], f"{rec['wall_s']}s", flush=True)
        out["points"].append(rec)
    # CLI contract
    r = subprocess.run([REF_BIN], capture_output=True, text=True)
    out["usage"] = {"stderr": r.stderr, "stdout": r.stdout, "rc": r.returncode}
    r = subprocess.run([REF_BIN, os.path.join(REF_CFG, "config_4_N15.cfg"), "bogus", "4", "2", "2"], capture_output=True, text=True)
    out["unknown_op"] = {"stdout_tail": r.stdout.splitlines()[-1], "rc": r.returncode}
    with open(os.path.join(os.path.dirname(__file__), "structural.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
