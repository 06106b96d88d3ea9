#!/usr/bin/env python3
"""Parameter-set sweeps of the reference's script/ directory on the MI355X backend.

The reference ships 25 shell scripts (script/para{A,B,C,D}/micro24_<set>_<op>.sh + run.sh, script/README.md:17-22) that
launch one simulator process per level in the background and append to
    outLogs/para<set>/<cluster>/<op>/<maxLevel>_<alpha>/<op>_<maxLevel>_<alpha>_<level>.log
This runner produces the same directory layout and the same per-run stdout (it runs host/Homulator.run with the same
argv), sequentially — a level takes about a second on the GPU instead of minutes to hours of simulation.

    python script/sweep.py --set B                      # all five ops, every level, 4 clusters
    python script/sweep.py --set A --ops hmult --levels 28,14,2
    HOMULATOR_BACKEND=count python script/sweep.py --set C   # no GPU: plans and instruction totals only
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = {  # script/README.md:17-22
    "A": dict(cfg="config_4_N15.cfg", L=28, alpha=28),
    "B": dict(cfg="config_4.cfg", L=45, alpha=15),
    "C": dict(cfg="config_4.cfg", L=24, alpha=6),
    "D": dict(cfg="config_4.cfg", L=26, alpha=9),
    "motivation": dict(cfg="config_4.cfg", L=28, alpha=28),   # script/motivation/micro24_motivation.sh
}
OPS = ["hadd", "hmult", "hrotate", "padd", "pmult"]   # order of script/para*/run.sh


def min_level(op):
    return 2 if op == "hmult" else 1   # hmult rescales: it needs two limbs (the reference's hmult scripts also stop at 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", required=True, choices=sorted(SETS))
    ap.add_argument("--ops", default=",".join(OPS))
    ap.add_argument("--cluster", type=int, default=4)
    ap.add_argument("--levels", default="", help="comma-separated levels (default: maxLevel down to the op's minimum)")
    ap.add_argument("--out", default=os.path.join(ROOT, "outLogs"))
    args = ap.parse_args()
    p = SETS[args.set]
    cli = os.path.join(ROOT, "host", "Homulator.run")
    if not os.path.exists(cli):
        sys.exit("build first: make -C host")
    failed = 0
    for op in args.ops.split(","):
        levels = [int(x) for x in args.levels.split(",")] if args.levels else list(range(p["L"], min_level(op) - 1, -1))
        tag = "motivation" if args.set == "motivation" else f"para{args.set}"
        out_dir = os.path.join(args.out, tag, str(args.cluster), op, f"{p['L']}_{p['alpha']}")
        os.makedirs(out_dir, exist_ok=True)
        for lv in levels:
            if lv < min_level(op) or lv > p["L"]:
                continue
            log = os.path.join(out_dir, f"{op}_{p['L']}_{p['alpha']}_{lv}.log")
            with open(log, "a") as f:
                rc = subprocess.call([cli, os.path.join(ROOT, "config", p["cfg"]), op, str(p["L"]), str(lv), str(p["alpha"]),
                                      str(args.cluster)], stdout=f, stderr=subprocess.STDOUT)
            failed += rc != 0
            print(f"{tag} {op} L={p['L']} l={lv} alpha={p['alpha']}: rc={rc} -> {os.path.relpath(log, ROOT)}", flush=True)
    print("Completed start" if not failed else f"{failed} run(s) failed")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
