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
    python script/sweep.py --bench --set A,B,C,D,motivation --ops hmult,hrotate --chains mont32,survey   # round 6: device time per level

--bench (round 6) times every (set, op, level) on the GPU instead of writing the reference's logs: one instance, `--batch` ops per launch
replayed as a HIP graph, device time from the backend's own events (hh_op_execute), and beside it SURVEY.md 8(d)'s algorithmic bytes for
that shape and the fraction of the 8 TB/s HBM peak they amount to — per level, on both prime chains (mont32 = the default chain of primes
h 2^32 + 1; survey = SURVEY.md 8(d)'s chain as written, the generic arithmetic back-end).  `--plan r5` reruns the same point with the plan
of round 5 (fused conversion capped at 15 input limbs; at N = 2^15 no pass 7b and no small-launch forms) for a same-box A/B.
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


def alg_limb_polys(op, ell, alpha):
    """compulsory traffic with intra-stage fusion only, in limb-polys (N x 8 B each); returns (total, of which evaluation key)"""
    beta = -(-ell // alpha)
    E = ell + alpha
    conv = beta * E - ell                                     # converted limbs of the ModUp
    ks = 2 * ell + (ell + conv) + 2 * conv + (beta * E + 2 * beta * E + 2 * E) + 4 * alpha + (2 * alpha + 2 * ell) + 4 * ell
    evk = 2 * beta * E
    if op == "hmult":   # tensor 7l, key switch, sub x P^-1 + add 8l, rescale 4 + (2 + 2(l-1)) + 6(l-1)
        return 7 * ell + ks + 8 * ell + 4 + (2 + 2 * (ell - 1)) + 6 * (ell - 1), evk
    if op == "hrotate":   # automorphism 4l, key switch, final 7l
        return 4 * ell + ks + 7 * ell, evk
    return {"hadd": 6, "padd": 5, "pmult": 5}[op] * ell, 0   # element-wise ops: operands + result per limb (hadd: two polynomials of two inputs and one output)


assert alg_limb_polys("hmult", 35, 15) == (2103, 300) and alg_limb_polys("hrotate", 35, 15)[0] == 1685   # SURVEY.md 8(d)'s own figures


R5_PLAN = {"fuse_bconv_max_in": 15, "fuse_auto": 0}         # rounds 3-5: digits of more than 15 limbs kept a conversion launch of their own; hrotate ran its AUTO launch
R5_PLAN_N15 = {"fuse_bconv_max_in": 15, "fuse_ip_inv": 0, "fuse_auto": 0}   # ... and N = 2^15 had no pass 7b
R5_ENV_N15 = {"HOMULATOR_NTT_FUSED_SMALL": "0", "HOMULATOR_NTT_SMALL_LIMBS": "0", "HOMULATOR_NIP_SMALL": "0"}   # ... nor the small-launch forms


def bench(args):
    """one line per (set, op, level, chain): device us per op, algorithmic bytes, fraction of the HBM peak (and with the key charged once per launch)"""
    sys.path.insert(0, ROOT)
    from homulator_amd import host
    HBM = 8e12
    print(f"# script/sweep.py --bench: batch {args.batch} ops per launch, HIP graph, {args.iters} timed launches after {args.warm} warm-up; plan = {args.plan}")
    print("# set op L alpha level beta chain arith launches us_per_op alg_MB frac_of_8TBs frac_evk_once ops_per_s")
    for name in args.set.split(","):
        p = SETS[name]
        logN = 15 if "N15" in p["cfg"] else 16
        for op in args.ops.split(","):
            levels = [int(x) for x in args.levels.split(",")] if args.levels else list(range(p["L"], min_level(op) - 1, -1))
            for chain in args.chains.split(","):
                for lv in levels:
                    if lv < min_level(op) or lv > p["L"]:
                        continue
                    ov = {"batch": args.batch, "graph": 1}
                    if chain != "mont32":
                        ov["chain_bits"] = 60 if chain == "survey" else int(chain)
                    env = {}
                    if args.plan == "r5":
                        ov.update(R5_PLAN_N15 if logN == 15 else R5_PLAN)
                        env = R5_ENV_N15 if logN == 15 else {}
                    old = {k: os.environ.get(k) for k in env}
                    os.environ.update(env)
                    try:
                        h = host.Op(p["cfg"], op, p["L"], lv, p["alpha"], overrides=ov)
                        for _ in range(args.warm):
                            h.execute(1)
                        ns = min(h.execute(args.iters) for _ in range(args.repeat))
                        arith, launches = h.backend_counter("arith"), h.launch_count()
                        h.close()
                    finally:
                        for k, v in old.items():
                            if v is None:
                                os.environ.pop(k, None)
                            else:
                                os.environ[k] = v
                    us = ns * 1e-3 / args.batch
                    lp, evk = alg_limb_polys(op, lv, p["alpha"])
                    lpb = (8 << logN)
                    alg, once = lp * lpb, (lp - evk * (1 - 1 / args.batch)) * lpb
                    beta = -(-lv // p["alpha"])
                    print(f"{name} {op} {p['L']} {p['alpha']} {lv} {beta} {chain} {'generic' if arith else 'mont32'} {launches} {us:.2f} {alg / 1e6:.1f} "
                          f"{alg / (us * 1e-6) / HBM:.3f} {once / (us * 1e-6) / HBM:.3f} {1e6 / us:.0f}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bench", action="store_true", help="time every level on the GPU (device us per op, SURVEY.md 8(d) bytes, fraction of the HBM peak) instead of writing the reference's logs")
    ap.add_argument("--chains", default="mont32", help="--bench: comma-separated prime chains: mont32 (default chain), survey (SURVEY.md 8(d) as written), or a bit width")
    ap.add_argument("--batch", type=int, default=8, help="--bench: ops per launch")
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--warm", type=int, default=2)
    ap.add_argument("--repeat", type=int, default=2, help="--bench: timed groups of --iters launches; the fastest group counts")
    ap.add_argument("--plan", default="head", choices=["head", "r5"], help="--bench: r5 = the launch plan of round 5 for a same-box A/B")
    ap.add_argument("--set", required=True, help="A | B | C | D | motivation (--bench: a comma-separated list)")
    ap.add_argument("--ops", default=",".join(OPS))
    ap.add_argument("--cluster", type=int, default=4)
    ap.add_argument("--levels", default="", help="comma-separated levels (default: maxLevel down to the op's minimum)")
    ap.add_argument("--out", default=os.path.join(ROOT, "outLogs"))
    args = ap.parse_args()
    if args.bench:
        if args.ops == ",".join(OPS):
            args.ops = "hmult,hrotate"
        return bench(args)
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
