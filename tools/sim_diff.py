"""Differential test of the `sim` backend against the compiled, unmodified reference (oracle/_ref/ForgeHomulator.run, build
container only): random operations, levels, cluster counts AND random timing / sizing constants in the .cfg, comparing the
cycle count and every counter of the stat block.  The reference runs with MALLOC_PERTURB_ set (see
tests/golden/make_structural.py for why); `--stock` runs it as it is, which shows how often its uninitialised scoreboard operands
change a result.  usage: python tools/sim_diff.py [points] [seed] [jobs] [--stock]"""
import os
import random
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ForgeHomulator.run")
CLI = os.path.join(ROOT, "host", "Homulator.run")
BASE = {"N": 32768, "cluster": 4, "hasHPIPU": 0, "batchSize": 256, "elementBitWidth": 36, "memCount": 1, "memSize": 128, "entryCount": 1,
        "offDelay": 2, "memDramFifo": 4, "memUnitsFifo": 4, "ewe_mult_delay": 4, "ewe_madd_delay": 2, "ewe_num_mul": 4, "ewe_num_add": 2,
        "ewe_full_pipeline": 1, "bconv_num_high": 2, "bconv_num_width": 6, "bconv_mac_delay": 20, "bconv_fifo_delay": 4, "butterfly_delay": 5,
        "phase1_step1_depth": 4, "phase1_step2_depth": 4, "phase2_step1_depth": 4, "phase2_step2_depth": 4, "intraTrans_delay": 16,
        "interTrans_delay": 256, "ntt_stall_delay": 0, "VecPECount": 4, "MacCount": 6, "MacDelay": 1, "auto_stages": 6, "auto_delay": 6,
        "memlinestatistic": 0, "NocOpt": 1}


def draw(rng):
    cfg = dict(BASE)
    cfg["N"] = rng.choice([8192, 16384, 32768])
    cfg["batchSize"] = rng.choice([128, 256])
    if rng.random() < 0.7:
        for key, choices in (("offDelay", [1, 2, 3, 5]), ("memDramFifo", [1, 2, 4, 8]), ("memUnitsFifo", [1, 2, 4, 8]), ("ewe_mult_delay", [1, 3, 4, 6]),
                             ("ewe_madd_delay", [1, 2, 3]), ("bconv_num_high", [1, 2, 3]), ("bconv_num_width", [2, 4, 6, 7]),
                             ("bconv_mac_delay", [4, 12, 20]), ("bconv_fifo_delay", [1, 4]), ("butterfly_delay", [2, 5]),
                             ("phase1_step2_depth", [2, 4]), ("intraTrans_delay", [4, 16]), ("interTrans_delay", [32, 256]),
                             ("ntt_stall_delay", [0, 0, 3, 8]), ("auto_stages", [1, 3, 6]), ("auto_delay", [1, 6]), ("hasHPIPU", [0, 1]),
                             ("memSize", [128, 128, 2, 1])):
            if rng.random() < 0.35:
                cfg[key] = rng.choice(choices)
    op = rng.choice(["hmult", "hmult", "hrotate", "hrotate", "hadd", "pmult", "padd"])
    alpha = rng.choice([1, 2, 3, 4])
    level = rng.randint(2, 7)
    L = level + rng.randint(0, 3)
    cluster = rng.choice([None, None, 1, 2, 3, 5, 8])
    return cfg, op, L, level, alpha, cluster


def parse(out):
    lines = out.split("\n")
    cyc = [ln for ln in lines if ln.startswith("FHE-Sim Total simulated")]
    if not cyc or "Start outPut statistic informations:" not in lines:
        return None, {}
    i = lines.index("Start outPut statistic informations:")
    return int(cyc[0].split("\t")[1].split()[0]), {ln.split(" :\t")[0]: int(ln.split(" :\t")[1]) for ln in lines[i + 2:] if " :\t" in ln}


def one(args):
    idx, (cfg, op, L, level, alpha, cluster) = args
    with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
        for k, v in cfg.items():
            f.write(f"{k} = {v}\n")
        path = f.name
    argv = [path, op, str(L), str(level), str(alpha)] + ([str(cluster)] if cluster else [])
    try:
        renv = dict(os.environ)
        renv.pop("MALLOC_PERTURB_", None)
        if not STOCK:
            renv["MALLOC_PERTURB_"] = "85"
        ref = subprocess.run([REF] + argv, capture_output=True, text=True, env=renv, timeout=900)
        mine = subprocess.run([CLI] + argv, capture_output=True, text=True, env=dict(os.environ, HOMULATOR_BACKEND="sim"), timeout=900)
    except subprocess.TimeoutExpired:
        return idx, "timeout", argv, cfg
    finally:
        os.unlink(path)
    rc, rs = parse(ref.stdout)
    mc, ms = parse(mine.stdout)
    delta = {k: v for k, v in cfg.items() if BASE[k] != v}
    if rc is None:  # the reference itself gave up (its dead-lock exit prints no stat block in some states, or it crashed)
        dead = "We have executed 0 instruction(s) in this period!" in ref.stdout
        mdead = "We have executed 0 instruction(s) in this period!" in mine.stdout
        return idx, "both-deadlock" if dead and mdead else f"ref-no-result(rc={ref.returncode}, dead={dead}) mine={mc} mine_dead={mdead} {mine.stderr[-200:]}", argv[1:], delta
    if (rc, rs) == (mc, ms):
        return idx, "ok", argv[1:], delta
    diff = {k: (rs.get(k), ms.get(k)) for k in set(rs) | set(ms) if rs.get(k) != ms.get(k)}
    return idx, f"DIFF cycles ref={rc} mine={mc} counters={dict(list(diff.items())[:6])} {mine.stderr[-300:]}", argv[1:], delta


STOCK = "--stock" in sys.argv   # compare with the reference as it is (its uninitialised scoreboard operands included): how often do they matter?
if STOCK:
    sys.argv.remove("--stock")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    jobs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rng = random.Random(seed)
    pts = [draw(rng) for _ in range(n)]
    bad = 0
    with ThreadPoolExecutor(jobs) as ex:
        for idx, verdict, argv, delta in ex.map(one, enumerate(pts)):
            print(idx, verdict, " ".join(argv), delta, flush=True)
            bad += verdict not in ("ok", "both-deadlock")
    print(f"{n - bad} of {n} points agree in the cycle count and every counter")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
