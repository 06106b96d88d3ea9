"""backend = sim: the headline configuration on 1..16 clusters of the simulated accelerator (the reference's argv[6]).
usage: python tools/sim_clusters.py > profiles/rNN_sim_cluster_scaling.txt"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homulator_amd import host

rows = []
for cl in (1, 2, 4, 8, 16):
    for op in ("hmult", "hrotate"):
        t = time.time()
        o = host.Op("config_4.cfg", op, 45, 35, 15, backend=host.BACKEND_SIM, overrides={"cluster": cl})
        r = o.sim_run()
        o.close()
        busy = lambda u: sum(v for k, v in r["stats"].items() if k.startswith(u + "_(")) / (cl * r["cycles"])
        rows.append((cl, op, r["cycles"], r["drained"], busy("NTT"), busy("EWE"), busy("BCONV"), r["stats"].get("NoC_Mem_Chip", 0), time.time() - t))
print("# backend = sim: config_4.cfg <op> 45 35 15 (N = 2^16) on 1..16 clusters of the simulated accelerator (argv[6]).")
print("# busy = unit busy cycles / (clusters x cycles); speedup is against the .cfg's 4 clusters.  Host only: each line is one run of the")
print("# build's cycle model (DESIGN.md section 10).  One cluster cannot hold the working set in its 116 508-line scratchpad: the model")
print("# stops retiring and takes the reference's dead-lock exit (no instruction in 2000 cycles).")
print(f"{'clusters':>8} {'op':>8} {'cycles':>9} {'speedup':>8} {'NTT busy':>9} {'EWE busy':>9} {'BCONV busy':>11} {'NoC lines':>10} {'host s':>7}")
base = {op: cyc for cl, op, cyc, *_ in rows if cl == 4}
for cl, op, cyc, dr, n, e, b, noc, t in rows:
    sp = f"{base[op] / cyc:>8.2f}" if dr else f"{'-':>8}"
    print(f"{cl:>8} {op:>8} {cyc:>9} {sp} {n:>9.3f} {e:>9.3f} {b:>11.3f} {noc:>10} {t:>7.2f}" + ("" if dr else "  (dead-lock exit)"))
