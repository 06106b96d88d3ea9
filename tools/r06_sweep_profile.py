"""profiles/r06_sweep_sets.txt from the outputs of tools/r06_sweep_full.sh (gpurun_out/r06_sweep/*.txt)"""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = os.path.join(ROOT, "gpurun_out", "r06_sweep")


def load(f):
    rows = []
    for ln in open(f):
        if ln.startswith('#') or not ln.strip():
            continue
        p = ln.split()
        rows.append(dict(set=p[0], op=p[1], lv=int(p[4]), beta=int(p[5]), chain=p[6], launches=int(p[8]), us=float(p[9]), mb=float(p[10]), frac=float(p[11]), once=float(p[12]), ops=float(p[13]), raw=ln.rstrip()))
    return rows


out = ["""round 6 — the reference's parameter sets on the MI355X back-end at HEAD: every level, both prime chains (VERDICT r5 item 1)
=========================================================================================================================
script/sweep.py --bench (tools/r06_sweep_full.sh, ONE gpurun lease): sets A (N = 2^15, L 28, alpha 28), B (N = 2^16, 45 / 15), C (2^16, 24 / 6),
D (2^16, 26 / 9) of script/README.md:17-22 and `motivation` (script/motivation/micro24_motivation.sh: N = 2^16, L 28, alpha 28) x hmult / hrotate x
every level x {mont32 = the default chain of primes h 2^32 + 1, survey = SURVEY.md 8(d)'s chain as written (generic arithmetic back-end)}.
One instance, 8 ops per launch replayed as a HIP graph, 2 x 6 timed launches after 2 warm-up (the faster group counts); device time from the
back-end's own events.  alg_MB = SURVEY.md 8(d)'s algorithmic bytes for the shape (the formula of section 8d, per level); frac = alg_MB / time /
8 TB/s; frac_evk_once = the same with the evaluation key charged once per launch.  (bench.py's headline shape is 2 instances x 10 ops: ~8 % more
than one instance x 8.)  The planner's defaults are in force: at N = 2^16 digits of more than 15 limbs keep a conversion launch of their own
(cap_bconv_col_pref_in), at N = 2^15 every digit converts inside its first pass.
"""]
for s in ("A", "B", "C", "D", "motivation"):
    rows = load(f"{S}/head_{s}.txt")
    out.append(f"## set {s}: summary (fraction of the 8 TB/s peak by the algorithmic bytes; levels from the top down)")
    for chain in ("mont32", "survey"):
        for op in ("hmult", "hrotate"):
            x = sorted([r for r in rows if r['chain'] == chain and r['op'] == op], key=lambda r: -r['lv'])
            fr = [r['frac'] for r in x]
            dips = [x[i]['lv'] for i in range(1, len(x) - 1) if fr[i] < 0.6 * min(fr[i - 1], fr[i + 1])]
            out.append(f"{s:10s} {op:8s} {chain:7s} top level {x[0]['lv']:2d}: {x[0]['us']:7.1f} us/op {x[0]['ops']:6.0f} ops/s frac {x[0]['frac']:.3f} ({x[0]['once']:.3f} key once) | "
                       f"frac over the levels: max {max(fr):.3f} min {min(fr):.3f} (level {x[fr.index(min(fr))]['lv']}) | levels below 0.6 x their neighbours: {dips or 'none'}")
    out.append("")
out.append("## same box, interleaved: HEAD against the launch plan of round 5 (--plan r5: fused conversion capped at 15 input limbs, the automorphism as a launch; at N = 2^15 also no pass 7b and no small-launch forms), mont32, us per op")
out.append("## (set A: the wide conversion inside the first pass + the small-launch forms at N = 2^15; motivation: the two plans coincide since the default at N = 2^16 stops at 15 limbs —")
out.append("##  the run with the wide conversion forced inside, 1-5 % slower, is profiles/r06_wide_stage_times.txt and the first sweep of the round: gpurun_out/r06_sweep_ab)")
for s in ("A", "motivation"):
    h = load(f"{S}/head2_{s}.txt")
    r = load(f"{S}/r5_{s}.txt")
    for op in ("hmult", "hrotate"):
        hh = {x['lv']: x for x in h if x['op'] == op}
        rr = {x['lv']: x for x in r if x['op'] == op}
        out.append(f"{s} {op}: level:r5/HEAD  " + " ".join(f"{lv}:{rr[lv]['us']:.1f}/{hh[lv]['us']:.1f}={rr[lv]['us'] / hh[lv]['us']:.2f}" for lv in sorted(hh, reverse=True)))
out.append("")
out.append("## every point (set op L alpha level beta chain arith launches us_per_op alg_MB frac_of_8TBs frac_evk_once ops_per_s)")
for s in ("A", "B", "C", "D", "motivation"):
    out += [r['raw'] for r in load(f"{S}/head_{s}.txt")]
open(os.path.join(ROOT, "profiles", "r06_sweep_sets.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[1:36]))
