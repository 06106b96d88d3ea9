"""profiles/roofline_inputs.json from the profiler outputs of tools/profile_round.sh (bench.py reads it; nothing is typed in
by hand).  usage: python tools/make_roofline_inputs.py gpurun_out/prof_<round> <round>"""
import json, os, re, sys
d, r = sys.argv[1], sys.argv[2]
sweep = open(os.path.join(d, f"{r}_pmc_ntt_sweep.txt")).read()
fetch = [float(x) for x in re.findall(r"FETCH_SIZE=([0-9.e+]+)", sweep)]
write = [float(x) for x in re.findall(r"WRITE_SIZE=([0-9.e+]+)", sweep)]
assert len(fetch) >= 1 and len(fetch) == len(write), sweep
traffic = int((2 * sum(fetch) + sum(write)) * 1024)   # KiB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md §HBM)
bf = open(os.path.join(d, f"{r}_bflyrate.txt")).read()
wb = float(re.search(r"\swave_butterfly_ns \(full chip, sustained\) = ([0-9.]+)", bf).group(1))
whole = {}
wp = os.path.join(d, f"{r}_pmc_whole_op.txt")
if os.path.exists(wp):   # "<COUNTER> total <KiB> KiB per op = ... ; shape: batch B x instances S"
    t = open(wp).read()
    f = re.search(r"FETCH_SIZE total ([0-9.]+) KiB per op", t)
    w = re.search(r"WRITE_SIZE total ([0-9.]+) KiB per op", t)
    sh = re.search(r"shape: batch (\d+) x instances (\d+)", t)
    if f and w and sh:
        whole = {"whole_op_bytes": int((2 * float(f.group(1)) + float(w.group(1))) * 1024),
                 "whole_op_fetch_kib": float(f.group(1)), "whole_op_write_kib": float(w.group(1)),
                 "whole_op_batch": int(sh.group(1)), "whole_op_instances": int(sh.group(2)),
                 "whole_op_source": f"profiles/{r}_pmc_whole_op.txt: (2 x FETCH_SIZE + WRITE_SIZE) KiB over every kernel of hmult at the timed region's launch shape, per op, separate --pmc passes"}
if os.path.exists(wp):   # per-kernel lines "<kernel> <KiB> KiB per op" under each counter: the NTT_IP launch = k_bconv_col* + k_ntt_row_ip
    t = open(wp).read()
    def kernel_sum(counter):
        tot, seen = 0.0, False
        for ln in t.splitlines():
            if ln.startswith(f"{counter} total"):
                return tot
            m = re.match(r"(k_.*?)\s+([0-9.]+) (KiB )?per op$", ln)   # (kernel names hold ", ")
            if m:
                if m.group(1).startswith("k_bconv_col") or m.group(1).startswith("k_ntt_row_ip"):
                    tot += float(m.group(2))
            if "total" in ln and not ln.startswith(counter):
                tot = 0.0   # the next counter's block starts behind a total line
        return tot
    f, w, v = kernel_sum("FETCH_SIZE"), kernel_sum("WRITE_SIZE"), kernel_sum("SQ_INSTS_VALU")
    if f and w:
        whole["ntt_ip_bytes_per_op"] = int((2 * f + w) * 1024)
        whole["ntt_ip_valu_per_op"] = int(v)
out = {**whole, "ntt_sweep50_traffic_bytes": traffic,
       "ntt_sweep50_traffic_source": f"profiles/{r}_pmc_ntt_sweep.txt: (2 x FETCH_SIZE + WRITE_SIZE) KiB summed over the two pass kernels, separate --pmc passes",
       "wave_butterfly_ns": wb,
       "wave_butterfly_source": f"profiles/{r}_bflyrate.txt (tools/bflyrate, V3 = the shipped butterflies): kernel time / wave-butterflies with all 1024 SIMDs busy, clocks settled"}
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "roofline_inputs.json"), "w"), indent=1)
print(out)
