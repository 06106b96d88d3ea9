"""profiles/roofline_inputs.json from the profiler outputs of tools/profile_round.sh (bench.py reads it; nothing is typed in
by hand).  usage: python tools/make_roofline_inputs.py gpurun_out/prof_<round> <round>"""
import json, os, re, sys
d, r = sys.argv[1], sys.argv[2]
sweep = open(os.path.join(d, f"{r}_pmc_ntt_sweep.txt")).read()
fetch = [float(x) for x in re.findall(r"FETCH_SIZE=([0-9.e+]+)", sweep)]
write = [float(x) for x in re.findall(r"WRITE_SIZE=([0-9.e+]+)", sweep)]
assert len(fetch) == 2 and len(write) == 2, sweep
traffic = int((2 * sum(fetch) + sum(write)) * 1024)   # KiB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md §HBM)
bf = open(os.path.join(d, f"{r}_bflyrate.txt")).read()
wb = float(re.search(r"wave_butterfly_ns \(full chip, sustained\) = ([0-9.]+)", bf).group(1))
out = {"ntt_sweep50_traffic_bytes": traffic,
       "ntt_sweep50_traffic_source": f"profiles/{r}_pmc_ntt_sweep.txt: (2 x FETCH_SIZE + WRITE_SIZE) KiB summed over the two pass kernels, separate --pmc passes",
       "wave_butterfly_ns": wb,
       "wave_butterfly_source": f"profiles/{r}_bflyrate.txt (tools/bflyrate, V3 = the shipped butterflies): kernel time / wave-butterflies with all 1024 SIMDs busy, clocks settled"}
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "roofline_inputs.json"), "w"), indent=1)
print(out)
