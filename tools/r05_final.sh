#!/bin/bash
# round 5 closing measurements at HEAD: profile round (kernel trace + stats of the bench command, sweep / whole-op / 128-limb counters), per-kernel
# counters of the batched op, the one-op-at-a-time kernel trace, the bench at the driver's command (twice) and at 200 steps
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_final; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/profile_round.sh r05 > $OUT/profile_round.log 2>&1; echo "profile_round rc=$?"
bash tools/pmc_kernels.sh r05_pmc_kernels > $OUT/pmc_kernels.log 2>&1; echo "pmc_kernels rc=$?"
bash tools/r05_trace1.sh > $OUT/trace1.txt 2>&1; echo "trace1 rc=$?"
for r in 1 2; do timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_k20_$r.json 2> $OUT/bench_k20_$r.err; echo "bench $r rc=$?"; done
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_k200.json 2> $OUT/bench_k200.err
python3 - <<P
import json
for f in ("bench_k20_1", "bench_k20_2", "bench_k200"):
    d = json.load(open("$OUT/" + f + ".json"))
    print(f, round(d["value"], 1), "sustained", round(d["sustained_ops_per_s"], 1), "single", round(d["single_stream_ops_per_s"], 1), "frac", round(d["hmult_frac_of_hbm_peak"], 3), "evk_once", round(d["hmult_frac_evk_once"], 3),
          "sweep", round(d["roofline"]["us_per_launch"], 1), round(d["roofline"]["frac"], 3), "in_op", round(d["roofline"]["in_op"]["us_per_limb"], 3), "hrotate", round(d["hrotate"]["ops_per_s"], 1), "generic", round(d["generic_chain_ops_per_s"], 1), round(d["generic_chain_frac_evk_once"], 3))
P
