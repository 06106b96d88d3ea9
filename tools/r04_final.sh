#!/bin/bash
# round 4 closing measurements at HEAD: profile round, co-run pairs, the bench at the driver's command (twice) and at 200 steps, hrotate
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04_final; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/profile_round.sh r04 > $OUT/profile_round.log 2>&1
timeout -k 10 300 python3 tools/corun.py > $OUT/r04_corun.txt 2>&1
for r in 1 2; do timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_k20_$r.json 2> $OUT/bench_k20_$r.err; done
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_k200.json 2> $OUT/bench_k200.err
timeout -k 10 300 python3 bench.py --op hrotate --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_hrotate.json 2> $OUT/bench_hrotate.err
# same box, interleaved: HEAD against the library of the round's first half (Shoup butterflies, 16-byte twiddles: ab_builds/libhm_shoup.so, built from a7bf0ac)
if [ -f ab_builds/libhm_shoup.so ]; then PARITY=0 TIME_WRONG=1 ROUNDS=3 bash tools/r03_bench_ab.sh r04_final_ab shoup > $OUT/r04_montgomery_ab.txt 2>&1; fi
python3 - <<P
import json
for f in ("bench_k20_1", "bench_k20_2", "bench_k200", "bench_hrotate"):
    d = json.load(open("$OUT/" + f + ".json"))
    print(f, round(d["value"], 1), "sustained", d["sustained_ops_per_s"] and round(d["sustained_ops_per_s"], 1), "single", round(d["single_stream_ops_per_s"], 1), "frac", round(d["hmult_frac_of_hbm_peak"], 3), "evk_once", round(d["hmult_frac_evk_once"], 3),
          "sweep", round(d["roofline"]["us_per_launch"], 1), round(d["roofline"]["frac"], 3), "in_op", round(d["roofline"]["in_op"]["us_per_limb"], 3))
P
