#!/bin/bash
# round 5, second lease: GPU tests with both arithmetic back-ends behind the one ABI, bench on the default chain
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_second; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_k20.json 2> $OUT/bench_k20.err; echo "bench rc=$?"
python3 - <<P
import json
d = json.load(open("$OUT/bench_k20.json"))
print("value", round(d["value"], 1), "sustained", round(d["sustained_ops_per_s"], 1), "single", round(d["single_stream_ops_per_s"], 1), "evk_once", round(d["hmult_frac_evk_once"], 3),
      "sweep", round(d["roofline"]["us_per_launch"], 1), round(d["roofline"]["frac"], 3), "hrotate", round(d["hrotate"]["ops_per_s"], 1), "cross", d["ntt_cross_xcd"], d["config"]["moduli"][:20])
P
