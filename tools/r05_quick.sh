#!/bin/bash
# quick lease: kernel + op parity, then the bench at the driver's command (no CPU baseline)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_quick; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_ops.py tests/test_gpu_ntt_fused.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
for r in 1 2; do
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_k20_$r.json 2> $OUT/bench_k20.err; echo "bench rc=$?"
python3 - <<P
import json
d = json.load(open("$OUT/bench_k20_$r.json"))
print("value", round(d["value"], 1), "sustained", round(d["sustained_ops_per_s"], 1), "single", round(d["single_stream_ops_per_s"], 1), "evk_once", round(d["hmult_frac_evk_once"], 3),
      "sweep", round(d["roofline"]["us_per_launch"], 1), round(d["roofline"]["frac"], 3), "hrotate", round(d["hrotate"]["ops_per_s"], 1), "generic", round(d.get("generic_chain_ops_per_s") or 0, 1))
print(d["stage_us_per_op_batched"]); print(d["stage_us"])
P
done
