#!/bin/bash
# full GPU parity suite + the driver's bench command (round-3 working script)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r03s}; mkdir -p $OUT
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; else rc=0; echo skipped > $OUT/tests.log; fi
tail -5 $OUT/tests.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR|E  )" $OUT/tests.log | head -30; exit $rc; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20.json 2> $OUT/bench20.err; python3 - <<P
import json
d=json.load(open("$OUT/bench20.json"))
print({k:d[k] for k in ("value","ms_per_step","single_stream_ops_per_s")}, d["config"]["launches_per_op"])
print(d["stage_us"])
print(d["roofline"]["us_per_launch"], d["roofline"]["in_op"])
P
HOMULATOR_FUSE_HPIP=0 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20_nohpip.json 2> $OUT/bench20_nohpip.err; python3 - <<P
import json
d=json.load(open("$OUT/bench20_nohpip.json"))
print("fuse_hpip=0", {k:d[k] for k in ("value","ms_per_step","single_stream_ops_per_s")}, d["config"]["launches_per_op"])
print(d["stage_us"])
P
