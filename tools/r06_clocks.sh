export TMPDIR=/tmp
python3 bench.py --steps 4000 --warmup 20 --no-cpu-baseline > /tmp/b.json 2>/dev/null &
PID=$!
sleep 6
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | head -6; echo ---; sleep 0.7; done
wait $PID
python3 -c "
import json;d=json.load(open('/tmp/b.json'));print(round(d['value'],1))"
