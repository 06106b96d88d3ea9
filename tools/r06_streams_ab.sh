export TMPDIR=/tmp
for r in 1 2; do for cfg in "2 10" "1 20" "1 10" "2 5" "3 7" "2 16"; do set -- $cfg
timeout -k 10 200 python3 bench.py --steps 840 --warmup 20 --streams $1 --batch $2 --no-cpu-baseline > /tmp/b.json 2>/dev/null
python3 -c "
import json;d=json.load(open('/tmp/b.json'));print('streams=$1 batch=$2', round(d['value'],1), round(d['sustained_ops_per_s'],1))"
done; done
