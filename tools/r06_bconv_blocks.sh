export TMPDIR=/tmp
for r in 1 2; do for w in 3072 2048 6144 12288; do
HOMULATOR_BCONV_BLOCKS=$w timeout -k 10 200 python3 bench.py --steps 400 --warmup 20 --no-cpu-baseline > /tmp/b.json 2>/dev/null
python3 -c "
import json;d=json.load(open('/tmp/b.json'));print('blocks=$w', round(d['value'],1), round(d['sustained_ops_per_s'],1), round(d['single_stream_ops_per_s'],1), [x[2] for x in d['stage_us_per_op_batched'] if x[0]=='BCONV'], [x[2] for x in d['stage_us'] if x[0]=='BCONV'])"
done; done
