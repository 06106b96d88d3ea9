ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_closing; mkdir -p $OUT; export TMPDIR=/tmp
for r in 1 2 3; do
  timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/k20_$r.json 2> $OUT/k20_$r.err
  python3 -c "
import json;d=json.load(open('$OUT/k20_$r.json'));print('K=20', 'value', round(d['value'],1), [round(x,1) for x in d['value_min_median_max']], 'generic', round(d['generic_chain_ops_per_s'],1), 'hrotate', round(d['hrotate']['ops_per_s'],1), 'single', round(d['single_stream_ops_per_s'],1), 'sweep', round(d['roofline']['us_per_launch'],2), round(d['roofline']['frac'],3), 'evk once', round(d['roofline']['op_frac_evk_once'],3), 'measured', round(d['roofline']['op_measured_frac'],3), 'cpu', round(d['cpu_baseline']['value'],2), 'ew', [round(d['elementwise'][k]['frac_of_hbm_peak'],2) for k in ('hadd','pmult','padd')])"
done
timeout -k 10 400 python3 bench.py > $OUT/default.json 2> $OUT/default.err
python3 -c "
import json;d=json.load(open('$OUT/default.json'));print('default (K=%d)' % d['steps'], 'value', round(d['value'],1), [round(x,1) for x in d['value_min_median_max']], 'generic', round(d['generic_chain_ops_per_s'],1), 'hrotate', round(d['hrotate']['ops_per_s'],1), 'single', round(d['single_stream_ops_per_s'],1), 'sweep', round(d['roofline']['us_per_launch'],2))"
