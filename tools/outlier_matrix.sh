# N full default 20-step lines per configuration on one box, legs included: tools/outlier_matrix.sh N "ENV=VAL args" ...
n=$1; shift
for cfg in "$@"; do
  for i in $(seq 1 $n); do
    env $(echo "$cfg" | tr ' ' '\n' | grep = | tr '\n' ' ') python bench.py --steps 20 --warmup 5 --no-cpu-baseline $(echo "$cfg" | tr ' ' '\n' | grep -v = | tr '\n' ' ') > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err && python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('[$cfg] run $i', round(d['value'],1), d['value_pass'], 'seq', round(d['sequential']['value'],1), {k:(v['value'],v['sequential_value']) for k,v in d['config']['legs'].items()})" || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
  done
done
