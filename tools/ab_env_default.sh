# A/B of one environment variable on whole default lines (headline + legs) of one box:
#   tools/ab_env_default.sh STEPS REPS VAR v1 v2 ...      (alternating REPS times)
steps=$1; reps=$2; var=$3; shift 3
for rep in $(seq $reps); do
for v in "$@"; do
env $var=$v python bench.py --steps $steps --no-cpu-baseline > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err && python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); L=d['config']['legs']
print('$var=$v chair', round(d['value'],1), 'seq', round(d['sequential']['value'],1), '|', ' '.join('%s %.1f seq %.1f' % (k, L[k]['value'], L[k]['sequential_value']) for k in L))" || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
done
done
