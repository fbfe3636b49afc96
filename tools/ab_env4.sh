# A/B of one environment variable on one box, both passes: tools/ab_env4.sh WORKLOAD STEPS VAR v1 v2 ... (alternating twice)
wl=$1; steps=$2; var=$3; shift 3
for rep in 1 2; do
for v in "$@"; do
env $var=$v python bench.py --workload $wl --steps $steps --warmup 4 --no-cpu-baseline --no-solo-probe --no-extra-workloads > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err && python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$wl $var=$v in flight', round(d['value'],1), 'sequential', round(d['sequential']['value'],1), 'identical', d.get('batches_in_flight',{}).get('identical_results'), {k: round(v,1) for k,v in d['kernel_ms'].items() if v > 0})" || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
done
done
