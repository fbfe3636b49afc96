# A/B of bench.py argument sets on one box (alternating twice): tools/ab_args.sh WORKLOAD STEPS "args1" "args2" ...
wl=$1; steps=$2; shift 2
for rep in 1 2; do
for a in "$@"; do
python bench.py --workload $wl --steps $steps --warmup 4 --no-cpu-baseline --no-solo-probe --no-extra-workloads $a > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err && python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$wl [$a] in flight', round(d['value'],1), 'sequential', round(d['sequential']['value'],1), 'identical', d.get('batches_in_flight',{}).get('identical_results'))" || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
done
done
