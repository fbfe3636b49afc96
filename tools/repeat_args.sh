# the same bench.py command N times on one box (outlier hunting): tools/repeat_args.sh N WORKLOAD STEPS "args"
n=$1; wl=$2; steps=$3; a=$4
for i in $(seq 1 $n); do
python bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline --no-solo-probe --no-extra-workloads $a > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err && python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$wl [$a] run $i in flight', round(d['value'],1), 'sequential', round(d['sequential']['value'],1), d['torch_allocator'] if 'torch_allocator' in d else '', d.get('scratch_pool'))" || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
done
