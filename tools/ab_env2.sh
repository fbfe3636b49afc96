# A/B of one environment variable on one box, any workload: tools/ab_env2.sh WORKLOAD VAR v1 v2 ... (alternating twice)
wl=$1; var=$2; shift 2
for rep in 1 2; do
for v in "$@"; do
env $var=$v python bench.py --workload $wl --no-cpu-baseline --no-overlap-probe --no-solo-probe --no-extra-workloads > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err && python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$wl $var=$v', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['kernel_ms'].items() if v > 0}, (d.get('ransac_prefilter') or {}).get('survivors'), round(d['roofline']['avg_launch_ms'],4))" || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
done
done
