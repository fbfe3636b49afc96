# like ab_env2.sh, printing result fingerprints (mean iterations, mean RRE, survivors) instead of kernel times
wl=$1; var=$2; shift 2
for rep in 1 2; do
for v in "$@"; do
env $var=$v python bench.py --workload $wl --no-cpu-baseline --no-overlap-probe --no-solo-probe --no-extra-workloads > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err && python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); c=d['config']; print('$wl $var=$v', round(d['value'],1), repr(c['ransac_mean_iters']), repr(c['rre_mean_deg']), c['rre_15'], (d.get('ransac_prefilter') or {}).get('survivors'))" || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
done
done
