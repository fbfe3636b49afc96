# A/B of one environment variable on one box: tools/ab_env.sh VAR v1 v2 ...
var=$1; shift
i=0
for v in "$@"; do
i=$((i+1))
env $var=$v python bench.py --no-cpu-baseline --no-overlap-probe --no-solo-probe --no-extra-workloads > gpurun_out/b_$i.json 2> gpurun_out/b_$i.err && python -c "
import json,sys; d=json.load(open('gpurun_out/b_$i.json')); print('$var=$v', round(d['value'],1), d['ms_per_step'], d['kernel_ms']['ransac_pre'], d['kernel_ms']['ransac_eval'], d['kernel_ms']['ransac_hyp'], d['ransac_prefilter']['survivors'], d['roofline']['avg_launch_ms'])" || exit 1
done
