# A/B of library builds on one box, whole pipeline (both passes): tools/ab_lib2.sh WORKLOAD STEPS REPS name1 name2 ...
# for corsair_amd/csrc/libcorsair_hip_<name>.so ("cur" = the library in place)
wl=$1; steps=$2; reps=$3; shift 3
L=$PWD/corsair_amd/csrc
for rep in $(seq $reps); do
for v in "$@"; do
lib=$L/libcorsair_hip_$v.so; [ $v = cur ] && lib=$L/libcorsair_hip.so
CORSAIR_HIP_LIB=$lib python bench.py --workload $wl --steps $steps --warmup 4 --no-cpu-baseline --no-solo-probe --no-extra-workloads > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err && python -c "
import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$wl $v in flight', round(d['value'],1), 'sequential', round(d['sequential']['value'],1), 'identical', d.get('batches_in_flight',{}).get('identical_results'), {k: round(v,1) for k,v in d['kernel_ms'].items() if v > 0}, 'survivors', d.get('ransac_prefilter',{}).get('survivors'))" || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
done
done
