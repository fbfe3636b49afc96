# A/B of library builds on one box, RANSAC overlaps off (kernel times are stand-alone):
# tools/ab_lib.sh name1 name2 ... for corsair_amd/csrc/libcorsair_hip_<name>.so
run() {
CORSAIR_HIP_LIB=$2 CORSAIR_SPLIT_RANSAC=0 CS_RANSAC_OVERLAP=0 python bench.py --no-cpu-baseline --no-overlap-probe --no-solo-probe --no-extra-workloads > gpurun_out/b_$1.json 2> gpurun_out/b_$1.err && python -c "
import json,sys; d=json.load(open('gpurun_out/b_$1.json')); print('$1', round(d['value'],1), d['kernel_ms']['ransac_pre'], d['kernel_ms']['ransac_eval'], d['kernel_ms']['ransac_hyp'], d['ransac_prefilter']['survivors'], d['roofline']['avg_launch_ms'])"
}
L=$PWD/corsair_amd/csrc
for v in "$@"; do run $v $L/libcorsair_hip_$v.so || exit 1; done
