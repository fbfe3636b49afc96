# A/B of library builds on one box: prints q/s, prefilter and exact-count ms, survivors
run() { # name lib
CORSAIR_HIP_LIB=$2 python bench.py --no-cpu-baseline --no-overlap-probe --no-solo-probe > gpurun_out/b_$1.json 2> gpurun_out/b_$1.err && python -c "
import json,sys; d=json.load(open('gpurun_out/b_$1.json')); print('$1', round(d['value'],1), d['kernel_ms']['ransac_pre'], d['kernel_ms']['ransac_eval'], d['ransac_prefilter']['survivors'], d['ransac_prefilter']['hypotheses'], d['roofline']['avg_launch_ms'])"
}
L=$PWD/corsair_amd/csrc
for v in "$@"; do run $v $L/libcorsair_hip_$v.so || exit 1; done
