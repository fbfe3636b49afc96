#!/bin/bash
timeout -k 10 600 python3 -m pytest tests/test_gpu_sparse.py tests/test_gpu_real_clouds.py tests/test_gpu_randomized.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size.py tests/test_gpu_harness.py -x -q > gpurun_out/r7_test.log 2>&1 || { tail -30 gpurun_out/r7_test.log; exit 1; }
tail -1 gpurun_out/r7_test.log
for w in chair stress; do
timeout -k 10 300 python3 bench.py --workload $w --steps 8 --no-cpu-baseline --no-overlap-probe --no-solo-probe > gpurun_out/r7_$w.json 2> gpurun_out/r7_$w.err || { tail -5 gpurun_out/r7_$w.err; exit 1; }
python3 -c "
import json
l=json.loads(open('gpurun_out/r7_$w.json').read().strip().splitlines()[-1]); print('$w', round(l['value'],1), round(l['ms_per_step'],2), l['roofline_by_kernel']['conv'])"
done
