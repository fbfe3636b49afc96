#!/bin/bash
timeout -k 10 600 python3 -m pytest tests/test_gpu_post.py tests/test_gpu_sympose.py -x -q > gpurun_out/r4a_test.log 2>&1 || { tail -30 gpurun_out/r4a_test.log; exit 1; }
tail -2 gpurun_out/r4a_test.log
for w in chair table; do for hs in 1 0 1 0; do
  CS_RANSAC_HYP_STREAM=$hs timeout -k 10 300 python3 bench.py --workload $w --steps 10 --no-cpu-baseline > gpurun_out/r4a_${w}_hs$hs.json 2> gpurun_out/r4a_${w}_hs$hs.err || { tail -5 gpurun_out/r4a_${w}_hs$hs.err; exit 1; }
  python3 - <<PY
import json
l=json.loads(open("gpurun_out/r4a_${w}_hs$hs.json").read().strip().splitlines()[-1])
print("$w hyp_stream=$hs", round(l["value"],1), round(l["ms_per_step"],2))
PY
done; done
