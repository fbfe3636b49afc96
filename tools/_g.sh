#!/bin/bash
set -e
tools/prof_bench.sh gpurun_out/r4d_stress --workload stress --steps 3 > gpurun_out/r4d_stress.log 2>&1
head -45 gpurun_out/r4d_stress.log
python3 -c "
import json
l=json.loads(open('gpurun_out/r4d_stress_line.json').read().strip().splitlines()[-1]); print(l['value'], l['ms_per_step'], l['roofline'])"
