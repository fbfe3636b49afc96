#!/bin/bash
# final collection at HEAD: smoke, bench lines + kernel stats, PMC passes (chair / table / stress)
set -e
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > gpurun_out/r4x_smoke.log 2>&1
tail -1 gpurun_out/r4x_smoke.log
for w in chair table stress; do
  tools/prof_bench.sh gpurun_out/r4x_$w --workload $w > gpurun_out/r4x_$w.log 2>&1
  cp gpurun_out/r4x_$w/*/*kernel_stats.csv gpurun_out/r4x_${w}_kernel_stats.csv
  rm -rf gpurun_out/r4x_$w
  python3 -c "
import json
l=json.loads(open('gpurun_out/r4x_${w}_line.json').read().strip().splitlines()[-1]); print('$w', round(l['value'],1), round(l['ms_per_step'],2), l['roofline']['kernel'], round(l['roofline']['frac'],3))"
done
for w in chair table stress; do
  tools/pmc_collect.sh gpurun_out/r4x_pmc_$w --workload $w > gpurun_out/r4x_pmc_$w.log 2>&1
  rm -rf gpurun_out/r4x_pmc_$w/FETCH_SIZE gpurun_out/r4x_pmc_$w/WRITE_SIZE gpurun_out/r4x_pmc_$w/SQ_VALU_MFMA_BUSY_CYCLES
  echo "pmc $w done"
done
