#!/bin/bash
# usage (GPU box, repo root): tools/pmc_bench.sh <outdir> <counter>  -- one rocprofv3 --pmc pass over bench.py (3 steps)
out=$1; ctr=$2
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$root/$out" -- python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-overlap-probe --no-solo-probe --no-extra-workloads > "$root/$out.log" 2>&1
cd "$root" && python3 tools/pmc_summary.py "$out"
