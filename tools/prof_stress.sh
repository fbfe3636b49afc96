#!/bin/bash
# usage (GPU box, repo root): tools/prof_stress.sh <outdir>: config-C5 stress shapes un-profiled, then under rocprofv3
out=$1
root=$(pwd)
python3 tools/stress.py > "$out".json 2> "$out".err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out" -- python3 "$root/tools/stress.py" > "$root/$out"_profiled.json 2> "$root/$out"_profiled.err || exit 1
cd "$root" && python3 tools/kernel_stats.py "$out" 14
