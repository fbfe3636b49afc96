#!/bin/bash
# usage (on the GPU box, from the repo root): tools/pmc_run.sh <outdir> <counters...> -- <python script + args>
# one rocprofv3 --pmc pass (kernel trace only, as the pool requires), summary printed per kernel
out=$1; shift
ctr=()
while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "${ctr[@]}" --output-format csv -d "$root/$out" -- python3 "$root/$1" "${@:2}" > "$root/$out.log" 2>&1
cd "$root" && python3 tools/pmc_summary.py "$out" | grep -i "prefilter\|ransac_count" 
