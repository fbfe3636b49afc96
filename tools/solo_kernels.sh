#!/bin/bash
# usage (GPU box, repo root): tools/solo_kernels.sh <outdir> [bench args]: kernel times with NOTHING overlapped (RANSAC rounds
# on one stream, one registration thread, sequential pass only) under rocprofv3 --kernel-trace --stats: what each kernel costs alone
out=$1; shift
root=$(pwd)
export CS_RANSAC_OVERLAP=0 CORSAIR_SPLIT_RANSAC=0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out" -- python3 "$root/bench.py" "$@" --no-cpu-baseline --no-overlap-probe --no-solo-probe --no-extra-workloads > "$root/$out"_profiled.json 2> "$root/$out"_profiled.err || exit 1
cd "$root"
python3 tools/kernel_stats.py "$out" 24
