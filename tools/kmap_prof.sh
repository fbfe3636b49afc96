#!/bin/bash
# usage (GPU box, repo root): tools/kmap_prof.sh <outdir> <stress|chair> [ENV=VAL ...]: rocprofv3 kernel trace of 33 map builds
out=$1; kind=$2; shift 2
root=$(pwd)
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out" -- python3 "$root/tools/kmap_bench.py" $kind 10 maps-only > "$root/$out.log" 2>&1 || exit 1
cd "$root"
python3 tools/kernel_stats.py "$out" 22
