#!/bin/bash
# usage (GPU box, repo root): tools/topk_prof.sh <outdir> [topk_ab.py arguments]: rocprofv3 kernel trace of tools/topk_ab.py
out=$1; shift
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out" -- python3 "$root/tools/topk_ab.py" "$@" > "$root/$out.log" 2>&1 || exit 1
cd "$root"
python3 tools/kernel_stats.py "$out" 12
