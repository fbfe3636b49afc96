#!/bin/bash
# usage (GPU box, repo root): tools/prof_bench.sh <outdir> [bench args]: rocprofv3 kernel trace + stats of bench.py,
# then top kernels and idle gaps.  The un-profiled line of the same box goes to <outdir>_line.json first.
out=$1; shift
root=$(pwd)
python3 bench.py "$@" > "$out"_line.json 2> "$out"_line.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out" -- python3 "$root/bench.py" "$@" --no-cpu-baseline --no-overlap-probe --no-solo-probe --no-extra-workloads > "$root/$out"_profiled.json 2> "$root/$out"_profiled.err || exit 1
cd "$root"
python3 tools/kernel_stats.py "$out" 30
python3 tools/gap_report.py "$out" 0.5
