#!/bin/bash
# usage (GPU box, repo root): tools/launch_census.sh <outdir> [bench args]
# Launches per timed step by kernel: rocprofv3 --kernel-trace --stats on bench.py at 2 and at 10 timed steps
# (same warm-up, probes off); tools/launch_census.py subtracts the two and divides by 8.
out=$1; shift
root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for steps in 2 10; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/s$steps" -- python3 "$root/bench.py" "$@" --steps $steps --warmup 2 --no-cpu-baseline --no-overlap-probe --no-solo-probe --no-extra-workloads > "$root/$out/s$steps.log" 2>&1 || { tail -5 "$root/$out/s$steps.log"; exit 1; }
done
cd "$root"
for steps in 2 10; do cp $(ls $out/s$steps/*/*kernel_stats.csv | head -1) $out/kernel_stats_$steps.csv; done
python3 tools/launch_census.py $out
