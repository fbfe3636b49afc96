#!/bin/bash
# usage (GPU box, repo root): tools/pmc_collect.sh <outdir> [bench args, e.g. --workload table]
# Three separate rocprofv3 --pmc passes over the same bench.py command (FETCH_SIZE and WRITE_SIZE do not fit one
# pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"; the SQ set is a third), kernel trace only, probes off; then
# tools/pmc_to_json.py distils <outdir>/pmc.json for the bench's dominant kernel.
out=$1; shift
root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$root/$out/$tag" -- python3 "$root/bench.py" "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-overlap-probe --no-solo-probe --no-extra-workloads > "$root/$out/$tag.log" 2>&1 || exit 1
  echo "[pmc] pass $tag done"
done
cd "$root" && python3 tools/pmc_to_json.py "$out" "$@"
