#!/bin/bash
# usage (GPU box, repo root): tools/conv_pmc.sh <outdir> [batch points voxel]
# rocprofv3 --pmc passes (separate: FETCH_SIZE does not fit beside the SQ set) over tools/conv_layers.py;
# tools/conv_pmc.py prints one line per convolution kernel instance.
out=$1; shift
root=$(pwd)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
            "FETCH_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$root/$out/$tag" -- python3 "$root/tools/conv_layers.py" "$@" > "$root/$out/$tag.log" 2>&1 || { tail -5 "$root/$out/$tag.log"; exit 1; }
done
cd "$root" && python3 tools/conv_pmc.py "$out"
