#!/bin/bash
# usage (build container, repo root): tools/publish_profiles.sh <tag> -- copies what tools/final_profiles.sh <tag> brought back under
# gpurun_out/<tag>/ into profiles/ (the tracked copies DESIGN.md / profiles/README.md quote) and prints the summary.
tag=$1
o=gpurun_out/$tag
for f in $o/${tag}_*_kernel_stats.csv $o/${tag}_*_line.json $o/${tag}_pmc_*_table.txt $o/${tag}_chair_launch_census.txt $o/${tag}_default_line.json; do cp $f profiles/; done
for w in chair table stress; do
  cp $o/pmc_$w.json profiles/pmc_$w.json
  cp $o/${tag}_${w}_profiled.json profiles/${tag}_${w}_line_profiled.json
done
python3 tools/summarize_profiles.py $o $tag
