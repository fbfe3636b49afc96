#!/bin/bash
# usage (GPU box, repo root): tools/final_profiles.sh <tag>   e.g. r4z
# The measurement set DESIGN.md / profiles/README.md quote, on ONE box: three --pmc passes per workload first (their
# pmc.json is put where bench.py looks for it), then per workload the unprofiled line and the same command under
# rocprofv3 --kernel-trace --stats, the launch census of the chair step, and the default (driver) line.  Raw traces are
# deleted as soon as they are distilled (gpurun brings back at most 64 MiB).
tag=$1
o=gpurun_out/$tag
mkdir -p $o
for w in chair table stress; do
  extra=""; [ $w != chair ] && extra="--workload $w"
  bash tools/pmc_collect.sh $o/pmc_$w $extra > $o/pmc_$w.log 2>&1 || { echo "pmc $w failed"; tail -5 $o/pmc_$w.log; exit 1; }
  cp $o/pmc_$w/pmc.json profiles/pmc_$w.json
  cp $o/pmc_$w/pmc.json $o/pmc_$w.json
  cp $o/pmc_$w/pmc_table.txt $o/${tag}_pmc_${w}_table.txt
  rm -rf $o/pmc_$w $o/pmc_$w.log
  echo "[final] pmc $w done"
done
for w in chair table stress; do
  extra=""; [ $w != chair ] && extra="--workload $w"
  bash tools/prof_bench.sh $o/${tag}_$w $extra --no-extra-workloads > $o/${tag}_${w}_summary.txt 2>&1 || { echo "prof $w failed"; tail -5 $o/${tag}_${w}_summary.txt; exit 1; }
  cp $(ls $o/${tag}_$w/*/*kernel_stats.csv | head -1) $o/${tag}_${w}_kernel_stats.csv
  rm -rf $o/${tag}_$w
  echo "[final] prof $w done"
done
bash tools/launch_census.sh $o/census > $o/${tag}_chair_launch_census.txt 2>&1 || { echo "census failed"; tail -5 $o/${tag}_chair_launch_census.txt; }
rm -rf $o/census
python3 bench.py > $o/${tag}_default_line.json 2> $o/${tag}_default_line.err || { echo "default line failed"; tail -5 $o/${tag}_default_line.err; exit 1; }
du -sh $o
echo "[final] done"
