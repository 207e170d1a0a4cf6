#!/bin/bash
# Copies the summaries tools/collect_profiles.sh left under gpurun_out/prof_<name> into profiles/<round>/<dir> (the raw passes stay
# in gpurun_out/, which is scratch).   usage: tools/adopt_profiles.sh r3 r3:. r3soup:soup r3gi:gi_wall r3blob:blob
ROUND=$1; shift
for d in "$@"; do
  src=gpurun_out/prof_${d%%:*}; dst=profiles/$ROUND/${d##*:}
  mkdir -p $dst
  for f in issue_profile.json hbm_traffic.json kernel_stats.csv pmc_summary.txt bench_line_under_profiler.json; do cp $src/$f $dst/$f; done
done
