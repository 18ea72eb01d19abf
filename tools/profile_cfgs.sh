#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats of the other training configs (bench.py --cfg 1/2/4/5).
# usage: tools/profile_cfgs.sh <tag>
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in 1 2 4 5; do
  OUT=$R/gpurun_out/prof_${TAG}_cfg$c
  mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --cfg $c --steps 3 --warmup 1 > $OUT/bench.log 2>&1
  cp $(find $OUT -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_cfg${c}_kernel_stats.csv
  python3 $R/tools/timeline_gaps.py $(find $OUT -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/${TAG}_cfg${c}_timeline.txt 2>&1 || true
  rm -f $(find $OUT -name "*kernel_trace.csv")
done
