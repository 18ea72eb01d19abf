#!/bin/bash
# Kernel-level profile of the other BASELINE configs (rocprofv3 --kernel-trace --stats of bench.py --cfg N): where
# a step of cfg1 / cfg2 / cfg4 / cfg5 goes.  tools/profile_configs.sh [tag]   (on the GPU box, from the repo root)
TAG=${1:-r03}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_cfg_$TAG
mkdir -p $O
for c in 1 2 4 5; do
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg$c -o cfg$c -- python3 $OLDPWD/bench.py --cfg $c --steps 3 --warmup 2 --pool 2 > $O/cfg$c.log 2>&1) || { tail -5 $O/cfg$c.log; exit 1; }
  f=$(find $O/cfg$c -name '*kernel_stats.csv' | head -1)
  echo "== cfg$c"; head -9 "$f" | cut -d, -f1-4 | cut -c1-150
  cp "$f" gpurun_out/${TAG}_cfg${c}_kernel_stats.csv
done
