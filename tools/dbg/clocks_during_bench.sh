#!/bin/bash
# Diagnostic: shader / memory clocks reported by rocm-smi while the cfg3 bench runs (is the persistent scan held at a low clock?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 150 python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $R/gpurun_out/clk_bench.log 2>&1 &
BP=$!
sleep 20
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -4
  rocm-smi --showpower --showuse 2>/dev/null | grep -i "power\|busy" | head -3
  echo ---
  sleep 1.5
done
wait $BP
tail -n 1 $R/gpurun_out/clk_bench.log | cut -c1-200
