#!/bin/bash
# On the GPU box: the measurements behind bench.py's roofline.latency_floor_us.
#   tools/ubench/build_evidence.sh (here)  ->  gpurun -- tools/latency_evidence.sh r03
# writes gpurun_out/<tag>_xchg_floor.txt (one chip-wide all-gather under the sc1 counter protocol of scan_common.h,
# 2 groups x 128 workgroups, in isolation; issue interval of a dependent v_mfma_f32_16x16x4_f32) and
# gpurun_out/<tag>_scan_stamps.json (in-kernel phase stamps of the two HyperLSTM scans, stamped build).
TAG=${1:-r03}
cd "$(dirname "$0")/.."
O=gpurun_out/${TAG}_xchg_floor.txt
B=tools/ubench/bin
{
  echo "# $(date -u +%F) one MI355X; every line is the binary's own output (tools/ubench/*.hip)"
  for b in xchg2_w0 xchg2_w20 xchg3_128_128 xchg3_128_128_w20 xchg3_64_64 xchg3_32_32 xchg4 mfma_rate; do
    echo "## $b"; timeout -k 10 120 $B/$b || exit 1
  done
} > $O 2>&1 || { tail -5 $O; exit 1; }
ASR_AMD_LIB=$PWD/asr_amd/libasr_amd_stamps.so timeout -k 10 200 python tools/scan_stamps.py --json gpurun_out/${TAG}_scan_stamps.json > gpurun_out/${TAG}_scan_stamps.txt 2>&1 || { tail -5 gpurun_out/${TAG}_scan_stamps.txt; exit 1; }
tail -30 $O; cat gpurun_out/${TAG}_scan_stamps.txt
