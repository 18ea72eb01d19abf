#!/bin/bash
# Builds (here, no GPU needed) the exchange / MFMA micro-benchmarks whose output tools/latency_evidence.sh records
# on the GPU box, plus the stamped library for tools/scan_stamps.py.  Binaries go to tools/ubench/bin (git-ignored).
set -e
cd "$(dirname "$0")/../.."
B=tools/ubench/bin; mkdir -p $B
I="-I asr_amd/csrc"
H="hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value $I"
$H tools/ubench/xchg2.hip -o $B/xchg2_w0 &
$H -DWORK=20 tools/ubench/xchg2.hip -o $B/xchg2_w20 &
$H tools/ubench/xchg3.hip -o $B/xchg3_128_128 &
$H -DNG=64 -DNR=64 tools/ubench/xchg3.hip -o $B/xchg3_64_64 &
$H -DNG=32 -DNR=32 tools/ubench/xchg3.hip -o $B/xchg3_32_32 &
$H -DWORK=20 tools/ubench/xchg3.hip -o $B/xchg3_128_128_w20 &
$H tools/ubench/xchg4.hip -o $B/xchg4 &
$H tools/ubench/mfma_rate.hip -o $B/mfma_rate &
wait
python -m asr_amd.build --stamps
ls -la $B
