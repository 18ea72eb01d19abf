#!/bin/bash
# Timing ablations of gemm_split8_kernel's main loop (results are WRONG by construction; timing only):
#   G8_ABL=1 no global loads in the loop, 2 no split + LDS stores, 3 no fragment reads / MFMAs / joins, 4 neither loads nor
#   stores (multiply phase + barrier only), SPL_ABL=2 products accumulated straight into the running sum (no joins).
# Builds asr_amd/libasr_amd_abl*.so; run with ASR_AMD_LIB=... python tools/ubench/gemm_bench.py   (see profiles/r03_gemm_ablation.txt)
set -e
cd "$(dirname "$0")/../.."
for v in 1 2 3 4; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -DG8_ABL=$v -shared -o asr_amd/libasr_amd_abl$v.so asr_amd/csrc/*.hip &
done
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -DSPL_ABL=2 -shared -o asr_amd/libasr_amd_abl5.so asr_amd/csrc/*.hip &
wait
ls -la asr_amd/libasr_amd_abl*.so
