#!/bin/bash
# Comparison builds of the library with the round-2 fast activations (ASR_ACT=0) and with ocml's expf / tanhf and
# IEEE division (ASR_ACT=2), next to the product's forms (common.h).  Used once in round 3 to price the accurate
# activations (profiles/r03_act_variants.txt); select one with ASR_AMD_LIB=asr_amd/libasr_amd_act0.so.
set -e
cd "$(dirname "$0")/.."
for v in 0 2; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -DASR_ACT=$v -shared \
        -o asr_amd/libasr_amd_act$v.so asr_amd/csrc/*.hip &
done
wait
ls -la asr_amd/libasr_amd_act*.so
