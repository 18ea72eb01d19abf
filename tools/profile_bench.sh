#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of the default bench, then the two HBM-traffic
# PMC passes and two MFMA-utilisation passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).
# usage: tools/profile_bench.sh <tag>
set -e
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
# MFMA utilisation of the GEMMs (derived MfmaUtil, and the raw counters it is built from), own passes
timeout -s KILL 240 rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $OUT/pmc_mfmautil -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_mfmautil.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfmaraw -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/pmc_mfmaraw.log 2>&1
find $OUT -name "*.csv" | head -20
python3 $R/tools/summarize_prof.py $OUT $TAG $R/gpurun_out/prof_summary_$TAG
python3 $R/tools/timeline_gaps.py $(find $OUT/stats -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/prof_summary_$TAG/${TAG}_bench_timeline.txt 2>&1 || true
ls -la $R/gpurun_out/prof_summary_$TAG
