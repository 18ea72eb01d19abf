#!/bin/bash
# Round-end measurement campaign on the GPU box (one gpurun call per stage keeps each under the limits):
#   tools/final_measure.sh suite | bench | sweep | configs | soak | dp
TAG=r03
cd "$(dirname "$0")/.."
O=gpurun_out
case "$1" in
  suite)   timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 900 --timeout-method thread > $O/${TAG}_gpu_suite.log 2>&1; rc=$?
           tail -4 $O/${TAG}_gpu_suite.log; python -c "import __graft_entry__ as g; g.smoke()" >> $O/${TAG}_gpu_suite.log 2>&1 || rc=1; tail -1 $O/${TAG}_gpu_suite.log; exit $rc;;
  bench)   timeout -k 10 500 python bench.py > $O/${TAG}_bench_line.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }; cat $O/${TAG}_bench_line.json
           timeout -k 10 300 python bench.py --length-sort --no-cpu-baseline --steps 20 --warmup 5 > $O/${TAG}_bench_line_length_sort.json 2>> $O/bench.err || exit 1; cut -c1-160 $O/${TAG}_bench_line_length_sort.json;;
  sweep)   tools/batch_sweep.sh > $O/${TAG}_batch_sweep.jsonl 2> $O/sweep.err || { tail -5 $O/sweep.err; exit 1; }
           python -c "
import json
for l in open('$O/${TAG}_batch_sweep.jsonl'):
    d=json.loads(l); r=d['roofline']; print(d['config']['per_gpu_batch'], d['ASR_SCAN_NC'], d['config']['micro_batches'], round(d['value']), round(d['ms_per_step'],1), round(r['avg_launch_ms'],2), round(r['us_per_timestep'],2), round(r['fwd_kernel_avg_launch_ms'],2))";;
  configs) : > $O/${TAG}_other_configs.jsonl
           for spec in "1 16" "2 16" "4 16" "5 16" "1 32" "2 32" "4 32" "5 32"; do set -- $spec
             timeout -k 10 300 python bench.py --cfg $1 --batch $2 --steps 8 --warmup 3 --pool 3 2>> $O/configs.err >> $O/${TAG}_other_configs.jsonl || { tail -5 $O/configs.err; exit 1; }; done
           for c in 1 2; do timeout -k 10 300 python bench.py --cfg $c --forward-only --steps 8 --warmup 3 --pool 3 2>> $O/configs.err >> $O/${TAG}_other_configs.jsonl || exit 1; done
           python -c "
import json
for l in open('$O/${TAG}_other_configs.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:40], d['config']['per_gpu_batch'], round(d['value']), round(d['ms_per_step'],1))";;
  soak)    timeout -k 10 1000 python -u tools/soak_determinism.py 40 > $O/${TAG}_soak_determinism.log 2>&1; rc=$?; tail -20 $O/${TAG}_soak_determinism.log; exit $rc;;
  dp)      : > $O/${TAG}_dp_gpu_check.log
           # two processes on the one GPU, gloo: bounded (a run that stalls is killed after 120 s) and printing as it goes
           for ov in 1 0; do echo "== ASR_DP_OVERLAP=$ov" | tee -a $O/${TAG}_dp_gpu_check.log
             ASR_DP_OVERLAP=$ov timeout -k 10 120 python -u -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29533 + ov)) tools/dp_gpu_check.py 2>&1 | grep -v "^W1\|^\[W\|^\*\*\*\|amdgpu.ids" | tee -a $O/${TAG}_dp_gpu_check.log
             [ ${PIPESTATUS[0]} -eq 0 ] || exit 1; done;;
  *) echo "usage: $0 suite|bench|sweep|configs|soak|dp"; exit 2;;
esac
