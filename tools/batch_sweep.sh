#!/bin/bash
# cfg3 training throughput on one GPU against the rows per GPU: the scans carry two 16-row
# chunks per workgroup above 16 rows; ASR_SCAN_NC=1 lines show the one-chunk-per-workgroup path on the same box; above 32 rows the batch runs
# as length-sorted micro-batches of 32 rows with gradient accumulation (bench.py --micro-batch).
#   tools/batch_sweep.sh > gpurun_out/batch_sweep.jsonl     (copied to profiles/r03_batch_sweep.jsonl)
cd "$(dirname "$0")/.."
for spec in "0 16" "0 32" "1 32" "0 64" "0 128"; do
  set -- $spec
  if [ "$1" = 1 ]; then export ASR_SCAN_NC=1; else unset ASR_SCAN_NC; fi
  timeout -k 10 400 python bench.py --steps 6 --warmup 2 --global-batch $2 --no-cpu-baseline --pool 2 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); d['ASR_SCAN_NC']='$1' if '$1'=='1' else 'default'; print(json.dumps(d))" || exit 1
done
