#!/bin/bash
# does the random-atomic rate of the box move together with the bench figure?
run() { python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench', 'kernel_ms %.3f' % r['kernel_avg_ms'], 'copy %.0f' % r['box_copy_GBps'])"; }
for i in 1 2 3; do
  run
  timeout -k 10 100 tools/atomic_rate 2>&1 | grep -E "^range +16384|16 GiB, ordered by region \(    1 "
done
