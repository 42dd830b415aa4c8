#!/bin/bash
run() { python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'kernel_ms %.3f' % r['kernel_avg_ms'], 'copy %.0f' % r['box_copy_GBps'])"; }
for i in 1 2 3 4 5; do run "run $i"; done
