#!/bin/bash
# how the bench figure moves with what the GPU did just before (heat? memory state?)
run() { python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'kernel_ms %.3f' % r['kernel_avg_ms'], 'copy %.0f' % r['box_copy_GBps'], 'sclk %.0f' % r['sclk_mhz'])"; }
run "cold start"
run "straight after"
run "straight after"
python bench.py --steps 300 --warmup 3 --no-cpu > /dev/null 2>&1   # ~2.5 s of solid load
run "after 300 steps of load"
sleep 30; run "after 30 s idle"
sleep 90; run "after 90 s idle"
rocm-smi --showtemp 2>/dev/null | grep -i "junction\|memory\|hbm" | head -6
