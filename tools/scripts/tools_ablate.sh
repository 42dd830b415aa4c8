#!/bin/bash
# perf-debug: time the hot kernel with phases switched off (results are wrong when ablated)
for ab in ${ABL:-0x0 0x1 0x4 0x8 0x24 0x2c 0x2d 0x2f 0x6f 0xef}; do
  BC_ABLATE=$ab python bench.py --reads 20000000 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$ab', 'kernel_ms %.3f' % d['roofline']['kernel_avg_ms'], 'Greads/s %.2f' % (d['roofline']['kernel_reads_per_s']/1e9))
except Exception as e: print('$ab', 'failed', e)
"
done
