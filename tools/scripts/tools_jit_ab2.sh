#!/bin/bash
for ab in 0x0 0x4 0x8 0xc 0x2c; do
for j in 0 1; do
  BC_ABLATE=$ab BC_JIT=$j python bench.py --reads 20000000 --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ablate=$ab BC_JIT=$j', 'kernel_ms %.3f' % d['roofline']['kernel_avg_ms'], 'Greads/s %.2f' % (d['roofline']['kernel_reads_per_s']/1e9))
"
done; done
