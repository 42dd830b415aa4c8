#!/bin/bash
# perf-debug: generic vs scheme-specialised kernel, interleaved in one box
for rep in 1 2; do
for j in 0 1; do
  BC_JIT=$j python bench.py --reads 20000000 --steps 5 --warmup 2 --no-cpu 2>gpurun_out/jit_err_$j.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('BC_JIT=$j', 'kernel_ms %.3f' % d['roofline']['kernel_avg_ms'], 'Greads/s %.2f' % (d['roofline']['kernel_reads_per_s']/1e9), d['outcomes']['matched'])
"
done; done
tail -3 gpurun_out/jit_err_1.txt
