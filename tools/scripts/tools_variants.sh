#!/bin/bash
# perf-debug: time variant builds of the engine library (build_variants/*.so) on the same box
for lib in build_variants/*.so; do
  for rep in 1 2; do
  BC_LIB=$PWD/$lib python bench.py --reads 20000000 --steps 5 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', 'kernel_ms %.3f' % d['roofline']['kernel_avg_ms'], 'Greads/s %.2f' % (d['roofline']['kernel_reads_per_s']/1e9))
"
  done
done
