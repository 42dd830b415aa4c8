#!/bin/bash
# A/B of library builds on the default bench (100M reads per step): tools_ab.sh label=lib.so[,ENV=VAL...] ...
mkdir -p gpurun_out
export BC_JIT_CACHE=$PWD/gpurun_out/jitc
CFG=${CFG:-config3}
for rep in 1 2; do
for spec in "$@"; do
  label=${spec%%=*}; rest=${spec#*=}
  lib=${rest%%,*}; envs=""
  if [[ "$rest" == *,* ]]; then envs=$(echo "${rest#*,}" | tr ',' ' '); fi
  env BC_LIB=$PWD/$lib $envs timeout -k 10 300 python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$CFG $label', d['roofline']['kernel'], 'kernel_ms %.3f' % d['roofline']['kernel_avg_ms'], 'Greads/s %.2f' % (d['roofline']['kernel_reads_per_s']/1e9), 'value %.4g' % d['value'], 'copy %.0f' % (d['roofline'].get('box_copy_GBps') or 0), 'sclk %.0f' % (d['roofline'].get('sclk_mhz') or 0))
" | tee -a gpurun_out/ab.txt
done; done
