#!/bin/bash
# every BASELINE workload, generic vs specialised kernel, LDS tables off/on
mkdir -p gpurun_out
export BC_JIT_CACHE=$PWD/gpurun_out/jitc
run() { # label, config, reads, env...
  local label=$1 cfg=$2 n=$3; shift 3
  env "$@" timeout -k 10 200 python bench.py --config $cfg --reads $n --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg $label', 'kernel_ms %.3f' % d['roofline']['kernel_avg_ms'], 'Greads/s %.2f' % (d['roofline']['kernel_reads_per_s']/1e9), 'value %.3g' % d['value'])
" | tee -a gpurun_out/allcfg.txt
}
for cfg in config5; do
  n=20000000
  run "prev" $cfg $n BC_LIB=$PWD/build_variants/libprev.so BC_JIT=0
  run "generic" $cfg $n BC_JIT=0
  run "jit lhash=0" $cfg $n BC_JIT=force BC_LHASH=0
  run "jit lhash=1" $cfg $n BC_JIT=force BC_LHASH=1
done
