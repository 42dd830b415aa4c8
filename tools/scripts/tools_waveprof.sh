#!/bin/bash
mkdir -p gpurun_out
export BC_JIT_CACHE=$PWD/gpurun_out/jitc
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
BC_JIT=force python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
BC_LHASH=0 BC_JIT=force python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for lh in 0 1; do
BC_LIB=$PWD/build_variants/libprof.so BC_JIT=force BC_LHASH=$lh timeout -k 10 200 python tools/wave_profile.py config3 20000000 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/waveprof.txt
done
run() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 120 python bench.py --reads 20000000 --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', 'kernel_ms %.3f' % d['roofline']['kernel_avg_ms'], 'Greads/s %.2f' % (d['roofline']['kernel_reads_per_s']/1e9))
" | tee -a gpurun_out/waveprof.txt
}
run "prev jit=0" BC_LIB=$PWD/build_variants/libprev.so BC_JIT=0
run "lhash=1 jit" BC_JIT=force BC_LHASH=1
run "lhash=0 jit" BC_JIT=force BC_LHASH=0
run "lhash=1 generic" BC_JIT=0 BC_LHASH=1
run "lhash=0 generic" BC_JIT=0 BC_LHASH=0
