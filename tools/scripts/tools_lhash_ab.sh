#!/bin/bash
# A/B of the LDS exact-match table, pipelined fetch and the specialised kernel
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
BC_JIT=force python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
run() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 120 python bench.py --reads 20000000 --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', 'kernel_ms %.3f' % d['roofline']['kernel_avg_ms'], 'Greads/s %.2f' % (d['roofline']['kernel_reads_per_s']/1e9))
" | tee -a gpurun_out/lhash_ab.txt
}
run "prev jit=0" BC_LIB=$PWD/build_variants/libprev.so BC_JIT=0
for lh in 0 1; do
for j in 0 cached; do
  run "lhash=$lh BC_JIT=$j" BC_LHASH=$lh BC_JIT=$j
done; done
ls -la ngs-barcode-count_amd/csrc/jit_cache/ | tail -5
