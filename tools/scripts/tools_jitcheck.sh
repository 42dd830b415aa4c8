#!/bin/bash
mkdir -p gpurun_out
BC_LHASH=0 BC_JIT=force BC_JIT_SRC=$PWD/gpurun_out/jit_real.hip BC_JIT_DUMP=$PWD/gpurun_out/jit_real.co timeout -k 10 120 python bench.py --reads 2000000 --steps 1 --warmup 1 --no-cpu > /dev/null 2>&1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes gpurun_out/jit_real.co | grep -E "private_segment|vgpr_count|sgpr_spill"
