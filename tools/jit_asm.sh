#!/bin/bash
# tools/jit_asm.sh <config> <out-prefix> [stride read_len]: rebuilds the library, compiles the specialised kernel of a
# workload and disassembles it (<out-prefix>.s); prints the register counts
set -e
cd "$(dirname "$0")/.."
make -C ngs-barcode-count_amd/csrc -s all
python tools/jit_dump.py "$1" "$2.co" ${3:-100} ${4:-${3:-100}} | tail -1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$2.co" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$2.elf"
/opt/rocm/lib/llvm/bin/llvm-objdump -d "$2.elf" > "$2.s"
wc -l < "$2.s"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$2.elf" | grep -E "vgpr_count|sgpr_count|spill_count|private_segment_fixed"
