#!/bin/bash
# tools/ab.sh <lib>...: the headline bench (config 3, no CPU leg, no extras) once per library, on the same box
# (run it under gpurun; BC_LIB selects the library, each compiles its own specialised kernel during the warm-up)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
for rep in 1 2; do
for lib in "$@"; do
  name=$(basename "$lib" .so)
  BC_LIB="$PWD/$lib" BC_JIT_CACHE="$PWD/gpurun_out/ab/cache_$name" python bench.py --no-cpu --no-extra ${AB_ARGS} > gpurun_out/ab/$name.$rep.json 2> gpurun_out/ab/$name.$rep.err
  python - "$name" gpurun_out/ab/$name.$rep.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    r = d["roofline"]
    print("%-28s %s  kernel %.3f ms  step %.3f ms  frac %.3f  %s" % (sys.argv[1], d["config"]["config"], r["kernel_avg_ms"], d["ms_per_step"], r["frac"], r["kernel"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
done
