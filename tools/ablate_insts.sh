#!/bin/bash
# tools/ablate_insts.sh <config> <mask> [<mask> ...]: wave-instructions per 64-read tile of the match kernel with phases
# switched off (BC_ABLATE masks of an -DBC_EXPERIMENT build: counts are then wrong, only the instruction mix is of
# interest) -- where the VALU work of a VALU-bound kernel goes.  Run under gpurun; results in gpurun_out/ablate_<config>.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CFG=$1; shift
OUT=$R/gpurun_out/ablate_$CFG
mkdir -p $OUT
export BC_LIB=$R/build_variants/libexp.so
: > $OUT.txt
for m in "$@"; do
  export BC_ABLATE=$m BC_JIT_CACHE=$OUT/cache_$m
  mkdir -p $BC_JIT_CACHE
  python3 $R/bench.py --config $CFG --no-cpu --no-extra --steps 2 --warmup 1 > $OUT/plain_$m.json 2> $OUT/err_$m.txt || { tail -3 $OUT/err_$m.txt; continue; }
  (cd /tmp && BC_JIT=cached rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_$m -- python3 $R/bench.py --config $CFG --no-cpu --no-extra --steps 2 --warmup 1 > /dev/null 2>> $OUT/err_$m.txt)
  python3 - <<PY >> $OUT.txt
import csv, glob, json, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_$m/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "match_count" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
b = json.load(open("$OUT/plain_$m.json"))
tiles = b["config"]["reads_per_step_per_gpu"] / 64
g = lambda k: sum(agg[k]) / max(1, len(agg[k])) / tiles
print("ablate %-10s  %.3f ms   VALU %6.0f  SALU %5.0f  LDS %4.0f  VMEM %4.0f per tile" % ("$m", b["roofline"]["kernel_avg_ms"], g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM")))
PY
done
cat $OUT.txt
