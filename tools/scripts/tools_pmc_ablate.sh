#!/bin/bash
# perf-debug: VALU/SALU instruction counts of the hot kernel per ablation mask (counts are additive, times are not)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for ab in ${ABL:-0x0 0x1 0x4 0x8 0x24 0x2c 0x2d 0x6d 0xed 0xfd}; do
  OUT=$R/gpurun_out/pmcab_$ab
  BC_ABLATE=$ab rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT -- python3 $R/bench.py --reads 20000000 --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'match_count' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
print("$ab", " ".join("%s=%.0f" % (k.replace("SQ_INSTS_",""), sum(v)/len(v)/312500) for k,v in sorted(agg.items())), "(per 64-read wave tile)")
PY
done
