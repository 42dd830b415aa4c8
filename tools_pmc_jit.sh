#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for j in 0 1; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  OUT=$R/gpurun_out/pmcjit_${j}_$n
  BC_JIT=$j rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --reads 20000000 --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
  done
  python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmcjit_${j}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'match_count' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            agg['_vgpr'].append(float(r['VGPR_Count'])); agg['_sgpr'].append(float(r['SGPR_Count'])); agg['_scratch'].append(float(r['Scratch_Size'])); agg['_grid'].append(float(r['Grid_Size']))
print("BC_JIT=$j", " ".join("%s=%.4g" % (k.replace("SQ_",""), (sum(v)/len(v))/(312500 if k.startswith("SQ_INSTS") else 1)) for k,v in sorted(agg.items())))
PY
done
