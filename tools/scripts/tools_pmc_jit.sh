#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export BC_JIT_CACHE=$R/gpurun_out/jitc
# warm the kernel cache outside the profiler
for lh in 0 1; do BC_LHASH=$lh BC_JIT=force python3 $R/bench.py --reads 2000000 --steps 1 --warmup 1 --no-cpu > /dev/null 2>&1; done
cd /tmp
for cfg in "0 0" "force 0" "force 1"; do
  set -- $cfg; j=$1; lh=$2
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  n=$(echo $set | cut -d' ' -f1)
  OUT=$R/gpurun_out/pmcjit_${j}_${lh}_$n
  BC_LHASH=$lh BC_JIT=$j timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --reads 20000000 --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
  echo "done $j $lh $n" >> $R/gpurun_out/pmc_progress.txt
  done
  python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmcjit_${j}_${lh}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'match_count' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            agg['_vgpr'].append(float(r['VGPR_Count'])); agg['_lds'].append(float(r['LDS_Block_Size']))
print("BC_JIT=$j LHASH=$lh", " ".join("%s=%.4g" % (k.replace("SQ_",""), (sum(v)/len(v))/312500) for k,v in sorted(agg.items())))
PY
done
