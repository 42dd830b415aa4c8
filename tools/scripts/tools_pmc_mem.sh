#!/bin/bash
# perf-debug: memory-path counters of the hot kernel (generic kernel, JIT off)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for set in "GRBM_GUI_ACTIVE GRBM_TA_BUSY" "GRBM_TC_BUSY GRBM_EA_BUSY" "TCC_REQ_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LEVEL_WAVES SQ_WAVE_CYCLES"; do
  n=$(echo $set | cut -d' ' -f1)
  echo "set: $set" >> $R/gpurun_out/pmcmem_progress.txt
  BC_JIT=${BC_JIT:-0} BC_ABLATE=${BC_ABLATE:-0} timeout -k 10 90 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcmem_$n -- python3 $R/bench.py --reads 20000000 --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pmcmem_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'match_count' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print("%-40s %.4g" % (k, sum(v)/len(v)))
PY
