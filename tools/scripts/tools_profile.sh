#!/bin/bash
# Collects the judged profile of one round: rocprofv3 kernel-trace stats of bench.py and, in separate
# passes (MI355X_MICROARCH.md rocprofv3 section), the HBM traffic counters.  Usage: tools_profile.sh <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/bench.err
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections, json
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if 'match_count' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
out={k:{"dispatches":len(v),"mean_per_dispatch":sum(v)/len(v)} for k,v in sorted(agg.items())}
f=out.get("FETCH_SIZE",{}).get("mean_per_dispatch"); w=out.get("WRITE_SIZE",{}).get("mean_per_dispatch")
if f is not None and w is not None:
    out["hbm_traffic_bytes_per_dispatch"]={"fetch_corrected_x2": f*1024*2, "write": w*1024, "total": f*1024*2+w*1024,
      "note": "FETCH_SIZE/WRITE_SIZE are KB; gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM), so it is doubled; the scattered 4-byte atomics are uncalibrated"}
out["_workload"]="bench.py default: config3, 100,000,000 reads per dispatch"
json.dump(out, open("$OUT/pmc_summary.json","w"), indent=1)
print(json.dumps(out, indent=1))
PY
cat $OUT/stats/*/*kernel_stats.csv | cut -c1-160
