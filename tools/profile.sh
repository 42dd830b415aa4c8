#!/bin/bash
# tools/profile.sh <tag> [config]: the judged profile of one workload (run it under gpurun).
#   1. a plain bench run (also warms the kernel cache: nothing is compiled under the profiler),
#   2. rocprofv3 --kernel-trace --stats of the same command,
#   3. PMC counters in passes of their own (MI355X_MICROARCH.md, rocprofv3 section: FETCH_SIZE and WRITE_SIZE cannot
#      share a pass; no tracing options next to --pmc).
# Results: gpurun_out/profile_<tag>/{bench.json, bench_under_rocprof.json, kernel_stats.csv, pmc_summary.json}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
CFG=${2:-config3}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
# the driver's own step / warm-up counts (python3 bench.py --gpus 1 --steps 20 --warmup 5), so that the profiled run is the
# judged run; STEPS / WARMUP override
ARGS="--config $CFG --no-cpu --no-extra --steps ${STEPS:-20} --warmup ${WARMUP:-5}"
python3 $R/bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
export BC_JIT=cached
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/bench_under_rocprof.json 2>> $OUT/bench.err
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --config $CFG --no-cpu --no-extra --steps 3 --warmup 1 > /dev/null 2>> $OUT/bench.err
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "match_count" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {"dispatches": len(v), "mean_per_dispatch": sum(v) / len(v)} for k, v in sorted(agg.items())}
g = lambda k: out.get(k, {}).get("mean_per_dispatch")
f, w = g("FETCH_SIZE"), g("WRITE_SIZE")
if f is not None and w is not None:
    out["hbm_traffic_bytes_per_dispatch"] = {
        "fetch_corrected_x2": f * 1024 * 2, "write": w * 1024, "total": f * 1024 * 2 + w * 1024,
        "note": "FETCH_SIZE/WRITE_SIZE are KB; gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads "
                "(MI355X_MICROARCH.md, HBM), so it is doubled; the scattered 4-byte atomics are uncalibrated"}
if g("SQ_WAVE_CYCLES") and g("SQ_BUSY_CYCLES"):
    d = {}
    # quad-cycle counters (MI355X_MICROARCH.md): shares of a wave's lifetime
    wc = g("SQ_WAVE_CYCLES")
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA",
              "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
        if g(k) is not None:
            d[k + "_share_of_wave_cycles"] = g(k) / wc
    if g("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share_of_lds_cycles"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        d["l2_hit_rate"] = g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("SQ_WAVES"):
        d["waves_per_dispatch"] = g("SQ_WAVES")
    out["derived"] = d
b = json.load(open("$OUT/bench.json"))
out["_workload"] = "bench.py --config %s: %s reads per dispatch" % ("$CFG", "{:,}".format(b["config"]["reads_per_step_per_gpu"]))
out["_kernel"] = b["roofline"]["kernel"]
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
print(json.dumps(out.get("derived"), indent=1))
PY
cut -c1-170 $OUT/kernel_stats.csv | head -8
