import json, sys
b=json.load(open(sys.argv[1])); r=b["roofline"]
print("value %.4g ms/step %.3f kernel_avg %.3f frac %.4f first %.3f med %.3f last %.3f max %.3f" % (b["value"], b["ms_per_step"], r["kernel_avg_ms"], r["frac"], r["kernel_ms_first"], r["kernel_ms_median"], r["kernel_ms_last"], r["kernel_ms_max"]))
print(r["kernel_ms_each"]); print(b["reset_ms"], b["reset_in_region_ms"], b["config"]["resets_in_region"], b.get("table_rows_at_end"), b["outcomes"])
