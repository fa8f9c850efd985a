#!/usr/bin/env python3
"""Mean per-launch PMC counter values per kernel from a rocprofv3 --pmc run (csv)."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("wfst::", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k][r["Counter_Name"]] += 1
for k in agg:
    if "kernel" not in k:
        continue
    a = {c: v / n[k][c] for c, v in agg[k].items()}
    line = "%-16s launches %d  " % (k, max(n[k].values())) + "  ".join("%s=%.3g" % (c, v) for c, v in sorted(a.items()))
    if "SQ_WAVE_CYCLES" in a and a["SQ_WAVE_CYCLES"]:
        line += "  | wait/wave_cycles %.2f" % (a.get("SQ_WAIT_ANY", 0) / a["SQ_WAVE_CYCLES"])
        line += " active/wave_cycles %.3f" % (a.get("SQ_ACTIVE_INST_ANY", 0) / a["SQ_WAVE_CYCLES"])
    print(line)
