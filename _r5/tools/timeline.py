#!/usr/bin/env python3
"""Per-kernel durations and inter-kernel gaps of the frame loop from a rocprofv3 kernel trace."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].split("(")[0].replace("wfst::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "wfst::" in r["Kernel_Name"]]
loop = [k for k in ks if k[0] in ("expand_kernel", "plan_kernel", "insert_kernel", "closure_kernel")][-1201:]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for i, (n, s, e) in enumerate(loop):
    dur[n].append(e - s)
    if i:
        gap[n].append(s - loop[i - 1][2])
print("span ms %.2f over %d kernels" % ((loop[-1][2] - loop[0][1]) / 1e6, len(loop)))
for n in dur:
    print("%-16s mean %.1f us  sum %.2f ms | gap before: mean %.2f us  sum %.2f ms" % (
        n, sum(dur[n]) / len(dur[n]) / 1e3, sum(dur[n]) / 1e6, sum(gap[n]) / max(len(gap[n]), 1) / 1e3, sum(gap[n]) / 1e6))
