"""Summary of a rocprofv3 kernel (+ memory copy) trace of the drop-in CLI: how long the device was busy (union of kernel intervals)
inside the decode window, the idle gaps by size, kernels by name.  tools/dropin_trace.sh writes the trace."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main(d):
    kf = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in kf:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    rows.sort()
    # the decode window: from the first expansion launch to the last kernel
    first = next(i for i, r in enumerate(rows) if "expand" in r[2])
    rows = rows[first:]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, gaps, cur_s, cur_e = 0, [], rows[0][0], rows[0][1]
    for s, e, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e - t0))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    by = defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        by[n][0] += 1
        by[n][1] += e - s
    out = {"window_ms": (t1 - t0) / 1e6, "busy_ms": busy / 1e6, "idle_ms": (t1 - t0 - busy) / 1e6, "kernels": len(rows)}
    edges = [0, 5e3, 2e4, 1e5, 5e5, 2e6, 1e12]
    hist = []
    for lo, hi in zip(edges[:-1], edges[1:]):
        g = [x for x, _ in gaps if lo <= x < hi]
        hist.append({"gap_us": "%g-%g" % (lo / 1e3, hi / 1e3), "n": len(g), "ms": sum(g) / 1e6})
    out["gaps"] = hist
    out["largest_gaps_ms_at_ms"] = [(round(g / 1e6, 3), round(at / 1e6, 1)) for g, at in sorted(gaps, reverse=True)[:12]]
    out["by_kernel"] = {n: {"n": c, "ms": round(t / 1e6, 2), "avg_us": round(t / c / 1e3, 1)} for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:10]}
    mf = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
    cp = defaultdict(lambda: [0, 0, 0])
    for f in mf:
        for r in csv.DictReader(open(f)):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if s < t0:
                continue
            k = r.get("Direction", "?")
            cp[k][0] += 1
            cp[k][1] += e - s
            cp[k][2] += int(r.get("Bytes", 0) or 0)
    out["copies"] = {k: {"n": c, "ms": round(t / 1e6, 2), "MB": round(b / 1e6, 1)} for k, (c, t, b) in cp.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
