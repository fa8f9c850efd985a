#!/usr/bin/env python3
"""Per channel group: what its frame chain spent where in the profiled step (WFST_PROFILE_DUMP of a WFST_AB_SWITCHES build)."""
import collections, sys
KIND = {0: "expand", 1: "insert", 2: "closure", 10: "prune:raw", 11: "prune:walk", 12: "prune:flags", 13: "prune:move"}
rows = [l.strip().split(",") for l in open(sys.argv[1]) if l.strip()]
rows = [(int(k), int(g), float(a), float(b)) for k, g, a, b in rows]
groups = sorted({r[1] for r in rows})
print("launches %d, span %.2f ms" % (len(rows), max(r[3] for r in rows) - min(r[2] for r in rows)))
for g in groups:
    R = sorted([r for r in rows if r[1] == g], key=lambda r: r[2])
    tot = collections.defaultdict(float); cnt = collections.Counter(); mx = collections.defaultdict(float)
    gap = 0.0
    for i, (k, _, a, b) in enumerate(R):
        tot[k] += b - a; cnt[k] += 1; mx[k] = max(mx[k], b - a)
        if i: gap += max(0.0, a - R[i - 1][3])
    print("group %d: start %.2f end %.2f ms, kernels %.2f ms, gaps %.2f ms" % (g, R[0][2], R[-1][3], sum(tot.values()), gap))
    for k in sorted(tot):
        print("    %-12s n=%4d sum %7.2f ms  mean %7.1f us  max %7.1f us" % (KIND.get(k, str(k)), cnt[k], tot[k], 1e3 * tot[k] / cnt[k], 1e3 * mx[k]))
