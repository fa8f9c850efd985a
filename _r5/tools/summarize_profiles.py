#!/usr/bin/env python3
"""Turn the rocprofv3 outputs under gpurun_out/prof/ (tools/profile_round.sh) into the small tracked files under profiles/.

    python tools/summarize_profiles.py r03        # round tag

Per configuration (headline / biglm / lattice_beam15): <tag>[_<config>]_kernel_stats.csv, <tag>[_<config>]_bench_under_rocprof.json,
and in <tag>_pmc_summary.json the FETCH_SIZE / WRITE_SIZE sums per kernel.  traffic_latest.json: per configuration and kernel
CLASS (expand / insert / closure = the three event classes of bench.py) the HBM bytes per launch, (2 x FETCH_SIZE + WRITE_SIZE) x
1024 (MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 128-B fabric requests at 64 B on gfx950)."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof")
OUT = os.path.join(ROOT, "profiles")


def klass(k):
    if k.startswith("expand_kernel"):
        return "expand"
    if k.startswith("insert_kernel"):
        return "insert"
    if k.startswith("closure_kernel") or k.startswith("lattice_prune"):
        return "closure"
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    os.makedirs(OUT, exist_ok=True)
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        head = "unknown"
    summary = {"unit": "KB as reported by rocprofv3 (FETCH_SIZE / WRITE_SIZE), summed over the launches of one bench step", "configs": {}}
    traffic = {"note": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch per kernel class: MI355X_MICROARCH.md HBM section (FETCH_SIZE counts "
                       "128-B fabric requests at 64 B on gfx950); separate --pmc passes (tools/profile_round.sh); round " + tag,
               "origin": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of round %s, summarised at commit %s" % (tag, head)}
    for name in ("headline", "biglm", "lattice_beam15"):
        suffix = "" if name == "headline" else "_" + name
        ks = sorted(glob.glob(os.path.join(P, "kt_" + name, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
        if ks:
            shutil.copy(ks[-1], os.path.join(OUT, "%s%s_kernel_stats.csv" % (tag, suffix)))
        bj = os.path.join(P, "bench_kt_%s.json" % name)
        if os.path.exists(bj) and os.path.getsize(bj) > 0:
            shutil.copy(bj, os.path.join(OUT, "%s%s_bench_under_rocprof.json" % (tag, suffix)))
        fj, wj = os.path.join(P, "fetch_%s.json" % name), os.path.join(P, "write_%s.json" % name)
        if not (os.path.exists(fj) and os.path.exists(wj)):
            continue
        f, w = json.load(open(fj)), json.load(open(wj))
        per = {}
        cls = {}
        for k in sorted(set(f) | set(w)):
            n = f.get(k, w.get(k))["launches"]
            fk, wk = f.get(k, {}).get("sum_kb", 0.0), w.get(k, {}).get("sum_kb", 0.0)
            per[k] = {"launches": n, "FETCH_SIZE_KB_per_launch": fk / n, "WRITE_SIZE_KB_per_launch": wk / n,
                      "hbm_bytes_per_launch_corrected": (2.0 * fk + wk) * 1024.0 / n}
            c = klass(k)
            if c:
                a = cls.setdefault(c, [0, 0.0])
                a[0] += n
                a[1] += (2.0 * fk + wk) * 1024.0
        summary["configs"][name] = per
        traffic[name] = {c + "_bytes_per_launch": v / n for c, (n, v) in cls.items()}
        traffic[name]["launches"] = {c: n for c, (n, v) in cls.items()}
    k2 = sorted(glob.glob(os.path.join(P, "kt_groups2", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if k2:
        shutil.copy(k2[-1], os.path.join(OUT, "%s_kernel_stats_two_groups_traced.csv" % tag))
    json.dump(traffic, open(os.path.join(OUT, "traffic_latest.json"), "w"), indent=1)
    json.dump(summary, open(os.path.join(OUT, "%s_pmc_summary.json" % tag), "w"), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
