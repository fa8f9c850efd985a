#!/usr/bin/env python3
"""Turn the rocprofv3 outputs under gpurun_out/prof/ into the small tracked files under profiles/.

    python tools/summarize_profiles.py r01        # round tag

Inputs (produced on the GPU box, see profiles/README.md for the exact commands):
  gpurun_out/prof/kt/*/..._kernel_stats.csv            rocprofv3 --kernel-trace --stats
  gpurun_out/prof/pmc_fetch|pmc_write/*/..._counter_collection.csv   separate --pmc passes
  gpurun_out/prof/calib/*/..._counter_collection.csv   FETCH_SIZE calibration (tools/ubench_fetch_calib)
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof")
OUT = os.path.join(ROOT, "profiles")


def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:] if fs else []


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        m = re.search(r"(\w+_kernel|calib_\w+)", r["Kernel_Name"])  # "void wfst::insert_kernel<false>(...)" -> insert_kernel
        k = m.group(1) if m else r["Kernel_Name"].split("(")[0].replace("wfst::", "")
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return {k: {"launches": n, "kb_per_launch": v / n} for k, (n, v) in agg.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(OUT, exist_ok=True)
    ks = newest(os.path.join(P, "kt", "*", "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], os.path.join(OUT, "%s_kernel_stats.csv" % tag))
    k2 = newest(os.path.join(P, "kt_groups2", "*", "*_kernel_stats.csv"))
    if k2:
        shutil.copy(k2[0], os.path.join(OUT, "%s_kernel_stats_two_groups_traced.csv" % tag))
    kl = newest(os.path.join(P, "kt_lattice", "*", "*_kernel_stats.csv"))
    if kl:
        shutil.copy(kl[0], os.path.join(OUT, "%s_lattice_mode_kernel_stats.csv" % tag))
    bj = os.path.join(P, "bench_kt.json")
    if os.path.exists(bj):
        shutil.copy(bj, os.path.join(OUT, "%s_bench_under_rocprof.json" % tag))
    fetch = newest(os.path.join(P, "pmc_fetch", "*", "*_counter_collection.csv"))
    write = newest(os.path.join(P, "pmc_write", "*", "*_counter_collection.csv"))
    summary = {"unit": "KB per launch as reported by rocprofv3 (FETCH_SIZE / WRITE_SIZE)", "kernels": {}}
    calib = newest(os.path.join(P, "calib", "*", "*_counter_collection.csv"))
    if calib:
        c = per_kernel(calib[0])
        summary["fetch_size_calibration"] = {
            "stream16_1GiB_reported_KB": c.get("calib_stream16", {}).get("kb_per_launch"),
            "gather16_8388608_requests_reported_KB": c.get("calib_gather16", {}).get("kb_per_launch"),
            "gather8_8388608_requests_reported_KB": c.get("calib_gather8", {}).get("kb_per_launch"),
            "reading": "a 16 B/lane coalesced stream is reported at exactly 1/2 of its bytes (the guide's gfx950 "
                       "correction); a random 8 or 16 B gather is reported as 64 B per request",
        }
    traffic = {}
    if fetch and write:
        f, w = per_kernel(fetch[0]), per_kernel(write[0])
        for k in sorted(set(f) | set(w)):
            if not k.endswith("_kernel"):
                continue
            fk, wk = f.get(k, {}).get("kb_per_launch", 0.0), w.get(k, {}).get("kb_per_launch", 0.0)
            summary["kernels"][k] = {"launches": f.get(k, {}).get("launches"), "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
                                     "hbm_bytes_per_launch_raw": (fk + wk) * 1024.0,
                                     "hbm_bytes_per_launch_corrected": (2.0 * fk + wk) * 1024.0}
            traffic[k.replace("_kernel", "") + "_bytes_per_launch"] = (2.0 * fk + wk) * 1024.0
        traffic["note"] = ("(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch: MI355X_MICROARCH.md HBM section (FETCH_SIZE counts "
                           "128-B fabric requests at 64 B on gfx950); separate --pmc passes; round " + tag)
        try:
            import subprocess
            head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
        except Exception:
            head = "unknown"
        traffic["origin"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of round %s, summarised at commit %s" % (tag, head)
        json.dump(traffic, open(os.path.join(OUT, "traffic_latest.json"), "w"), indent=1)
    json.dump(summary, open(os.path.join(OUT, "%s_pmc_summary.json" % tag), "w"), indent=1)
    print(json.dumps(summary, indent=1)[:1500])


if __name__ == "__main__":
    main()
