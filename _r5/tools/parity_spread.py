#!/usr/bin/env python3
"""Where max_active / min_active bind (the reference service's 7000 / 200), how far is the order-free restatement -- what the GPU
computes bit for bit (tests/test_gpu_parity.py) -- from the reference, next to how far the reference is from ITSELF when only its
hash table size changes (hash_ratio 2, 2.5, 3: other visiting orders of the same algorithm, base-inl.h:188-226, 237-244)?
CPU only (oracle/_ref + oracle/_build): every pair among {reference @ 2, @ 2.5, @ 3, order-free oracle} over the 128 utterances of
bench.py's service-point workloads.

    python tools/parity_spread.py [--workload single|calibrated] [--utts 128] [--out profiles/r05_parity_spread.json]
"""
import argparse
import importlib
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=128)
    ap.add_argument("--states", type=int, default=2850000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_parity_spread.json"))
    a = ap.parse_args()
    import bench
    import pyoracle

    synth = importlib.import_module("asr-decoder_amd").synth
    gp = "/tmp/wfst_bench_graph_%d.bin" % a.states
    if os.path.exists(gp):
        g = synth.Graph.read(gp)
    else:
        g = synth.make_hclg_like(a.states, seed=7, n_tid=6000)
        g.write(gp)
    m = synth.default_tid2pdf(6000)
    pyoracle.build_oracle()
    ref, orc = pyoracle.RefDecoder(), pyoracle.OracleDecoder()
    out = {"what": __doc__.strip().split("\n\n")[0], "utterances": a.utts, "config": "beam 13, max_active 7000, min_active 200, lattice_beam 7",
           "workloads": {}}
    nth = os.cpu_count() or 1
    for name, mu in (("single_planted_path_mu_-2", -2.0), ("calibrated_mu_-2.6", -2.6)):
        mats = [synth.make_loglikes(g, 300, 3000, m, seed=u, mu=mu, sigma=1.0)[0] for u in range(a.utts)]
        cd = dict(beam=13.0, max_active=7000, min_active=200, lattice_beam=7.0, prune_interval=25, beam_delta=0.5)
        res = {}
        for hr in (2.0, 2.5, 3.0):
            res["reference@%g" % hr] = bench.cpu_decode_all(ref, gp, dict(cd, hash_ratio=hr), mats, m, nth)
        orc.set_order_free(True)
        res["order_free"] = bench.cpu_decode_all(orc, gp, cd, mats, m, nth)
        orc.set_order_free(False)
        as_gpu = lambda rs: [dict(ok=r.ok, words=r.words, tids=r.tids, tot_score=r.tot_score) for r in rs]
        pairs = {}
        for x, y in itertools.combinations(list(res), 2):
            d = bench.divergence(as_gpu(res[x]), res[y])
            pairs["%s vs %s" % (x, y)] = {k: d[k] for k in ("bit_identical", "same_words", "word_errors", "ref_words", "wer", "max_rel_cost_gap")}
            pairs["%s vs %s" % (x, y)]["cost_sign"] = {k: d["signed_rel_cost_gap"][k] for k in ("first_cheaper", "second_cheaper", "equal", "mean")}
        rr = [pairs[k]["wer"] for k in pairs if "order_free" not in k]
        oo = [pairs[k]["wer"] for k in pairs if "order_free" in k]
        out["workloads"][name] = {"pairs": pairs, "reference_vs_reference_wer": {"min": min(rr), "max": max(rr)},
                                  "order_free_vs_reference_wer": {"min": min(oo), "max": max(oo)},
                                  "order_free_inside_the_references_own_spread": bool(max(oo) <= max(rr))}
        print(name, json.dumps(out["workloads"][name]["reference_vs_reference_wer"]), json.dumps(out["workloads"][name]["order_free_vs_reference_wer"]), flush=True)
        for k, v in pairs.items():
            print("   %-34s identical %3d/%d  wer %.4f  cheaper %d / %d" % (k, v["bit_identical"], a.utts, v["wer"], v["cost_sign"]["first_cheaper"], v["cost_sign"]["second_cheaper"]), flush=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1, default=lambda o: o.item() if isinstance(o, np.generic) else str(o))


if __name__ == "__main__":
    main()
