"""Loader + comparators for tests/golden/*.npz (reference-generated vectors)."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_NAMES = ["hclg600", "quirk_parallel_arcs", "eps_chains", "no_final", "dead_end"]


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.z = z
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.graph_bytes = bytes(z["graph"])
        t = z["tid2pdf"]
        self.tid2pdf = t if t.size else None
        self.utts = [z["ll_%d" % i] for i in range(int(z["n_utt"]))]

    def write_graph(self, path):
        with open(path, "wb") as f:
            f.write(self.graph_bytes)
        return path

    def cases(self):
        for k, c in enumerate(self.meta["cases"]):
            md = dict(self.meta["modes"][c["mode"]])
            yield k, dict(self.meta["cfgs"][c["cfg"]]), md, c["utt"]

    def expected(self, k):
        p = "c%d_" % k
        e = {n[len(p):]: self.z[n] for n in self.z.files if n.startswith(p)}
        return e


def bits(a):
    return np.asarray(a, np.float32).view(np.int32)


def check_result(r, e, what="", check_counts=True):
    """Bit-exact comparison of a decode Result with a golden entry."""
    assert bool(r.ok) == bool(int(e["ok"])), what
    assert np.array_equal(r.words, e["words"]), what + " words"
    assert np.array_equal(r.tids, e["tids"]), what + " tids"
    assert np.array_equal(r.path_ilabel, e["path_ilabel"]), what + " path ilabels"
    assert np.array_equal(r.path_olabel, e["path_olabel"]), what + " path olabels"
    assert np.array_equal(bits(r.path_graph), bits(e["path_graph"])), what + " graph costs"
    assert np.array_equal(bits(r.path_ac), bits(e["path_ac"])), what + " acoustic costs"
    assert np.array_equal(bits([r.tot_score, r.lm_score]), bits(e["scores"])), what + " scores"
    if check_counts:
        assert [r.num_toks_end, r.num_links_end] == list(e["toks_links_end"]), what + " tok/link counts"
        if "frame_ntoks" in e and r.frame_ntoks is not None:
            assert np.array_equal(r.frame_ntoks, e["frame_ntoks"]), what + " tokens per frame"
            assert np.array_equal(bits(r.frame_best), bits(e["frame_best"])), what + " best cost per frame"
