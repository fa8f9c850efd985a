"""The service's n-best on top of a raw lattice (SURVEY.md 8(f) rank 2 is not rebuilt; the reference's
own host functions run on the lattice this decoder returns).  Helper shared by the CPU and GPU
tests: write a lattice in the reference's on-disk format, run the reference pipeline
(oracle/_ref: Lattice::Read, LatticeCheckFormat, DeterminizeLatticeWrapper, NShortestPath,
ConvertNbestToVector, LatticeToVector) and compare with tests/golden/nbest_hclg600.npz, which holds
what the same pipeline gives on the REFERENCE decoder's own lattice."""
import os

import numpy as np

import pyoracle
from golden_util import GOLDEN_DIR


def golden_nbest(ci, ui):
    z = np.load(os.path.join(GOLDEN_DIR, "nbest_hclg600.npz"))
    key = "c%d_u%d_" % (ci, ui)
    lens = z[key + "lens"]
    words = np.split(z[key + "words"], np.cumsum(lens)[:-1]) if len(lens) else []
    return [(w, float(s[0]), float(s[1])) for w, s in zip(words, z[key + "scores"])], int(z["n"])


def check_nbest_of_lattice_bytes(ref, blob, ci, ui, tmp_path, what=""):
    want, n = golden_nbest(ci, ui)
    p = str(tmp_path / ("nbest_%d_%d.lat" % (ci, ui)))
    with open(p, "wb") as f:
        f.write(blob)
    got = pyoracle.ref_nbest_from_lattice_file(ref, p, 0, n)
    assert got is not None, what + " lattice rejected by the reference's LatticeCheckFormat / determinizer"
    paths = got[0]
    assert len(paths) == len(want), what + " number of paths"
    for k, (a, b) in enumerate(zip(paths, want)):
        assert np.array_equal(a[0], b[0]), "%s path %d words" % (what, k)
        assert abs(a[1] - b[1]) <= 1e-4 * abs(b[1]) and abs(a[2] - b[2]) <= 1e-4 * max(1.0, abs(b[2])), "%s path %d scores" % (what, k)
