#!/usr/bin/env python3
"""Development helper (with tools/det_bench.hip): raw lattices in the reference's on-disk format -> the flat files det_bench reads.
    python tools/det_bench_data.py out_dir lattice_file [lattice_file ...]
Needs oracle/_ref (the reference's Lattice::Read)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    ref = pyoracle.RefDecoder()
    for p in sys.argv[2:]:
        L = pyoracle.ref_lattice_read(ref, p, 0)
        q = os.path.join(out, os.path.splitext(os.path.basename(p))[0] + ".bin")
        with open(q, "wb") as f:
            np.asarray([L.n_states, len(L.a_src)], np.int32).tofile(f)
            np.asarray(L.st_final, np.int32).tofile(f)
            rec = np.zeros(len(L.a_src), dtype=[("src", "<i4"), ("dst", "<i4"), ("il", "<i4"), ("ol", "<i4"), ("g", "<f4"), ("ac", "<f4")])
            rec["src"], rec["dst"], rec["il"], rec["ol"], rec["g"], rec["ac"] = L.a_src, L.a_dst, L.a_il, L.a_ol, L.a_graph, L.a_ac
            rec.tofile(f)
        print(q, L.n_states, len(L.a_src))


if __name__ == "__main__":
    main()
