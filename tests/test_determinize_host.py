"""CPU: the lattice determinization code the device runs (asr-decoder_amd/csrc/wfst_determinize.h, compiled for the
host by tests/det_host.cc) against the reference's DeterminizeLatticeWrapper (newfst/lattice-determinize-api.cc:5-21,
newfst/lattice-determinize.h:300-1468): same number of states and final states, same arcs as a multiset -- labels
and (graph, acoustic) float costs bit for bit -- on the reference-generated goldens (tests/golden/det_hclg600.npz:
the reference's own raw lattices and what its determinizer made of them) and, where oracle/_ref is built, on
fresh lattices."""
import os

import numpy as np
import pytest

import pyoracle
from golden_util import GOLDEN_DIR


def _same(D, counts, arcs, what):
    assert [D.n_states, int(D.st_final.sum()), len(D.a_src)] == list(counts), what + " counts"
    assert np.array_equal(D.arc_multiset(), arcs), what + " arcs"
    assert np.all(D.a_il == 0), what
    # finals: exactly the extra states the final weights lead to; no arc leaves them
    fin = np.nonzero(D.st_final)[0]
    assert not np.isin(D.a_src, fin).any(), what


def test_host_build_reproduces_the_reference_goldens():
    lib = pyoracle.build_det_host()
    z = np.load(os.path.join(GOLDEN_DIR, "det_hclg600.npz"))
    n = 0
    for ci in z["cfgs"]:
        for ui in range(3):
            key = "c%d_u%d_" % (ci, ui)
            (L,) = pyoracle.parse_lattice_file(bytes(z[key + "raw"]))
            for low in (0, 1024, 5):   # (the closure's fast buffers: none / the device's size / so small that closures outgrow them)
                rc, D = pyoracle.det_host_run(lib, L, cap_scale=32, low_tmp=low)
                assert rc == 0, key
                _same(D, z[key + "counts"], z[key + "arcs"], key + " low %d" % low)
            n += 1
    assert n == 9


def test_host_build_equals_the_reference_on_fresh_lattices(refdec, synth, tmp_path):
    lib = pyoracle.build_det_host()
    n = 0
    for seed, (S, T, beam, lb) in enumerate([(600, 40, 13.0, 7.0), (600, 40, 13.0, 2.0), (3000, 60, 12.0, 5.0), (6000, 80, 11.0, 4.0)]):
        g = synth.make_hclg_like(S, seed=31 + seed, n_tid=600, n_words=500)
        m = synth.default_tid2pdf(600)
        gp = str(tmp_path / ("g%d.bin" % seed))
        g.write(gp)
        h = refdec.load_graph(gp)
        cd = dict(beam=beam, max_active=1000000, min_active=0, lattice_beam=lb)
        for u in range(3):
            ll = synth.make_loglikes(g, T, 300, m, seed=200 * seed + u, mu=-2.2)[0]
            p = str(tmp_path / "l.lat")
            if os.path.exists(p):
                os.remove(p)
            if not pyoracle.ref_lattice_write(refdec, h, pyoracle.Config(**cd), ll, p, m):
                continue
            R = pyoracle.ref_determinize_lattice_file(refdec, p, 0)
            L = pyoracle.ref_lattice_read(refdec, p, 0)
            rc, D = pyoracle.det_host_run(lib, L, cap_scale=32)
            if rc == 1 and os.environ.get("WFST_SYNTH_SEED_OFFSET", "0") not in ("", "0"):
                continue   # (another draw's lattice beyond this workspace: the unpruned construction is exponential on some)
            assert rc == 0 and R is not None
            _same(D, [R.n_states, int(R.st_final.sum()), len(R.a_src)], R.arc_multiset(), "seed %d utt %d" % (seed, u))
            n += 1
        refdec.free_graph(h)
    assert n >= 10


def test_workspace_overflow_is_reported():
    lib = pyoracle.build_det_host()
    z = np.load(os.path.join(GOLDEN_DIR, "det_hclg600.npz"))
    (L,) = pyoracle.parse_lattice_file(bytes(z["c0_u0_raw"]))
    rc, _ = pyoracle.det_host_run(lib, L, cap_scale=0)   # 1024-entry tables: too small for this lattice
    assert rc in (0, 1)
