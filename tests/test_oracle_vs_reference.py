"""Pin the oracle against the reference decoder itself (oracle/_ref, built from
/root/reference by oracle/Makefile) on fresh seeded inputs, beyond the committed goldens.
Skipped where neither the reference tree nor a prebuilt oracle/_ref exists."""
import os

import numpy as np
import pytest

import pyoracle
from golden_util import bits


def _same(a, b):
    assert a.ok == b.ok
    assert np.array_equal(a.words, b.words) and np.array_equal(a.tids, b.tids)
    assert np.array_equal(a.path_ilabel, b.path_ilabel) and np.array_equal(a.path_olabel, b.path_olabel)
    assert np.array_equal(bits(a.path_graph), bits(b.path_graph))
    assert np.array_equal(bits(a.path_ac), bits(b.path_ac))
    assert np.array_equal(bits([a.tot_score, a.lm_score]), bits([b.tot_score, b.lm_score]))
    assert (a.num_toks_end, a.num_links_end) == (b.num_toks_end, b.num_links_end)
    if a.frame_ntoks is not None:
        assert np.array_equal(a.frame_ntoks, b.frame_ntoks)
        assert np.array_equal(bits(a.frame_best), bits(b.frame_best))


CFGS = [
    dict(beam=13.0, max_active=1000000, min_active=0, lattice_beam=7.0),
    dict(beam=13.0, max_active=7000, min_active=200, lattice_beam=7.0),
    dict(beam=10.0, max_active=1500, min_active=200, lattice_beam=7.0),
    dict(beam=4.0, max_active=100000, min_active=2000, lattice_beam=2.0, prune_interval=10),
    dict(beam=15.0, max_active=4000, min_active=0, lattice_beam=10.0, beam_delta=0.25, hash_ratio=1.5),
]


@pytest.mark.parametrize("ci", range(len(CFGS)))
def test_oracle_matches_reference_bitwise(ci, oracle, refdec, synth, tmp_path, capfd):
    g = synth.make_hclg_like(9000, seed=3)
    path = str(tmp_path / "g.bin")
    g.write(path)
    m = synth.default_tid2pdf(6000)
    hr, ho = refdec.load_graph(path), oracle.load_graph(path)
    cfg = pyoracle.Config(**CFGS[ci])
    for seed in range(2):
        ll, _ = synth.make_loglikes(g, 90, 3000, m, seed=100 + seed, mu=-2.3 - 0.4 * seed)
        for kw in (dict(trace=True), dict(chunk=0), dict(chunk=11, finalize=False),
                   dict(chunk=0, finalize=False, use_final_probs=False)):
            _same(refdec.decode(hr, cfg, ll, m, **kw), oracle.decode(ho, cfg, ll, m, **kw))
    refdec.free_graph(hr)
    oracle.free_graph(ho)


@pytest.mark.parametrize("block", range(4))
def test_oracle_matches_reference_on_random_graphs(block, oracle, refdec, synth, tmp_path):
    """The graphs of tests/test_gpu_fuzz.py (dense forward epsilon structure, parallel arcs, several
    final states, tight beams): best path, token/link counts and the raw lattice's arc multiset of
    the oracle against the reference decoder itself."""
    from test_gpu_fuzz import random_graph

    rng = np.random.default_rng(int(os.environ.get("WFST_FUZZ_SEED", "99")) + block)   # WFST_FUZZ_SEED: other campaigns
    n = 0
    for case in range(10):
        n_states = int(rng.integers(4, 70))
        n_labels = int(rng.integers(3, 12))
        g = random_graph(synth, rng, n_states, n_labels)
        path = str(tmp_path / ("g%d.bin" % case))
        g.write(path)
        hr, ho = refdec.load_graph(path), oracle.load_graph(path)
        cd = dict(beam=float(rng.uniform(3.0, 14.0)), max_active=int(rng.choice([1000000, 40, 12])), min_active=int(rng.choice([0, 5])),
                  lattice_beam=float(rng.uniform(0.5, 8.0)), prune_interval=int(rng.integers(3, 30)))
        cfg = pyoracle.Config(**cd)
        for T in (int(rng.integers(1, 45)), int(rng.integers(1, 45))):
            x = rng.normal(-1.5, 1.0, size=(T, n_labels + 1)).astype(np.float32)
            r = refdec.decode(hr, cfg, x, None, chunk=0)
            if not r.ok:       # the reference aborts in PruneForwardLinks when every token died; skip those
                continue
            _same(r, oracle.decode(ho, cfg, x, None, chunk=0))
            R = pyoracle.ref_raw_lattice(refdec, hr, cfg, x, None)
            O = pyoracle.oracle_raw_lattice(oracle, ho, cfg, x, None)
            assert R.ok == O.ok
            if R.ok:
                assert (R.n_states, int(R.st_final.sum())) == (O.n_states, int(O.st_final.sum()))
                assert np.array_equal(R.arc_multiset(), O.arc_multiset())
            n += 1
        refdec.free_graph(hr)
        oracle.free_graph(ho)
    assert n >= 10
