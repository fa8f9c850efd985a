"""Pin the oracle against the reference decoder itself (oracle/_ref, built from
/root/reference by oracle/Makefile) on fresh seeded inputs, beyond the committed goldens.
Skipped where neither the reference tree nor a prebuilt oracle/_ref exists."""
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
