"""-m gpu: PruneActiveTokens on the device: every prune_interval
frames the recorded forward links are pruned backwards by lattice_beam and the token arena / link store are
COMPACTED, so that (a) GetRawLattice can be served at any time, like the reference's (base-inl.h:869-975, called
mid-utterance by kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:58,81), and (b) lattice mode runs in bounded
memory whatever the utterance length.

The reference walks backwards while a frame's extra costs moved by more than delta = lattice_beam *
prune_scale (base-inl.h:439-480, 541-542), judging sweep by sweep over its token list -- with delta > 0 an
order-dependent judgement where surviving tokens of a frame are chained by epsilon links.  The device judges on
the frame's exact fixpoint against the values before the pass; the oracle's order-free mode states exactly that
(oracle/wfst_oracle.c prune_forward_links) and is pinned to the reference on the goldens
(tests/test_oracle_lattice.py).  Against it the mid-utterance lattices are IDENTICAL state by state, for the
default prune_scale and for prune_scale -> 0 (the exact walk); after FinalizeDecoding (delta 0 in the reference
too) the lattices are identical in any case.
"""
import numpy as np
import pytest

import pyoracle
from golden_util import bits

pytestmark = pytest.mark.gpu


def _setup(synth, tmp_path, n_states=6000, seed=13):
    import gpu_util as G

    g = synth.make_hclg_like(n_states, seed=seed, n_tid=2000, n_words=3000)
    m = synth.default_tid2pdf(2000)
    path = str(tmp_path / "g.bin")
    g.write(path)
    graph = G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    return G, g, m, path, graph


def _same_lattice(L, O, what):
    from test_gpu_lattice import nodes

    assert np.array_equal(nodes(L), nodes(O)), what + " states"
    assert np.array_equal(L.labelled_arcs(), O.labelled_arcs()), what + " arcs"


@pytest.mark.parametrize("raw_pass", [False, True])
@pytest.mark.parametrize("prune_scale", [1e-9, 0.1])
def test_mid_utterance_raw_lattice(prune_scale, raw_pass, synth, oracle, tmp_path):
    """raw_pass: wfst_options.debug 0x800 -- the never-priced frames of EVERY channel go through lattice_prune_raw_kernel (several
    workgroups per channel, meetings at a counter), which by default only channels with 800 k such links take; same lattices."""
    from test_gpu_lattice import as_raw

    G, g, m, path, graph = _setup(synth, tmp_path)
    cd = dict(beam=11.0, max_active=1000000, min_active=0, lattice_beam=5.0, prune_interval=10, prune_scale=prune_scale)
    T = [97, 64, 97, 31]
    mats = [synth.make_loglikes(g, t, 1000, m, seed=40 + i, mu=-2.3)[0] for i, t in enumerate(T)]
    dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(mats), max_frames=128, max_tokens_per_frame=32768,
                                 arena_tokens=1 << 20, lattice_links=1 << 21,
                                 options=G.wfstdec.Options(debug=0x800 | 0x300) if raw_pass else None)   # (0x300: and 8 workgroups per channel in the closure launches)
    dev = G.upload(mats)
    dec.init()
    h = oracle.load_graph(path)
    n_checked = 0
    try:
        oracle.set_order_free(True)
        # after InitDecoding only: the reference asserts num_frames > 0 (base-inl.h:896); the device returns "no lattice"
        assert dec.raw_lattice(0, True) is None
        for r in (7, 10, 11, 25, 40, 41, 64, 90, 97):
            dec.advance([t.data_ptr() for t in dev], [min(r, t) for t in T], 1000)
            for c in range(len(mats)):
                k = min(r, T[c])
                for ufp in (True, False):
                    d = dec.raw_lattice(c, ufp)
                    O = pyoracle.oracle_raw_lattice(oracle, h, pyoracle.Config(**cd), mats[c][:k], m, finalize=False, use_final_probs=ufp)
                    assert (d is not None) == O.ok, (r, c, ufp)
                    if d is None:
                        continue
                    L = as_raw(d)
                    what = "frame %d channel %d use_final_probs %s" % (r, c, ufp)
                    assert L.st_frame[0] == 0 and L.st_frame.max() == k and np.all(L.a_dst > L.a_src), what
                    _same_lattice(L, O, what)
                    n_checked += 1
            # the partial best path does not care about the pruning
            part = dec.best_paths(use_final_probs=False)
            for c in range(len(mats)):
                o = oracle.decode(h, pyoracle.Config(**cd), mats[c][: min(r, T[c])], m, finalize=False, use_final_probs=False)
                assert np.array_equal(part[c]["tids"], o.tids) and np.array_equal(bits(part[c]["graph"]), bits(o.path_graph)), (r, c)
        # mid-utterance n-best (the service's partial result): its 1-best is the best path with final-probs
        nb = dec.nbest(3)
        best = dec.best_paths(use_final_probs=True)
        for c in range(len(mats)):
            assert len(nb[c]) >= 1 and np.array_equal(nb[c][0]["words"], best[c]["words"]), c
        dec.finalize()
        fin = [dec.raw_lattice(c, True) for c in range(len(mats))]
        for c in range(len(mats)):
            O = pyoracle.oracle_raw_lattice(oracle, h, pyoracle.Config(**cd), mats[c], m)
            _same_lattice(as_raw(fin[c]), O, "final lattice of channel %d" % c)
            assert dec.raw_lattice(c, False) is None   # finalized && !use_final_probs (base-inl.h:879-884)
    finally:
        oracle.set_order_free(False)
        oracle.free_graph(h)
        dec.free()
        graph.free()
    assert n_checked >= 60


def test_long_utterance_in_a_fixed_arena(synth, oracle, tmp_path):
    """6 000 frames (one minute of speech) in lattice mode with room for ~40 frames of raw tokens and links: without the running
    back-pruning and compaction the arena would need 60x as much.  Streaming chunks of 48 frames; the final
    lattice and best path equal the oracle's; a ragged companion channel finishes early and idles."""
    from test_gpu_lattice import as_raw

    G, g, m, path, graph = _setup(synth, tmp_path, n_states=5000, seed=17)
    cd = dict(beam=10.0, max_active=1000000, min_active=0, lattice_beam=5.0)
    T = [6000, 333]
    mats = [synth.make_loglikes(g, t, 1000, m, seed=70 + i, mu=-2.5)[0] for i, t in enumerate(T)]
    h = oracle.load_graph(path)
    try:
        oracle.set_order_free(True)
        want = [oracle.decode(h, pyoracle.Config(**cd), x, m) for x in mats]
        # room for the tokens that survive the back-pruning (the lattice's own size, 3x what FinalizeDecoding keeps:
        # between passes more is alive) plus ~60 frames of raw tokens -- a small fraction of what the utterance creates
        per_frame = max(o.extra["tokens_created"] for o in want) / float(max(T))
        arena = int(3 * max(o.num_toks_end for o in want) + 60 * max(64, per_frame))
        print("tokens created %d, alive after FinalizeDecoding %d, arena %d" % (max(o.extra["tokens_created"] for o in want),
                                                                               max(o.num_toks_end for o in want), arena))
        default_data = __import__("os").environ.get("WFST_SYNTH_SEED_OFFSET", "0") in ("", "0")   # (how much more than the arena the DEFAULT utterance creates)
        assert max(o.extra["tokens_created"] for o in want) > (8 if default_data else 4) * arena
        dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), 2, max_frames=6100, max_tokens_per_frame=16384,
                                     arena_tokens=arena, lattice_links=4 * arena)
        dev = G.upload(mats)
        dec.init()
        for r in list(range(48, max(T), 48)) + [max(T)]:
            dec.advance([t.data_ptr() for t in dev], [min(r, t) for t in T], 1000)
        dec.sync()
        st = [dec.stats(c) for c in range(2)]
        dec.finalize()
        best = dec.best_paths(cap=2 * max(T) + 64)
        for c in range(2):
            assert np.array_equal(best[c]["tids"], want[c].tids) and np.array_equal(best[c]["words"], want[c].words), c
            assert np.array_equal(bits(best[c]["graph"]), bits(want[c].path_graph)) and np.array_equal(bits(best[c]["ac"]), bits(want[c].path_ac)), c
            O = pyoracle.oracle_raw_lattice(oracle, h, pyoracle.Config(**cd), mats[c], m, max_states=1 << 22, max_arcs=1 << 23)
            _same_lattice(as_raw(dec.raw_lattice(c, True)), O, "channel %d" % c)
        assert st[0]["tokens"] > (8 if default_data else 4) * arena   # it really did create that many
        dec.free()
    finally:
        oracle.set_order_free(False)
        oracle.free_graph(h)
        graph.free()


def test_abandoned_raw_pass_falls_back_without_an_error(synth, oracle, tmp_path):
    """VERDICT r5 next #5 / ADVICE r5: a meeting of lattice_prune_raw_kernel's workgroups that does not complete (forced here --
    wfst_options.debug 0x400: one workgroup of every channel stays away from the first meeting; on a shared chip: a sibling that is not
    resident) must not fail the utterance.  The workgroups give up after their 40 ms, the pass is marked abandoned in the channel's
    parameter block (not in ChanCtl::error, which the launch's share-out table reads), and lattice_prune_kernel prices the raw frames
    on its one-workgroup path: the same lattices as the oracle's, mid-utterance and final, and a count of the passes done over."""
    from test_gpu_lattice import as_raw

    G, g, m, path, graph = _setup(synth, tmp_path)
    cd = dict(beam=11.0, max_active=1000000, min_active=0, lattice_beam=5.0, prune_interval=10, prune_scale=0.1)
    T = [64, 47, 64]
    mats = [synth.make_loglikes(g, t, 1000, m, seed=140 + i, mu=-2.3)[0] for i, t in enumerate(T)]
    dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(mats), max_frames=128, max_tokens_per_frame=32768,
                                 arena_tokens=1 << 20, lattice_links=1 << 21, options=G.wfstdec.Options(debug=0x800 | 0x400))
    dev = G.upload(mats)
    dec.init()
    h = oracle.load_graph(path)
    try:
        oracle.set_order_free(True)
        for r in (25, 41, 64):
            dec.advance([t.data_ptr() for t in dev], [min(r, t) for t in T], 1000)
            for c in range(len(mats)):
                k = min(r, T[c])
                d = dec.raw_lattice(c, True)
                O = pyoracle.oracle_raw_lattice(oracle, h, pyoracle.Config(**cd), mats[c][:k], m, finalize=False, use_final_probs=True)
                assert (d is not None) == O.ok, (r, c)
                if d is not None:
                    _same_lattice(as_raw(d), O, "frame %d channel %d" % (r, c))
        ab = [dec.prune_raw_abandoned(c) for c in range(len(mats))]
        assert all(a >= 3 for a in ab), ab   # every running pass of every channel was abandoned and done over (passes at 10, 20, ...)
        dec.finalize()
        best = dec.best_paths()
        for c in range(len(mats)):
            assert best[c]["ok"]   # no WFST_E_DEVICE, no channel error
            O = pyoracle.oracle_raw_lattice(oracle, h, pyoracle.Config(**cd), mats[c], m)
            _same_lattice(as_raw(dec.raw_lattice(c, True)), O, "final lattice of channel %d" % c)
        # ... and without the switch nothing is abandoned
        dec2 = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(mats), max_frames=128, max_tokens_per_frame=32768,
                                      arena_tokens=1 << 20, lattice_links=1 << 21, options=G.wfstdec.Options(debug=0x800))
        dec2.init()
        dec2.advance([t.data_ptr() for t in dev], T, 1000)
        dec2.sync()
        assert [dec2.prune_raw_abandoned(c) for c in range(len(mats))] == [0, 0, 0]
        dec2.free()
    finally:
        oracle.set_order_free(False)
        oracle.free_graph(h)
        dec.free()
        graph.free()
