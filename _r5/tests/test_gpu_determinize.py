"""-m gpu: the determinized lattice on the device (BASELINE configs[4]) --
wfst_decoder_get_determinized_lattice = the reference's GetLattice (base-inl.h:850-866: GetRawLattice +
DeterminizeLatticeWrapper) -- through the C ABI.  Checked arc for arc (multiset of labels and float costs, bit
for bit; state and final-state counts) against
  * the reference's own determinizer run on the raw lattice the device returned (oracle/_ref, where built),
  * the same algorithm compiled for the host (tests/det_host.cc), always,
  * the reference-generated goldens, where the device's raw lattice IS the reference's (beam-only cases)."""
import os

import numpy as np
import pytest

import pyoracle
from golden_util import GOLDEN_DIR, Golden

pytestmark = pytest.mark.gpu


def as_det(d):
    return pyoracle.RawLattice(True, d["n_states"], 0, d["st_final"], d["a_src"], d["a_dst"], d["a_ilabel"], d["a_olabel"],
                               d["a_graph"], d["a_acoustic"])


def _check_against(G, dec, c, raw, lib, ref, tmp_path, what, ref_must_accept=True):
    from test_gpu_lattice import as_raw

    d = dec.determinized_lattice(c)
    assert (d is not None) == (raw is not None), what
    if d is None:
        return None
    D = as_det(d)
    L = as_raw(raw)
    rc, H = pyoracle.det_host_run(lib, L, cap_scale=32)
    assert rc == 0, what
    assert [D.n_states, int(D.st_final.sum())] == [H.n_states, int(H.st_final.sum())] and np.array_equal(D.arc_multiset(), H.arc_multiset()), what + " vs host build"
    if ref is not None:
        p = str(tmp_path / "raw.lat")
        with open(p, "wb") as f:
            f.write(G.pkg.shard.lattice_to_bytes(raw))
        R = pyoracle.ref_determinize_lattice_file(ref, p, 0)
        if R is None and not ref_must_accept:
            return D   # dead ends in the unpruned frames of a mid-utterance raw lattice fail the reference's LatticeCheckFormat
        assert R is not None, what
        assert [D.n_states, int(D.st_final.sum())] == [R.n_states, int(R.st_final.sum())], what + " vs reference (counts)"
        assert np.array_equal(D.arc_multiset(), R.arc_multiset()), what + " vs reference (arcs)"
    # deterministic on words: no two arcs of a state with the same word; start state 0; arcs carry ilabel 0
    k = np.stack([D.a_src, D.a_ol], axis=1)[D.a_ol != 0]
    assert len(np.unique(k, axis=0)) == len(k), what + " not deterministic"
    assert np.all(D.a_il == 0), what
    return D


def _ref_or_none():
    return pyoracle.RefDecoder() if os.path.exists(pyoracle.REF_SO) else None


def test_golden_utterances(tmp_path):
    import gpu_util as G

    lib = pyoracle.build_det_host()
    ref = _ref_or_none()
    g = Golden("lattice_hclg600")
    z = np.load(os.path.join(GOLDEN_DIR, "det_hclg600.npz"))
    graph = G.wfstdec.Graph.load(g.write_graph(str(tmp_path / "g.bin")))
    graph.set_tid2pdf(g.tid2pdf)
    lim = dict(max_frames=64, max_tokens_per_frame=16384, arena_tokens=1 << 19, lattice_links=1 << 20)
    n = n_gold = 0
    for ci in (0, 1, 2):
        cd = dict(g.meta["cfgs"][ci])
        dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(g.utts), **lim)
        dev = G.upload(g.utts)
        dec.init()
        T = [int(x.shape[0]) for x in g.utts]
        if cd.get("prune_interval", 25) == 10:
            # mid-utterance request (the service's partial n-best goes through GetLattice too), one frame after a pruning pass
            dec.advance([t.data_ptr() for t in dev], [min(31, t) for t in T], int(g.utts[0].shape[1]))
            _check_against(G, dec, 0, dec.raw_lattice(0), lib, ref, tmp_path, "cfg %d mid-utterance" % ci, ref_must_accept=False)
        dec.advance([t.data_ptr() for t in dev], T, int(g.utts[0].shape[1]))
        dec.finalize()
        for ui in range(len(g.utts)):
            raw = dec.raw_lattice(ui)
            D = _check_against(G, dec, ui, raw, lib, ref, tmp_path, "cfg %d utt %d" % (ci, ui))
            key = "c%d_u%d_" % (ci, ui)
            (Lref,) = pyoracle.parse_lattice_file(bytes(z[key + "raw"]))
            if D is not None and len(raw["a_src"]) == len(Lref.a_src) and cd["max_active"] >= 1000:
                # the device's raw lattice is the reference's own (no order-dependent extras in it): so is the determinized one
                assert [D.n_states, int(D.st_final.sum()), len(D.a_src)] == list(z[key + "counts"]), key
                assert np.array_equal(D.arc_multiset(), z[key + "arcs"]), key
                n_gold += 1
            n += 1
        dec.free()
    graph.free()
    assert n == 9 and n_gold >= 4


def test_mid_size_batch_and_refusals(synth, tmp_path):
    import gpu_util as G

    lib = pyoracle.build_det_host()
    ref = _ref_or_none()
    g = synth.make_hclg_like(20000, seed=7, n_tid=2000, n_words=3000)
    m = synth.default_tid2pdf(2000)
    path = str(tmp_path / "g.bin")
    g.write(path)
    graph = G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    cd = dict(beam=12.0, max_active=1000000, min_active=0, lattice_beam=5.0)
    mats = [synth.make_loglikes(g, T, 1000, m, seed=60 + i, mu=-2.4)[0] for i, T in enumerate([120, 80, 120, 9, 120, 55])]
    dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(mats), max_frames=128, max_tokens_per_frame=32768, arena_tokens=1 << 20,
                                 lattice_links=1 << 21)
    dev = G.upload(mats)
    dec.init()
    dec.advance([t.data_ptr() for t in dev], [int(x.shape[0]) for x in mats], 1000)
    dec.finalize()
    best = dec.best_paths()
    for c in range(len(mats)):
        D = _check_against(G, dec, c, dec.raw_lattice(c), lib, ref, tmp_path, "channel %d" % c)
        # the best path of the determinized lattice is the decoder's best path: cheapest word sequence, same cost
        S = D.n_states
        dist = np.full(S, np.inf)
        dist[0] = 0.0
        order = np.argsort(D.a_src, kind="stable")
        for _ in range(S):   # Bellman-Ford (the numbering need not be topological)
            nd = dist.copy()
            np.minimum.at(nd, D.a_dst[order], dist[D.a_src[order]] + (D.a_graph[order].astype(np.float64) + D.a_ac[order]))
            if np.array_equal(nd, dist):
                break
            dist = nd
        sp = dist[D.st_final == 1].min()
        assert abs(sp - best[c]["tot_score"]) <= 1e-3 * abs(sp), c
    # the binding's one-sweep fetch hands out the same lattices as the per-channel calls
    for c, L in enumerate(dec.determinized_lattices()):
        one = dec.determinized_lattice(c)
        assert (L is None) == (one is None)
        if L is not None:
            assert L["n_states"] == one["n_states"] and all(np.array_equal(L[k], one[k]) for k in one if k != "n_states")
    assert dec.determinized_lattice(0, use_final_probs=False) is None   # finalized && !use_final_probs
    dec.free()
    # not in lattice mode: refused
    d2 = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), 1, max_frames=128, max_tokens_per_frame=32768, arena_tokens=1 << 20)
    d2.init()
    with pytest.raises(G.wfstdec.WfstError):
        d2.determinized_lattice(0)
    d2.free()
    graph.free()


def test_prefetched_determinization_gives_the_same_lattices(synth, tmp_path):
    """wfst_decoder_prefetch_determinized changes WHEN the determinizer runs (right after FinalizeDecoding, on a side stream,
    beside the best paths and the n-best lists), not what anything returns: best paths, n-best lists, raw and determinized
    lattices equal those of a decoder that never prefetched -- with one channel group and with two, with channels initialised
    anew while a prefetch is in flight, and as a decoder's first determinizer use."""
    import gpu_util as G

    g = synth.make_hclg_like(20000, seed=17, n_tid=2000, n_words=3000)
    m = synth.default_tid2pdf(2000)
    path = str(tmp_path / "g.bin")
    g.write(path)
    graph = G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    cd = dict(beam=12.0, max_active=1000000, min_active=0, lattice_beam=5.0)
    mats = [synth.make_loglikes(g, T, 1000, m, seed=160 + i, mu=-2.4)[0] for i, T in enumerate([120, 80, 120, 9, 120, 55, 100, 33])]
    lim = dict(max_frames=128, max_tokens_per_frame=32768, arena_tokens=1 << 20, lattice_links=1 << 21)
    dev = G.upload(mats)
    ptrs, T = [t.data_ptr() for t in dev], [int(x.shape[0]) for x in mats]

    def run(dec, prefetch):
        dec.init()
        dec.advance(ptrs, T, 1000)
        dec.finalize()
        if prefetch:
            dec.prefetch_determinized()
        best = dec.best_paths()
        nb = dec.nbest(4)
        det = [dec.determinized_lattice(c) for c in range(len(mats))]
        raw = [dec.raw_lattice(c) for c in range(len(mats))]
        return best, nb, det, raw

    def same(a, b, what):
        for x, y in zip(a[0], b[0]):
            assert np.array_equal(x["words"], y["words"]) and x["tot_score"] == y["tot_score"], what
        for x, y in zip(a[1], b[1]):
            assert len(x) == len(y), what + " n-best"
            for px, py in zip(x, y):
                assert np.array_equal(px["words"], py["words"]) and px["tot_score"] == py["tot_score"] and px["lm_score"] == py["lm_score"], what + " n-best"
        for x, y in zip(a[2], b[2]):
            assert (x is None) == (y is None), what
            if x is not None:
                for key in x:
                    assert np.array_equal(x[key], y[key]), (what, key)
        for x, y in zip(a[3], b[3]):   # (a raw lattice lists its states in arena order, which is a run's own: the same states and arcs)
            assert np.array_equal(np.sort(x["st_state"]), np.sort(y["st_state"])) and len(x["a_src"]) == len(y["a_src"]), what + " raw"
            assert np.array_equal(np.sort(x["a_graph"]), np.sort(y["a_graph"])) and np.array_equal(np.sort(x["a_acoustic"]), np.sort(y["a_acoustic"])), what + " raw"

    plain = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(mats), **lim)
    want = run(plain, False)
    plain.free()
    for opt in (dict(channel_groups=1), dict(channel_groups=2)):
        dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(mats), options=G.wfstdec.Options(**opt), **lim)
        same(run(dec, True), want, "first use %s" % opt)      # the decoder's first determinizer use IS the prefetch
        same(run(dec, True), want, "second utterance %s" % opt)
        # a prefetch nobody harvests before the channels are initialised anew: init waits for it, the next utterance is unharmed
        dec.init()
        dec.advance(ptrs, T, 1000)
        dec.finalize()
        dec.prefetch_determinized()
        same(run(dec, False), want, "after an abandoned prefetch %s" % opt)
        # twice in a row, then a batched second pass right behind it: the slots are harvested before they are reused
        dec.init()
        dec.advance(ptrs, T, 1000)
        dec.finalize()
        dec.prefetch_determinized()
        dec.prefetch_determinized()
        det = [dec.determinized_lattice(c) for c in range(len(mats))]
        for x, y in zip(det, want[2]):
            for key in x:
                assert np.array_equal(x[key], y[key]), key
        dec.free()
    # a decoder freed with a prefetch in flight
    dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(mats), **lim)
    dec.init()
    dec.advance(ptrs, T, 1000)
    dec.finalize()
    dec.prefetch_determinized()
    dec.free()
    # not in lattice mode: refused
    d2 = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), 1, max_frames=128, max_tokens_per_frame=32768, arena_tokens=1 << 20)
    d2.init()
    with pytest.raises(G.wfstdec.WfstError):
        d2.prefetch_determinized()
    d2.free()
    graph.free()


def test_detached_prefetch_keeps_the_lattices_of_the_utterance_before(synth, tmp_path):
    """wfst_decoder_prefetch_determinized_detached: the channels go on to their NEXT utterances while the determinizer works on
    the lattices of the ones they just finalized; wfst_decoder_get_prefetched_lattice returns those -- array for array what a
    decoder that determinizes on request returns for the same utterances -- whatever the channels are doing by then (initialised
    anew, mid-utterance, finalized again), and the utterances decoded beside the determinizer are unharmed."""
    import gpu_util as G

    g = synth.make_hclg_like(20000, seed=23, n_tid=2000, n_words=3000)
    m = synth.default_tid2pdf(2000)
    path = str(tmp_path / "g.bin")
    g.write(path)
    graph = G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    cd = dict(beam=12.0, max_active=1000000, min_active=0, lattice_beam=5.0)
    Ts = [120, 80, 120, 9, 120, 55]
    sets = [[synth.make_loglikes(g, T, 1000, m, seed=300 + 10 * k + i, mu=-2.4)[0] for i, T in enumerate(Ts)] for k in range(3)]
    lim = dict(max_frames=128, max_tokens_per_frame=32768, arena_tokens=1 << 20, lattice_links=1 << 21)

    def decode(dec, mats):
        dev = G.upload(mats)
        dec.init()
        dec.advance([t.data_ptr() for t in dev], [int(x.shape[0]) for x in mats], 1000)
        dec.finalize()
        return dev

    plain = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(Ts), **lim)
    want = []
    for mats in sets:
        decode(plain, mats)
        want.append(([dict(words=b["words"].copy(), tot=b["tot_score"]) for b in plain.best_paths()], [plain.determinized_lattice(c) for c in range(len(Ts))]))
    plain.free()

    def same_det(got, exp, what):
        for x, y in zip(got, exp):
            assert (x is None) == (y is None), what
            if x is not None:
                for key in x:
                    assert np.array_equal(x[key], y[key]), (what, key)

    for opt in (dict(channel_groups=1), dict(channel_groups=2)):
        dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(Ts), options=G.wfstdec.Options(**opt), **lim)
        with pytest.raises(G.wfstdec.WfstError):
            dec.prefetched_lattice(0)               # nothing prefetched yet
        keep = []
        for k, mats in enumerate(sets):
            keep.append(decode(dec, mats))
            dec.prefetch_determinized(detached=True)   # harvests utterance k - 1, starts utterance k
            best = dec.best_paths()
            for b, w in zip(best, want[k][0]):
                assert np.array_equal(b["words"], w["words"]) and b["tot_score"] == w["tot"], "best paths beside the determinizer %s" % opt
            if k > 0:
                same_det([dec.prefetched_lattice(c) for c in range(len(Ts))], want[k - 1][1], "utterance %d, fetched while %d is finalized %s" % (k - 1, k, opt))
        # the last utterance's lattices: fetched while the channels are already mid-way through another one
        dev = G.upload(sets[0])
        dec.init()
        dec.advance([t.data_ptr() for t in dev], [min(5, int(x.shape[0])) for x in sets[0]], 1000)
        same_det([dec.prefetched_lattice(c) for c in range(len(Ts))], want[1][1], "not harvested yet: still the utterance before %s" % opt)
        dec.harvest_prefetched()
        same_det([dec.prefetched_lattice(c) for c in range(len(Ts))], want[2][1], "the last utterance, channels live again %s" % opt)
        # ... and a channel that still holds its utterance serves them through GetLattice too
        decode(dec, sets[1])
        dec.prefetch_determinized(detached=True)
        same_det([dec.determinized_lattice(c) for c in range(len(Ts))], want[1][1], "GetLattice behind a detached prefetch %s" % opt)
        dec.prefetch_determinized(detached=True)       # nothing left to do
        dec.free()
    graph.free()


def test_a_channels_device_error_stays_with_that_channel(synth, tmp_path):
    """a prefetched determinization is harvested in front of InitDecoding / AdvanceDecoding / FinalizeDecoding of ANY channel;
    a capacity error of one channel's finished utterance must not make those calls fail for the others, nor lose the other channels'
    lattices of the same launch -- it is reported when THAT channel's lattice is asked for.  And a detached harvest forgets the
    lattices of channels it did not cover (they belong to utterances further back)."""
    import gpu_util as G

    W = G.wfstdec
    g = synth.make_hclg_like(6000, seed=29, n_tid=2000, n_words=3000)
    m = synth.default_tid2pdf(2000)
    path = str(tmp_path / "g.bin")
    g.write(path)
    graph = W.Graph.load(path)
    graph.set_tid2pdf(m)
    cd = dict(beam=12.0, max_active=1000000, min_active=0, lattice_beam=5.0)
    Ts = [110, 8, 110, 8]
    mats = [synth.make_loglikes(g, T, 1000, m, seed=500 + i, mu=-2.4)[0] for i, T in enumerate(Ts)]
    lim = dict(max_frames=128, max_tokens_per_frame=32768, arena_tokens=1 << 20)

    def decode(dec, channels=None):
        dev = G.upload(mats)
        dec.init()
        dec.advance([t.data_ptr() for t in dev], Ts, 1000)
        dec.finalize()
        return dev

    roomy = W.BatchDecoder(graph, G.gpu_config(cd), len(Ts), lattice_links=1 << 21, **lim)
    keep = decode(roomy)
    roomy.sync()
    links = [roomy.lattice_stats(c)["links_recorded"] for c in range(len(Ts))]
    want = [roomy.determinized_lattice(c) for c in range(len(Ts))]
    roomy.free()
    short, long_ = max(links[1], links[3]), min(links[0], links[2])
    assert 4 * short < long_, links
    tight = W.BatchDecoder(graph, G.gpu_config(cd), len(Ts), lattice_links=int(2 * short + 64), **lim)   # the long utterances outgrow it
    keep2 = decode(tight)
    tight.prefetch_determinized()           # GetLattice's determinizer, started behind FinalizeDecoding
    tight.init(channels=[1])                 # harvests it: channel 0 / 2's overflow is not channel 1's business
    dev1 = G.upload([mats[1]])
    tight.advance([dev1[0].data_ptr()], [Ts[1]], 1000, channels=[1])
    tight.finalize(channels=[1])
    with pytest.raises(W.WfstError) as e:
        tight.determinized_lattice(0)
    assert e.value.code == -4 and "forward links" in str(e.value)   # WFST_E_CAPACITY
    got = tight.determinized_lattice(3)      # the same launch's other lattice was kept
    assert got is not None and all(np.array_equal(got[k], want[3][k]) for k in got)
    tight.free()

    # a detached harvest that does not cover a channel forgets what an earlier one left for it
    dec = W.BatchDecoder(graph, G.gpu_config(cd), len(Ts), lattice_links=1 << 21, **lim)
    keep3 = decode(dec)
    dec.prefetch_determinized(detached=True)
    dec.init()
    dec.advance([t.data_ptr() for t in keep3], Ts, 1000)
    dec.finalize(channels=[1])               # only channel 1 finishes its next utterance
    dec.prefetch_determinized(detached=True)   # harvests the first prefetch (all four), starts channel 1's
    assert all(dec.prefetched_lattice(c) is not None for c in (0, 2))
    dec.harvest_prefetched()                 # ... which covered channel 1 alone
    assert dec.prefetched_lattice(1) is not None
    for c in (0, 2, 3):
        with pytest.raises(W.WfstError):
            dec.prefetched_lattice(c)
    dec.free()
    graph.free()
    del keep, keep2
