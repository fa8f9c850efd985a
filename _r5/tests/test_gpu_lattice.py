"""-m gpu: lattice mode of the HIP path (wfst_limits.lattice_links > 0) through the C ABI.
FinalizeDecoding's lattice_beam pruning + GetRawLattice (reference base-inl.h:725-975).

The reference admits an arc against next_cutoff as it stands when the arc is visited
(base-inl.h:326-333), so a handful of links above a frame's FINAL cutoff exist or not depending on
its hash-list order; the GPU applies the final cutoff to every arc.  Hence:
(1) against the reference-generated vectors the GPU lattice must be a sub-multiset of the
    reference's arcs, and equal to it whenever the arc counts agree;
(2) against the C oracle in its order-free mode (oracle_set_order_free; pinned to the reference by
    tests/test_oracle_lattice.py) it must be IDENTICAL up to the numbering of states inside a
    frame: same (frame, graph state) nodes with bit-equal forward costs and final flags, same arcs
    with bit-equal graph / acoustic costs."""
import numpy as np
import pytest

import pyoracle
from golden_util import Golden, bits

pytestmark = pytest.mark.gpu

LIM = dict(max_frames=512, max_tokens_per_frame=32768, arena_tokens=1 << 21, lattice_links=1 << 22)


def gpu_lattices(G, graph, cd, mats, use_final_probs=True, limits=LIM, nbest=0, det_out=None):
    """det_out: a list that receives the determinized lattices (GetLattice) and, last, the seconds they took"""
    dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(mats), **limits)
    dev = G.upload(mats)
    dec.init()
    dec.advance([t.data_ptr() for t in dev], [int(m.shape[0]) for m in mats], int(mats[0].shape[1]))
    dec.finalize()
    lats = [dec.raw_lattice(c, use_final_probs) for c in range(len(mats))]
    best = dec.best_paths(use_final_probs=True)
    nb = dec.nbest(nbest) if nbest else None
    if det_out is not None:
        import time

        t0 = time.time()
        det_out.extend(dec.determinized_lattice(c, use_final_probs) for c in range(len(mats)))
        det_out.append(time.time() - t0)
    dec.free()
    return (lats, best, nb) if nbest else (lats, best)


def as_raw(d):
    return pyoracle.RawLattice(True, d["n_states"], 0, d["st_final"], d["a_src"], d["a_dst"], d["a_ilabel"], d["a_olabel"],
                               d["a_graph"], d["a_acoustic"], d["st_frame"], d["st_state"], d["st_cost"])


def multiset_contains(big, small):
    from collections import Counter

    cb, cs = Counter(map(tuple, big)), Counter(map(tuple, small))
    return all(cb[k] >= v for k, v in cs.items())


def nodes(L):
    k = np.stack([L.st_frame, L.st_gstate, L.st_final, bits(L.st_cost)], axis=1)
    return k[np.lexsort(k.T[::-1])]


@pytest.mark.parametrize("name", ["lattice_hclg600", "lattice_eps_chains"])
def test_gpu_lattice_reproduces_golden(name, tmp_path):
    import gpu_util as G

    g = Golden(name)
    graph = G.wfstdec.Graph.load(g.write_graph(str(tmp_path / "g.bin")))
    if g.tid2pdf is not None:
        graph.set_tid2pdf(g.tid2pdf)
    groups = {}
    for k, cd, md, ui in g.cases():
        if md["finalize"] and cd["max_active"] >= 1000:   # beam-only regime (DESIGN.md 'Deviations')
            groups.setdefault((g.meta["cases"][k]["cfg"], md["use_final_probs"]), []).append((k, ui))
    n = n_equal = 0
    for (ci, ufp), items in groups.items():
        cd = dict(g.meta["cfgs"][ci])
        lats, _ = gpu_lattices(G, graph, cd, [g.utts[ui] for _, ui in items], ufp)
        for (k, ui), d in zip(items, lats):
            ok, ns, nf, na = (int(x) for x in g.z["c%d_counts" % k][:4])
            what = "%s case %d" % (name, k)
            assert (d is not None) == bool(ok), what
            if d is None:
                continue
            L = as_raw(d)
            ref_arcs = g.z["c%d_arcs" % k]
            assert L.n_states <= ns and int(L.st_final.sum()) == nf and len(L.a_src) <= na, what + " counts"
            assert multiset_contains(ref_arcs, L.arc_multiset()), what + " arcs not among the reference's"
            if len(L.a_src) == na:
                assert L.n_states == ns and np.array_equal(L.arc_multiset(), ref_arcs), what
                n_equal += 1
            assert np.all(L.a_dst > L.a_src) and np.all(np.diff(L.a_src) >= 0), what + " numbering"
            n += 1
    graph.free()
    assert n > 0 and n_equal >= n - 2


@pytest.mark.parametrize("lattice_beam", [0.5, 4.0, 8.0])
def test_gpu_lattice_equals_oracle_state_by_state(lattice_beam, oracle, synth, tmp_path):
    import gpu_util as G

    g = synth.make_hclg_like(20000, seed=7, n_tid=2000, n_words=3000)
    m = synth.default_tid2pdf(2000)
    path = str(tmp_path / "g.bin")
    g.write(path)
    graph = G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    ho = oracle.load_graph(path)
    cd = dict(beam=12.0, max_active=1000000, min_active=0, lattice_beam=lattice_beam, prune_interval=25)
    mats = [synth.make_loglikes_multi(g, T, 1000, m, seed=40 + i)[0] for i, T in enumerate((60, 33, 1, 80))]
    lats, best = gpu_lattices(G, graph, cd, mats)
    for i, (d, ll) in enumerate(zip(lats, mats)):
        try:
            oracle.set_order_free(True)
            O = pyoracle.oracle_raw_lattice(oracle, ho, pyoracle.Config(**cd), ll, m)
        finally:
            oracle.set_order_free(False)
        assert (d is not None) == O.ok
        L = as_raw(d)
        what = "utt %d lattice_beam %g" % (i, lattice_beam)
        assert np.array_equal(nodes(L), nodes(O)), what + " states"
        assert np.array_equal(L.labelled_arcs(), O.labelled_arcs()), what + " arcs"
        assert np.all(L.a_dst > L.a_src), what
        # the gather payload of the N>1 path (shard.lattice_to_bytes = the reference's on-disk format)
        (P,) = pyoracle.parse_lattice_file(G.pkg.shard.lattice_to_bytes(d))
        assert P.n_states == L.n_states and P.start == 0 and np.array_equal(P.st_final, L.st_final), what
        assert np.array_equal(P.a_src, L.a_src) and np.array_equal(P.a_dst, L.a_dst) and np.array_equal(P.arc_multiset(), L.arc_multiset()), what
        # the best path is unchanged by lattice mode
        r = oracle.decode(ho, pyoracle.Config(**cd), ll, m)
        assert np.array_equal(best[i]["words"], r.words) and np.array_equal(best[i]["tids"], r.tids), what
    oracle.free_graph(ho)
    graph.free()


def test_gpu_lattice_errors_are_loud(synth, tmp_path):
    import gpu_util as G

    g = synth.make_hclg_like(600, seed=11, n_tid=600, n_words=500)
    m = synth.default_tid2pdf(600)
    path = str(tmp_path / "g.bin")
    g.write(path)
    graph = G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    cd = dict(beam=13.0, max_active=1000000, min_active=0, lattice_beam=7.0)
    ll = synth.make_loglikes(g, 40, 300, m, seed=0, mu=-2.2, sigma=1.0)[0]
    # not in lattice mode
    dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), 1, max_frames=64, max_tokens_per_frame=8192, arena_tokens=1 << 18)
    dev = G.upload([ll])
    dec.init()
    dec.advance([dev[0].data_ptr()], [40], ll.shape[1])
    dec.finalize()
    with pytest.raises(G.wfstdec.WfstError):
        dec.raw_lattice(0)
    dec.free()
    # lattice mode: before InitDecoding -> state error (mid-utterance it is served: tests/test_gpu_running_prune.py);
    # link capacity too small -> capacity error
    dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), 1, max_frames=64, max_tokens_per_frame=8192, arena_tokens=1 << 18,
                                 lattice_links=1 << 20)
    with pytest.raises(G.wfstdec.WfstError):
        dec.raw_lattice(0)
    dec.init()
    dec.advance([dev[0].data_ptr()], [40], ll.shape[1])
    assert dec.raw_lattice(0) is not None
    dec.free()
    dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), 1, max_frames=64, max_tokens_per_frame=8192, arena_tokens=1 << 18,
                                 lattice_links=64)
    dec.init()
    with pytest.raises(G.wfstdec.WfstError):
        dec.advance([dev[0].data_ptr()], [40], ll.shape[1])
        dec.finalize()
        dec.raw_lattice(0)
    dec.free()
    graph.free()


def test_gpu_lattice_feeds_the_references_nbest(refdec, tmp_path):
    """Downstream check of the lattice: the GPU's GetRawLattice, written in the reference's on-disk
    format, through the reference's own DeterminizeLatticeWrapper + NShortestPath (oracle/_ref)
    gives the n-best word sequences and scores the reference gets from its own lattice."""
    import gpu_util as G
    from nbest_util import check_nbest_of_lattice_bytes

    g = Golden("lattice_hclg600")
    graph = G.wfstdec.Graph.load(g.write_graph(str(tmp_path / "g.bin")))
    graph.set_tid2pdf(g.tid2pdf)
    for ci in (0, 1):
        lats, _ = gpu_lattices(G, graph, dict(g.meta["cfgs"][ci]), g.utts)
        for ui, d in enumerate(lats):
            check_nbest_of_lattice_bytes(refdec, G.pkg.shard.lattice_to_bytes(d), ci, ui, tmp_path, "cfg %d utt %d" % (ci, ui))
    graph.free()


def _same_nbest(got, want, what):
    """same word sequences in the same order, scores within 1e-4 relative; entries whose total
    scores are closer than that may come out in either order"""
    assert len(got) == len(want), "%s: %d paths, reference %d" % (what, len(got), len(want))
    used = [False] * len(want)
    for k, a in enumerate(got):
        hit = None
        for j, b in enumerate(want):
            if not used[j] and np.array_equal(a["words"], b[0]) and abs(a["tot_score"] - b[1]) <= 1e-4 * abs(b[1]):
                hit = j
                break
        assert hit is not None, "%s: path %d (%s, %.4f) is not in the reference's list" % (what, k, a["words"].tolist(), a["tot_score"])
        used[hit] = True
        b = want[hit]
        assert abs(a["lm_score"] - b[2]) <= 1e-4 * max(1.0, abs(b[2])), "%s path %d lm_score" % (what, k)
        if hit != k:  # only a near-tie may swap places
            assert abs(want[k][1] - b[1]) <= 2e-4 * abs(b[1]), "%s: path %d out of order" % (what, k)


def test_gpu_nbest_matches_the_reference_golden(tmp_path):
    """wfst_decoder_get_nbest (k-best distinct word sequences on the device) against the n-best the
    reference gets from ITS OWN lattice with its determinizer + NShortestPath
    (tests/golden/nbest_hclg600.npz)."""
    import gpu_util as G
    from nbest_util import golden_nbest

    g = Golden("lattice_hclg600")
    graph = G.wfstdec.Graph.load(g.write_graph(str(tmp_path / "g.bin")))
    graph.set_tid2pdf(g.tid2pdf)
    for ci in (0, 1):
        dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(dict(g.meta["cfgs"][ci])), len(g.utts), **LIM)
        dev = G.upload(g.utts)
        dec.init()
        dec.advance([t.data_ptr() for t in dev], [int(m.shape[0]) for m in g.utts], int(g.utts[0].shape[1]))
        dec.finalize()
        want_n = golden_nbest(ci, 0)[1]
        got = dec.nbest(want_n)
        best = dec.best_paths()
        for ui in range(len(g.utts)):
            want, _ = golden_nbest(ci, ui)
            _same_nbest(got[ui], want, "cfg %d utt %d" % (ci, ui))
            assert np.array_equal(got[ui][0]["words"], best[ui]["words"])   # 1-best == GetBestPath
        # a sub-list of channels, other n
        one = dec.nbest(2, channels=[2])
        assert len(one) == 1 and [p["words"].tolist() for p in one[0]] == [p["words"].tolist() for p in got[2][:2]]
        dec.free()
    graph.free()


@pytest.mark.parametrize("lattice_beam", [3.0, 8.0])
def test_gpu_nbest_equals_reference_pipeline_on_its_own_lattice(lattice_beam, refdec, synth, tmp_path):
    """Bigger lattices, n = 10: the device n-best against the reference's DeterminizeLatticeWrapper
    + NShortestPath + LatticeToVector run (oracle/_ref) on the very lattice the device returns."""
    import gpu_util as G

    g = synth.make_hclg_like(20000, seed=7, n_tid=2000, n_words=3000)
    m = synth.default_tid2pdf(2000)
    path = str(tmp_path / "g.bin")
    g.write(path)
    graph = G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    cd = dict(beam=12.0, max_active=1000000, min_active=0, lattice_beam=lattice_beam)
    mats = [synth.make_loglikes_multi(g, T, 1000, m, seed=90 + i)[0] for i, T in enumerate((60, 33, 1, 80, 47))]
    dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(mats), **LIM)
    dev = G.upload(mats)
    dec.init()
    dec.advance([t.data_ptr() for t in dev], [int(x.shape[0]) for x in mats], int(mats[0].shape[1]))
    dec.finalize()
    got = dec.nbest(10)
    for i in range(len(mats)):
        p = str(tmp_path / ("l%d.lat" % i))
        with open(p, "wb") as f:
            f.write(G.pkg.shard.lattice_to_bytes(dec.raw_lattice(i)))
        ref = pyoracle.ref_nbest_from_lattice_file(refdec, p, 0, 10)
        assert ref is not None
        _same_nbest(got[i], ref[0], "utt %d lattice_beam %g" % (i, lattice_beam))
    dec.free()
    graph.free()


def test_gpu_lattice_channel_reuse_ragged_and_no_final_state(oracle, synth, tmp_path):
    """Lattice mode across utterances: the same decoder object decodes a second, different batch
    after InitDecoding (links, offsets and compacted lattices of the first must not leak), with
    ragged lengths and a subset of channels; and a graph whose final state is unreachable (every
    last-frame token is final, base-inl.h:936-940)."""
    import gpu_util as G

    g = synth.make_hclg_like(3000, seed=17, n_tid=400, n_words=300)
    m = synth.default_tid2pdf(400)
    path = str(tmp_path / "g.bin")
    g.write(path)
    graph = G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    ho = oracle.load_graph(path)
    cd = dict(beam=11.0, max_active=1000000, min_active=0, lattice_beam=5.0)
    dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), 4, **LIM)
    try:
        oracle.set_order_free(True)
        for rnd, (chans, lens) in enumerate((([0, 1, 2, 3], [40, 25, 7, 33]), ([2, 0], [12, 50]), ([3, 1, 0], [1, 30, 30]))):
            mats = [synth.make_loglikes(g, T, 200, m, seed=1000 * rnd + i, mu=-2.0)[0] for i, T in enumerate(lens)]
            dev = G.upload(mats)
            dec.init(chans)
            dec.advance([t.data_ptr() for t in dev], lens, int(mats[0].shape[1]), channels=chans)
            dec.finalize(chans)
            for c, x in zip(chans, mats):
                O = pyoracle.oracle_raw_lattice(oracle, ho, pyoracle.Config(**cd), x, m)
                d = dec.raw_lattice(c)
                assert (d is not None) == O.ok
                L = as_raw(d)
                assert np.array_equal(nodes(L), nodes(O)) and np.array_equal(L.labelled_arcs(), O.labelled_arcs()), "round %d channel %d" % (rnd, c)
            nb = dec.nbest(3, channels=chans)
            bp = dec.best_paths(channels=chans)
            for k in range(len(chans)):
                assert np.array_equal(nb[k][0]["words"], bp[k]["words"])
    finally:
        oracle.set_order_free(False)
        oracle.free_graph(ho)
    dec.free()
    graph.free()
    # no final state reachable: golden graph of tests/golden/no_final.npz
    gn = Golden("no_final")
    graph = G.wfstdec.Graph.load(gn.write_graph(str(tmp_path / "nf.bin")))
    ho = oracle.load_graph(str(tmp_path / "nf.bin"))
    cdn = dict(gn.meta["cfgs"][0])
    lats, _ = gpu_lattices(G, graph, cdn, [gn.utts[1]])
    try:
        oracle.set_order_free(True)
        O = pyoracle.oracle_raw_lattice(oracle, ho, pyoracle.Config(**cdn), gn.utts[1], None)
    finally:
        oracle.set_order_free(False)
    L = as_raw(lats[0])
    assert O.ok and np.array_equal(nodes(L), nodes(O)) and np.array_equal(L.labelled_arcs(), O.labelled_arcs())
    assert L.st_final.sum() == (L.st_frame == gn.utts[1].shape[0]).sum()   # every last-frame state is final
    oracle.free_graph(ho)
    graph.free()


def test_zero_frame_utterance_in_lattice_mode(synth, oracle, tmp_path):
    """An utterance with no frames next to a normal one: FinalizeDecoding, GetBestPath, GetRawLattice
    and GetNbest must not disturb each other (the reference asserts num_frames > 0 in GetRawLattice,
    base-inl.h:899; here that is 'no lattice')."""
    import gpu_util as G

    g = synth.make_hclg_like(2000, seed=41, n_tid=400, n_words=300)
    m = synth.default_tid2pdf(400)
    path = str(tmp_path / "g.bin")
    g.write(path)
    graph = G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    cd = dict(beam=11.0, max_active=1000000, min_active=0, lattice_beam=5.0)
    x = synth.make_loglikes(g, 20, 200, m, seed=5, mu=-2.0)[0]
    dev = G.upload([x])
    dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), 2, **LIM)
    dec.init()
    dec.advance([dev[0].data_ptr(), dev[0].data_ptr()], [20, 0], int(x.shape[1]))
    dec.finalize()
    assert dec.raw_lattice(1) is None
    nb = dec.nbest(3)
    assert nb[1] == [] and len(nb[0]) >= 1
    bp = dec.best_paths()
    ho = oracle.load_graph(path)
    o = oracle.decode(ho, pyoracle.Config(**cd), x, m)
    oracle.free_graph(ho)
    assert np.array_equal(bp[0]["tids"], o.tids) and np.array_equal(nb[0][0]["words"], o.words)
    assert len(bp[1]["tids"]) == 0
    dec.free()
    graph.free()


def test_closure_launch_workgroups_per_channel_give_one_lattice(oracle, synth, tmp_path):
    """A lattice decoder's closure launch shares a frame's epsilon links out over 4 workgroups per channel (wfst_options.debug
    0x100 / 0x200 / 0x300: 1 / 2 / 8); the one that finishes last closes the frame.  Wide frames (a beam that keeps most of a
    20 000-state graph alive: thousands of emitters a frame, several sweeps per workgroup) must give the SAME lattice whatever the
    count -- and the oracle's."""
    import gpu_util as G

    g = synth.make_hclg_like(20000, seed=11, n_tid=2000, n_words=3000)
    m = synth.default_tid2pdf(2000)
    path = str(tmp_path / "g.bin")
    g.write(path)
    graph = G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    cd = dict(beam=40.0, max_active=1000000, min_active=0, lattice_beam=6.0, prune_interval=10)
    mats = [synth.make_loglikes_multi(g, T, 1000, m, seed=70 + i)[0] for i, T in enumerate((34, 21))]
    lim = dict(max_frames=64, max_tokens_per_frame=65536, arena_tokens=1 << 22, lattice_links=1 << 23)
    got = {}
    peak = 0
    for dbg in (0, 0x100, 0x200, 0x300):
        dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(mats), options=G.wfstdec.Options(debug=dbg) if dbg else None, **lim)
        dev = G.upload(mats)
        dec.init()
        dec.advance([t.data_ptr() for t in dev], [int(x.shape[0]) for x in mats], int(mats[0].shape[1]))
        dec.finalize()
        got[dbg] = [as_raw(dec.raw_lattice(c, True)) for c in range(len(mats))]
        peak = max(peak, max(dec.lattice_stats(c)["links_recorded"] for c in range(len(mats))))
        dec.free()
    assert peak > 200000, peak   # (wide frames: the epsilon links alone are thousands a frame)
    for dbg in (0x100, 0x200, 0x300):
        for c in range(len(mats)):
            what = "debug %#x channel %d" % (dbg, c)
            assert np.array_equal(nodes(got[dbg][c]), nodes(got[0][c])), what + " states"
            assert np.array_equal(got[dbg][c].labelled_arcs(), got[0][c].labelled_arcs()), what + " arcs"
    ho = oracle.load_graph(path)
    try:
        oracle.set_order_free(True)
        O = pyoracle.oracle_raw_lattice(oracle, ho, pyoracle.Config(**cd), mats[1], m)
    finally:
        oracle.set_order_free(False)
        oracle.free_graph(ho)
    assert np.array_equal(nodes(got[0][1]), nodes(O)) and np.array_equal(got[0][1].labelled_arcs(), O.labelled_arcs())
    graph.free()
