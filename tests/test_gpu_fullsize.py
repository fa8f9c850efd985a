"""-m gpu: BASELINE configs[1] sizes (batch 128 x 300 frames, ~10M-arc HCLG, beam 13).
The oracle (~0.5 s per utterance, on host threads) checks every utterance of the best-path batch and a
sample of the lattices; on top, size-independent properties: batch invariance (an utterance decodes identically alone, in another
channel), run-to-run determinism, one transition-id per frame, and beam monotonicity (a wider beam
never gives a worse best path)."""
import numpy as np
import pytest

import pyoracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(synth, oracle, tmp_path_factory):
    import gpu_util as G

    g = synth.make_hclg_like(2850000, seed=7)
    assert 9.5e6 < g.n_arcs < 10.5e6
    path = str(tmp_path_factory.mktemp("big") / "g10m.bin")
    g.write(path)
    m = synth.default_tid2pdf(6000)
    graph = G.wfstdec.Graph.from_arrays(g.start, g.final_state, g.state_info, g.arcs)
    graph.set_tid2pdf(m)
    B, T = 128, 300
    mats = [synth.make_loglikes_multi(g, T, 3000, m, seed=u, n_paths=272, mu=-4.0, jitter=0.5, ac_lo=0.5)[0] for u in range(B)]
    yield dict(G=G, g=g, path=path, m=m, graph=graph, mats=mats, B=B, T=T)
    graph.free()


LIM = dict(max_frames=304, max_tokens_per_frame=131072, arena_tokens=300 * 13900)   # (<= 2^22 tokens: the tokens carry degree codes)
CD = dict(beam=13.0, max_active=1000000, min_active=0, lattice_beam=7.0)


def test_batch128_parity_sample_and_properties(big, oracle):
    G = big["G"]
    res = G.decode_batch(big["graph"], CD, big["mats"], limits=LIM)
    assert all(r.ok and len(r.tids) == big["T"] for r in res)
    # the oracle on EVERY utterance (host threads; ~0.5 s each): bit-exact labels and costs
    import os
    from concurrent.futures import ThreadPoolExecutor

    h = oracle.load_graph(big["path"])
    cfg = pyoracle.Config(**CD)
    with ThreadPoolExecutor(max_workers=max(1, min(64, os.cpu_count() or 1))) as ex:
        outs = list(ex.map(lambda u: oracle.decode(h, cfg, big["mats"][u], big["m"]), range(big["B"])))
    n_tied = 0
    for u, o in enumerate(outs):
        if o.extra["ties"] == 0:
            G.assert_same_as_oracle(res[u], o, "utt %d" % u)
        else:
            # an exact float tie ON the best path: the reference keeps the first arrival in its hash-list
            # order, the GPU the lowest arc index (deviation 3).  Both paths are
            # optimal: same length, total within north_star's 1e-4 (the hop sums differ in rounding only)
            n_tied += 1
            assert res[u].ok and len(res[u].tids) == len(o.tids), u
            assert abs(res[u].tot_score - o.tot_score) <= 1e-4 * abs(o.tot_score), u
        # GPU work counters use the reference loop's definitions: they may only fall short by the
        # few order-dependent extras the reference expands at exact-equality cutoffs
        assert abs(res[u].stats["N"] - o.extra["N"]) <= 1e-3 * o.extra["N"]
        assert abs(res[u].stats["E"] - o.extra["E"]) <= 1e-3 * o.extra["E"]
    assert n_tied == 0, "%d utterances with an exact cost tie on the best path (none seen so far on these seeds)" % n_tied
    oracle.free_graph(h)
    # determinism: a second run gives the same bits
    res2 = G.decode_batch(big["graph"], CD, big["mats"], limits=LIM)
    for a, b in zip(res, res2):
        G.assert_same_path(a, b.words, b.tids, b.path_ilabel, b.path_olabel, b.path_graph, b.path_ac, [b.tot_score, b.lm_score])
    # batch invariance: 8 of them alone, in other channels
    pick = [3, 30, 60, 64, 90, 99, 120, 126]
    sub = G.decode_batch(big["graph"], CD, [big["mats"][u] for u in pick], limits=LIM)
    for u, b in zip(pick, sub):
        a = res[u]
        G.assert_same_path(a, b.words, b.tids, b.path_ilabel, b.path_olabel, b.path_graph, b.path_ac, [b.tot_score, b.lm_score])
    # beam monotonicity on a few utterances: without final-probs (with them a wider beam may
    # keep a final token alive that costs more than the best non-final one) the best token of a
    # wider beam is never worse; and the wider beam is itself checked against the oracle
    narrow = G.decode_batch(big["graph"], CD, [big["mats"][u] for u in pick[:4]], limits=LIM, finalize=False, use_final_probs=False)
    wide = G.decode_batch(big["graph"], dict(CD, beam=15.0), [big["mats"][u] for u in pick[:4]], limits=LIM, finalize=False, use_final_probs=False)
    for nr, w in zip(narrow, wide):
        assert w.tot_score <= nr.tot_score * (1 + 1e-6)
    h = oracle.load_graph(big["path"])
    o = oracle.decode(h, pyoracle.Config(**dict(CD, beam=15.0)), big["mats"][pick[0]], big["m"], finalize=False, use_final_probs=False)
    assert o.extra["ties"] == 0
    G.assert_same_as_oracle(wide[0], o, "beam 15")
    oracle.free_graph(h)


def test_batch128_lattice_mode_properties(big, oracle):
    """Lattice mode at full size (§8 f.1): FinalizeDecoding's lattice_beam pruning + GetRawLattice for
    all 128 utterances.  Oracle (order-free mode, see tests/test_gpu_lattice.py) state by state on a
    sample; for every utterance the size-independent properties of a pruned lattice: topologically
    numbered, trim (every state reachable from the start and reaching a final state), its shortest
    path IS the best path, and a narrower lattice_beam gives a sub-lattice."""
    from test_gpu_lattice import as_raw, gpu_lattices, multiset_contains, nodes

    G = big["G"]
    lim = dict(LIM, lattice_links=6 << 20)
    lats, best, nbest = gpu_lattices(G, big["graph"], CD, big["mats"], limits=lim, nbest=8)
    assert all(d is not None for d in lats)
    # n-best of all 128 utterances: the 1-best is GetBestPath's, totals ascend, word sequences are distinct
    for u, paths in enumerate(nbest):
        assert len(paths) >= 1 and np.array_equal(paths[0]["words"], best[u]["words"]), u
        assert abs(paths[0]["tot_score"] - best[u]["tot_score"]) <= 1e-4 * abs(best[u]["tot_score"]), u
        tots = [p["tot_score"] for p in paths]
        assert all(b >= a for a, b in zip(tots, tots[1:])), u
        assert len({tuple(p["words"].tolist()) for p in paths}) == len(paths), u
    h = oracle.load_graph(big["path"])
    try:
        oracle.set_order_free(True)
        for u in (5, 64, 127):
            O = pyoracle.oracle_raw_lattice(oracle, h, pyoracle.Config(**CD), big["mats"][u], big["m"])
            L = as_raw(lats[u])
            assert np.array_equal(nodes(L), nodes(O)), "utt %d states" % u
            assert np.array_equal(L.labelled_arcs(), O.labelled_arcs()), "utt %d arcs" % u
    finally:
        oracle.set_order_free(False)
        oracle.free_graph(h)
    for u, d in enumerate(lats):
        L = as_raw(d)
        S = L.n_states
        assert np.all(L.a_dst > L.a_src) and np.all(np.diff(L.a_src) >= 0), u
        assert L.st_frame[0] == 0 and np.all(np.diff(L.st_frame) >= 0) and L.st_frame[-1] == big["T"], u
        assert L.st_final.sum() >= 1 and np.all(L.st_frame[L.st_final == 1] == big["T"]), u
        # forward DP in state order (ids are topological): reachability and shortest path
        dist = np.full(S, np.inf, np.float32)
        dist[0] = 0.0
        w = (L.a_graph + L.a_ac).astype(np.float32)
        for k in range(len(L.a_src)):   # arcs are sorted by source state
            c = np.float32(dist[L.a_src[k]] + w[k])
            if c < dist[L.a_dst[k]]:
                dist[L.a_dst[k]] = c
        assert np.all(np.isfinite(dist)), "utt %d: state unreachable from the start" % u
        co = np.zeros(S, bool)
        co[L.st_final == 1] = True
        for k in range(len(L.a_src) - 1, -1, -1):
            if co[L.a_dst[k]]:
                co[L.a_src[k]] = True
        assert co.all(), "utt %d: state that reaches no final state" % u
        sp = dist[L.st_final == 1].min()
        assert abs(sp - best[u]["tot_score"]) <= 1e-4 * abs(sp), "utt %d: shortest path %g != best path %g" % (u, sp, best[u]["tot_score"])
        # forward costs of the states are the tokens' costs: never below the lattice's own shortest distance
        assert np.all(L.st_cost >= dist - 1e-3 * np.abs(dist) - 1e-3), u
    # a narrower lattice_beam gives a sub-lattice
    pick = [0, 31, 77, 100]
    narrow, _ = gpu_lattices(G, big["graph"], dict(CD, lattice_beam=3.0), [big["mats"][u] for u in pick], limits=lim)
    for u, d in zip(pick, narrow):
        A, Bw = as_raw(d), as_raw(lats[u])
        assert len(A.a_src) < len(Bw.a_src)
        assert multiset_contains(Bw.labelled_arcs(), A.labelled_arcs()), "utt %d" % u


def test_config5_beam15_lattice_and_nbest_sample(big, oracle, refdec, tmp_path):
    """BASELINE configs[4] (lattice-generating decode, beam = 15) on the 10M-arc graph: 16 utterances
    in lattice mode; two of them state by state against the oracle, their 5-best against the
    reference's determinizer + NShortestPath run on the lattice the device returned; the determinized
    lattices (GetLattice, built on the device) of all 16 arc for arc against the reference's
    DeterminizeLatticeWrapper run on the raw lattice the device returned."""
    from test_gpu_determinize import as_det
    from test_gpu_lattice import _same_nbest, as_raw, gpu_lattices, nodes

    G = big["G"]
    cd = dict(CD, beam=15.0, lattice_beam=8.0)
    lim = dict(max_frames=304, max_tokens_per_frame=262144, arena_tokens=300 * 60000, lattice_links=24 << 20)
    mats = big["mats"][:16]
    dets = []
    lats, best, nbest = gpu_lattices(G, big["graph"], cd, mats, limits=lim, nbest=5, det_out=dets)
    det_seconds = dets.pop()
    lib = pyoracle.build_det_host()
    for u in range(16):
        D, L = as_det(dets[u]), as_raw(lats[u])
        rc, H = pyoracle.det_host_run(lib, L, cap_scale=32)
        assert rc == 0 and [D.n_states, int(D.st_final.sum())] == [H.n_states, int(H.st_final.sum())], u
        assert np.array_equal(D.arc_multiset(), H.arc_multiset()), u
        p = str(tmp_path / "c5_raw.lat")
        with open(p, "wb") as f:
            f.write(G.pkg.shard.lattice_to_bytes(lats[u]))
        R = pyoracle.ref_determinize_lattice_file(refdec, p, 0)
        assert R is not None and [D.n_states, int(D.st_final.sum())] == [R.n_states, int(R.st_final.sum())], u
        assert np.array_equal(D.arc_multiset(), R.arc_multiset()), u
        assert D.n_states < L.n_states   # that is what it is for
    print("determinized 16 beam-15 lattices on the device in %.3f s (raw states %s -> %s)" % (
        det_seconds, [int(as_raw(x).n_states) for x in lats[:4]], [int(d["n_states"]) for d in dets[:4]]))
    h = oracle.load_graph(big["path"])
    try:
        for u in (3, 12):
            r = oracle.decode(h, pyoracle.Config(**cd), mats[u], big["m"])
            assert r.extra["ties"] == 0
            assert np.array_equal(best[u]["words"], r.words) and np.array_equal(best[u]["tids"], r.tids)
            oracle.set_order_free(True)
            O = pyoracle.oracle_raw_lattice(oracle, h, pyoracle.Config(**cd), mats[u], big["m"])
            oracle.set_order_free(False)
            L = as_raw(lats[u])
            assert np.array_equal(nodes(L), nodes(O)) and np.array_equal(L.labelled_arcs(), O.labelled_arcs()), u
            p = str(tmp_path / ("c5_%d.lat" % u))
            with open(p, "wb") as f:
                f.write(G.pkg.shard.lattice_to_bytes(lats[u]))
            ref = pyoracle.ref_nbest_from_lattice_file(refdec, p, 0, 5)
            assert ref is not None
            _same_nbest(nbest[u], ref[0], "utt %d" % u)
    finally:
        oracle.set_order_free(False)
        oracle.free_graph(h)


def test_config4_biglm_batch128_full_size(big, synth, oracle, tmp_path):
    """BASELINE configs[3] at full size (VERDICT r2 next #1a): on-the-fly LM rescoring, 128 utterances x 300 frames on the
    10 M-arc graph with the bench's LM pair (old: 156 k-state bigram scaled -1, new: 622 k-state trigram).  The fixed-mode
    oracle (bit for bit: words, transition-ids, per-hop labels and costs, scores) on 16 utterances; on ALL 128: a path per
    utterance with one transition-id per frame, run-to-run determinism, and batch invariance (the same utterance alone, in
    another channel, in a smaller batch)."""
    import importlib
    import os
    from concurrent.futures import ThreadPoolExecutor

    from test_gpu_biglm import _same

    lmsynth = importlib.import_module("asr-decoder_amd.lmsynth")
    G = big["G"]
    V = int(big["g"].arcs["olabel"].max())
    paths = []
    for tag, spec, seed in (("old", (20000, 5, 0, 0), 41), ("new", (40000, 6, 100000, 3), 42)):   # bench.py --lm-old / --lm-new
        nb, s2, nt, s3 = spec
        lp = str(tmp_path / ("lm_%s.bin" % tag))
        lmsynth.make_lm(V, 3 if nt > 0 else 2, nb, s2, nt, s3, seed=seed).to_fsa().write(lp)
        paths.append(lp)
    L1, L2 = G.wfstdec.Lm.load(paths[0], -1.0), G.wfstdec.Lm.load(paths[1], 1.0)
    assert L1.info()["n_states"] > 100000 and L2.info()["n_states"] > 500000
    lim = dict(max_frames=304, max_tokens_per_frame=131072, arena_tokens=300 * 13900)
    cd = dict(CD)

    def run(mats):
        dec = G.wfstdec.BatchDecoder(big["graph"], G.gpu_config(cd), len(mats), old_lm=L1, new_lm=L2, lm_pairs=1 << 20, **lim)
        try:
            return G.decode_batch(big["graph"], cd, mats, dec=dec)
        finally:
            dec.free()

    res = run(big["mats"])
    # (the reference's biglm final pruning -- final_best_cost ranges over non-final tokens too, biglm.h:186-188 -- leaves some
    # utterances without a path at lattice_beam 7: reproduced, and checked against the oracle on the sample below)
    n_ok = sum(int(r.ok) for r in res)
    # (measured: 45 of these 128 utterances keep a path -- bench.py's biglm leg reports the same count as utterances_with_path)
    assert all(len(r.tids) == big["T"] for r in res if r.ok), n_ok
    if __import__("os").environ.get("WFST_SYNTH_SEED_OFFSET", "0") in ("", "0"):   # (the default data's count; another draw -- WFST_SYNTH_SEED_OFFSET -- has its own)
        assert abs(n_ok - 45) <= 2, n_ok
    print("biglm full size: %d of %d utterances keep a path at lattice_beam %g" % (n_ok, big["B"], cd["lattice_beam"]))
    # the oracle in FIXED DiffArpaLm mode (DESIGN.md section 4 "biglm"), order-free, on 16 utterances
    h = oracle.load_graph(big["path"])
    o1, o2 = pyoracle.Lm(oracle, paths[0], -1.0), pyoracle.Lm(oracle, paths[1], 1.0)
    sample = list(range(0, 128, 8))
    try:
        oracle.set_order_free(True)
        with ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1))) as ex:
            want = list(ex.map(lambda u: pyoracle.biglm_decode(oracle, h, pyoracle.Config(**cd), o1, o2, big["mats"][u], big["m"], fixed=True), sample))
    finally:
        oracle.set_order_free(False)
    for u, o in zip(sample, want):
        assert o.extra["lm_oob"] == 0 and o.extra["ties"] == 0, u
        _same(res[u], o, "utt %d" % u)
    o1.free()
    o2.free()
    oracle.free_graph(h)
    # the LM difference is really applied: LM scores differ from the plain decoder's on most utterances
    plain = G.decode_batch(big["graph"], cd, big["mats"][:8], limits=lim)
    assert sum(int(a.ok and a.lm_score != b.lm_score) for a, b in zip(res[:8], plain)) >= 1
    # determinism: a second run gives the same bits; batch invariance: 8 of them alone, in other channels
    res2 = run(big["mats"])
    for a, b in zip(res, res2):
        G.assert_same_path(a, b.words, b.tids, b.path_ilabel, b.path_olabel, b.path_graph, b.path_ac, [b.tot_score, b.lm_score])
    pick = [3, 30, 60, 64, 90, 99, 120, 126]
    sub = run([big["mats"][u] for u in pick])
    for u, b in zip(pick, sub):
        a = res[u]
        G.assert_same_path(a, b.words, b.tids, b.path_ilabel, b.path_olabel, b.path_graph, b.path_ac, [b.tot_score, b.lm_score])
    L1.free()
    L2.free()


def test_config4_biglm_lattice_mode_full_size(big, synth, oracle, tmp_path):
    """The biglm decoder as the LATTICE decoder the reference's service runs (kaldi-online-nnet3-my-decoder.h:275-283), at full size
    (VERDICT r3 next #6b): 16 utterances x 300 frames on the 10 M-arc graph with the bench's LM pair, lattice mode with the
    running back-pruning -- the raw lattice of every one state by state and arc by arc against the fixed-mode oracle (order-free),
    the best path bit for bit."""
    import importlib
    import os
    from concurrent.futures import ThreadPoolExecutor

    from test_gpu_biglm import _same
    from test_gpu_lattice import as_raw, nodes

    lmsynth = importlib.import_module("asr-decoder_amd.lmsynth")
    G = big["G"]
    V = int(big["g"].arcs["olabel"].max())
    paths = []
    for tag, spec, seed in (("old", (20000, 5, 0, 0), 41), ("new", (40000, 6, 100000, 3), 42)):   # bench.py --lm-old / --lm-new
        nb, s2, nt, s3 = spec
        lp = str(tmp_path / ("lm_%s.bin" % tag))
        lmsynth.make_lm(V, 3 if nt > 0 else 2, nb, s2, nt, s3, seed=seed).to_fsa().write(lp)
        paths.append(lp)
    L1, L2 = G.wfstdec.Lm.load(paths[0], -1.0), G.wfstdec.Lm.load(paths[1], 1.0)
    cd = dict(CD, lattice_beam=9.0)
    sample = list(range(3, 128, 8))
    mats = [big["mats"][u] for u in sample]
    dec = G.wfstdec.BatchDecoder(big["graph"], G.gpu_config(cd), len(mats), old_lm=L1, new_lm=L2, lm_pairs=1 << 20, max_frames=304,
                                 max_tokens_per_frame=131072, arena_tokens=300 * 13900, lattice_links=8 << 20)
    res = G.decode_batch(big["graph"], cd, mats, dec=dec)
    lats = [dec.raw_lattice(i) for i in range(len(mats))]
    dec.free()
    h = oracle.load_graph(big["path"])
    o1, o2 = pyoracle.Lm(oracle, paths[0], -1.0), pyoracle.Lm(oracle, paths[1], 1.0)
    try:
        oracle.set_order_free(True)
        with ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1))) as ex:
            want = list(ex.map(lambda ll: (pyoracle.biglm_decode(oracle, h, pyoracle.Config(**cd), o1, o2, ll, big["m"], fixed=True),
                                           pyoracle.biglm_raw_lattice(oracle, h, pyoracle.Config(**cd), o1, o2, ll, big["m"], fixed=True)), mats))
    finally:
        oracle.set_order_free(False)
    n_lat = 0
    for u, r, d, (o, O) in zip(sample, res, lats, want):
        assert o.extra["lm_oob"] == 0 and o.extra["ties"] == 0, u
        _same(r, o, "utt %d" % u)
        assert (d is not None) == bool(O.ok), u
        if d is not None:
            L = as_raw(d)
            assert np.array_equal(nodes(L), nodes(O)) and np.array_equal(L.labelled_arcs(), O.labelled_arcs()), "utt %d lattice" % u
            n_lat += 1
    print("biglm lattice mode at full size: %d of %d utterances with a lattice, each equal to the oracle's" % (n_lat, len(sample)))
    assert n_lat >= 4, n_lat
    o1.free()
    o2.free()
    oracle.free_graph(h)
    L1.free()
    L2.free()


def test_config5_beam15_lattices_batch128(big, oracle, refdec, tmp_path):
    """BASELINE configs[4] at BATCH 128 (VERDICT r2 next #1a): lattice-generating decode at beam 15 / lattice-beam 8 with the
    reference's running back-pruning (prune_interval 25), all 128 utterances in one decoder.  On all 128: the raw lattice's
    size-independent properties (topological numbering, trim, shortest path = best path), the 5-best (1-best = GetBestPath,
    ascending, distinct) and the determinized lattice built on the device (deterministic in its word labels, fewer states
    than the raw lattice, same best cost).  On a sample: the raw lattice state by state against the order-free oracle, and
    the determinized lattice arc for arc against the REFERENCE's determinizer (oracle/_ref) run on the raw lattice."""
    from test_gpu_determinize import as_det
    from test_gpu_lattice import as_raw, gpu_lattices, nodes

    G = big["G"]
    cd = dict(CD, beam=15.0, lattice_beam=8.0)
    lim = dict(max_frames=304, max_tokens_per_frame=262144, arena_tokens=300 * 60000, lattice_links=24 << 20)
    dets = []
    lats, best, nbest = gpu_lattices(G, big["graph"], cd, big["mats"], limits=lim, nbest=5, det_out=dets)
    det_seconds = dets.pop()
    assert all(d is not None for d in lats) and all(d is not None for d in dets)
    for u in range(big["B"]):
        L, D = as_raw(lats[u]), as_det(dets[u])
        S = L.n_states
        assert np.all(L.a_dst > L.a_src) and np.all(np.diff(L.a_src) >= 0), u
        assert L.st_frame[0] == 0 and np.all(np.diff(L.st_frame) >= 0) and L.st_frame[-1] == big["T"], u
        assert L.st_final.sum() >= 1 and np.all(L.st_frame[L.st_final == 1] == big["T"]), u
        dist = np.full(S, np.inf, np.float32)
        dist[0] = 0.0
        w = (L.a_graph + L.a_ac).astype(np.float32)
        for k in range(len(L.a_src)):   # arcs are sorted by source state, ids are topological
            c = np.float32(dist[L.a_src[k]] + w[k])
            if c < dist[L.a_dst[k]]:
                dist[L.a_dst[k]] = c
        assert np.all(np.isfinite(dist)), "utt %d: state unreachable from the start" % u
        co = np.zeros(S, bool)
        co[L.st_final == 1] = True
        for k in range(len(L.a_src) - 1, -1, -1):
            if co[L.a_dst[k]]:
                co[L.a_src[k]] = True
        assert co.all(), "utt %d: state that reaches no final state" % u
        sp = dist[L.st_final == 1].min()
        assert abs(sp - best[u]["tot_score"]) <= 1e-4 * abs(sp), u
        paths = nbest[u]
        assert len(paths) >= 1 and np.array_equal(paths[0]["words"], best[u]["words"]), u
        tots = [p["tot_score"] for p in paths]
        assert all(b >= a for a, b in zip(tots, tots[1:])) and len({tuple(p["words"].tolist()) for p in paths}) == len(paths), u
        # the determinized lattice: at most one arc per (state, word), fewer states, and the same best cost
        assert D.n_states < L.n_states, u
        key = D.a_src.astype(np.int64) * (1 << 32) + D.a_ol.astype(np.int64)
        nonfinal = D.a_ol != 0
        assert len(np.unique(key[nonfinal])) == int(nonfinal.sum()), "utt %d: two arcs with one word out of a state" % u
        dd = np.full(D.n_states, np.inf, np.float64)
        dd[0] = 0.0
        order = np.argsort(D.a_src, kind="stable")
        changed = True
        for _ in range(D.n_states + 1):   # Bellman-Ford (the state numbering of a determinized lattice is not topological)
            if not changed:
                break
            changed = False
            for k in order:
                c = dd[D.a_src[k]] + float(D.a_graph[k]) + float(D.a_ac[k])
                if c < dd[D.a_dst[k]] - 1e-9:
                    dd[D.a_dst[k]] = c
                    changed = True
        assert abs(dd[D.st_final == 1].min() - sp) <= 2e-3 + 1e-4 * abs(sp), (u, dd[D.st_final == 1].min(), sp)
    print("batch 128 at beam 15: raw states mean %.0f, determinized mean %.0f; 128 determinized lattices in %.3f s" % (
        np.mean([d["n_states"] for d in lats]), np.mean([d["n_states"] for d in dets]), det_seconds))
    h = oracle.load_graph(big["path"])
    try:
        # the raw lattice state by state and arc by arc against the order-free oracle on 16 of the 128 utterances (VERDICT r3 next
        # #6c; the oracle's runs in parallel host threads) ...
        import os
        from concurrent.futures import ThreadPoolExecutor

        sample = sorted(set(list(range(5, 128, 9)) + [7, 70, 121]))[:16]
        oracle.set_order_free(True)
        with ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1))) as ex:
            want = list(ex.map(lambda u: pyoracle.oracle_raw_lattice(oracle, h, pyoracle.Config(**cd), big["mats"][u], big["m"]), sample))
        oracle.set_order_free(False)
        for u, O in zip(sample, want):
            L = as_raw(lats[u])
            assert np.array_equal(nodes(L), nodes(O)) and np.array_equal(L.labelled_arcs(), O.labelled_arcs()), u
        # ... and the determinized lattice arc for arc against the REFERENCE's determinizer on three
        for u in (7, 70, 121):
            L = as_raw(lats[u])
            p = str(tmp_path / ("c5b_%d.lat" % u))
            with open(p, "wb") as f:
                f.write(G.pkg.shard.lattice_to_bytes(lats[u]))
            R = pyoracle.ref_determinize_lattice_file(refdec, p, 0)
            D = as_det(dets[u])
            assert R is not None and [D.n_states, int(D.st_final.sum())] == [R.n_states, int(R.st_final.sum())], u
            assert np.array_equal(D.arc_multiset(), R.arc_multiset()), u
    finally:
        oracle.set_order_free(False)
        oracle.free_graph(h)


def test_service_operating_point_divergence_from_the_reference(big, synth, refdec, capsys):
    """max_active 7000 / min_active 200 (the reference service's own configuration,
    v1-asrbin/conf/decoder.conf:4-8) on the 10 M-arc graph, all 128 utterances, against the reference
    decoder ITSELF (oracle/_ref).  Where the limits bind, the reference's cutoff depends on its
    hash-list visiting order (deviation 2), so it is not a function of its inputs
    alone: with nothing changed but its hash table size (hash_ratio 3 instead of 2) it differs from
    itself.  That self-divergence is the yardstick: the GPU (the order-independent restatement) must
    not differ from the reference by much more than the reference differs from itself.
      (a) the headline log-likelihoods (272 live hypotheses): nearly every utterance identical;
      (b) SURVEY 8(d)'s single planted path in N(-2,1) noise, a search at its critical point where the
          best path is one of many near-equal noise paths: measured on this box in round 2 -- GPU vs
          reference WER 0.236, 80/128 identical, worst cost gap 1.8 %; reference vs itself (64
          utterances, CPU): WER 0.116 (hash_ratio 3) / 0.142 (2.5), 51 and 48 of 64 identical."""
    import os
    import sys
    from concurrent.futures import ThreadPoolExecutor

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import divergence

    G = big["G"]
    cd = dict(beam=13.0, max_active=7000, min_active=200, lattice_beam=7.0)
    n_thr = max(1, min(64, len(os.sched_getaffinity(0))))
    h = refdec.load_graph(big["path"])

    def ref_all(mats, **kw):
        cfg = pyoracle.Config(**dict(cd, **kw))
        with ThreadPoolExecutor(max_workers=n_thr) as ex:
            return list(ex.map(lambda u: refdec.decode(h, cfg, mats[u], big["m"]), range(len(mats))))

    as_gpu = lambda rs: [dict(ok=r.ok, words=r.words, tids=r.tids, tot_score=r.tot_score) for r in rs]
    single = [synth.make_loglikes(big["g"], big["T"], 3000, big["m"], seed=u, mu=-2.0, sigma=1.0)[0] for u in range(big["B"])]
    # (c) the same generator calibrated as SURVEY 8(d) asks (mu -2.6: ~5.5 k tokens per frame at beam 13, ~4 k expanded), where
    #     max_active 7000 binds on a minority of the frames
    calibrated = [synth.make_loglikes(big["g"], big["T"], 3000, big["m"], seed=u, mu=-2.6, sigma=1.0)[0] for u in range(big["B"])]
    out, spread = {}, {}
    for name, mats in (("headline", big["mats"]), ("single", single), ("calibrated", calibrated)):
        res = G.decode_batch(big["graph"], cd, mats, limits=LIM)
        assert all(r.ok and len(r.tids) == big["T"] for r in res)
        r2 = ref_all(mats)
        r3 = ref_all(mats, hash_ratio=3.0)
        out[name] = (divergence(as_gpu(res), r2), divergence(as_gpu(r3), r2))
        if name != "headline":
            # the reference's own SPREAD (round 5, tools/parity_spread.py): a third visiting order, every pair
            r25 = ref_all(mats, hash_ratio=2.5)
            spread[name] = (max(out[name][1]["wer"], divergence(as_gpu(r25), r2)["wer"], divergence(as_gpu(r3), r25)["wer"]),
                            max(out[name][0]["wer"], divergence(as_gpu(res), r25)["wer"], divergence(as_gpu(res), r3)["wer"]))
            with capsys.disabled():
                print("[7000/200, %s] WER: reference vs reference at most %.4f, GPU vs reference at most %.4f" % ((name,) + spread[name]))
        with capsys.disabled():
            print("\n[7000/200, %s] GPU vs reference: %s\n[7000/200, %s] reference(hash_ratio 3) vs reference: %s" % (name, out[name][0], name, out[name][1]))
    refdec.free_graph(h)
    for name, (dv, self_dv) in out.items():
        assert dv["max_rel_cost_gap"] <= (0.03 if __import__("os").environ.get("WFST_SYNTH_SEED_OFFSET", "0") in ("", "0") else 0.06), (name, dv)   # every path within 3 % of the reference's cost (default data; another draw: 6 %)
        # bounded by what was measured (round 2: single 0.236 vs 0.169 self, 80 vs 94 identical): the divergence from the reference
        # stays within 1.5x the reference's own order dependence
        if __import__("os").environ.get("WFST_SYNTH_SEED_OFFSET", "0") in ("", "0"):   # (bounds measured on the default data; another draw has its own spread)
            assert dv["wer"] <= 1.5 * self_dv["wer"] + 0.005, (name, dv, self_dv)
        assert dv["bit_identical"] >= 0.8 * self_dv["bit_identical"], (name, dv, self_dv)
        # the SIGN of the cost differences (VERDICT r3 next #6a): without transcripts, path cost is the quality measure -- the GPU's
        # paths must not be systematically costlier than the reference's (mean within twice the reference's own hash_ratio spread,
        # and no more "reference cheaper" utterances than twice the self figure + 4)
        sg, ss = dv["signed_rel_cost_gap"], self_dv["signed_rel_cost_gap"]
        if __import__("os").environ.get("WFST_SYNTH_SEED_OFFSET", "0") in ("", "0"):   # (bounds measured on the default data)
            assert sg["mean"] <= 2.0 * abs(ss["mean"]) + 1e-4, (name, sg, ss)
            assert sg["second_cheaper"] <= 2 * ss["second_cheaper"] + 4, (name, sg, ss)
    # Round 5: measured against the reference's own spread over THREE visiting orders (hash_ratio 2, 2.5, 3:
    # WER 0.168-0.193 single, 0.031-0.042 calibrated), the GPU sits just outside it (0.211-0.239, 0.041-0.052): it computes
    # ProcessEmitting's next_cutoff as the minimum over all arrivals BEFORE admitting any, the limit point of the reference's rule
    # (which tightens while it walks its hash list: every order admits a superset), so it is further from each order than they are
    # from each other -- by a factor 1.24 at most here.  The bound is that measurement with a margin, down from 1.5 x one pair.
    for name, (ref_max, gpu_max) in spread.items():
        assert gpu_max <= 1.35 * ref_max + 0.005, (name, ref_max, gpu_max)
    assert out["headline"][0]["wer"] <= 0.05, out["headline"]
