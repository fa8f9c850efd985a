"""GetNbest as the service defines it (VERDICT r2 missing #3): NShortestPath over the lattice GetLattice returns, every path a linear
lattice carrying the lattice's own arcs (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:97-105, newfst/lattice-to-nbest.cc:15-199), for
any n -- with and without the second LM pass.

CPU: oracle/pyoracle.py's restatement (nshortest_paths, on the host determinizer's output) against the compiled reference run on the
same raw lattice.  GPU: wfst_decoder_get_nbest_paths (nbest_paths_kernel over the device's own determinized / rescored lattice)
against the reference: the same paths in the same order, labels and both costs of every arc bit for bit."""
import importlib

import numpy as np
import pytest

import pyoracle
from test_compose_lattice import _as_dict, _setup

shard = importlib.import_module("asr-decoder_amd.shard")


def _strip(p):
    """a reference path: one epsilon arc in front (the second Reverse's start state), then the lattice's arcs, the final weight's arc,
    the super-final arc and the first Reverse's epsilon -- all three <eps>:<eps>, the last two free"""
    il, ol, g, ac = p["ilabel"], p["olabel"], p["graph"], p["acoustic"]
    assert np.all(il == 0)
    assert ol[0] == 0 and g[0] == 0 and ac[0] == 0
    assert np.all(ol[-2:] == 0) and np.all(g[-2:] == 0) and np.all(ac[-2:] == 0)
    return ol[1:-2], g[1:-2], ac[1:-2]


def _same_paths(got, ref, what):
    assert len(got) == len(ref), "%s: %d paths, the reference has %d" % (what, len(got), len(ref))
    # same order wherever the costs differ; paths of exactly equal cost may swap (the reference's heap order)
    key = lambda ol, g, ac: (tuple(ol.tolist()), tuple(g.tolist()), tuple(ac.tolist()))
    R = [_strip(p) for p in ref]
    rt = [float(np.float32(sum(np.float32(x) + np.float32(y) for x, y in zip(g, ac)))) for (_, g, ac) in R]
    i = 0
    while i < len(got):
        j = i
        while j + 1 < len(got) and abs(got[j + 1]["tot"] - got[i]["tot"]) <= 1e-5 * max(1.0, abs(got[i]["tot"])):
            j += 1
        a = sorted(key(p["olabel"], p["graph"], p["acoustic"]) for p in got[i : j + 1])
        b = sorted(key(*r) for r in R[i : j + 1])
        assert a == b, "%s: paths %d..%d differ" % (what, i, j)
        i = j + 1
    tots = [p["tot"] for p in got]
    assert all(y >= x - 1e-5 * max(1.0, abs(x)) for x, y in zip(tots, tots[1:])), what
    for p, t in zip(got, rt):
        assert abs(p["tot"] - t) <= 2e-5 * max(1.0, abs(t)), what


def test_nbest_restatement_equals_the_compiled_reference(oracle, refdec, synth, tmp_path):
    lib = pyoracle.build_det_host()
    n_checked = 0
    for seed in range(2):
        g, m, gp, p1, p2, lls = _setup(synth, tmp_path, seed)
        h = oracle.load_graph(gp)
        r1, r2 = pyoracle.Lm(refdec, p1, -1.0), pyoracle.Lm(refdec, p2, 1.0)
        o1, o2 = pyoracle.Lm(oracle, p1, -1.0), pyoracle.Lm(oracle, p2, 1.0)
        cd = dict(beam=11.0, max_active=7000, min_active=0, lattice_beam=6.0)
        try:
            oracle.set_order_free(True)
            for u, ll in enumerate(lls):
                O = pyoracle.oracle_raw_lattice(oracle, h, pyoracle.Config(**cd), ll, m)
                if not O.ok:
                    continue
                p = str(tmp_path / "raw.lat")
                with open(p, "wb") as f:
                    f.write(shard.lattice_to_bytes(_as_dict(O)))
                rc, D = pyoracle.det_host_run(lib, O, cap_scale=32)
                assert rc == 0
                for n in (1, 4, 30):
                    ref = pyoracle.ref_nbest_paths_from_lattice_file(refdec, p, 0, n)
                    assert ref is not None and 1 <= len(ref) <= n
                    _same_paths(pyoracle.nshortest_paths(D, n), ref, "seed %d utt %d n %d" % (seed, u, n))
                C2 = pyoracle.compose_lattice(pyoracle.compose_lattice(D, o1), o2)
                ref = pyoracle.ref_nbest_paths_from_lattice_file(refdec, p, 0, 12, r1, r2)
                assert ref is not None
                _same_paths(pyoracle.nshortest_paths(C2, 12), ref, "seed %d utt %d second pass" % (seed, u))
                n_checked += 1
        finally:
            oracle.set_order_free(False)
            oracle.free_graph(h)
            for L in (r1, r2, o1, o2):
                L.free()
    assert n_checked >= 4


@pytest.mark.gpu
def test_device_nbest_paths_equal_the_reference(oracle, refdec, synth, tmp_path):
    import gpu_util as G

    W = G.wfstdec
    n_checked = 0
    most = 0
    for seed in range(2):
        g, m, gp, p1, p2, lls = _setup(synth, tmp_path, seed)
        graph = W.Graph.load(gp)
        graph.set_tid2pdf(m)
        L1, L2 = W.Lm.load(p1, -1.0), W.Lm.load(p2, 1.0)
        r1, r2 = pyoracle.Lm(refdec, p1, -1.0), pyoracle.Lm(refdec, p2, 1.0)
        cd = dict(beam=11.0, max_active=7000, min_active=0, lattice_beam=6.0)
        dec = W.BatchDecoder(graph, G.gpu_config(cd), len(lls), max_frames=64, max_tokens_per_frame=32768, arena_tokens=1 << 20, lattice_links=1 << 21)
        dev = G.upload(lls)
        dec.init()
        dec.advance([t.data_ptr() for t in dev], [20] * len(lls), 300)
        # mid-utterance: served, ascending, the cheapest path first
        mid = dec.nbest_paths(0, 7, use_final_probs=False)
        assert 1 <= len(mid) <= 7 and all(b["tot"] >= a["tot"] for a, b in zip(mid, mid[1:]))
        dec.advance([t.data_ptr() for t in dev], [40] * len(lls), 300)
        dec.finalize()
        for c in reversed(range(len(lls))):   # (not the lowest finalized channel first: see test_compose_lattice)
            raw = dec.raw_lattice(c)
            if raw is None:
                assert dec.nbest_paths(c, 5) == []
                continue
            p = str(tmp_path / "raw.lat")
            with open(p, "wb") as f:
                f.write(shard.lattice_to_bytes(raw))
            for n in (1, 5, 40, 700, 4096):   # (4096: the sort buffer's limit -- candidate lists merged in chunks)
                ref = pyoracle.ref_nbest_paths_from_lattice_file(refdec, p, 0, n)
                assert ref is not None
                got = dec.nbest_paths(c, n)
                _same_paths(got, ref, "seed %d utt %d n %d" % (seed, c, n))
                most = max(most, len(got))
            for n in (3, 60):
                ref = pyoracle.ref_nbest_paths_from_lattice_file(refdec, p, 0, n, r1, r2)
                assert ref is not None
                _same_paths(dec.nbest_paths(c, n, L1, L2), ref, "seed %d utt %d n %d second pass" % (seed, c, n))
            # the short list on the raw lattice agrees on the word sequences
            short = dec.nbest(5)[c]
            long_ = dec.nbest_paths(c, 5)
            assert [tuple(x["words"].tolist()) for x in short] == [tuple(int(w) for w in q["olabel"] if w != 0) for q in long_]
            n_checked += 1
        with pytest.raises(Exception):
            dec.nbest_paths(0, 5000)
        with pytest.raises(Exception):
            dec.nbest_paths(0, 5, L1, None)
        dec.free()
        for L in (r1, r2):
            L.free()
        L1.free()
        L2.free()
        graph.free()
    assert n_checked >= 4
    # (what the DEFAULT data covers; WFST_SYNTH_SEED_OFFSET draws other graphs: the comparisons above hold there, this count is that draw's)
    assert most > 2048 or not (__import__("os").environ.get("WFST_SYNTH_SEED_OFFSET", "0") in ("", "0")), "no lattice with enough paths to fill more than half the sort buffer (%d)" % most


@pytest.mark.gpu
def test_batched_postprocessing_equals_the_per_channel_calls(synth, tmp_path):
    """The service's post-processing as a batch (VERDICT r3 missing #2): wfst_decoder_rescore_lattices and
    wfst_decoder_nbest_paths_batch -- one launch per stage for all finalized channels, a workgroup per lattice -- give, channel by
    channel, what the per-channel calls compute one lattice at a time (on a second decoder fed the same utterances: arc for arc, bit
    for bit), for the second LM pass and for the n-best with and without it; a channel list, and the kept results' lifetime."""
    import time

    import gpu_util as G

    W = G.wfstdec
    g, m, gp, p1, p2, lls = _setup(synth, tmp_path, 0)
    lls = lls + [ll[:30] for ll in lls]   # (a second helping, shorter: other lattices)
    graph = W.Graph.load(gp)
    graph.set_tid2pdf(m)
    L1, L2 = W.Lm.load(p1, -1.0), W.Lm.load(p2, 1.0)
    cd = dict(beam=11.0, max_active=7000, min_active=0, lattice_beam=6.0)
    decs = []
    for _ in range(2):
        dec = W.BatchDecoder(graph, G.gpu_config(cd), len(lls), max_frames=64, max_tokens_per_frame=32768, arena_tokens=1 << 20, lattice_links=1 << 21)
        dev = G.upload(lls)
        dec.init()
        dec.advance([t.data_ptr() for t in dev], [int(x.shape[0]) for x in lls], 300)
        dec.finalize()
        decs.append(dec)
    A, B = decs   # A: batched; B: one channel at a time
    C = len(lls)
    # (A as the service would run it: GetLattice's determinizer started right behind FinalizeDecoding, beside the best paths -- the
    # batched second pass below finds the determinized lattices in the workspace slots and starts from them)
    A.prefetch_determinized()
    A.best_paths()
    eq = lambda x, y: (x is None and y is None) or (x is not None and y is not None and all(np.array_equal(x[k], y[k]) for k in x))
    same_paths = lambda x, y: len(x) == len(y) and all(np.array_equal(p[k], q[k]) for p, q in zip(x, y) for k in ("olabel", "graph", "acoustic")) \
        and all(np.float32(p["tot"]).tobytes() == np.float32(q["tot"]).tobytes() for p, q in zip(x, y))
    t0 = time.perf_counter()
    A.rescore_lattices(L1, L2)
    got = [A.rescored_lattice(c, L1, L2) for c in range(C)]
    t_batch = time.perf_counter() - t0
    t0 = time.perf_counter()
    want = [B.rescored_lattice(c, L1, L2) for c in range(C)]
    t_single = time.perf_counter() - t0
    assert sum(x is not None for x in want) >= C // 2
    for c in range(C):
        assert eq(got[c], want[c]), "second pass, channel %d" % c
    for n, lms in ((10, (L1, L2)), (5, (None, None)), (300, (None, None))):
        A.nbest_paths_batch(n, *lms)
        for c in range(C):
            assert same_paths(A.nbest_paths(c, n, *lms), B.nbest_paths(c, n, *lms)), "n-best %d, channel %d" % (n, c)
    # the single-channel call right behind a batch: the batch's offset buffer holds (n + 1) x C words, its totals buffer n x C -- a
    # request alone with n between the two has to grow the totals buffer on its own size (it wrote past it)
    A.nbest_paths_batch(10)
    assert same_paths(A.nbest_paths(1, 10 * C + 1), B.nbest_paths(1, 10 * C + 1)), "n between the batch's two buffer sizes"
    # a channel list; a request the batch did not cover is computed alone (and agrees)
    A.nbest_paths_batch(7, L1, L2, channels=[C - 1, 0])
    for c in (0, C - 1, 1):
        assert same_paths(A.nbest_paths(c, 7, L1, L2), B.nbest_paths(c, 7, L1, L2)), c
    # the kept results die with the utterance
    A.init(channels=[0])
    with pytest.raises(W.WfstError):
        A.nbest_paths_batch(5, channels=[0])   # not finalized
    print("second pass of %d lattices: batched %.1f ms, one at a time %.1f ms" % (C, 1e3 * t_batch, 1e3 * t_single))
    for d in decs:
        d.free()
    L1.free()
    L2.free()
    graph.free()
