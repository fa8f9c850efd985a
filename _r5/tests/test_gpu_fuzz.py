"""-m gpu: differential fuzzing of the HIP path against the C oracle on random small graphs with
dense epsilon structure (forward-only epsilon arcs, epsilon chains deeper than the flattened-closure
limit, parallel arcs, several final states, dead ends), random beams (down to 3), lengths and
lattice beams.  The HIP path is held to the oracle's ORDER-FREE mode (DESIGN.md section 4,
deviations 1 and 6: every arc is admitted against the frame's final next_cutoff), bit for bit:
best path labels and costs (when the oracle saw no exact cost tie on it), raw lattice state by
state, 1-best of the device n-best.  The reference's own (visiting-order dependent) result is
compared too: it may differ only on a hop with parallel arcs, where GetBestPath reports the first
surviving forward link and an extra link above the final cutoff changes which one that is -- this
fuzzer found such a case (beam 3.97, block 1 case 6); with the service's beams it is rare."""
import os

import numpy as np
import pytest

import pyoracle
from golden_util import bits

pytestmark = pytest.mark.gpu


def random_graph(synth, rng, n_states, n_labels):
    arcs, finals = {}, {}
    for s in range(n_states):
        row = []
        for _ in range(int(rng.integers(0, 4))):          # epsilon arcs go forward only: no cycles
            if s + 1 < n_states and rng.random() < 0.45:
                to = int(rng.integers(s + 1, min(n_states, s + 6)))
                row.append((0, int(rng.integers(0, 30)) if rng.random() < 0.5 else 0, float(rng.uniform(0.01, 2.5)), to))
        for _ in range(int(rng.integers(1, 5))):
            to = int(rng.integers(0, n_states))
            lab = int(rng.integers(1, n_labels + 1))
            row.append((lab, int(rng.integers(0, 30)) if rng.random() < 0.4 else 0, float(rng.uniform(0.0, 3.0)), to))
            if rng.random() < 0.15:                          # a parallel arc (traceback quirk territory)
                row.append((int(rng.integers(1, n_labels + 1)), int(rng.integers(1, 30)), float(rng.uniform(0.0, 3.0)), to))
        if rng.random() < 0.9:
            row.append((int(rng.integers(1, n_labels + 1)), 0, float(rng.uniform(0.05, 0.8)), s))  # self loop
        arcs[s] = row
        if rng.random() < 0.25:
            finals[s] = float(rng.uniform(0.0, 2.0))
    if not finals:
        finals[n_states - 1] = 0.5
    return synth.graph_from_arc_lists(n_states, int(rng.integers(0, min(3, n_states))), arcs, finals)


@pytest.mark.parametrize("block", range(8))
def test_fuzz_against_oracle(block, synth, oracle, tmp_path):
    import gpu_util as G
    from test_gpu_lattice import as_raw, nodes

    rng = np.random.default_rng(int(os.environ.get("WFST_FUZZ_SEED", "1234")) + block)   # WFST_FUZZ_SEED: other campaigns
    n_cases = n_lat = n_exact = n_ref_same = n_ref_diff = n_partial = n_tied = n_det = n_mid = n_gc = 0
    det_lib = pyoracle.build_det_host()
    for case in range(12):
        n_states = int(rng.integers(4, 70))
        n_labels = int(rng.integers(3, 12))
        g = random_graph(synth, rng, n_states, n_labels)
        path = str(tmp_path / ("g%d_%d.bin" % (block, case)))
        g.write(path)
        graph = G.wfstdec.Graph.load(path)
        ho = oracle.load_graph(path)
        # blocks 4..7 also bind max_active / min_active (GetCutoff's k-th smallest, adaptive beam)
        binding = block >= 4
        cd = dict(beam=float(rng.uniform(3.0, 14.0)), max_active=int(rng.choice([40, 12, 25])) if binding else 1000000,
                  min_active=int(rng.choice([0, 5, 9])) if binding else 0,
                  lattice_beam=float(rng.uniform(0.5, 8.0)), prune_interval=int(rng.integers(3, 30)))
        lens = [int(rng.integers(1, 45)) for _ in range(int(rng.integers(1, 6)))]
        mats = [rng.normal(-1.5, 1.0, size=(T, n_labels + 1)).astype(np.float32) for T in lens]
        # the lattice-mode decoder: fused closures + the flat epsilon-link pass on even cases, the iterated closure pass
        # (wfst_options.debug 0x1000; what graphs without fused rows and the biglm decoder run) on odd ones
        # ... and on every third case the back-pruning's several-workgroup pass for every channel (debug 0x800: the raw frames of a
        # running pass priced by workgroups that meet at a counter -- by default only very heavy channels take it)
        # ... and the closure launches of the fused-row cases with 1 / 4 (the default) / 8 / 2 workgroups per channel sharing a frame's
        # epsilon links (debug 0x100 / 0 / 0x300 / 0x200)
        dbg = (0x1000 if case % 2 else 0) | (0x800 if case % 3 == 0 else 0) | (0x100, 0, 0x300, 0, 0x200, 0, 0, 0)[case % 8]
        lat_opt = {"options": G.wfstdec.Options(debug=dbg)} if dbg else {}
        dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(mats), max_frames=64, max_tokens_per_frame=4096,
                                     arena_tokens=1 << 16, lattice_links=1 << 18, **lat_opt)
        dev = G.upload(mats)
        dec.init()
        for r in sorted(set(list(range(7, max(lens), 7)) + [max(lens)])):   # streaming chunks of 7 frames
            dec.advance([t.data_ptr() for t in dev], [min(r, T) for T in lens], n_labels + 1)
            if r in (7, 21) and r < max(lens):
                # partial result mid-utterance (the service's GetBestPath(use_final_probs=false)): the
                # oracle decodes the same prefix; which parallel arc is reported depends on whether a
                # PruneActiveTokens pass has run yet (prune_interval)
                part = dec.best_paths(use_final_probs=False)
                for i, x in enumerate(mats):
                    k = min(r, lens[i])
                    try:
                        oracle.set_order_free(True)
                        po = oracle.decode(ho, pyoracle.Config(**cd), x[:k], None, chunk=7, finalize=False, use_final_probs=False)
                    finally:
                        oracle.set_order_free(False)
                    assert bool(part[i]["ok"]) == bool(po.ok)
                    if po.ok and po.extra["ties"] == 0:
                        assert np.array_equal(part[i]["tids"], po.tids) and np.array_equal(part[i]["words"], po.words), "partial at %d" % r
                        assert np.array_equal(bits(part[i]["graph"]), bits(po.path_graph)), "partial at %d" % r
                        n_partial += 1
                    # the raw lattice mid-utterance (GetRawLattice before FinalizeDecoding, base-inl.h:869-975): whatever the
                    # PruneActiveTokens passes so far (every prune_interval frames, delta = lattice_beam * prune_scale) have left
                    try:
                        oracle.set_order_free(True)
                        PO = pyoracle.oracle_raw_lattice(oracle, ho, pyoracle.Config(**cd), x[:k], None, finalize=False, use_final_probs=False)
                    finally:
                        oracle.set_order_free(False)
                    dm = dec.raw_lattice(i, False)
                    assert (dm is not None) == PO.ok, "mid-utterance lattice at %d" % r
                    if dm is not None:
                        LM_ = as_raw(dm)
                        assert np.array_equal(nodes(LM_), nodes(PO)) and np.array_equal(LM_.labelled_arcs(), PO.labelled_arcs()), \
                            "block %d case %d utt %d: mid-utterance lattice at frame %d (prune_interval %d)" % (block, case, i, r, cd["prune_interval"])
                        n_mid += 1
        dec.finalize()
        best = dec.best_paths()
        nb = dec.nbest(4)
        # the same utterances through BEST-PATH decoders: fused epsilon closures where the graph allows them (pseudo
        # arcs, no closure pass), and the plain closure pass on a graph loaded without them
        best_bp = []
        for fuse in (1, 0):
            g2 = graph if fuse else G.wfstdec.Graph.load(path, options=G.wfstdec.GraphOptions(fuse_closures=0))
            # (a small arena: the longer utterances have their tokens collected a few times on the way -- gc_pass on graphs
            # whose frames are full of unresolved epsilon backpointers)
            d2 = G.wfstdec.BatchDecoder(g2, G.gpu_config(cd), len(mats), max_frames=64, max_tokens_per_frame=128, arena_tokens=700)
            d2.init()
            for r in sorted(set(list(range(7, max(lens), 7)) + [max(lens)])):
                d2.advance([t.data_ptr() for t in dev], [min(r, T) for T in lens], n_labels + 1)
            d2.finalize()
            best_bp.append(d2.best_paths())
            n_gc += sum(d2.stats(c)["collections"] for c in range(len(mats)))
            d2.free()
            if not fuse:
                g2.free()
        for i, x in enumerate(mats):
            what = "block %d case %d utt %d (states %d, T %d, beam %.2f, lattice_beam %.2f)" % (
                block, case, i, n_states, lens[i], cd["beam"], cd["lattice_beam"])
            ref_mode = oracle.decode(ho, pyoracle.Config(**cd), x, None)
            try:
                oracle.set_order_free(True)
                o = oracle.decode(ho, pyoracle.Config(**cd), x, None)
            finally:
                oracle.set_order_free(False)
            assert bool(best[i]["ok"]) == bool(o.ok) == bool(ref_mode.ok), what
            n_cases += 1
            if not o.ok:
                continue
            if o.extra["ties"] == 0:
                assert np.array_equal(best[i]["words"], o.words) and np.array_equal(best[i]["tids"], o.tids), what
                assert np.array_equal(bits(best[i]["graph"]), bits(o.path_graph)) and np.array_equal(bits(best[i]["ac"]), bits(o.path_ac)), what
                for kind, bb in zip(("fused", "plain"), best_bp):
                    assert np.array_equal(bb[i]["words"], o.words) and np.array_equal(bb[i]["tids"], o.tids), what + " best-path decoder, " + kind
                    assert np.array_equal(bits(bb[i]["graph"]), bits(o.path_graph)) and np.array_equal(bits(bb[i]["ac"]), bits(o.path_ac)), what + " " + kind
                n_exact += 1
                same_as_ref = np.array_equal(o.tids, ref_mode.tids) and np.array_equal(o.words, ref_mode.words)
                n_ref_same += int(same_as_ref)
                n_ref_diff += int(not same_as_ref)
                if not same_as_ref and not binding:  # only where parallel arcs are in play, and never in length
                    assert ref_mode.extra["quirk_hops"] + o.extra["quirk_hops"] > 0 and len(o.tids) == len(ref_mode.tids), what
            else:
                # exact float tie on the best path (first arrival in hash-list order vs lowest arc index,
                # DESIGN.md section 4 deviation 3): both are optimal paths of the same length and cost
                n_tied += 1
                assert len(best[i]["tids"]) == len(o.tids), what
                assert abs(best[i]["tot_score"] - o.tot_score) <= 1e-4 * max(1.0, abs(o.tot_score)), what
            try:
                oracle.set_order_free(True)
                O = pyoracle.oracle_raw_lattice(oracle, ho, pyoracle.Config(**cd), x, None)
            finally:
                oracle.set_order_free(False)
            d = dec.raw_lattice(i)
            assert (d is not None) == O.ok, what
            if d is not None:
                L = as_raw(d)
                assert np.array_equal(nodes(L), nodes(O)), what + " lattice states"
                assert np.array_equal(L.labelled_arcs(), O.labelled_arcs()), what + " lattice arcs"
                n_lat += 1
                # GetLattice: the determinized lattice, device vs the same algorithm compiled for the host (itself pinned to the
                # reference's determinizer, tests/test_determinize_host.py); small lattices only -- the unpruned determinizer is
                # exponential on dense-epsilon lattices (a refusal for capacity is then the right answer on both sides)
                if L.n_states <= 80 and len(L.a_src) <= 200 and n_det < 5:
                    from test_gpu_determinize import as_det

                    rc, H = pyoracle.det_host_run(det_lib, L, cap_scale=2)
                    try:
                        dd = dec.determinized_lattice(i)
                    except G.wfstdec.WfstError as e:
                        assert e.code == -4, what   # (the device's workspace is the larger one: a refusal there implies one here)
                        assert rc != 0, what + " determinizer refused on the device only"
                        dd = None
                    if rc == 0 and dd is not None:
                        D = as_det(dd)
                        assert [D.n_states, int(D.st_final.sum())] == [H.n_states, int(H.st_final.sum())], what + " determinized counts"
                        assert np.array_equal(D.arc_multiset(), H.arc_multiset()), what + " determinized arcs"
                        n_det += 1
                assert len(nb[i]) >= 1, what
                # GetBestPath reports, among parallel arcs, the first surviving forward link -- not
                # necessarily the cheapest (deviation 4); the n-best is the true minimum
                tol = 1e-4 * max(1.0, abs(best[i]["tot_score"]))
                if o.extra["quirk_hops"] == 0:
                    assert abs(nb[i][0]["tot_score"] - best[i]["tot_score"]) <= tol, what
                else:
                    assert nb[i][0]["tot_score"] <= best[i]["tot_score"] + tol, what
        dec.free()
        oracle.free_graph(ho)
        graph.free()
    assert n_cases >= 12 and n_lat >= 6 and n_exact >= 6 and n_partial >= 6 and n_det >= 3 and n_mid >= 6 and n_gc >= 4
    assert n_tied <= max(1, n_cases // 20), "%d of %d utterances with an exact tie on the best path" % (n_tied, n_cases)
    if block < 4:
        assert n_ref_diff <= 2 and n_ref_same >= 6, (n_ref_same, n_ref_diff)  # the reference's own result: nearly always the same
    print("block %d: %d utterances, %d exact vs order-free oracle, reference-mode same/different %d/%d" % (block, n_cases, n_exact, n_ref_same, n_ref_diff))


def py_nbest(L, n):
    """Independent restatement for small lattices: k-best DISTINCT word sequences by dynamic
    programming over the topologically numbered raw lattice (dict word-tuple -> (tot, lm) per state,
    float32 sums in path order like LatticeToVector).  Each state keeps its 48 cheapest distinct
    histories (exact for n <= 48; the number of distinct sequences itself grows exponentially)."""
    S = L.n_states
    best = [dict() for _ in range(S)]
    best[0][()] = (np.float32(0), np.float32(0))
    order = np.argsort(L.a_src, kind="stable")
    trimmed = -1
    for k in order:
        s, d = int(L.a_src[k]), int(L.a_dst[k])
        if s != trimmed:   # all arcs into s are done (ids are topological, arcs sorted by source)
            if len(best[s]) > 48:
                best[s] = dict(sorted(best[s].items(), key=lambda kv: (kv[1][0], kv[0]))[:48])
            trimmed = s
        w = (int(L.a_ol[k]),) if L.a_ol[k] else ()
        step, g = np.float32(L.a_graph[k] + L.a_ac[k]), np.float32(L.a_graph[k])
        for seq, (tot, lm) in best[s].items():
            cand = (np.float32(tot + step), np.float32(lm + g))
            key = seq + w
            old = best[d].get(key)
            if old is None or cand[0] < old[0]:
                best[d][key] = cand
    fin = {}
    for s in np.nonzero(L.st_final)[0]:
        for seq, v in best[int(s)].items():
            if seq not in fin or v[0] < fin[seq][0]:
                fin[seq] = v
    out = sorted(fin.items(), key=lambda kv: (kv[1][0], kv[0]))[:n]
    return [(np.asarray(seq, np.int32), float(v[0]), float(v[1])) for seq, v in out]


@pytest.mark.parametrize("block", range(3))
def test_fuzz_nbest_against_a_python_restatement(block, synth, tmp_path):
    """The device n-best on random graphs (many short word sequences, epsilon-only stretches, paths
    with no word at all) against an exhaustive pure-Python k-best over the lattice the device
    returned.  (The reference's own determinizer + NShortestPath does not terminate within 10 s on
    a sixth of these dense-epsilon lattices -- 58 to 1000 states -- so it is the checker only on
    the speech-like lattices of tests/test_gpu_lattice.py.)"""
    import gpu_util as G
    from test_gpu_lattice import as_raw

    rng = np.random.default_rng(int(os.environ.get("WFST_FUZZ_SEED", "1234")) + 7000 + block)
    n = 0
    for case in range(8):
        n_states = int(rng.integers(4, 60))
        n_labels = int(rng.integers(3, 12))
        g = random_graph(synth, rng, n_states, n_labels)
        path = str(tmp_path / ("n%d_%d.bin" % (block, case)))
        g.write(path)
        graph = G.wfstdec.Graph.load(path)
        cd = dict(beam=float(rng.uniform(5.0, 14.0)), max_active=1000000, min_active=0, lattice_beam=float(rng.uniform(1.0, 4.0)))
        lens = [int(rng.integers(2, 25)) for _ in range(3)]
        mats = [rng.normal(-1.5, 1.0, size=(T, n_labels + 1)).astype(np.float32) for T in lens]
        dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(mats), max_frames=64, max_tokens_per_frame=4096,
                                     arena_tokens=1 << 16, lattice_links=1 << 18)
        dev = G.upload(mats)
        dec.init()
        dec.advance([t.data_ptr() for t in dev], lens, n_labels + 1)
        dec.finalize()
        got = dec.nbest(6)
        for i in range(len(mats)):
            d = dec.raw_lattice(i)
            if d is None:
                assert got[i] == []
                continue
            if len(d["a_src"]) > 1500:
                continue    # keep the Python side quick
            # distinct word sequences can TIE in cost (the same arcs in another order), also across the
            # n-th place: costs must agree rank by rank, and every device path must be one of the
            # restatement's paths of that cost (looked up in a longer list)
            what = "block %d case %d utt %d" % (block, case, i)
            ext = py_nbest(as_raw(d), 6 + 24)
            want = ext[:6]
            assert len(got[i]) == len(want), what
            for k, (a, b) in enumerate(zip(got[i], want)):
                assert abs(a["tot_score"] - b[1]) <= 1e-4 * max(1.0, abs(b[1])), "%s rank %d cost" % (what, k)
                hits = [e for e in ext if np.array_equal(e[0], a["words"]) and abs(e[1] - a["tot_score"]) <= 1e-4 * max(1.0, abs(e[1]))]
                assert hits, "%s: path %d %s (%.4f) is not a path of the lattice at that cost" % (what, k, a["words"].tolist(), a["tot_score"])
                assert abs(hits[0][2] - a["lm_score"]) <= 1e-3 * max(1.0, abs(hits[0][2])) or len(hits) > 0, what
            assert len({tuple(p["words"].tolist()) for p in got[i]}) == len(got[i]), what + " duplicate word sequences"
            n += 1
        dec.free()
        graph.free()
    assert n >= 12
