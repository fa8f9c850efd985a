"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on fresh seeded inputs:
BASELINE configs[0]-sized graph (~50k arcs), ragged batches, streaming chunks, partial results,
host-fed matrices, channel reuse, the traceback quirk with parallel arcs, error behaviour.

Bar: best-path word ids, transition-ids, per-hop labels AND float costs, tot/lm score all
bit-identical to the oracle (itself pinned bit-exact to the reference decoder).  The only
tolerated difference is on utterances the oracle flags as having an exact float TIE on the best
path (the reference resolves ties by hash-list arrival order, the GPU by lowest arc index; DESIGN.md
'Deviations'): there the total cost must still be equal."""
import numpy as np
import pytest

import pyoracle
from golden_util import bits

pytestmark = pytest.mark.gpu

BEAM_ONLY = dict(beam=13.0, max_active=1000000, min_active=0, lattice_beam=7.0)


@pytest.fixture(scope="module")
def setup50k(tmp_path_factory, synth, oracle):
    import gpu_util as G

    g = synth.make_hclg_like(14000, seed=7)  # ~50k arcs: BASELINE configs[0]
    path = str(tmp_path_factory.mktemp("g") / "g50k.bin")
    g.write(path)
    m = synth.default_tid2pdf(6000)
    graph = G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    h = oracle.load_graph(path)
    yield dict(g=g, path=path, m=m, graph=graph, h=h, G=G)
    graph.free()
    oracle.free_graph(h)


def _utts(synth, s, lengths, seed0, mu=-2.6):
    return [synth.make_loglikes(s["g"], T, 3000, s["m"], seed=seed0 + i, mu=mu)[0] for i, T in enumerate(lengths)]


def _check(G, r, o, what):
    if o.extra.get("ties", 0):
        assert np.array_equal(bits([r.tot_score]), bits([o.tot_score])) or abs(r.tot_score - o.tot_score) <= 1e-4 * abs(o.tot_score), what
        return False
    G.assert_same_as_oracle(r, o, what)
    return True


def test_config0_single_utterance_3s(setup50k, synth, oracle):
    """BASELINE configs[0]: one 3-second utterance, ~50k-arc HCLG, best path."""
    s = setup50k
    ll = _utts(synth, s, [300], 1000)
    r = s["G"].decode_batch(s["graph"], BEAM_ONLY, ll)[0]
    o = oracle.decode(s["h"], pyoracle.Config(**BEAM_ONLY), ll[0], s["m"])
    assert r.ok and o.ok and len(o.tids) == 300
    s["G"].assert_same_as_oracle(r, o, "config0")


def test_ragged_batch_matches_oracle(setup50k, synth, oracle):
    s = setup50k
    lengths = [120, 1, 37, 300, 64, 2, 255, 90, 199, 10, 150, 77, 31, 280, 5, 111]
    mats = _utts(synth, s, lengths, 2000)
    res = s["G"].decode_batch(s["graph"], BEAM_ONLY, mats)
    cfg = pyoracle.Config(**BEAM_ONLY)
    for i, (r, ll) in enumerate(zip(res, mats)):
        o = oracle.decode(s["h"], cfg, ll, s["m"])
        s["G"].assert_same_as_oracle(r, o, "utt %d (T=%d)" % (i, lengths[i]))
        assert r.stats["frames"] == lengths[i]


@pytest.mark.parametrize("chunk", [1, 7, 25, 64])
def test_streaming_chunks_and_partial_results(setup50k, synth, oracle, chunk):
    """NumFramesReady grows chunk by chunk (kaldi-online-nnet3-my-decoder.cc:32-46); the partial
    best path without final-probs (use_final_probs=false before FinalizeDecoding) must match too."""
    s = setup50k
    mats = _utts(synth, s, [100, 63, 100, 29], 3000)
    cfg = pyoracle.Config(**BEAM_ONLY)
    res = s["G"].decode_batch(s["graph"], BEAM_ONLY, mats, chunk=chunk, finalize=False, use_final_probs=False)
    for i, (r, ll) in enumerate(zip(res, mats)):
        o = oracle.decode(s["h"], cfg, ll, s["m"], chunk=chunk, finalize=False, use_final_probs=False)
        s["G"].assert_same_as_oracle(r, o, "chunk %d utt %d" % (chunk, i))
    res = s["G"].decode_batch(s["graph"], BEAM_ONLY, mats, chunk=chunk, finalize=True)
    for i, (r, ll) in enumerate(zip(res, mats)):
        o = oracle.decode(s["h"], cfg, ll, s["m"], chunk=chunk, finalize=True)
        s["G"].assert_same_as_oracle(r, o, "chunk %d utt %d finalized" % (chunk, i))


def test_host_fed_matrices_equal_device_resident(setup50k, synth):
    s = setup50k
    mats = _utts(synth, s, [80, 80, 41], 4000)
    a = s["G"].decode_batch(s["graph"], BEAM_ONLY, mats, chunk=16)
    b = s["G"].decode_batch(s["graph"], BEAM_ONLY, mats, chunk=16, host_feed=True)
    for x, y in zip(a, b):
        s["G"].assert_same_path(x, y.words, y.tids, y.path_ilabel, y.path_olabel, y.path_graph, y.path_ac,
                                [y.tot_score, y.lm_score])
    # one long hand-over (> 96 frames): advance_host uploads and decodes it in 48-frame slices
    mats = _utts(synth, s, [230, 101, 97, 12], 4100)
    a = s["G"].decode_batch(s["graph"], BEAM_ONLY, mats)
    b = s["G"].decode_batch(s["graph"], BEAM_ONLY, mats, host_feed=True)
    for x, y in zip(a, b):
        s["G"].assert_same_path(x, y.words, y.tids, y.path_ilabel, y.path_olabel, y.path_graph, y.path_ac,
                                [y.tot_score, y.lm_score])


def test_page_locked_rows_in_one_block_go_up_as_2d_copies(setup50k, synth):
    """wfst_decoder_advance_host with PAGE-LOCKED rows that lie equally spaced in host memory (one matrix of utterances; the channel
    pool's row slab): runs of consecutive channels covering the same frames are uploaded as one 2-D copy each, the call returns when
    enqueued.  Ragged lengths (runs break where a channel has ended), a channel subset with a gap, a history that has to regrow
    mid-utterance (max_frames above 1024: it starts at 256 rows): the same paths as from device-resident matrices."""
    import torch

    s = setup50k
    G = s["G"]
    lengths = [300, 300, 300, 41, 300, 280, 300, 7]
    mats = _utts(synth, s, lengths, 4200)
    stride = int(mats[0].shape[1])
    block = torch.zeros((len(mats), max(lengths), stride), dtype=torch.float32).pin_memory()
    for i, x in enumerate(mats):
        block[i, : x.shape[0]] = torch.from_numpy(x)
    rows = [block[i, : lengths[i]].numpy() for i in range(len(mats))]
    assert all(r.ctypes.data - rows[0].ctypes.data == i * max(lengths) * stride * 4 for i, r in enumerate(rows))
    want = G.decode_batch(s["graph"], BEAM_ONLY, mats, limits=dict(max_frames=2000, max_tokens_per_frame=32768, arena_tokens=1 << 22))
    dec = G.wfstdec.BatchDecoder(s["graph"], G.gpu_config(BEAM_ONLY), len(mats), max_frames=2000, max_tokens_per_frame=32768, arena_tokens=1 << 22)
    for channels in (None, [0, 1, 2, 4, 5, 7]):   # (all channels; a list with gaps: runs 0-2, 4-5, 7)
        ch = list(range(len(mats))) if channels is None else channels
        dec.init(channels)
        for r in (16, 100, 200, 300):   # (256 rows per channel to begin with: the third call regrows the history, rows in flight or not)
            dec.advance_host([rows[c] for c in ch], [min(r, lengths[c]) for c in ch], channels=channels)
        dec.finalize(channels)
        got = [G.GpuResult(d) for d in dec.best_paths(channels)]
        for c, y in zip(ch, got):
            x = want[c]
            G.assert_same_path(y, x.words, x.tids, x.path_ilabel, x.path_olabel, x.path_graph, x.path_ac, [x.tot_score, x.lm_score], "channel %d" % c)
    dec.free()


def test_per_frame_best_cost_and_token_subset(setup50k, synth, oracle):
    """Frame by frame: identical best cost; GPU token set is a subset of the reference's (which
    keeps order-dependent extras) and every GPU token cost equals the oracle's for that state."""
    s = setup50k
    ll = _utts(synth, s, [60], 5000)[0]
    cfg = pyoracle.Config(**BEAM_ONLY)
    G = s["G"]
    dec = G.wfstdec.BatchDecoder(s["graph"], G.gpu_config(BEAM_ONLY), 1, max_frames=128, max_tokens_per_frame=32768, arena_tokens=1 << 20)
    dev = G.upload([ll])
    dec.init()
    for f in range(0, 61, 5):
        if f:
            dec.advance([dev[0].data_ptr()], [f], ll.shape[1])
        st, co = dec.frontier(0)
        o = oracle.decode(s["h"], cfg, ll[: max(f, 1)] if f else ll[:1], s["m"], dump_frame=f, dump_cap=1 << 20)
        ost, oco, on = o.dump
        assert on == len(ost)
        ref = dict(zip(ost.tolist(), oco.view(np.int32).tolist()))
        assert len(st) <= on
        assert len(set(st.tolist())) == len(st), "duplicate state in frontier"
        for a, b in zip(st.tolist(), co.view(np.int32).tolist()):
            assert ref.get(a) == b, "frame %d state %d" % (f, a)
        assert co.min() == oco.min()
    dec.free()


def test_channel_reuse_and_partial_channel_lists(setup50k, synth, oracle):
    """InitDecoding again on a used channel; advance only a subset of channels."""
    s = setup50k
    G = s["G"]
    cfg = pyoracle.Config(**BEAM_ONLY)
    mats = _utts(synth, s, [50, 70, 40, 60], 6000)
    dev = G.upload(mats)
    dec = G.wfstdec.BatchDecoder(s["graph"], G.gpu_config(BEAM_ONLY), 4, max_frames=128, max_tokens_per_frame=32768, arena_tokens=1 << 20)
    dec.init()
    dec.advance([dev[1].data_ptr(), dev[3].data_ptr()], [70, 60], 3000, channels=[1, 3])
    assert [dec.num_frames_decoded(c) for c in range(4)] == [0, 70, 0, 60]
    dec.advance([dev[0].data_ptr()], [50], 3000, channels=[0])
    dec.finalize(channels=[0, 1, 3])
    got = dec.best_paths(channels=[0, 1, 3])
    for c, r in zip([0, 1, 3], got):
        o = oracle.decode(s["h"], cfg, mats[c], s["m"])
        G.assert_same_as_oracle(G.GpuResult(r), o, "channel %d" % c)
    # reuse channel 1 for utterance 2, leave the others alone
    dec.init(channels=[1])
    dec.advance([dev[2].data_ptr()], [40], 3000, channels=[1])
    dec.finalize(channels=[1])
    r = G.GpuResult(dec.best_paths(channels=[1])[0])
    G.assert_same_as_oracle(r, oracle.decode(s["h"], cfg, mats[2], s["m"]), "reused channel")
    dec.free()


def test_parallel_arcs_traceback_quirk(synth, oracle, tmp_path):
    """Graphs WITH parallel (src,dst) arcs: GetBestPath must report the same arc the reference's
    first-matching-forward-link search does, with and without FinalizeDecoding."""
    import gpu_util as G

    g = synth.make_hclg_like(3000, seed=21, n_tid=600, n_words=300, allow_parallel=True)
    # force many parallel arcs: redirect every 3rd emitting arc to its predecessor's target
    arcs = g.arcs.copy()
    off = g.row_offsets()
    for st in range(0, 3000, 2):
        b, ne, na = int(off[st]), int(g.state_info["niepsilons"][st]), int(g.state_info["num_arcs"][st])
        for a in range(b + ne + 1, b + na, 2):
            arcs["to"][a] = arcs["to"][a - 1]
    g2 = synth.Graph(g.start, g.final_state, g.state_info, arcs)
    path = str(tmp_path / "par.bin")
    g2.write(path)
    m = synth.default_tid2pdf(600)
    graph = G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    h = oracle.load_graph(path)
    quirks = 0
    for lat_beam in (0.5, 7.0):
        cd = dict(beam=11.0, max_active=1000000, min_active=0, lattice_beam=lat_beam)
        mats = [synth.make_loglikes(g2, 70, 300, m, seed=50 + i, mu=-2.0)[0] for i in range(6)]
        for fin in (True, False):
            res = G.decode_batch(graph, cd, mats, finalize=fin)
            for i, r in enumerate(res):
                o = oracle.decode(h, pyoracle.Config(**cd), mats[i], m, finalize=fin)
                G.assert_same_as_oracle(r, o, "lattice_beam %g finalize %s utt %d" % (lat_beam, fin, i))
                # a quirk hop: the reported arc is not the arg-min one, so tot_score != best token cost
                quirks += int(not np.isclose(o.tot_score, float(np.sum(o.path_graph + o.path_ac))))
    oracle.free_graph(h)
    graph.free()


@pytest.mark.parametrize("cd", [dict(beam=13.0, max_active=2000, min_active=200, lattice_beam=7.0),
                                dict(beam=13.0, max_active=700, min_active=0, lattice_beam=7.0),
                                dict(beam=6.0, max_active=100000, min_active=1500, lattice_beam=4.0)])
def test_max_active_min_active_binding_is_exact_in_order_free_terms(cd, setup50k, synth, oracle):
    """max_active / min_active binding (GetCutoff's k-th smallest cost, adaptive beam).  The
    reference's k-th smallest runs over a token list that still holds its visiting-order dependent
    extras, so ITS result is order dependent; the order-independent
    restatement of the same algorithm (oracle, order-free mode: every arc admitted against the
    frame's final next_cutoff) is reproduced by the GPU bit for bit.  Against the reference's own
    result: a valid path that is not worse by more than 1 % (the reference's extras use up part of
    its max_active budget, so its own path can be the worse one), several utterances identical."""
    s = setup50k
    mats = _utts(synth, s, [120] * 8, 7000, mu=-2.3)
    res = s["G"].decode_batch(s["graph"], cd, mats)
    same = 0
    for i, (r, ll) in enumerate(zip(res, mats)):
        o = oracle.decode(s["h"], pyoracle.Config(**cd), ll, s["m"])
        try:
            oracle.set_order_free(True)
            f = oracle.decode(s["h"], pyoracle.Config(**cd), ll, s["m"])
        finally:
            oracle.set_order_free(False)
        assert r.ok and len(r.tids) == 120
        assert f.extra["ties"] == 0, "exact cost tie on the best path (utt %d)" % i
        s["G"].assert_same_as_oracle(r, f, "utt %d (order-free)" % i)
        default_data = __import__("os").environ.get("WFST_SYNTH_SEED_OFFSET", "0") in ("", "0")   # (the two bounds below were measured on the default data)
        assert r.tot_score <= o.tot_score + 0.01 * abs(o.tot_score) or not default_data
        same += int(np.array_equal(r.words, o.words))
    # measured on these seeds (MI355X, round 2): see the assertion message if it moves
    assert same >= 3 or not default_data, "only %d/8 utterances with the reference's own words" % same


@pytest.mark.parametrize("limit,cd", [(1500, BEAM_ONLY), (64, BEAM_ONLY),
                                      (1200, dict(beam=13.0, max_active=2147483647, min_active=200, lattice_beam=7.0))])
def test_per_frame_limit_degrades_like_max_active(limit, cd, setup50k, synth, oracle):
    """wfst_limits.max_tokens_per_frame reached mid-utterance: the reference never refuses a frame -- it grows its hash
    (base-inl.h:237-244) and pools (util/mem-pool.h:17-65) and tightens with max_active (:188-203).  A best-path decoder here
    keeps the frame's tokens and goes on from the limit-th cheapest: bit for bit the (order-free) reference algorithm run
    with max_active = the limit; wfst_decoder_get_degraded_frames says on how many frames."""
    s = setup50k
    W = s["G"].wfstdec
    mats = _utts(synth, s, [120] * 4, 7100, mu=-2.3)
    dec = W.BatchDecoder(s["graph"], s["G"].gpu_config(cd), len(mats), max_frames=512, max_tokens_per_frame=limit, arena_tokens=1 << 22)
    res = s["G"].decode_batch(s["graph"], cd, mats, dec=dec)
    ocd = dict(cd, max_active=limit)
    try:
        oracle.set_order_free(True)
        for i, (r, ll) in enumerate(zip(res, mats)):
            f = oracle.decode(s["h"], pyoracle.Config(**ocd), ll, s["m"])
            assert r.ok and f.ok and len(r.tids) == 120
            assert f.extra["ties"] == 0, "exact cost tie on the best path (utt %d)" % i
            s["G"].assert_same_as_oracle(r, f, "utt %d (limit %d as max_active)" % (i, limit))
            assert dec.degraded_frames(i) > 0, "the limit never bound (utt %d)" % i
    finally:
        oracle.set_order_free(False)
    # an utterance the limit does not touch reports none
    dec.free()
    dec = W.BatchDecoder(s["graph"], s["G"].gpu_config(cd), 1, max_frames=512, max_tokens_per_frame=32768, arena_tokens=1 << 22)
    s["G"].decode_batch(s["graph"], cd, mats[:1], dec=dec)
    assert dec.degraded_frames(0) == 0
    dec.free()


def test_errors_are_loud(setup50k, synth):
    s = setup50k
    G = s["G"]
    W = G.wfstdec
    ll = _utts(synth, s, [30], 8000)[0]
    dev = G.upload([ll])
    dec = W.BatchDecoder(s["graph"], G.gpu_config(BEAM_ONLY), 2, max_frames=64, max_tokens_per_frame=4096, arena_tokens=1 << 18)
    with pytest.raises(W.WfstError) as e:  # AdvanceDecoding before InitDecoding
        dec.advance([dev[0].data_ptr()], [30], 3000, channels=[0])
    assert e.value.code == -5
    dec.init()
    with pytest.raises(W.WfstError) as e:  # stride smaller than the columns the graph reads
        dec.advance([dev[0].data_ptr()], [30], 100, channels=[0])
    assert e.value.code == -1
    with pytest.raises(W.WfstError) as e:  # longer than max_frames
        dec.advance([dev[0].data_ptr()], [100], 3000, channels=[0])
    assert e.value.code == -4
    dec.advance([dev[0].data_ptr()], [30], 3000, channels=[0])
    dec.finalize(channels=[0])
    with pytest.raises(W.WfstError) as e:  # reference: LOG_ERR in BestPathEnd (base-inl.h:1100-1102)
        dec.best_paths(channels=[0], use_final_probs=False)
    assert e.value.code == -5
    with pytest.raises(W.WfstError) as e:  # AdvanceDecoding after FinalizeDecoding
        dec.advance([dev[0].data_ptr()], [30], 3000, channels=[0])
    assert e.value.code == -5
    assert dec.best_paths(channels=[1])[0]["ok"] is False  # no frames decoded -> GetBestPath false
    dec.free()
    # an utterance that does not fit its token arena must fail, not silently drop tokens (the per-frame token limit of a
    # best-path decoder is no such capacity: test_per_frame_limit_degrades_like_max_active)
    tiny = W.BatchDecoder(s["graph"], G.gpu_config(BEAM_ONLY), 1, max_frames=64, max_tokens_per_frame=4096, arena_tokens=1 << 10)
    tiny.init()
    tiny.advance([dev[0].data_ptr()], [30], 3000)
    with pytest.raises(W.WfstError) as e:
        tiny.sync()
    assert e.value.code == -4
    tiny.free()
    with pytest.raises(W.WfstError) as e:
        W.Graph.load("/nonexistent/graph.bin")
    assert e.value.code == -2


def test_openfst_vector_and_const_graphs_decode_like_the_flat_graph(synth, oracle, tmp_path):
    """Graph ingestion (8(f) rank 4) end to end: the same graph loaded from the reference's flat
    format, an OpenFst vector fst and an OpenFst const fst (plain and 16-byte aligned) through
    wfst_graph_load gives bit-identical decodes, equal to the oracle's on the flat file."""
    import gpu_util as G

    g = synth.make_hclg_like(5000, seed=13, n_tid=600, n_words=900)
    m = synth.default_tid2pdf(600)
    flat = str(tmp_path / "g.flat")
    g.write(flat)
    files = {"flat": flat}
    for name, kw in (("vector", dict(fst_type="vector")), ("const", dict(fst_type="const")),
                     ("const_aligned", dict(fst_type="const", aligned=True))):
        files[name] = str(tmp_path / (name + ".fst"))
        with open(files[name], "wb") as f:
            f.write(synth.to_openfst_bytes(g, **kw))
    cd = dict(beam=12.0, max_active=1000000, min_active=0, lattice_beam=6.0)
    mats = [synth.make_loglikes(g, T, 300, m, seed=700 + i, mu=-2.2)[0] for i, T in enumerate((70, 33, 5))]
    h = oracle.load_graph(flat)
    want = [oracle.decode(h, pyoracle.Config(**cd), x, m) for x in mats]
    oracle.free_graph(h)
    for name, path in files.items():
        graph = G.wfstdec.Graph.load(path)
        graph.set_tid2pdf(m)
        assert graph.info()["n_states"] == g.n_states and graph.info()["n_arcs"] == g.n_arcs, name
        for r, o in zip(G.decode_batch(graph, cd, mats), want):
            G.assert_same_as_oracle(r, o, name)
        graph.free()
    with pytest.raises(G.wfstdec.WfstError):
        bad = str(tmp_path / "sym.fst")
        with open(bad, "wb") as f:
            f.write(synth.to_openfst_bytes(g, "const", flags=2))
        G.wfstdec.Graph.load(bad)


def test_channel_groups_decode_identically(synth, oracle, tmp_path):
    """wfst_options.channel_groups = 2/3 (one hipGraph + stream per channel group) and use_hip_graph = 0 are
    scheduling choices only: same bits as the oracle, ragged lengths included."""
    import gpu_util as G

    g = synth.make_hclg_like(6000, seed=29, n_tid=600, n_words=900)
    m = synth.default_tid2pdf(600)
    path = str(tmp_path / "g.bin")
    g.write(path)
    graph = G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    cd = dict(beam=12.0, max_active=1000000, min_active=0, lattice_beam=6.0)
    mats = [synth.make_loglikes(g, T, 300, m, seed=800 + i, mu=-2.2)[0] for i, T in enumerate((60, 9, 41, 60, 1, 33, 17))]
    h = oracle.load_graph(path)
    want = [oracle.decode(h, pyoracle.Config(**cd), x, m) for x in mats]
    oracle.free_graph(h)
    lim = dict(max_frames=512, max_tokens_per_frame=32768, arena_tokens=1 << 22)
    for opt in (dict(channel_groups=2), dict(channel_groups=3), dict(use_hip_graph=0), dict(channel_groups=2, use_hip_graph=0),
                dict(log2_partitions=0, log2_lds_slots=8), dict(joint_max=64, expand_workgroups=3, insert_workgroups=7),
                dict(upload_slice_frames=0)):
        for chunk in (0, 7):
            dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(mats), options=G.wfstdec.Options(**opt), **lim)
            for r, o in zip(G.decode_batch(graph, cd, mats, chunk=chunk, dec=dec, host_feed="upload_slice_frames" in opt), want):
                G.assert_same_as_oracle(r, o, "%s chunk %d" % (opt, chunk))
            dec.free()
    with pytest.raises(G.wfstdec.WfstError):
        G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), 2, options=G.wfstdec.Options(channel_groups=9), **lim)
    # unflattened closures / packed rows: graph upload choices, same bits
    # unflattened closures / packed rows / no fused closures (the separate closure pass): graph upload choices, same bits
    for go in (dict(flatten_closures=0, row_align_slots=1), dict(fuse_closures=0), dict(fuse_closures=0, flatten_closures=0)):
        g2 = G.wfstdec.Graph.load(path, options=G.wfstdec.GraphOptions(**go))
        g2.set_tid2pdf(m)
        for chunk in (0, 7):
            for r, o in zip(G.decode_batch(g2, cd, mats, chunk=chunk), want):
                G.assert_same_as_oracle(r, o, "graph options %s chunk %d" % (go, chunk))
        g2.free()
    graph.free()


def test_long_calls_split_into_head_and_rest_graphs(synth, oracle, tmp_path):
    """An AdvanceDecoding call of 128 frames and more is launched as a 32-frame head graph and the rest (wfst_capi.cc: the
    device sees the first frames while the calling thread is still submitting the others); shorter calls, calls that start
    mid-utterance, ragged lengths (channels that end inside the head, inside the rest, or before the call) and the tile size
    knob (wfst_options.tile_tokens) are scheduling only: same bits as the oracle, for best-path and lattice decoders, on one
    channel group and on three."""
    import gpu_util as G

    g = synth.make_hclg_like(6000, seed=31, n_tid=600, n_words=900)
    m = synth.default_tid2pdf(600)
    path = str(tmp_path / "g.bin")
    g.write(path)
    graph = G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    cd = dict(beam=11.0, max_active=1000000, min_active=0, lattice_beam=6.0)
    Ts = (190, 133, 20, 160, 128, 45, 175, 1)
    mats = [synth.make_loglikes(g, T, 300, m, seed=900 + i, mu=-2.2)[0] for i, T in enumerate(Ts)]
    h = oracle.load_graph(path)
    want = [oracle.decode(h, pyoracle.Config(**cd), x, m) for x in mats]
    oracle.free_graph(h)
    for lat in (0, 1 << 21):
        lim = dict(max_frames=256, max_tokens_per_frame=32768, arena_tokens=1 << 22, lattice_links=lat)
        for opt in (dict(channel_groups=1), dict(channel_groups=3), dict(channel_groups=2, tile_tokens=64), dict(tile_tokens=136)):
            for chunk in (0, 150, 40):   # one call of 190 frames (split); 150 + 40 (split + whole); five short calls
                dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(mats), options=G.wfstdec.Options(**opt), **lim)
                for r, o in zip(G.decode_batch(graph, cd, mats, chunk=chunk, dec=dec), want):
                    G.assert_same_as_oracle(r, o, "lattice_links %d %s chunk %d" % (lat, opt, chunk))
                dec.free()
    graph.free()
