"""-m gpu: BASELINE configs[1] sizes (batch 128 x 300 frames, ~10M-arc HCLG, beam 13).
The oracle needs ~0.5 s per utterance here, so it checks a sample; the whole batch is covered by
size-independent properties: batch invariance (an utterance decodes identically alone, in another
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


LIM = dict(max_frames=304, max_tokens_per_frame=131072, arena_tokens=300 * 20000)
CD = dict(beam=13.0, max_active=1000000, min_active=0, lattice_beam=7.0)


def test_batch128_parity_sample_and_properties(big, oracle):
    G = big["G"]
    res = G.decode_batch(big["graph"], CD, big["mats"], limits=LIM)
    assert all(r.ok and len(r.tids) == big["T"] for r in res)
    # oracle on a sample: bit-exact labels and costs
    h = oracle.load_graph(big["path"])
    cfg = pyoracle.Config(**CD)
    for u in (0, 17, 42, 77, 101, 127):
        o = oracle.decode(h, cfg, big["mats"][u], big["m"])
        if o.extra["ties"] == 0:
            G.assert_same_as_oracle(res[u], o, "utt %d" % u)
        # GPU work counters use the reference loop's definitions: they may only fall short by the
        # few order-dependent extras the reference expands at exact-equality cutoffs
        assert abs(res[u].stats["N"] - o.extra["N"]) <= 1e-3 * o.extra["N"]
        assert abs(res[u].stats["E"] - o.extra["E"]) <= 1e-3 * o.extra["E"]
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
    if o.extra["ties"] == 0:
        G.assert_same_as_oracle(wide[0], o, "beam 15")
    oracle.free_graph(h)
