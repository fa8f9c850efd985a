"""-m gpu: token collection of BEST-PATH decoders ("tokens are never reclaimed").  A best-path decoder keeps
no forward links to prune its arena by; when the arena is half full it keeps what the frontier's backpointers reach
and moves it down (gc_pass in wfst_kernels.hip).  GetBestPath must not notice: long utterances in arenas a small
fraction of what they create decode to the oracle's path bit for bit -- fused and plain closure passes, biglm,
ragged batches, chunked advances with partial results in between."""
import numpy as np
import pytest

import pyoracle
from golden_util import bits

pytestmark = pytest.mark.gpu


def _graph(synth, tmp_path, n_states=5000, seed=17, **go):
    import gpu_util as G

    g = synth.make_hclg_like(n_states, seed=seed, n_tid=2000, n_words=3000)
    m = synth.default_tid2pdf(2000)
    path = str(tmp_path / "g.bin")
    g.write(path)
    graph = G.wfstdec.Graph.load(path, options=G.wfstdec.GraphOptions(**go)) if go else G.wfstdec.Graph.load(path)
    graph.set_tid2pdf(m)
    return G, g, m, path, graph


def _same(r, o, what):
    assert np.array_equal(r["tids"], o.tids) and np.array_equal(r["words"], o.words), what
    assert np.array_equal(bits(r["graph"]), bits(o.path_graph)) and np.array_equal(bits(r["ac"]), bits(o.path_ac)), what
    assert np.float32(r["tot_score"]).tobytes() == np.float32(o.tot_score).tobytes(), what


@pytest.mark.parametrize("fuse", [1, 0])
def test_long_utterances_in_a_small_arena(fuse, synth, oracle, tmp_path):
    G, g, m, path, graph = _graph(synth, tmp_path, fuse_closures=fuse)
    cd = dict(beam=10.0, max_active=1000000, min_active=0, lattice_beam=5.0)
    T = [3000, 1777, 333]
    mats = [synth.make_loglikes(g, t, 1000, m, seed=70 + i, mu=-2.5)[0] for i, t in enumerate(T)]
    h = oracle.load_graph(path)
    try:
        want = [oracle.decode(h, pyoracle.Config(**cd), x, m) for x in mats]
        created = max(o.extra["tokens_created"] for o in want)
        arena = int(created // 12)
        assert all(o.extra["ties"] == 0 for o in want)
        dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), len(T), max_frames=3100, max_tokens_per_frame=16384, arena_tokens=arena)
        dev = G.upload(mats)
        dec.init()
        n_partial = 0
        for r in list(range(97, max(T), 97)) + [max(T)]:
            dec.advance([t.data_ptr() for t in dev], [min(r, t) for t in T], 1000)
            if r % (97 * 8) == 0:   # partial results (use_final_probs = false) after collections have run
                part = dec.best_paths(use_final_probs=False, cap=2 * max(T) + 64)
                for c in range(len(T)):
                    o = oracle.decode(h, pyoracle.Config(**cd), mats[c][: min(r, T[c])], m, finalize=False, use_final_probs=False)
                    assert np.array_equal(part[c]["tids"], o.tids) and np.array_equal(bits(part[c]["graph"]), bits(o.path_graph)), (r, c)
                    n_partial += 1
        dec.sync()
        st = [dec.stats(c) for c in range(len(T))]
        dec.finalize()
        best = dec.best_paths(cap=2 * max(T) + 64)
        for c in range(len(T)):
            _same(best[c], want[c], "channel %d" % c)
        print("created", created, "arena", arena, "collections", [s["collections"] for s in st])
        assert st[0]["tokens"] > 8 * arena and st[0]["collections"] >= 8 and st[2]["collections"] < st[0]["collections"] and n_partial >= 6
        # the channels are reusable after it
        dec.init()
        dec.advance([t.data_ptr() for t in dev], T, 1000)
        dec.finalize()
        again = dec.best_paths(cap=2 * max(T) + 64)
        for c in range(len(T)):
            _same(again[c], want[c], "second pass, channel %d" % c)
        dec.free()
    finally:
        oracle.free_graph(h)
        graph.free()


def test_binding_limits_and_a_full_arena(synth, oracle, tmp_path):
    """max_active binding (order-free oracle) with collections; and an arena that one collection cannot bring below
    its mark is refused loudly, not silently."""
    G, g, m, path, graph = _graph(synth, tmp_path, n_states=8000, seed=5)
    cd = dict(beam=11.0, max_active=600, min_active=100, lattice_beam=5.0)
    mats = [synth.make_loglikes(g, 1500, 1000, m, seed=30 + i, mu=-2.3)[0] for i in range(2)]
    h = oracle.load_graph(path)
    try:
        oracle.set_order_free(True)
        want = [oracle.decode(h, pyoracle.Config(**cd), x, m) for x in mats]
        arena = int(max(o.extra["tokens_created"] for o in want) // 10)
        dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), 2, max_frames=1600, max_tokens_per_frame=16384, arena_tokens=arena)
        dev = G.upload(mats)
        dec.init()
        dec.advance([t.data_ptr() for t in dev], [1500, 1500], 1000)
        dec.finalize()
        best = dec.best_paths(cap=4096)
        for c in range(2):
            if want[c].extra["ties"] == 0:
                _same(best[c], want[c], "channel %d" % c)
        assert dec.stats(0)["collections"] >= 5
        dec.free()
        tiny = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), 1, max_frames=1600, max_tokens_per_frame=16384, arena_tokens=1500)
        tiny.init()
        with pytest.raises(G.wfstdec.WfstError) as ei:
            tiny.advance([dev[0].data_ptr()], [1500], 1000)
            tiny.finalize()
            tiny.best_paths(cap=4096)
        assert "arena" in str(ei.value)
        tiny.free()
    finally:
        oracle.set_order_free(False)
        oracle.free_graph(h)
        graph.free()
