"""Exploration: lattice mode at BASELINE configs[1] size -- link counts, finalize (prune) time,
GetRawLattice extraction time, lattice sizes."""
import importlib, sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("asr-decoder_amd")
synth, wfstdec = pkg.synth, pkg.wfstdec
import torch
B, T = int(os.environ.get("B", 128)), 300
g = synth.make_hclg_like(2850000, seed=7)
m = synth.default_tid2pdf(6000)
graph = wfstdec.Graph.from_arrays(g.start, g.final_state, g.state_info, g.arcs)
graph.set_tid2pdf(m)
mats = [synth.make_loglikes_multi(g, T, 3000, m, seed=u, n_paths=272, mu=-4.0, jitter=0.5, ac_lo=0.5)[0] for u in range(B)]
dev = [torch.from_numpy(x).to("cuda:0") for x in mats]
ptrs = [t.data_ptr() for t in dev]
cfg = wfstdec.Config(beam=13.0, max_active=1000000, min_active=0, lattice_beam=float(os.environ.get("LB", 7.0)))
for links in (0, int(os.environ.get("LINKS", 6 << 20))):
    dec = wfstdec.BatchDecoder(graph, cfg, B, max_frames=304, max_tokens_per_frame=131072, arena_tokens=300 * 20000, lattice_links=links)
    for it in range(3):
        dec.init(); torch.cuda.synchronize()
        t0 = time.time(); dec.advance(ptrs, [T] * B, 3000); dec.sync(); t1 = time.time()
        dec.finalize(); dec.sync(); t2 = time.time()
        bp = dec.best_paths(); t3 = time.time()
    print("links cap %d: advance %.1f ms  finalize %.1f ms  best paths %.1f ms  -> %.0f frames/s" % (links, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, B * T / (t3 - t0)))
    if links:
        st = [dec.stats(c) for c in range(B)]
        print("links per utt: mean %.0f max %d; tokens mean %.0f" % (np.mean([s["links"] for s in st]), max(s["links"] for s in st), np.mean([s["tokens"] for s in st])))
        t0 = time.time(); L = dec.raw_lattices(); dt_lat = (time.time() - t0) * 1e3
        ts = []
        for it in range(5):
            t0 = time.time(); nb = dec.nbest(10 if it < 4 else 1); ts.append((time.time() - t0) * 1e3)
        print("nbest(10) all %d channels, 4 calls + nbest(1): %s ms; paths %s" % (B, [round(x, 1) for x in ts], [len(x) for x in nb[:8]]))
        print("raw_lattice of all %d channels: %.1f ms (%.2f ms per utterance); states %s arcs %s" % (
            B, dt_lat, dt_lat / B, [l["n_states"] for l in L[:8]], [len(l["a_src"]) for l in L[:8]]))
    dec.free()
