"""Experiment: several independent 128-utterance batches in flight on one GPU (one BatchDecoder and
one HIP stream each).  The headline bench keeps ONE batch in flight; this measures what a service
that overlaps batches gets."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("asr-decoder_amd")
synth, wfstdec = pkg.synth, pkg.wfstdec
import torch
B, T = 128, 300
g = synth.make_hclg_like(2850000, seed=7)
m = synth.default_tid2pdf(6000)
graph = wfstdec.Graph.from_arrays(g.start, g.final_state, g.state_info, g.arcs)
graph.set_tid2pdf(m)
cfg = wfstdec.Config(beam=13.0, max_active=1000000, min_active=0, lattice_beam=7.0)
NMAX = int(os.environ.get("NMAX", 4))
sets = []
for k in range(NMAX):
    mats = [synth.make_loglikes_multi(g, T, 3000, m, seed=k * B + u, n_paths=272, mu=-4.0, jitter=0.5, ac_lo=0.5)[0] for u in range(B)]
    dev = [torch.from_numpy(x).to("cuda:0") for x in mats]
    sets.append((dev, [t.data_ptr() for t in dev]))
for n in range(1, NMAX + 1):
    streams = [torch.cuda.Stream() for _ in range(n)]
    decs = [wfstdec.BatchDecoder(graph, cfg, B, max_frames=304, max_tokens_per_frame=131072, arena_tokens=300 * 20000,
                                 stream=streams[k].cuda_stream) for k in range(n)]
    best = None
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.time()
        for k, d in enumerate(decs):
            d.init(); d.advance(sets[k][1], [T] * B, 3000); d.finalize()
        res = [d.best_paths() for d in decs]
        dt = time.time() - t0
        best = dt if best is None else min(best, dt)
    print("%d batches in flight: %.1f ms per round of %d x 128 utterances -> %.0f frames/s" % (n, best * 1e3, n, n * B * T / best))
    for d in decs:
        d.free()
