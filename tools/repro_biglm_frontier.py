"""Hand-run (GPU box): the case of tools/repro_biglm_flake.py decoded N times frame by frame; after every frame the frontier
(state, cost) of the run is compared with the first run's -- prints the first frame at which a run's frontier parts from it."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import pyoracle
import gpu_util as G
from test_gpu_fuzz import random_graph
pkg = importlib.import_module("asr-decoder_amd")
synth = pkg.synth
lmsynth = importlib.import_module("asr-decoder_amd.lmsynth")
seed, block, want_case = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 50
lattice = os.environ.get("REPRO_LATTICE") == "1"
rng = np.random.default_rng(seed + block)
tmp = "/tmp/repro_biglm_%d" % os.getpid()
os.makedirs(tmp, exist_ok=True)
for case in range(10):
    n_states = int(rng.integers(4, 60)); n_labels = int(rng.integers(3, 12))
    g = random_graph(synth, rng, n_states, n_labels)
    gp = os.path.join(tmp, "g.bin"); g.write(gp)
    old = lmsynth.make_lm(30, int(rng.integers(1, 3)), int(rng.integers(3, 20)), 3, 0, 0, seed=int(rng.integers(1, 1 << 30)))
    new = lmsynth.make_lm(30, int(rng.integers(1, 4)), int(rng.integers(3, 25)), 3, int(rng.integers(2, 30)), 2, seed=int(rng.integers(1, 1 << 30)))
    p1, p2 = os.path.join(tmp, "old.bin"), os.path.join(tmp, "new.bin")
    old.to_fsa().write(p1); new.to_fsa().write(p2)
    binding = case % 3 == 2
    cd = dict(beam=float(rng.uniform(4.0, 14.0)), max_active=int(rng.choice([40, 15])) if binding else 1000000,
              min_active=int(rng.choice([0, 6])) if binding else 0, lattice_beam=float(rng.uniform(6.0, 30.0)), prune_interval=int(rng.integers(3, 30)))
    lens = [int(rng.integers(1, 40)) for _ in range(int(rng.integers(1, 5)))]
    mats = [rng.normal(-1.5, 1.0, size=(T, n_labels + 1)).astype(np.float32) for T in lens]
    chunk = int(rng.choice([0, 7]))
    if case != want_case:
        continue
    graph = G.wfstdec.Graph.load(gp)
    L1, L2 = G.wfstdec.Lm.load(p1, -1.0), G.wfstdec.Lm.load(p2, 1.0)
    x = mats[0]; T = x.shape[0]
    dev = G.upload([x])
    lim = dict(max_frames=64, max_tokens_per_frame=8192, arena_tokens=1 << 17)
    if lattice:
        lim["lattice_links"] = 1 << 19
    step = int(os.environ.get("REPRO_STEP", "1"))
    first = None
    hist = {}
    for rep in range(N):
        dec = G.wfstdec.BatchDecoder(graph, G.gpu_config(cd), 1, old_lm=L1, new_lm=L2, **lim)
        dec.init()
        fr = []
        for f in list(range(step, T, step)) + [T]:
            dec.advance([dev[0].data_ptr()], [f], x.shape[1])
            st, co = dec.frontier(0)
            o = np.lexsort((co.view(np.int32), st))
            fr.append((f, st[o].copy(), co[o].copy()))
        dec.free()
        if first is None:
            first = fr
            continue
        for (f, s0, c0), (_, s1, c1) in zip(first, fr):
            if len(s0) != len(s1) or not np.array_equal(s0, s1) or not np.array_equal(c0.view(np.int32), c1.view(np.int32)):
                a = set(zip(s0.tolist(), c0.tolist())); b = set(zip(s1.tolist(), c1.tolist()))
                key = (f, tuple(sorted(a - b))[:6], tuple(sorted(b - a))[:6])
                hist[key] = hist.get(key, 0) + 1
                break
    print("T", T, "cfg", cd, "frames of the first run:", [len(s) for _, s, _ in first])
    for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
        print("%d runs part from the first at frame %d: only in first %s | only in run %s" % (v, k[0], k[1], k[2]))
    print("%d of %d runs parted" % (sum(hist.values()), N - 1))
