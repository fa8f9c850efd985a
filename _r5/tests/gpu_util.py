"""Helpers for the -m gpu parity tests: drive the HIP path through the C ABI
(asr-decoder_amd/wfstdec.py -> libwfstdec.so) the way the reference CLI drives its decoder
(kaldi-nnet3bin/kaldi-hclg-my-decoder.cc:97-129)."""
import importlib

import numpy as np

import pyoracle
from golden_util import bits

pkg = importlib.import_module("asr-decoder_amd")
wfstdec = pkg.wfstdec


def gpu_config(cd):
    return wfstdec.Config(**cd)


def upload(mats):
    """float32 host matrices -> torch tensors in HBM (plumbing only)."""
    import torch

    return [torch.from_numpy(np.ascontiguousarray(m, dtype=np.float32)).to("cuda:0") for m in mats]


class GpuResult:
    def __init__(self, d):
        self.ok = d["ok"]
        self.words, self.tids = d["words"], d["tids"]
        self.path_ilabel, self.path_olabel = d["ilabel"], d["olabel"]
        self.path_graph, self.path_ac = d["graph"], d["ac"]
        self.tot_score, self.lm_score = d["tot_score"], d["lm_score"]
        self.frame_ntoks = self.frame_best = None
        self.num_toks_end = self.num_links_end = 0


def decode_batch(graph, cd, mats, chunk=0, finalize=True, use_final_probs=True, trace=False,
                 host_feed=False, limits=None, dec=None):
    """Decode len(mats) utterances (ragged lengths allowed) as one batch; returns GpuResults.
    chunk > 0: NumFramesReady grows by `chunk` per AdvanceDecoding call (streaming shape)."""
    B = len(mats)
    own = dec is None
    if own:
        dec = wfstdec.BatchDecoder(graph, gpu_config(cd), B, **(limits or dict(max_frames=512, max_tokens_per_frame=32768, arena_tokens=1 << 22)))
    T = [int(m.shape[0]) for m in mats]
    stride = int(mats[0].shape[1])
    dev = None if host_feed else upload(mats)
    ptrs = None if host_feed else [t.data_ptr() for t in dev]
    dec.init()
    fn = fb = None
    if trace:
        chunk = 1
        fn = [np.zeros(t + 1, np.int32) for t in T]
        fb = [np.zeros(t + 1, np.float32) for t in T]

        def snap(r):
            for c in range(B):
                k = min(r, T[c])
                if k == r or r == 0:
                    st, co = dec.frontier(c)
                    fn[c][k] = len(st)
                    fb[c][k] = co.min() if len(co) else np.inf
        snap(0)
    Tmax = max(T) if T else 0
    if chunk <= 0:
        steps = [Tmax]
    else:
        steps = list(range(chunk, Tmax, chunk)) + [Tmax]
    for r in steps:
        ready = [min(r, t) for t in T]
        if host_feed:
            dec.advance_host(mats, ready)
        else:
            dec.advance(ptrs, ready, stride)
        if trace:
            snap(r)
    if finalize:
        dec.finalize()
    res = [GpuResult(d) for d in dec.best_paths(use_final_probs=use_final_probs)]
    if trace:
        for c in range(B):
            res[c].frame_ntoks, res[c].frame_best = fn[c], fb[c]
    stats = [dec.stats(c) for c in range(B)]
    for r, s in zip(res, stats):
        r.stats = s
    if own:
        dec.free()
    return res


def assert_same_path(r, e_words, e_tids, e_il, e_ol, e_g, e_ac, e_scores, what=""):
    assert np.array_equal(r.words, e_words), what + " words"
    assert np.array_equal(r.tids, e_tids), what + " transition-ids"
    assert np.array_equal(r.path_ilabel, e_il), what + " path ilabels"
    assert np.array_equal(r.path_olabel, e_ol), what + " path olabels"
    assert np.array_equal(bits(r.path_graph), bits(e_g)), what + " graph costs"
    assert np.array_equal(bits(r.path_ac), bits(e_ac)), what + " acoustic costs"
    assert np.array_equal(bits([r.tot_score, r.lm_score]), bits(e_scores)), what + " scores"


def assert_same_as_oracle(r, o, what=""):
    assert bool(r.ok) == bool(o.ok), what + " ok"
    assert_same_path(r, o.words, o.tids, o.path_ilabel, o.path_olabel, o.path_graph, o.path_ac,
                     [o.tot_score, o.lm_score], what)
