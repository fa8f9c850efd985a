"""Seeded synthetic back-off n-gram language models for the biglm path (BASELINE configs[3]).

The reference's biglm decoder (``src/my-decoder/online-decoder-mempool-base-biglm.h``) rescores the
HCLG's word labels on the fly with ``new LM - old LM``; both LMs are ARPA models converted to the
reference's FSA (``src/newlm/arpa2fsa.{h,cc}``) and stored in its binary LM format
(``ArpaLm::Write``, ``newlm/arpa2fsa.h:399-439`` + ``Fsa::Write``, ``newlm/arpa2fsa.cc:8-60``):

    i32 bos, i32 eos, i32 unk, u64 n_orders, i32 ngram_count[n_orders],
    i32 n_states, {i32 arc_num, f32 backoff_prob, i32 backoff_id} x n_states,
    i32 n_arcs,   {i32 wordid, f32 weight, i32 tostateid} x n_arcs      (arcs state by state, wordid-sorted)

State 0 is the empty-history state; its arcs are indexed directly by word id (arc k = word k -> state
k + 1, ``Fsa::GetArc``, ``arpa2fsa.cc:244-262``); every n-gram line of higher order adds one state in
file order; weights are natural-log probabilities (``logprob * M_LN10``), costs are their negation.

``NgramLm.to_fsa()`` builds that FSA with the reference converter's state numbering
(``Arpa2Fsa::AddLineToFsa``, ``arpa2fsa.cc:519-633``) straight from the n-gram tables, so the GPU box
needs neither the ARPA text nor the reference converter; ``tests/test_lm_format.py`` holds it byte
for byte against the file the reference's own ``Arpa2Fsa`` writes from ``NgramLm.arpa_text()``.
numpy only.
"""
from __future__ import annotations

import struct

import numpy as np

M_LN10 = 2.302585092994045684017991454684
STATE_DT = np.dtype([("arc_num", "<i4"), ("backoff_prob", "<f4"), ("backoff_id", "<i4")])
ARC_DT = np.dtype([("wordid", "<i4"), ("weight", "<f4"), ("tostateid", "<i4")])


def _ln(x):
    """AnalyLine (arpa2fsa.cc:469,506): float *= double M_LN10, rounded back to float"""
    return np.float32(np.float64(np.float32(x)) * M_LN10)


class Fsa:
    """The reference's LM automaton as flat arrays (+ the ArpaLm header fields)."""

    def __init__(self, bos, eos, unk, num_gram, states, arcs):
        self.bos, self.eos, self.unk = int(bos), int(eos), int(unk)
        self.num_gram = [int(x) for x in num_gram]
        self.states = np.ascontiguousarray(states, dtype=STATE_DT)
        self.arcs = np.ascontiguousarray(arcs, dtype=ARC_DT)

    @property
    def n_states(self):
        return int(self.states.shape[0])

    @property
    def n_arcs(self):
        return int(self.arcs.shape[0])

    def arc_offsets(self):
        off = np.zeros(self.n_states + 1, np.int64)
        np.cumsum(self.states["arc_num"].astype(np.int64), out=off[1:])
        return off

    def to_bytes(self):
        out = [struct.pack("<iiiQ", self.bos, self.eos, self.unk, len(self.num_gram)),
               np.asarray(self.num_gram, "<i4").tobytes(), struct.pack("<i", self.n_states), self.states.tobytes(),
               struct.pack("<i", self.n_arcs), self.arcs.tobytes()]
        return b"".join(out)

    def write(self, path):
        with open(path, "wb") as f:
            f.write(self.to_bytes())

    @staticmethod
    def from_bytes(b):
        bos, eos, unk, n = struct.unpack_from("<iiiQ", b, 0)
        o = 20
        num_gram = np.frombuffer(b, "<i4", n, o)
        o += 4 * n
        (ns,) = struct.unpack_from("<i", b, o)
        o += 4
        states = np.frombuffer(b, STATE_DT, ns, o)
        o += 12 * ns
        (na,) = struct.unpack_from("<i", b, o)
        o += 4
        arcs = np.frombuffer(b, ARC_DT, na, o)
        return Fsa(bos, eos, unk, num_gram, states, arcs)

    @staticmethod
    def read(path):
        with open(path, "rb") as f:
            return Fsa.from_bytes(f.read())

    def rescaled(self, scale):
        """ArpaLm::Rescale (arpa2fsa.cc:264-275): arc weights and back-off weights times `scale`
        (float *= float); the biglm CLI rescales the OLD LM by -1 (kaldi-hclg-my-decoder-biglm.cc:59)."""
        st, ar = self.states.copy(), self.arcs.copy()
        if scale != 1:
            ar["weight"] = (ar["weight"] * np.float32(scale)).astype(np.float32)
            st["backoff_prob"] = (st["backoff_prob"] * np.float32(scale)).astype(np.float32)
        return Fsa(self.bos, self.eos, self.unk, self.num_gram, st, ar)


class NgramLm:
    """A back-off n-gram model as tables: grams[k] = list of (words tuple of length k+1,
    log10 prob, log10 back-off weight) in ARPA file order (lines of one context contiguous)."""

    def __init__(self, n_words, grams):
        self.V = int(n_words)          # real words have ids 1..V (the graph's olabels)
        self.bos, self.eos, self.unk = self.V + 1, self.V + 2, self.V + 3
        self.grams = grams

    def word_str(self, w):
        return {self.bos: "<s>", self.eos: "</s>", self.unk: "<unk>", 0: "<eps>"}.get(w, "w%d" % w)

    def wordlist_text(self):
        return "".join("%s %d\n" % (self.word_str(w), w) for w in range(0, self.V + 4))

    def arpa_text(self):
        out = ["\\data\\\n"]
        for k, g in enumerate(self.grams):
            out.append("ngram %d=%d\n" % (k + 1, len(g)))
        for k, g in enumerate(self.grams):
            out.append("\n\\%d-grams:\n" % (k + 1))
            last = k + 1 == len(self.grams)
            for words, lp, bo in g:
                s = "%.9g\t%s" % (lp, " ".join(self.word_str(w) for w in words))
                if not last:
                    s += "\t%.9g" % bo
                out.append(s + "\n")
        out.append("\n\\end\\\n")
        return "".join(out)

    def n_ngrams(self):
        return sum(len(g) for g in self.grams)

    def to_fsa(self):
        """Arpa2Fsa::ConvertArpa2Fsa with one thread (arpa2fsa.cc:519-739), from the tables."""
        order = len(self.grams)
        bo_prob, bo_id = [np.float32(0)], [0]           # per state
        arcs = [dict()]                                  # per state: wordid -> (weight, tostate); state 0 filled below
        start_to = []                                    # start state: arc k -> state k + 1 (created on demand, in id order)

        def new_state():
            bo_prob.append(np.float32(0))
            bo_id.append(0)
            arcs.append(dict())
            return len(arcs) - 1

        start_w = []
        for words, lp, bo in self.grams[0]:
            w = words[0]
            while len(start_to) - 1 < w:                 # arpa2fsa.cc:531-537
                start_to.append(new_state())
                start_w.append(np.float32(0))
            start_w[w] = _ln(lp)
            s = start_to[w]
            bo_id[s] = 0
            bo_prob[s] = _ln(bo) if order > 1 else np.float32(0)

        def walk(ws):
            """state reached from the start state over the words ws (None if an arc is missing)"""
            s = 0
            for i, w in enumerate(ws):
                if i == 0:
                    if w >= len(start_to):
                        return None
                    s = start_to[w]
                else:
                    a = arcs[s].get(w)
                    if a is None:
                        return None
                    s = a[1]
            return s

        for k in range(1, order):
            last = k + 1 == order
            for words, lp, bo in self.grams[k]:
                ctx = walk(words[:-1])
                if ctx is None:                           # "no A B, but have A B C": not added (:566-573)
                    continue
                to = new_state()
                arcs[ctx][words[-1]] = (_ln(lp), to)
                # back-off target: the longest proper suffix that is a path from the start state (:590-624)
                tgt = 0
                for bs in range(1, len(words)):
                    t = walk(words[bs:])
                    if t is not None:
                        tgt = t
                        break
                bo_id[to] = tgt
                bo_prob[to] = np.float32(0) if last else _ln(bo)
        n = len(arcs)
        st = np.zeros(n, STATE_DT)
        st["backoff_prob"] = np.asarray(bo_prob, np.float32)
        st["backoff_id"] = np.asarray(bo_id, np.int32)
        rows = []
        st["arc_num"][0] = len(start_to)
        rows.append(np.array([(w, start_w[w], start_to[w]) for w in range(len(start_to))], ARC_DT))
        for s in range(1, n):
            a = arcs[s]
            st["arc_num"][s] = len(a)
            if a:
                ks = sorted(a)
                rows.append(np.array([(w, a[w][0], a[w][1]) for w in ks], ARC_DT))
        allarcs = np.concatenate(rows) if rows else np.zeros(0, ARC_DT)
        return Fsa(self.bos, self.eos, self.unk, [len(g) for g in self.grams], st, allarcs)


def make_lm(n_words, order=3, n_bigram_ctx=2000, succ=6, n_trigram_ctx=1500, succ3=4, seed=0, sharp=1.0,
            unigram_only_words=None):
    """Random back-off LM over words 1..n_words (+ <s>, </s>).  Every word has a unigram (the
    reference's start state is indexed by word id without a bounds check); `n_bigram_ctx` contexts
    get `succ` bigram successors each, `n_trigram_ctx` existing bigrams get `succ3` trigram
    successors.  `sharp` scales how strongly the higher orders prefer their successors."""
    rng = np.random.default_rng(seed)
    V = int(n_words)
    bos, eos = V + 1, V + 2
    uni = []
    lp1 = rng.uniform(-5.0, -1.0, V + 3).astype(np.float32)
    bo1 = rng.uniform(-1.0, 0.0, V + 3).astype(np.float32)
    for w in range(1, V + 3):
        if w == bos:
            uni.append(((w,), np.float32(-99.0), bo1[w]))
        else:
            uni.append(((w,), lp1[w], np.float32(0) if w == eos else bo1[w]))
    grams = [uni]
    if order >= 2:
        ctxs = rng.choice(np.arange(1, V + 1), size=min(n_bigram_ctx, V), replace=False).tolist()
        ctxs.append(bos)
        bi = []
        for a in ctxs:
            nxt = rng.choice(np.arange(1, V + 1), size=min(succ, V), replace=False).tolist()
            if rng.random() < 0.3 and a != bos:
                nxt.append(eos)
            for b in sorted(set(nxt)):
                bi.append(((a, b), np.float32(rng.uniform(-2.5, -0.2) * sharp), np.float32(0) if b == eos else np.float32(rng.uniform(-0.8, 0.0))))
        grams.append(bi)
        if order >= 3:
            cand = [g[0] for g in bi if g[0][1] != eos]
            pick = rng.choice(len(cand), size=min(n_trigram_ctx, len(cand)), replace=False)
            tri = []
            for i in sorted(pick.tolist()):
                a, b = cand[i]
                nxt = rng.choice(np.arange(1, V + 1), size=min(succ3, V), replace=False).tolist()
                if rng.random() < 0.2:
                    nxt.append(eos)
                for c in sorted(set(nxt)):
                    tri.append(((a, b, c), np.float32(rng.uniform(-2.0, -0.1) * sharp), np.float32(0)))
            grams.append(tri)
    return NgramLm(V, grams)
