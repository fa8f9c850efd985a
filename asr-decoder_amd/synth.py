"""Seeded synthetic inputs for the WFST token-passing path.

The reference ships no model, graph or feature files, so every
test and bench input is generated here, in the reference's own formats:

* graph: the flat binary format read by ``Fst::ReadFst`` (reference
  ``src/newfst/optimize-fst.h:226-280``): six int32 ``{start, final_state, total_states,
  total_arcs, total_niepsilons, total_noepsilons}``, then ``StateInfo{u32 num_arcs,
  niepsilons, noepsilons}`` x S, then ``StdArc{i32 ilabel, i32 olabel, f32 w, i32 to}`` x A
  (``src/newfst/arc.h:17-26``).  One super-final state; a final state's arc 0 is the
  epsilon arc into it carrying the final weight
  (``src/fst_format_convert_tool/read_fst.c:110-135``); input-epsilon arcs precede
  emitting arcs and emitting arcs are ilabel-sorted.
* log-likelihoods: row-major ``float32[T][P]`` with the acoustic scale pre-applied, read
  through ``tid2pdf`` exactly like Kaldi's ``DecodableMatrixScaledMapped`` in the reference
  CLI (``src/kaldi-nnet3bin/kaldi-hclg-my-decoder.cc:107``).

The "hclg-like" recipe is SURVEY.md section 8(d).  numpy only; no GPU, no torch.
"""
from __future__ import annotations

import os

import numpy as np

ARC_DTYPE = np.dtype([("ilabel", "<i4"), ("olabel", "<i4"), ("w", "<f4"), ("to", "<i4")])
STATE_DTYPE = np.dtype([("num_arcs", "<u4"), ("niepsilons", "<u4"), ("noepsilons", "<u4")])


class Graph:
    """Flat graph in host memory (the arrays of the reference on-disk format)."""

    def __init__(self, start, final_state, state_info, arcs):
        self.start = int(start)
        self.final_state = int(final_state)
        self.state_info = np.ascontiguousarray(state_info, dtype=STATE_DTYPE)
        self.arcs = np.ascontiguousarray(arcs, dtype=ARC_DTYPE)

    @property
    def n_states(self):
        return int(self.state_info.shape[0])

    @property
    def n_arcs(self):
        return int(self.arcs.shape[0])

    def row_offsets(self):
        off = np.zeros(self.n_states + 1, dtype=np.int64)
        np.cumsum(self.state_info["num_arcs"].astype(np.int64), out=off[1:])
        return off

    def write(self, path):
        hdr = np.array(
            [
                self.start,
                self.final_state,
                self.n_states,
                self.n_arcs,
                int(self.state_info["niepsilons"].astype(np.int64).sum()),
                int(self.state_info["noepsilons"].astype(np.int64).sum()),
            ],
            dtype="<i4",
        )
        with open(path, "wb") as f:
            f.write(hdr.tobytes())
            f.write(self.state_info.tobytes())
            f.write(self.arcs.tobytes())

    @staticmethod
    def read(path):
        with open(path, "rb") as f:
            hdr = np.frombuffer(f.read(24), dtype="<i4")
            start, final_state, n_states, n_arcs = (int(x) for x in hdr[:4])
            si = np.frombuffer(f.read(12 * n_states), dtype=STATE_DTYPE)
            arcs = np.frombuffer(f.read(16 * n_arcs), dtype=ARC_DTYPE)
        if si.shape[0] != n_states or arcs.shape[0] != n_arcs:
            raise IOError("truncated graph file: %s" % path)
        return Graph(start, final_state, si, arcs)


# (test campaigns only: WFST_SYNTH_SEED_OFFSET shifts every generator seed of this module -- other graphs and log-likelihoods through
# the same tests; the bench and the default test runs leave it unset)
_SEED_OFFSET = int(os.environ.get("WFST_SYNTH_SEED_OFFSET", "0") or 0)

def to_openfst_bytes(g, fst_type="vector", aligned=False, flags=0):
    """The flat graph `g` as an OpenFst binary file (StdArc), the input side of the ingestion
    tests: the super-final construction is undone (a leading <eps>:<eps> arc into g.final_state
    becomes the state's final weight, every other state gets weight +inf = Zero) and the result is
    written as a "vector" fst {float final, int64 narcs, arcs} or a "const" fst {ConstState x S,
    arcs} (optionally 16-byte aligned, OpenFst's --fst_align).  Layout: OpenFst fst/fst.h
    FstHeader, fst/vector-fst.h, fst/const-fst.h."""
    import struct

    off = g.row_offsets()
    S = g.n_states - 1
    assert g.final_state == S and g.state_info["num_arcs"][S] == 0
    arcs, si = g.arcs, g.state_info
    first = off[:S]
    has = si["num_arcs"][:S] > 0
    lead = np.zeros(S, bool)
    fi = first[has]
    lead[has] = (arcs["ilabel"][fi] == 0) & (arcs["olabel"][fi] == 0) & (arcs["to"][fi] == S)
    final_w = np.full(S, np.inf, np.float32)
    final_w[lead] = arcs["w"][first[lead]]
    keep = np.ones(g.n_arcs, bool)
    keep[first[lead]] = False
    body = arcs[keep]
    narcs = si["num_arcs"][:S].astype(np.int64) - lead
    nie = si["niepsilons"][:S].astype(np.int64) - lead
    noe = si["noepsilons"][:S].astype(np.int64) - lead

    def hstr(x):
        return struct.pack("<i", len(x)) + x.encode()

    flags = int(flags) | (4 if (aligned and fst_type == "const") else 0)
    version = 2 if not (aligned and fst_type == "const") else 1
    head = (struct.pack("<i", 2125659606) + hstr(fst_type) + hstr("standard") + struct.pack("<ii", version, flags) +
            struct.pack("<Q", 0) + struct.pack("<qqq", g.start, S, int(body.shape[0])))
    out = [head]
    if fst_type == "vector":
        pos = np.zeros(S + 1, np.int64)
        np.cumsum(narcs, out=pos[1:])
        raw = body.tobytes()
        for s_ in range(S):
            out.append(struct.pack("<fq", float(final_w[s_]), int(narcs[s_])))
            out.append(raw[pos[s_] * 16:pos[s_ + 1] * 16])
    elif fst_type == "const":
        pos = np.zeros(S + 1, np.int64)
        np.cumsum(narcs, out=pos[1:])
        cs = np.zeros(S, np.dtype([("w", "<f4"), ("pos", "<u4"), ("narcs", "<u4"), ("nie", "<u4"), ("noe", "<u4")]))
        cs["w"], cs["pos"], cs["narcs"], cs["nie"], cs["noe"] = final_w, pos[:S], narcs, nie, noe
        n = len(head)
        if aligned:
            out.append(b"\0" * ((16 - n % 16) % 16))
            n += (16 - n % 16) % 16
        out.append(cs.tobytes())
        n += S * 20
        if aligned:
            out.append(b"\0" * ((16 - n % 16) % 16))
        out.append(body.tobytes())
    else:
        raise ValueError(fst_type)
    return b"".join(out)


def graph_from_arc_lists(n_states, start, arcs_by_state, final_weights):
    """Build a flat graph from python lists (small hand-made cases).

    arcs_by_state[s] = [(ilabel, olabel, w, to), ...]; final_weights = {state: weight}.
    A super-final state (id n_states) is appended; final states get the epsilon arc into it
    at index 0.  Remaining arcs are stable-sorted so that ilabel==0 arcs come first.
    """
    final_state = n_states
    si = np.zeros(n_states + 1, dtype=STATE_DTYPE)
    out = []
    for s in range(n_states):
        row = []
        if s in final_weights:
            row.append((0, 0, float(final_weights[s]), final_state))
        rest = sorted(arcs_by_state.get(s, []), key=lambda a: a[0])
        row.extend(rest)
        si[s] = (len(row), sum(1 for a in row if a[0] == 0), sum(1 for a in row if a[1] == 0))
        out.extend(row)
    arcs = np.array(out, dtype=ARC_DTYPE) if out else np.zeros(0, dtype=ARC_DTYPE)
    return Graph(start, final_state, si, arcs)


def make_hclg_like(
    n_states,
    seed=7,
    n_tid=6000,
    n_words=50000,
    p_final=0.01,
    p_eps=0.08,
    p_fanout=0.02,
    allow_parallel=False,
):
    """SURVEY.md 8(d) "hclg-like" graph with ``n_states`` regular states (+1 super-final).

    ~3.5 arcs/state: S=14k -> A~50k (config 1), S=2.85M -> A~10.1M (configs 2/3).
    """
    rng = np.random.default_rng(seed + _SEED_OFFSET)
    S = int(n_states)
    final_state = S

    has_final = rng.random(S) < p_final
    has_eps = rng.random(S) < p_eps
    has_eps[S - 1] = False  # forward-only epsilon arcs: no epsilon cycles
    n_emit = rng.integers(1, 3, size=S)
    fan = rng.random(S) < p_fanout
    n_emit[fan] = rng.integers(20, 81, size=int(fan.sum()))

    # emitting arcs: one self-loop per state + n_emit arcs to uniform random targets
    src_loop = np.arange(S, dtype=np.int64)
    src_out = np.repeat(np.arange(S, dtype=np.int64), n_emit)
    E = src_out.shape[0]
    dst_out = rng.integers(0, S, size=E).astype(np.int64)
    e_src = np.concatenate([src_loop, src_out])
    e_dst = np.concatenate([src_loop, dst_out])
    e_tid = rng.integers(1, n_tid + 1, size=S + E).astype(np.int32)
    e_w = np.concatenate(
        [rng.uniform(0.1, 0.6, size=S), rng.uniform(0.0, 4.0, size=E)]
    ).astype(np.float32)
    e_ol = np.where(rng.random(S + E) < 0.1, rng.integers(1, n_words + 1, size=S + E), 0).astype(np.int32)
    e_ol[:S] = 0  # self-loops carry no word
    if not allow_parallel:
        # drop duplicate (src, dst) pairs, keeping the first (self-loops come first)
        key = e_src * S + e_dst
        _, first = np.unique(key, return_index=True)
        keep = np.zeros(S + E, dtype=bool)
        keep[first] = True
        e_src, e_dst, e_tid, e_w, e_ol = e_src[keep], e_dst[keep], e_tid[keep], e_w[keep], e_ol[keep]
    # ilabel-sorted within each state (stable)
    order = np.lexsort((e_tid, e_src))
    e_src, e_dst, e_tid, e_w, e_ol = e_src[order], e_dst[order], e_tid[order], e_w[order], e_ol[order]
    emit_deg = np.bincount(e_src, minlength=S).astype(np.int64)

    # epsilon arcs
    f_w = rng.uniform(0.0, 2.0, size=S).astype(np.float32)
    eps_span = rng.integers(1, 1001, size=S)
    eps_dst = np.minimum(np.arange(S) + eps_span, S - 1)
    eps_w = rng.uniform(0.0, 3.0, size=S).astype(np.float32)
    eps_ol = np.where(rng.random(S) < 0.5, rng.integers(1, n_words + 1, size=S), 0).astype(np.int32)

    n_eps = has_final.astype(np.int64) + has_eps.astype(np.int64)
    deg = n_eps + emit_deg
    off = np.zeros(S + 1, dtype=np.int64)
    np.cumsum(deg, out=off[1:])
    A = int(off[-1])
    arcs = np.zeros(A, dtype=ARC_DTYPE)

    fs = np.nonzero(has_final)[0]
    pos = off[fs]
    arcs["ilabel"][pos] = 0
    arcs["olabel"][pos] = 0
    arcs["w"][pos] = f_w[fs]
    arcs["to"][pos] = final_state

    es = np.nonzero(has_eps)[0]
    pos = off[es] + has_final[es]
    arcs["ilabel"][pos] = 0
    arcs["olabel"][pos] = eps_ol[es]
    arcs["w"][pos] = eps_w[es]
    arcs["to"][pos] = eps_dst[es]

    emit_off = np.zeros(S + 1, dtype=np.int64)
    np.cumsum(emit_deg, out=emit_off[1:])
    rank = np.arange(e_src.shape[0], dtype=np.int64) - emit_off[e_src]
    pos = off[e_src] + n_eps[e_src] + rank
    arcs["ilabel"][pos] = e_tid
    arcs["olabel"][pos] = e_ol
    arcs["w"][pos] = e_w
    arcs["to"][pos] = e_dst

    si = np.zeros(S + 1, dtype=STATE_DTYPE)
    si["num_arcs"][:S] = deg
    si["niepsilons"][:S] = n_eps
    oeps = np.bincount(
        np.repeat(np.arange(S, dtype=np.int64), deg)[arcs["olabel"] == 0], minlength=S
    )
    si["noepsilons"][:S] = oeps
    return Graph(0, final_state, si, arcs)


def default_tid2pdf(n_tid=6000):
    """tid2pdf[t] = (t-1)//2 for t in 1..n_tid; entry 0 unused (epsilon)."""
    t = np.arange(n_tid + 1, dtype=np.int32)
    m = (t - 1) // 2
    m[0] = 0
    return m.astype(np.int32)


def make_loglikes(graph, T, n_pdf, tid2pdf, seed, mu=-2.6, sigma=1.0, p_eps_step=0.3):
    """``float32[T][n_pdf]`` ~ N(mu, sigma) with a planted path.

    The planted path is a random walk over the graph from the start state; at every frame
    the pdf of the emitting arc it takes is set to U(-1, 0).  Returns (loglikes, planted_tids).
    """
    rng = np.random.default_rng(seed + _SEED_OFFSET)
    ll = rng.normal(mu, sigma, size=(T, n_pdf)).astype(np.float32)
    off = graph.row_offsets()
    si = graph.state_info
    arcs = graph.arcs
    s = graph.start
    planted = np.zeros(T, dtype=np.int32)
    for t in range(T):
        # optionally hop over one (non-final) epsilon arc first
        for _ in range(4):
            b, ne, na = int(off[s]), int(si["niepsilons"][s]), int(si["num_arcs"][s])
            if ne and rng.random() < p_eps_step:
                a = arcs[b + int(rng.integers(0, ne))]
                if int(a["to"]) != graph.final_state:
                    s = int(a["to"])
                    continue
            break
        b, ne, na = int(off[s]), int(si["niepsilons"][s]), int(si["num_arcs"][s])
        if na - ne <= 0:
            break
        a = arcs[b + ne + int(rng.integers(0, na - ne))]
        tid = int(a["ilabel"])
        planted[t] = tid
        ll[t, int(tid2pdf[tid])] = np.float32(rng.uniform(-1.0, 0.0))
        s = int(a["to"])
    return ll, planted


def make_loglikes_multi(graph, T, n_pdf, tid2pdf, seed, n_paths=256, mu=-4.5, sigma=1.0,
                        drift=2.5, jitter=1.5, p_eps_step=0.2, ac_lo=-2.0, ac_hi=6.0, _cache={}):
    """Stable many-hypothesis workload: ``n_paths`` planted paths whose cumulative costs are
    steered to ``drift * t + U(-jitter, jitter)`` so that all of them stay inside the beam for
    the whole utterance (what real decoding looks like: many live hypotheses of similar score),
    over N(mu, sigma) background noise low enough that off-path tokens die within a few frames.

    A single planted path over Gaussian noise (``make_loglikes``) reaches thousands of active
    tokens only near the critical point of the branching search, where the per-frame token count
    is wildly heavy-tailed (median hundreds, bursts of 10^5 on the 10M-arc graph); this recipe
    gets the same mean from short-lived satellites around many live hypotheses instead, so the
    frontier size is proportional to ``n_paths`` and stable from frame to frame.
    Vectorised over paths (one numpy step per frame).  Returns (loglikes, None).
    """
    rng = np.random.default_rng(seed + _SEED_OFFSET)
    ll = rng.normal(mu, sigma, size=(T, n_pdf)).astype(np.float32)
    key = id(graph)
    if key not in _cache:
        _cache.clear()
        si = graph.state_info
        _cache[key] = (graph.row_offsets(), si["niepsilons"].astype(np.int64), si["num_arcs"].astype(np.int64))
    off, neps, narcs = _cache[key]
    arcs = graph.arcs
    t2p = np.asarray(tid2pdf, dtype=np.int64)
    state = np.full(n_paths, graph.start, dtype=np.int64)
    cum = np.zeros(n_paths, dtype=np.float64)
    for t in range(T):
        # optional hop over one (non-final) epsilon arc
        ne = neps[state]
        hop = (ne > 0) & (rng.random(n_paths) < p_eps_step)
        if hop.any():
            idx = np.nonzero(hop)[0]
            a = arcs[off[state[idx]] + (rng.random(idx.shape[0]) * ne[idx]).astype(np.int64)]
            ok = a["to"] != graph.final_state
            cum[idx[ok]] += a["w"][ok]
            state[idx[ok]] = a["to"][ok]
        ne = neps[state]
        nem = narcs[state] - ne
        live = nem > 0
        idx = np.nonzero(live)[0]
        a = arcs[off[state[idx]] + ne[idx] + (rng.random(idx.shape[0]) * nem[idx]).astype(np.int64)]
        target = drift * (t + 1) + rng.uniform(-jitter, jitter, size=idx.shape[0])
        want_ac = np.clip(target - cum[idx] - a["w"], ac_lo, ac_hi)
        ll[t, t2p[a["ilabel"]]] = (-want_ac).astype(np.float32)
        cum[idx] += a["w"] + want_ac
        state[idx] = a["to"]
    return ll, None
