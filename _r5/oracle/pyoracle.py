"""TEST INFRASTRUCTURE -- ctypes loaders for the two checkers.  Never imported by the product.

* ``RefDecoder``    -> oracle/_ref/libref_decoder.so   the unmodified reference decoder
                       (built by ``make -C oracle ref`` where /root/reference exists)
* ``OracleDecoder`` -> oracle/_build/libwfst_oracle.so  our plain-C restatement
                       (oracle/wfst_oracle.c, built by ``make -C oracle oracle``)

Both expose the same ``decode(...)`` returning a ``Result``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "_ref", "libref_decoder.so")
ORACLE_SO = os.path.join(HERE, "_build", "libwfst_oracle.so")


class Config(C.Structure):
    """Field-for-field ``LatticeFasterDecoderConfig`` (reference
    src/my-decoder/lattice-faster-decoder-conf.h:21-44), same defaults."""

    _fields_ = [
        ("beam", C.c_float),
        ("max_active", C.c_int),
        ("min_active", C.c_int),
        ("lattice_beam", C.c_float),
        ("prune_interval", C.c_int),
        ("beam_delta", C.c_float),
        ("hash_ratio", C.c_float),
        ("prune_scale", C.c_float),
    ]

    def __init__(self, beam=16.0, max_active=2147483647, min_active=200, lattice_beam=10.0,
                 prune_interval=25, beam_delta=0.5, hash_ratio=2.0, prune_scale=0.1):
        super().__init__(beam, max_active, min_active, lattice_beam, prune_interval, beam_delta,
                         hash_ratio, prune_scale)

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


@dataclass
class Result:
    ok: bool
    words: np.ndarray
    tids: np.ndarray
    tot_score: float
    lm_score: float
    path_ilabel: np.ndarray
    path_olabel: np.ndarray
    path_graph: np.ndarray
    path_ac: np.ndarray
    frame_ntoks: np.ndarray | None = None
    frame_best: np.ndarray | None = None
    dump: tuple | None = None
    num_toks_end: int = 0
    num_links_end: int = 0
    extra: dict = field(default_factory=dict)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int)) if a is not None else None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


def build_ref():
    subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


class _Base:
    PREFIX = ""
    SO = ""

    def __init__(self):
        if not os.path.exists(self.SO):
            raise FileNotFoundError(self.SO)
        self.lib = C.CDLL(self.SO)
        p = self.PREFIX
        self._load = getattr(self.lib, p + "_graph_load")
        self._load.restype = C.c_void_p
        self._load.argtypes = [C.c_char_p]
        self._free = getattr(self.lib, p + "_graph_free")
        self._free.argtypes = [C.c_void_p]
        self._decode = getattr(self.lib, p + "_decode")
        self._decode.restype = C.c_int
        self._graphs = {}

    def load_graph(self, path):
        h = self._load(path.encode())
        if not h:
            raise IOError("cannot read graph %s" % path)
        return h

    def free_graph(self, h):
        self._free(C.c_void_p(h))

    def decode(self, graph_handle, cfg, loglikes, tid2pdf=None, chunk=0, finalize=True,
               use_final_probs=True, trace=False, dump_frame=-1, dump_cap=0):
        ll = np.ascontiguousarray(loglikes, dtype=np.float32)
        T, stride = ll.shape
        if tid2pdf is not None:
            tid2pdf = np.ascontiguousarray(tid2pdf, dtype=np.int32)
            n_tid = int(tid2pdf.shape[0] - 1)
        else:
            n_tid = stride - 1
        max_path = 16 * T + 256   # hops: T emitting + the epsilon hops between them (dense-epsilon graphs: several per frame)
        pi = np.zeros(max_path, np.int32)
        po = np.zeros(max_path, np.int32)
        pg = np.zeros(max_path, np.float32)
        pa = np.zeros(max_path, np.float32)
        words = np.zeros(max_path, np.int32)
        tids = np.zeros(max_path, np.int32)
        n_path, n_words, n_tids = C.c_int(0), C.c_int(0), C.c_int(0)
        tot, lm = C.c_float(0), C.c_float(0)
        fn = fb = None
        if trace:
            chunk = 1
            fn = np.zeros(T + 1, np.int32)
            fb = np.zeros(T + 1, np.float32)
        ds = dc = None
        dn = C.c_int(0)
        if dump_frame >= 0:
            chunk = 1
            ds = np.zeros(max(dump_cap, 1), np.int32)
            dc = np.zeros(max(dump_cap, 1), np.float32)
        nt, nl = C.c_int(0), C.c_int(0)
        ok = self._decode(
            C.c_void_p(graph_handle), C.byref(cfg), _fp(ll), T, stride, _ip(tid2pdf), n_tid,
            int(chunk), int(bool(finalize)), int(bool(use_final_probs)),
            _ip(pi), _ip(po), _fp(pg), _fp(pa), max_path, C.byref(n_path),
            C.byref(tot), C.byref(lm), _ip(words), max_path, C.byref(n_words),
            _ip(tids), max_path, C.byref(n_tids),
            _ip(fn), _fp(fb), int(dump_frame), _ip(ds), _fp(dc), int(dump_cap), C.byref(dn),
            C.byref(nt), C.byref(nl))
        n = n_path.value
        if n > max_path:
            raise RuntimeError("best path of %d hops does not fit the binding's %d-hop buffers" % (n, max_path))
        dump = None
        if dump_frame >= 0:
            k = min(dn.value, dump_cap)
            dump = (ds[:k].copy(), dc[:k].copy(), dn.value)
        return Result(bool(ok), words[: n_words.value].copy(), tids[: n_tids.value].copy(),
                      float(tot.value), float(lm.value), pi[:n].copy(), po[:n].copy(),
                      pg[:n].copy(), pa[:n].copy(), fn, fb, dump, nt.value, nl.value)


class RefDecoder(_Base):
    PREFIX = "ref"
    SO = REF_SO


class OracleDecoder(_Base):
    """Adds the audit counters of oracle_decode_ex to Result.extra: N/E/Z work counts,
    ``ties`` = best-path tokens that saw an exact-cost rival (reference tie-break is arrival
    order), ``quirk_hops`` = hops where the reported arc is not the arg-min one."""

    PREFIX = "oracle"
    SO = ORACLE_SO

    def __init__(self):
        super().__init__()
        self._decode_plain = self._decode
        ex = self.lib.oracle_decode_ex
        ex.restype = C.c_int
        import threading

        self._tls = threading.local()

        def call(*args):
            self._tls.extra = np.zeros(8, np.int64)
            return ex(*args, self._tls.extra.ctypes.data_as(C.POINTER(C.c_int64)))

        self._decode = call

    def set_order_free(self, on):
        """See oracle/wfst_oracle.c `g_order_free`: apply each frame's FINAL next_cutoff to every arc
        (what the GPU computes) instead of the reference's visiting-order-dependent evolving one."""
        self.lib.oracle_set_order_free(int(bool(on)))

    def decode(self, *a, **kw):
        r = super().decode(*a, **kw)
        e = self._tls.extra
        r.extra = dict(N=int(e[0]), E=int(e[1]), Z=int(e[2]), tokens_created=int(e[3]), links_created=int(e[4]),
                       ties=int(e[5]), quirk_hops=int(e[6]))
        return r


@dataclass
class RawLattice:
    ok: bool
    n_states: int
    start: int
    st_final: np.ndarray
    a_src: np.ndarray
    a_dst: np.ndarray
    a_il: np.ndarray
    a_ol: np.ndarray
    a_graph: np.ndarray
    a_ac: np.ndarray
    st_frame: np.ndarray | None = None   # oracle / GPU only
    st_gstate: np.ndarray | None = None
    st_cost: np.ndarray | None = None

    def arc_multiset(self):
        """Isomorphism-invariant view: sorted rows (ilabel, olabel, graph bits, acoustic bits)."""
        k = np.stack([self.a_il, self.a_ol, self.a_graph.view(np.int32), self.a_ac.view(np.int32)], axis=1)
        return k[np.lexsort(k.T[::-1])]

    def labelled_arcs(self):
        """Rows (src frame, src graph state, dst frame, dst graph state, ilabel, olabel, graph bits, ac bits),
        sorted: equal for two implementations iff their lattices are identical up to state numbering."""
        f, g = self.st_frame, self.st_gstate
        k = np.stack([f[self.a_src], g[self.a_src], f[self.a_dst], g[self.a_dst], self.a_il, self.a_ol,
                      self.a_graph.view(np.int32), self.a_ac.view(np.int32)], axis=1)
        return k[np.lexsort(k.T[::-1])]


def _raw_lattice(lib, name, labelled, graph_handle, cfg, loglikes, tid2pdf, finalize, use_final_probs, max_states, max_arcs):
    ll = np.ascontiguousarray(loglikes, dtype=np.float32)
    T, stride = ll.shape
    n_tid = stride - 1
    if tid2pdf is not None:
        tid2pdf = np.ascontiguousarray(tid2pdf, dtype=np.int32)
        n_tid = int(tid2pdf.shape[0] - 1)
    ns, na, st = C.c_int(0), C.c_int(0), C.c_int(0)
    fin = np.zeros(max_states, np.int32)
    fr = np.zeros(max_states, np.int32)
    gs = np.zeros(max_states, np.int32)
    co = np.zeros(max_states, np.float32)
    src, dst, il, ol = (np.zeros(max_arcs, np.int32) for _ in range(4))
    gr, ac = np.zeros(max_arcs, np.float32), np.zeros(max_arcs, np.float32)
    f = getattr(lib, name)
    f.restype = C.c_int
    head = [C.c_void_p(graph_handle), C.byref(cfg), _fp(ll), T, stride, _ip(tid2pdf), n_tid, int(bool(finalize)),
            int(bool(use_final_probs)), max_states, C.byref(ns), C.byref(st), _ip(fin)]
    if labelled:
        head += [_ip(fr), _ip(gs), _fp(co)]
    ok = f(*head, max_arcs, C.byref(na), _ip(src), _ip(dst), _ip(il), _ip(ol), _fp(gr), _fp(ac))
    S, A = ns.value, na.value
    if S > max_states or A > max_arcs:
        raise ValueError("lattice larger than the caps: %d states, %d arcs" % (S, A))
    return RawLattice(bool(ok), S, st.value, fin[:S].copy(), src[:A].copy(), dst[:A].copy(), il[:A].copy(), ol[:A].copy(),
                      gr[:A].copy(), ac[:A].copy(), fr[:S].copy() if labelled else None, gs[:S].copy() if labelled else None,
                      co[:S].copy() if labelled else None)


def ref_raw_lattice(ref, graph_handle, cfg, loglikes, tid2pdf=None, finalize=True, use_final_probs=True,
                    max_states=1 << 20, max_arcs=1 << 21):
    return _raw_lattice(ref.lib, "ref_raw_lattice", False, graph_handle, cfg, loglikes, tid2pdf, finalize, use_final_probs,
                        max_states, max_arcs)


def oracle_raw_lattice(orc, graph_handle, cfg, loglikes, tid2pdf=None, finalize=True, use_final_probs=True,
                       max_states=1 << 20, max_arcs=1 << 21):
    return _raw_lattice(orc.lib, "oracle_raw_lattice", True, graph_handle, cfg, loglikes, tid2pdf, finalize, use_final_probs,
                        max_states, max_arcs)


def biglm_raw_lattice(dec, graph_handle, cfg, lm1, lm2, loglikes, tid2pdf=None, finalize=True, use_final_probs=True, fixed=True,
                      max_states=1 << 20, max_arcs=1 << 21):
    """GetRawLattice of the biglm decoder of `dec` (RefDecoder: the reference's OnlineLatticeDecoderMempoolBiglm as it is;
    OracleDecoder: the restatement, `fixed` chooses the DiffArpaLm mode, states labelled with frame / graph state / cost)."""
    ll = np.ascontiguousarray(loglikes, dtype=np.float32)
    T, stride = ll.shape
    n_tid = stride - 1
    if tid2pdf is not None:
        tid2pdf = np.ascontiguousarray(tid2pdf, dtype=np.int32)
        n_tid = int(tid2pdf.shape[0] - 1)
    is_ref = dec.PREFIX == "ref"
    ns, na, st = C.c_int(0), C.c_int(0), C.c_int(0)
    fin, fr, gs = (np.zeros(max_states, np.int32) for _ in range(3))
    co = np.zeros(max_states, np.float32)
    src, dst, il, ol = (np.zeros(max_arcs, np.int32) for _ in range(4))
    gr, ac = np.zeros(max_arcs, np.float32), np.zeros(max_arcs, np.float32)
    f = getattr(dec.lib, dec.PREFIX + "_biglm_raw_lattice")
    f.restype = C.c_int
    head = [C.c_void_p(graph_handle), C.byref(cfg), C.c_void_p(lm1.h), C.c_void_p(lm2.h)]
    if not is_ref:
        head.append(int(bool(fixed)))
    head += [_fp(ll), T, stride, _ip(tid2pdf), n_tid, int(bool(finalize)), int(bool(use_final_probs)), max_states, C.byref(ns),
             C.byref(st), _ip(fin)]
    if not is_ref:
        head += [_ip(fr), _ip(gs), _fp(co)]
    ok = f(*head, max_arcs, C.byref(na), _ip(src), _ip(dst), _ip(il), _ip(ol), _fp(gr), _fp(ac))
    S, A = ns.value, na.value
    if S > max_states or A > max_arcs:
        raise ValueError("lattice larger than the caps: %d states, %d arcs" % (S, A))
    lab = not is_ref
    return RawLattice(bool(ok), S, st.value, fin[:S].copy(), src[:A].copy(), dst[:A].copy(), il[:A].copy(), ol[:A].copy(),
                      gr[:A].copy(), ac[:A].copy(), fr[:S].copy() if lab else None, gs[:S].copy() if lab else None,
                      co[:S].copy() if lab else None)


def ref_lattice_write(ref, graph_handle, cfg, loglikes, path, tid2pdf=None):
    """Append the reference's GetRawLattice to `path` with the reference's own Lattice::Write."""
    ll = np.ascontiguousarray(loglikes, dtype=np.float32)
    T, stride = ll.shape
    n_tid = stride - 1
    if tid2pdf is not None:
        tid2pdf = np.ascontiguousarray(tid2pdf, dtype=np.int32)
        n_tid = int(tid2pdf.shape[0] - 1)
    f = ref.lib.ref_lattice_write
    f.restype = C.c_int
    return bool(f(C.c_void_p(graph_handle), C.byref(cfg), _fp(ll), T, stride, _ip(tid2pdf), n_tid, path.encode()))


def ref_lattice_read(ref, path, index, max_states=1 << 20, max_arcs=1 << 21):
    """Lattice number `index` of `path`, read with the reference's own Lattice::Read."""
    ns, na, st = C.c_int(0), C.c_int(0), C.c_int(0)
    fin = np.zeros(max_states, np.int32)
    src, dst, il, ol = (np.zeros(max_arcs, np.int32) for _ in range(4))
    gr, ac = np.zeros(max_arcs, np.float32), np.zeros(max_arcs, np.float32)
    f = ref.lib.ref_lattice_read
    f.restype = C.c_int
    ok = f(path.encode(), int(index), max_states, C.byref(ns), C.byref(st), _ip(fin), max_arcs, C.byref(na), _ip(src),
           _ip(dst), _ip(il), _ip(ol), _fp(gr), _fp(ac))
    S, A = ns.value, na.value
    return RawLattice(bool(ok), S, st.value, fin[:S].copy(), src[:A].copy(), dst[:A].copy(), il[:A].copy(), ol[:A].copy(),
                      gr[:A].copy(), ac[:A].copy())


def parse_lattice_file(data):
    """All lattices of a file in the reference's on-disk format (newfst/lattice-fst.cc:38-101): a
    list of RawLattice.  Pure numpy/struct restatement of Lattice::Read for the tests."""
    import struct

    out, o = [], 0
    while o < len(data):
        n, start = struct.unpack_from("<Qi", data, o)
        o += 12
        fin, src, rows = [], [], []
        for s in range(n):
            f, na = struct.unpack_from("<iQ", data, o)
            o += 12
            fin.append(f)
            a = np.frombuffer(data, dtype=np.dtype([("il", "<i4"), ("ol", "<i4"), ("g", "<f4"), ("ac", "<f4"), ("to", "<i4")]),
                              count=na, offset=o)
            o += 20 * na
            rows.append(a)
            src.append(np.full(na, s, np.int32))
        a = np.concatenate(rows) if rows else np.zeros(0, dtype=[("il", "<i4"), ("ol", "<i4"), ("g", "<f4"), ("ac", "<f4"), ("to", "<i4")])
        src = np.concatenate(src) if src else np.zeros(0, np.int32)
        out.append(RawLattice(True, n, start, np.asarray(fin, np.int32), src, a["to"].astype(np.int32), a["il"].astype(np.int32),
                              a["ol"].astype(np.int32), a["g"].astype(np.float32), a["ac"].astype(np.float32)))
    return out


REF_CONVERT = os.path.join(HERE, "_ref", "convert_fst")


def ref_convert_fst(in_path, out_path):
    """The reference's own OpenFst-vector -> flat converter (fst_format_convert_tool/convert_fst.c),
    compiled as it is into oracle/_ref/convert_fst.  It appends to out_path and prints every arc."""
    if os.path.exists(out_path):
        os.remove(out_path)
    subprocess.check_call([REF_CONVERT, in_path, out_path], stdout=subprocess.DEVNULL)


def ref_constfst_dump(ref, path, max_states=1 << 22, max_arcs=1 << 24):
    """ConstFst<StdArc,int>::Read + Fst(ConstFst) of the reference -> (start, final, state_info[S,3], arcs[A,4] int32)."""
    st, fin, ns, na = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    si = np.zeros((max_states, 3), np.uint32)
    arcs = np.zeros((max_arcs, 4), np.int32)
    f = ref.lib.ref_constfst_dump
    f.restype = C.c_int
    ok = f(path.encode(), C.byref(st), C.byref(fin), max_states, C.byref(ns), si.ctypes.data_as(C.POINTER(C.c_uint)), max_arcs,
           C.byref(na), _ip(arcs))
    if not ok:
        raise IOError("reference could not read %s" % path)
    return st.value, fin.value, si[: ns.value].copy(), arcs[: na.value].copy()


def ref_nbest_from_lattice_file(ref, path, index, n, max_len=512):
    """The reference's own determinize + n-shortest-paths on lattice `index` of a lattice file.
    Returns (list of (words, tot_score, lm_score), determinized states, determinized arcs) or None."""
    words = np.zeros((n, max_len), np.int32)
    nw = np.zeros(n, np.int32)
    sc = np.zeros((n, 2), np.float32)
    ds, da = C.c_int(0), C.c_int(0)
    f = ref.lib.ref_nbest_from_lattice_file
    f.restype = C.c_int
    k = f(path.encode(), int(index), int(n), int(max_len), _ip(words), _ip(nw), _fp(sc), C.byref(ds), C.byref(da))
    if k < 0:
        return None
    return [(words[i, : nw[i]].copy(), float(sc[i, 0]), float(sc[i, 1])) for i in range(k)], ds.value, da.value


# ---- biglm (BASELINE configs[3]) -----------------------------------------------------------------
class quiet_stdout:
    """the reference's ArpaLm::Read / Arpa2Fsa print progress on stdout (arpa2fsa.cc:128,161-171)"""

    def __enter__(self):
        import sys

        sys.stdout.flush()
        self.saved = os.dup(1)
        self.devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self.devnull, 1)

    def __exit__(self, *exc):
        os.dup2(self.saved, 1)
        os.close(self.saved)
        os.close(self.devnull)


def ref_arpa2fsa(ref, arpa_path, wordlist_path, out_path, nthread=1):
    """The reference's own ARPA -> binary LM converter (Arpa2Fsa::ConvertArpa2Fsa + ArpaLm::Write)."""
    if os.path.exists(out_path):
        os.remove(out_path)
    f = ref.lib.ref_arpa2fsa
    f.restype = C.c_int
    with quiet_stdout():
        ok = f(arpa_path.encode(), wordlist_path.encode(), out_path.encode(), int(nthread))
    if not ok:
        raise IOError("reference Arpa2Fsa failed on %s" % arpa_path)


class Lm:
    """An LM automaton loaded by one of the checkers (prefix 'ref' or 'oracle')."""

    def __init__(self, dec, path, scale=1.0):
        self.dec, self.p = dec, dec.PREFIX
        f = getattr(dec.lib, self.p + "_lm_load")
        f.restype = C.c_void_p
        with quiet_stdout():
            self.h = f(path.encode(), C.c_float(scale))
        if not self.h:
            raise IOError("cannot read LM %s" % path)

    def free(self):
        if self.h:
            getattr(self.dec.lib, self.p + "_lm_free")(C.c_void_p(self.h))
            self.h = None

    def start(self):
        f = getattr(self.dec.lib, self.p + "_lm_start")
        f.restype = C.c_int
        return int(f(C.c_void_p(self.h)))

    def final(self, s):
        f = getattr(self.dec.lib, self.p + "_lm_final")
        f.restype = C.c_float
        return float(f(C.c_void_p(self.h), int(s)))

    def getarc_many(self, states, words):
        """ComposeArpaLm::GetArc for every (state, word): (next states, Value1 costs)"""
        st = np.ascontiguousarray(states, np.int32)
        wd = np.ascontiguousarray(words, np.int32)
        nx = np.zeros(st.shape[0], np.int32)
        v1 = np.zeros(st.shape[0], np.float32)
        getattr(self.dec.lib, self.p + "_lm_getarc_many")(C.c_void_p(self.h), int(st.shape[0]), _ip(st), _ip(wd), _ip(nx), _fp(v1))
        return nx, v1


def biglm_decode(dec, graph_handle, cfg, lm1, lm2, loglikes, tid2pdf=None, chunk=0, finalize=True, use_final_probs=True,
                 trace=False, fixed=True):
    """One utterance through the biglm decoder of `dec` (RefDecoder: the reference's
    OnlineLatticeDecoderMempoolBiglm as it is; OracleDecoder: the restatement, `fixed` chooses the
    DiffArpaLm mode).  lm1 = old LM loaded with scale -1, lm2 = new LM."""
    ll = np.ascontiguousarray(loglikes, dtype=np.float32)
    T, stride = ll.shape
    if tid2pdf is not None:
        tid2pdf = np.ascontiguousarray(tid2pdf, dtype=np.int32)
        n_tid = int(tid2pdf.shape[0] - 1)
    else:
        n_tid = stride - 1
    max_path = 16 * T + 256
    pi, po, words, tids = (np.zeros(max_path, np.int32) for _ in range(4))
    pg, pa = np.zeros(max_path, np.float32), np.zeros(max_path, np.float32)
    n_path, n_words, n_tids = C.c_int(0), C.c_int(0), C.c_int(0)
    tot, lm = C.c_float(0), C.c_float(0)
    fn = fb = None
    if trace:
        chunk = 1
        fn = np.zeros(T + 1, np.int32)
        fb = np.zeros(T + 1, np.float32)
    nt, nl = C.c_int(0), C.c_int(0)
    is_ref = dec.PREFIX == "ref"
    f = getattr(dec.lib, dec.PREFIX + "_biglm_decode")
    f.restype = C.c_int
    head = [C.c_void_p(graph_handle), C.byref(cfg), C.c_void_p(lm1.h), C.c_void_p(lm2.h)]
    if not is_ref:
        head.append(int(bool(fixed)))
    ex = np.zeros(10, np.int64)
    args = head + [_fp(ll), T, stride, _ip(tid2pdf), n_tid, int(chunk), int(bool(finalize)), int(bool(use_final_probs)),
                   _ip(pi), _ip(po), _fp(pg), _fp(pa), max_path, C.byref(n_path), C.byref(tot), C.byref(lm), _ip(words), max_path,
                   C.byref(n_words), _ip(tids), max_path, C.byref(n_tids), _ip(fn), _fp(fb), C.byref(nt), C.byref(nl)]
    if not is_ref:
        args.append(ex.ctypes.data_as(C.POINTER(C.c_int64)))
    ok = f(*args)
    n = n_path.value
    if n > max_path:
        raise RuntimeError("best path of %d hops does not fit the binding's %d-hop buffers" % (n, max_path))
    r = Result(bool(ok), words[: n_words.value].copy(), tids[: n_tids.value].copy(), float(tot.value), float(lm.value),
               pi[:n].copy(), po[:n].copy(), pg[:n].copy(), pa[:n].copy(), fn, fb, None, nt.value, nl.value)
    if not is_ref:
        r.extra = dict(N=int(ex[0]), E=int(ex[1]), Z=int(ex[2]), tokens_created=int(ex[3]), links_created=int(ex[4]),
                       ties=int(ex[5]), quirk_hops=int(ex[6]), lm_pairs=int(ex[7] & ((1 << 40) - 1)), lm_oob=int(ex[7] >> 40), L=int(ex[8]), L_eps=int(ex[9]))
    return r


# ---- determinized lattices -----------------------------------------------------------
def ref_determinize_lattice_file(ref, path, index, max_states=1 << 20, max_arcs=1 << 21):
    """The reference's DeterminizeLatticeWrapper (newfst/lattice-determinize-api.cc:5-21) on lattice `index`
    of a file in its on-disk lattice format: RawLattice of the result (arcs: ilabel 0, olabel word), or None."""
    ns, na, st = C.c_int(0), C.c_int(0), C.c_int(0)
    fin = np.zeros(max_states, np.int32)
    src, dst, il, ol = (np.zeros(max_arcs, np.int32) for _ in range(4))
    gr, ac = np.zeros(max_arcs, np.float32), np.zeros(max_arcs, np.float32)
    f = ref.lib.ref_determinize_lattice_file
    f.restype = C.c_int
    ok = f(path.encode(), int(index), max_states, C.byref(ns), C.byref(st), _ip(fin), max_arcs, C.byref(na), _ip(src), _ip(dst),
           _ip(il), _ip(ol), _fp(gr), _fp(ac))
    if not ok:
        return None
    S, A = ns.value, na.value
    return RawLattice(True, S, st.value, fin[:S].copy(), src[:A].copy(), dst[:A].copy(), il[:A].copy(), ol[:A].copy(),
                      gr[:A].copy(), ac[:A].copy())


def ref_rescore_lattice_file(ref, path, index, lm1, lm2, max_states=1 << 20, max_arcs=1 << 21):
    """The service's GetLattice under --use-second (kaldi-online-nnet3-my-decoder.cc:53-78): determinize, then ComposeLattice with the
    old LM (loaded with scale -1) and with the new one, by the compiled reference; RawLattice of the result, or None."""
    ns, na, st = C.c_int(0), C.c_int(0), C.c_int(0)
    fin = np.zeros(max_states, np.int32)
    src, dst, il, ol = (np.zeros(max_arcs, np.int32) for _ in range(4))
    gr, ac = np.zeros(max_arcs, np.float32), np.zeros(max_arcs, np.float32)
    f = ref.lib.ref_rescore_lattice_file
    f.restype = C.c_int
    ok = f(path.encode(), int(index), C.c_void_p(lm1.h), C.c_void_p(lm2.h), max_states, C.byref(ns), C.byref(st), _ip(fin), max_arcs,
           C.byref(na), _ip(src), _ip(dst), _ip(il), _ip(ol), _fp(gr), _fp(ac))
    if not ok:
        return None
    S, A = ns.value, na.value
    return RawLattice(True, S, st.value, fin[:S].copy(), src[:A].copy(), dst[:A].copy(), il[:A].copy(), ol[:A].copy(),
                      gr[:A].copy(), ac[:A].copy())


def ref_nbest_paths_from_lattice_file(ref, path, index, n, lm1=None, lm2=None, max_arcs=1 << 21):
    """The service's GetNbest as lattices (kaldi-online-nnet3-my-decoder.cc:97-105): determinize [+ ComposeLattice with lm1, lm2] +
    NShortestPath + ConvertNbestToVector by the compiled reference.  List of paths, each a dict of per-arc arrays
    (ilabel, olabel, graph, acoustic) in the order the linear lattice is walked from its start; None on failure."""
    off = np.zeros(n + 2, np.int32)
    il, ol = np.zeros(max_arcs, np.int32), np.zeros(max_arcs, np.int32)
    gr, ac = np.zeros(max_arcs, np.float32), np.zeros(max_arcs, np.float32)
    na = C.c_int(0)
    f = ref.lib.ref_nbest_paths_from_lattice_file
    f.restype = C.c_int
    k = f(path.encode(), int(index), int(n), C.c_void_p(lm1.h if lm1 else None), C.c_void_p(lm2.h if lm2 else None), int(n), _ip(off),
          max_arcs, C.byref(na), _ip(il), _ip(ol), _fp(gr), _fp(ac))
    if k < 0 or na.value > max_arcs:
        return None
    return [dict(ilabel=il[off[i]:off[i + 1]].copy(), olabel=ol[off[i]:off[i + 1]].copy(), graph=gr[off[i]:off[i + 1]].copy(),
                 acoustic=ac[off[i]:off[i + 1]].copy()) for i in range(k)]


def nshortest_paths(L, n):
    """NShortestPath + ConvertNbestToVector (newfst/lattice-to-nbest.cc:15-199), restated on a RawLattice whose state 0 is the start
    (a determinized or rescored lattice): backward best costs in topological order, then best-first expansion of (state, forward
    cost) pairs ordered by forward + backward cost, a state expanded at most n times, until n paths have reached the super-final
    state (AddSuperFinalState: a 0-cost arc from every final state).  Forward costs are float32 sums front to back, an arc's cost
    is graph + acoustic in float32.  Returns the paths in the order they are found: dict(olabel, graph, acoustic, tot), arcs front
    to back (without the epsilon arcs the two Reverse calls and the super-final state add)."""
    import heapq

    S = L.n_states
    out = [[] for _ in range(S)]
    for k in range(len(L.a_src)):
        out[int(L.a_src[k])].append(k)
    val = (L.a_graph.astype(np.float32) + L.a_ac.astype(np.float32)).astype(np.float32)
    # topological order (TopSort, topsort.cc) -- any one will do for the backward costs
    indeg = np.zeros(S, np.int64)
    for k in range(len(L.a_src)):
        indeg[L.a_dst[k]] += 1
    order, stack = [], [s for s in range(S) if indeg[s] == 0]
    while stack:
        s = stack.pop()
        order.append(s)
        for k in out[s]:
            indeg[L.a_dst[k]] -= 1
            if indeg[L.a_dst[k]] == 0:
                stack.append(int(L.a_dst[k]))
    back = np.full(S + 1, np.inf, np.float32)   # [S] = the super-final state
    back[S] = 0.0
    for s in reversed(order):
        if L.st_final[s]:
            back[s] = np.float32(0.0) + back[S]   # its arc to the super-final state
        for k in out[s]:
            c = np.float32(val[k] + back[L.a_dst[k]])
            if c < back[s]:
                back[s] = c
    pairs = [(0, np.float32(0.0), -1, -1)]   # (state, forward cost, parent pair, arc)
    heap = [(np.float32(back[0]), 0, 0)]
    r = {}
    finals = []
    seq = 1
    while heap:
        _, _, pid = heapq.heappop(heap)
        st, w, _, _ = pairs[pid]
        r[st] = r.get(st, 0) + 1
        if len(finals) == n:
            break
        if r[st] > n:
            continue
        if st == S:
            finals.append(pid)
            continue
        succ = [(int(L.a_dst[k]), np.float32(w + val[k]), k) for k in out[st]]
        if L.st_final[st]:
            succ.append((S, np.float32(w + np.float32(0.0)), -2))
        for (to, w2, k) in succ:
            pairs.append((to, w2, pid, k))
            heapq.heappush(heap, (np.float32(w2 + back[to]), seq, len(pairs) - 1))
            seq += 1
    res = []
    for pid in finals:
        arcs = []
        tot = pairs[pid][1]
        while pid > 0:
            _, _, par, k = pairs[pid]
            if k >= 0:
                arcs.append(k)
            pid = par
        arcs.reverse()
        a = np.array(arcs, np.int64)
        res.append(dict(olabel=L.a_ol[a].astype(np.int32), graph=L.a_graph[a].astype(np.float32), acoustic=L.a_ac[a].astype(np.float32),
                        tot=float(tot)))
    return res


def compose_lattice(det, lm, scale=1.0):
    """ComposeLattice (newfst/compose-lat-inl.h:15-130) + Connect (newfst/connect-fst.cc:10-22), restated: `det` a RawLattice (state 0 =
    start), `lm` a pyoracle.Lm of the C oracle (ComposeArpaLm::GetArc / Final / Start = its getarc_many / final / start).  Pairs
    (lattice state, LM state) breadth first in the reference's order; float32 arithmetic in its operation order.  Pure Python: for the
    determinized lattices of the tests (a few hundred states)."""
    f32 = np.float32
    S = det.n_states
    order = np.argsort(det.a_src, kind="stable")
    by_src = [[] for _ in range(S)]
    for k in order:
        by_src[int(det.a_src[k])].append(int(k))
    ids = {(det.start, lm.start()): 0}
    queue = [(det.start, lm.start())]
    fin = [0]
    arcs = []
    qi = 0
    sc = f32(scale)
    while qi < len(queue):
        s1, s2 = queue[qi]
        sid = qi
        qi += 1
        for k in by_src[s1]:
            ol = int(det.a_ol[k])
            n1 = int(det.a_dst[k])
            n2, lw = s2, f32(0.0)
            if ol != 0:
                nx, v = lm.getarc_many(np.array([s2], np.int32), np.array([ol], np.int32))
                n2, lw = int(nx[0]), f32(v[0])
            key = (n1, n2)
            if key not in ids:
                ids[key] = len(queue)
                queue.append(key)
                fin.append(0)
            nid = ids[key]
            final_score = f32(0.0)
            if det.st_final[n1]:
                final_score = f32(lm.final(n2))
                if np.isinf(final_score):
                    final_score = f32(0.0)
                else:
                    fin[nid] = 1
            g, a = f32(det.a_graph[k]), f32(det.a_ac[k])
            if ol == 0:
                arcs.append((sid, nid, int(det.a_il[k]), 0, f32(g + f32(final_score * sc)), a))
            else:
                arcs.append((sid, nid, int(det.a_il[k]), ol, f32(g + f32(f32(lw + final_score) * sc)), f32(a + f32(f32(0.0) * sc))))
    n = len(queue)
    keep = list(fin)
    changed = True
    while changed:   # Connect: every composed state is accessible; keep the ones that reach a final state
        changed = False
        for (s, d, _, _, _, _) in arcs:
            if keep[d] and not keep[s]:
                keep[s] = 1
                changed = True
    renum = {}
    for i in range(n):
        if keep[i]:
            renum[i] = len(renum)
    kept = [(renum[s], renum[d], il, ol, g, a) for (s, d, il, ol, g, a) in arcs if s in renum and d in renum]
    fin2 = np.asarray([fin[i] for i in range(n) if keep[i]], np.int32)
    A = len(kept)
    col = lambda j, t: np.asarray([x[j] for x in kept], t) if A else np.zeros(0, t)
    return RawLattice(True, len(renum), 0, fin2, col(0, np.int32), col(1, np.int32), col(2, np.int32), col(3, np.int32), col(4, np.float32),
                      col(5, np.float32))


DET_HOST_SO = os.path.join(HERE, "_build", "libdet_host.so")


def build_det_host():
    """tests/det_host.cc: asr-decoder_amd/csrc/wfst_determinize.h (the algorithm the device runs) compiled for the host."""
    src = os.path.join(os.path.dirname(HERE), "tests", "det_host.cc")
    hdr = os.path.join(os.path.dirname(HERE), "asr-decoder_amd", "csrc", "wfst_determinize.h")
    if (not os.path.exists(DET_HOST_SO)) or os.path.getmtime(DET_HOST_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(DET_HOST_SO), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-o", DET_HOST_SO, src])
    return C.CDLL(DET_HOST_SO)


def det_host_run(lib, L, cap_scale=4, max_states=1 << 20, max_arcs=1 << 21, low_tmp=0):
    """The device's determinization code, run on the host, on the raw lattice L (RawLattice): (status, RawLattice).
    low_tmp > 0: the closure's fast buffers (LDS on the device) emulated at this many elements."""
    lib.det_host_set_low(int(low_tmp))
    ns, na = C.c_int(0), C.c_int(0)
    fin = np.zeros(max_states, np.int32)
    src, dst, il, ol = (np.zeros(max_arcs, np.int32) for _ in range(4))
    gr, ac = np.zeros(max_arcs, np.float32), np.zeros(max_arcs, np.float32)
    f = lib.det_host_run
    f.restype = C.c_int
    c = lambda a, t: np.ascontiguousarray(a, t)
    rc = f(int(L.n_states), _ip(c(L.st_final, np.int32)), int(len(L.a_src)), _ip(c(L.a_src, np.int32)), _ip(c(L.a_dst, np.int32)),
           _ip(c(L.a_il, np.int32)), _ip(c(L.a_ol, np.int32)), _fp(c(L.a_graph, np.float32)), _fp(c(L.a_ac, np.float32)),
           int(cap_scale), max_states, C.byref(ns), _ip(fin), max_arcs, C.byref(na), _ip(src), _ip(dst), _ip(il), _ip(ol), _fp(gr), _fp(ac))
    S, A = ns.value, na.value
    return rc, RawLattice(rc == 0, S, 0, fin[:S].copy(), src[:A].copy(), dst[:A].copy(), il[:A].copy(), ol[:A].copy(),
                          gr[:A].copy(), ac[:A].copy())
