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
        max_path = 4 * T + 64
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

    def decode(self, *a, **kw):
        r = super().decode(*a, **kw)
        e = self._tls.extra
        r.extra = dict(N=int(e[0]), E=int(e[1]), Z=int(e[2]), tokens_created=int(e[3]), links_created=int(e[4]),
                       ties=int(e[5]), quirk_hops=int(e[6]))
        return r
