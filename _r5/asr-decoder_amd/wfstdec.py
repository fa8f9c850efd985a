"""ctypes binding of the C ABI in include/wfst_decoder.h (libwfstdec.so).

Thin plumbing for tests, bench.py and the smoke entry: numpy in, numpy out; device matrices
are passed as raw device pointers (e.g. ``torch.Tensor.data_ptr()``).  There is no CPU
fallback: if the HIP library is missing or no MI355X is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# (eight hardware queues for the process's HIP streams where nothing else is asked for: has an effect if this module is imported
# before the HIP runtime starts -- a decoder with four channel groups has five streams, and the runtime's default is four queues)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# (WFST_LIB_VARIANT: kernel experiments only -- tools/ab_bench.sh builds variants of the library beside the product one with
# `build.py --variant NAME -D...` and times them on one box; the product path never sets it)
LIB_PATH = os.path.join(HERE, "lib", "libwfstdec%s.so" % (("_" + os.environ["WFST_LIB_VARIANT"]) if os.environ.get("WFST_LIB_VARIANT") else ""))

WFST_OK = 0
ERR_NAMES = {-1: "WFST_E_ARG", -2: "WFST_E_IO", -3: "WFST_E_DEVICE", -4: "WFST_E_CAPACITY",
             -5: "WFST_E_STATE", -6: "WFST_E_FORMAT"}

# every symbol include/wfst_decoder.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "wfst_config_default", "wfst_last_error", "wfst_device_count", "wfst_graph_load", "wfst_graph_convert_file",
    "wfst_graph_from_arrays", "wfst_graph_set_tid2pdf", "wfst_graph_info", "wfst_graph_free",
    "wfst_decoder_create", "wfst_decoder_free", "wfst_decoder_init", "wfst_decoder_advance",
    "wfst_decoder_advance_host", "wfst_decoder_finalize", "wfst_decoder_sync",
    "wfst_decoder_num_frames_decoded", "wfst_decoder_get_best_path", "wfst_lattice_to_vector", "wfst_lattice_to_vector_batch",
    "wfst_decoder_get_stats", "wfst_decoder_get_frontier", "wfst_decoder_set_profiling",
    "wfst_decoder_get_profile", "wfst_decoder_get_profile_busy", "wfst_decoder_channel_groups", "wfst_decoder_get_raw_lattice", "wfst_decoder_get_nbest",
    "wfst_options_default", "wfst_graph_options_default", "wfst_graph_load_ex", "wfst_graph_from_arrays_ex",
    "wfst_decoder_create_ex", "wfst_lm_load", "wfst_lm_from_arrays", "wfst_lm_info", "wfst_lm_free",
    "wfst_decoder_create_biglm", "wfst_decoder_get_determinized_lattice", "wfst_decoder_get_lattice_stats", "wfst_decoder_get_rescored_lattice", "wfst_decoder_get_nbest_paths",
    "wfst_decoder_get_degraded_frames", "wfst_decoder_get_path_flags", "wfst_decoder_rescore_lattices", "wfst_decoder_nbest_paths_batch",
    "wfst_decoder_prefetch_determinized", "wfst_lattice_labels_batch",
    "wfst_decoder_prefetch_determinized_detached", "wfst_decoder_get_prefetched_lattice", "wfst_decoder_harvest_prefetched",
]


class _LazyList(object):
    """A read-only sequence whose items are put together from the batch's host arrays when first looked at, and kept."""

    def __init__(self, n, make):
        self._items = [None] * n
        self._make = make

    def __len__(self):
        return len(self._items)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self._items)))]
        if i < 0:
            i += len(self._items)
        if self._items[i] is None:
            self._items[i] = self._make(i)
        return self._items[i]

    def __iter__(self):
        for i in range(len(self._items)):
            yield self[i]

    def __eq__(self, other):
        return list(self) == list(other)


class _BestPaths(object):
    """What best_paths() returns: a read-only sequence of per-channel dicts over the batch's hop arrays.  Everything the call
    computes -- hops, scores, the words and transition-ids of every path -- is in host arrays when it returns (the epsilons are
    dropped for the whole batch at once, wfst_lattice_labels_batch); an utterance's dict is put together when it is first looked at and kept (a service
    that decodes 128 utterances per step does not pay a Python loop over them inside the step)."""

    def __init__(self, il, ol, g, ac, nh, tot_s, lm_s, words, woff, tids, toff):
        self._a = (il, ol, g, ac, nh, tot_s, lm_s)
        self._words, self._woff, self._tids, self._toff = words, woff, tids, toff   # (wfst_lattice_labels_batch: path after path, hop order)
        self._items = [None] * len(nh)

    def __len__(self):
        return len(self._items)

    def _make(self, i):
        il, ol, g, ac, nh, tot_s, lm_s = self._a
        k = int(nh[i])
        return dict(ok=k > 0, ilabel=il[i, :k], olabel=ol[i, :k], graph=g[i, :k], ac=ac[i, :k],
                    words=self._words[self._woff[i]:self._woff[i + 1]], tids=self._tids[self._toff[i]:self._toff[i + 1]],
                    tot_score=float(tot_s[i]) if k else 0.0, lm_score=float(lm_s[i]) if k else 0.0)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self._items)))]
        if i < 0:
            i += len(self._items)
        if self._items[i] is None:
            self._items[i] = self._make(i)
        return self._items[i]

    def __iter__(self):
        for i in range(len(self._items)):
            yield self[i]


class WfstError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "WFST_E_?"), code, msg))
        self.code = code


class Config(C.Structure):
    """wfst_config == LatticeFasterDecoderConfig (reference
    src/my-decoder/lattice-faster-decoder-conf.h:21-44), reference defaults."""

    _fields_ = [("beam", C.c_float), ("max_active", C.c_int32), ("min_active", C.c_int32),
                ("lattice_beam", C.c_float), ("prune_interval", C.c_int32), ("beam_delta", C.c_float),
                ("hash_ratio", C.c_float), ("prune_scale", C.c_float)]

    def __init__(self, beam=16.0, max_active=2147483647, min_active=200, lattice_beam=10.0,
                 prune_interval=25, beam_delta=0.5, hash_ratio=2.0, prune_scale=0.1):
        super().__init__(beam, max_active, min_active, lattice_beam, prune_interval, beam_delta,
                         hash_ratio, prune_scale)


class Limits(C.Structure):
    _fields_ = [("max_frames", C.c_int32), ("max_tokens_per_frame", C.c_int32), ("arena_tokens", C.c_int64),
                ("lattice_links", C.c_int64), ("lm_pairs", C.c_int64), ("det_raw_states", C.c_int32), ("det_raw_arcs", C.c_int32),
                ("det_workspace_bytes", C.c_int64)]


class Options(C.Structure):
    """wfst_options: scheduling choices of a decoder (never a result bit); defaults from the library."""

    _fields_ = [("channel_groups", C.c_int32), ("use_hip_graph", C.c_int32), ("log2_partitions", C.c_int32),
                ("log2_lds_slots", C.c_int32), ("joint_max", C.c_int32), ("expand_workgroups", C.c_int32),
                ("insert_workgroups", C.c_int32), ("upload_slice_frames", C.c_int32), ("tile_tokens", C.c_int32), ("debug", C.c_int32)]

    def __init__(self, **kw):
        super().__init__()
        lib().wfst_options_default(C.byref(self))
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise TypeError("unknown wfst_options field %r" % k)
            setattr(self, k, int(v))


class GraphOptions(C.Structure):
    _fields_ = [("row_align_slots", C.c_int32), ("flatten_closures", C.c_int32), ("fuse_closures", C.c_int32)]

    def __init__(self, **kw):
        super().__init__()
        lib().wfst_graph_options_default(C.byref(self))
        for k, v in kw.items():
            if k not in dict(self._fields_):
                raise TypeError("unknown wfst_graph_options field %r" % k)
            setattr(self, k, int(v))


_lib = None


def lib():
    """Load libwfstdec.so; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  If
        # this library were dlopen'ed first it would bind /opt/rocm's copy and torch would then
        # bring up a second runtime ("No HIP GPUs are available").  Loading torch first lets the
        # loader satisfy our DT_NEEDED libamdhip64.so.7 with the copy already in the process.
        if os.environ.get("WFST_NO_TORCH", "0") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        L.wfst_last_error.restype = C.c_char_p
        L.wfst_decoder_get_frontier.restype = C.c_int
        _lib = L
    return _lib


def _check(rc):
    if rc != WFST_OK:
        raise WfstError(rc, lib().wfst_last_error().decode())


def _i32(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def _f32(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def device_count():
    return int(lib().wfst_device_count())


class Graph:
    """HCLG resident in HBM (replaces the reference's ``Fst``)."""

    def __init__(self, handle):
        self.h = handle

    @staticmethod
    def load(path, device=0, options=None):
        h = C.c_void_p()
        _check(lib().wfst_graph_load_ex(path.encode(), int(device), C.byref(options) if options is not None else None,
                                        C.byref(h)))
        return Graph(h)

    @staticmethod
    def from_arrays(start, final_state, state_info, arcs, device=0, options=None):
        si = np.ascontiguousarray(state_info)
        ar = np.ascontiguousarray(arcs)
        assert si.dtype.itemsize == 12 and ar.dtype.itemsize == 16
        h = C.c_void_p()
        _check(lib().wfst_graph_from_arrays_ex(int(start), int(final_state), int(si.shape[0]), int(ar.shape[0]),
                                               si.ctypes.data_as(C.c_void_p), ar.ctypes.data_as(C.c_void_p),
                                               int(device), C.byref(options) if options is not None else None,
                                               C.byref(h)))
        return Graph(h)

    def set_tid2pdf(self, tid2pdf):
        m = np.ascontiguousarray(tid2pdf, dtype=np.int32)
        _check(lib().wfst_graph_set_tid2pdf(self.h, _i32(m), int(m.shape[0] - 1)))

    def info(self):
        s, f, ns, na = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        b = C.c_int64()
        _check(lib().wfst_graph_info(self.h, C.byref(s), C.byref(f), C.byref(ns), C.byref(na), C.byref(b)))
        return dict(start=s.value, final_state=f.value, n_states=ns.value, n_arcs=na.value, device_bytes=b.value)

    def free(self):
        if self.h:
            lib().wfst_graph_free(self.h)
            self.h = None


class Lm:
    """A back-off n-gram LM automaton resident in HBM (replaces the reference's ``ArpaLm``); `scale`
    is ArpaLm::Rescale -- the biglm CLI loads the OLD LM with -1."""

    def __init__(self, handle):
        self.h = handle

    @staticmethod
    def load(path, scale=1.0, device=0):
        h = C.c_void_p()
        _check(lib().wfst_lm_load(path.encode(), C.c_float(scale), int(device), C.byref(h)))
        return Lm(h)

    @staticmethod
    def from_arrays(bos, eos, unk, states, arcs, scale=1.0, device=0):
        st = np.ascontiguousarray(states)
        ar = np.ascontiguousarray(arcs)
        assert st.dtype.itemsize == 12 and ar.dtype.itemsize == 12
        h = C.c_void_p()
        _check(lib().wfst_lm_from_arrays(int(bos), int(eos), int(unk), int(st.shape[0]), st.ctypes.data_as(C.c_void_p),
                                         int(ar.shape[0]), ar.ctypes.data_as(C.c_void_p), C.c_float(scale), int(device), C.byref(h)))
        return Lm(h)

    def info(self):
        v = [C.c_int32() for _ in range(5)]
        b = C.c_int64()
        _check(lib().wfst_lm_info(self.h, *[C.byref(x) for x in v], C.byref(b)))
        return dict(bos=v[0].value, eos=v[1].value, n_states=v[2].value, n_arcs=v[3].value, n_words=v[4].value, device_bytes=b.value)

    def free(self):
        if self.h:
            lib().wfst_lm_free(self.h)
            self.h = None


class BatchDecoder:
    """A batch of decoding channels (one channel == one reference decoder object).  old_lm / new_lm:
    biglm mode (the reference's OnlineLatticeDecoderMempoolBiglm)."""

    def __init__(self, graph, cfg, n_channels, max_frames=0, max_tokens_per_frame=0, arena_tokens=0, stream=None,
                 lattice_links=0, options=None, old_lm=None, new_lm=None, lm_pairs=0, det_raw_states=0, det_raw_arcs=0,
                 det_workspace_bytes=0):
        self.graph = graph
        self.n = int(n_channels)
        lim = Limits(int(max_frames), int(max_tokens_per_frame), int(arena_tokens), int(lattice_links), int(lm_pairs),
                     int(det_raw_states), int(det_raw_arcs), int(det_workspace_bytes))
        self.lattice_links = int(lattice_links)
        h = C.c_void_p()
        _check(lib().wfst_decoder_create_biglm(graph.h, C.byref(cfg), self.n, C.byref(lim),
                                               C.byref(options) if options is not None else None,
                                               old_lm.h if old_lm is not None else None, new_lm.h if new_lm is not None else None,
                                               C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h

    def free(self):
        if self.h:
            lib().wfst_decoder_free(self.h)
            self.h = None

    def _chan(self, channels):
        if channels is None:
            return None, 0
        ch = np.ascontiguousarray(channels, dtype=np.int32)
        return ch, int(ch.shape[0])

    def init(self, channels=None):
        ch, n = self._chan(channels)
        _check(lib().wfst_decoder_init(self.h, _i32(ch), n))

    def advance(self, ll_ptrs, n_frames_ready, stride, channels=None, max_num_frames=-1):
        """ll_ptrs: device addresses (ints) of row 0 of each listed channel's matrix."""
        ch, n = self._chan(channels)
        cnt = n if ch is not None else self.n
        ptrs = (C.c_void_p * cnt)(*[int(p) for p in ll_ptrs])
        nr = np.ascontiguousarray(n_frames_ready, dtype=np.int32)
        assert nr.shape[0] == cnt
        _check(lib().wfst_decoder_advance(self.h, _i32(ch), n, ptrs, _i32(nr), int(stride), int(max_num_frames)))

    def advance_host(self, mats, n_frames_ready, channels=None, max_num_frames=-1):
        """mats: list of C-contiguous float32 [frames][stride] numpy arrays (host)."""
        ch, n = self._chan(channels)
        cnt = n if ch is not None else self.n
        mats = [np.ascontiguousarray(m, dtype=np.float32) for m in mats]
        stride = int(mats[0].shape[1])
        assert all(m.shape[1] == stride for m in mats)
        ptrs = (C.c_void_p * cnt)(*[m.ctypes.data for m in mats])
        nr = np.ascontiguousarray(n_frames_ready, dtype=np.int32)
        _check(lib().wfst_decoder_advance_host(self.h, _i32(ch), n, ptrs, _i32(nr), stride, int(max_num_frames)))

    def finalize(self, channels=None):
        ch, n = self._chan(channels)
        _check(lib().wfst_decoder_finalize(self.h, _i32(ch), n))

    def sync(self):
        _check(lib().wfst_decoder_sync(self.h))

    def num_frames_decoded(self, channel):
        return int(lib().wfst_decoder_num_frames_decoded(self.h, int(channel)))

    def best_paths(self, channels=None, use_final_probs=True, cap=2048):
        """Returns one dict per listed channel: ok, ilabel, olabel, graph, ac, words, tids,
        tot_score, lm_score (LatticeToVector applied to the hop list)."""
        ch, n = self._chan(channels)
        cnt = n if ch is not None else self.n
        il = np.empty((cnt, cap), np.int32)    # (the call fills all of it)
        ol = np.empty((cnt, cap), np.int32)
        g = np.empty((cnt, cap), np.float32)
        ac = np.empty((cnt, cap), np.float32)
        nh = np.zeros(cnt, np.int32)
        _check(lib().wfst_decoder_get_best_path(self.h, _i32(ch), n, int(bool(use_final_probs)), int(cap),
                                                _i32(il), _i32(ol), _f32(g), _f32(ac), _i32(nh)))
        # LatticeToVector (newfst/lattice-functions.cc:179-217) for all channels at once: the float32 running sums
        # tot += (g + a), lm += g in forward order, by the C entry point (tests/test_capi_symbols.py holds it against
        # wfst_lattice_to_vector and against numpy's sequential cumsum)
        tot_s = np.zeros(cnt, np.float32)
        lm_s = np.zeros(cnt, np.float32)
        nw, nt = np.zeros(cnt, np.int32), np.zeros(cnt, np.int32)
        _check(lib().wfst_lattice_to_vector_batch(_i32(il), _i32(ol), _f32(g), _f32(ac), _i32(nh), cnt, int(cap), _f32(tot_s), _f32(lm_s),
                                                  _i32(nw), _i32(nt)))
        words, tids = np.empty(max(1, int(nw.sum())), np.int32), np.empty(max(1, int(nt.sum())), np.int32)
        woff, toff = np.zeros(cnt + 1, np.int32), np.zeros(cnt + 1, np.int32)
        _check(lib().wfst_lattice_labels_batch(_i32(il), _i32(ol), _i32(nh), cnt, int(cap), _i32(words), _i32(woff), _i32(tids), _i32(toff)))
        return _BestPaths(il, ol, g, ac, nh, tot_s, lm_s, words, woff, tids, toff)

    def stats(self, channel):
        s = (C.c_int64 * 8)()
        _check(lib().wfst_decoder_get_stats(self.h, int(channel), s))
        lat = self.lattice_links > 0
        return dict(frames=s[0], N=s[1], E=s[2], Z=s[3], tokens=s[4], peak_tokens=s[5], records=s[6], links=s[7] if lat else 0,
                    collections=0 if lat else s[7])

    def degraded_frames(self, channel):
        """Frames of the utterance on which the per-frame token limit acted as a max_active (wfst_decoder_get_degraded_frames)."""
        n = C.c_int32(0)
        _check(lib().wfst_decoder_get_degraded_frames(self.h, int(channel), C.byref(n)))
        return int(n.value)

    def lattice_stats(self, channel):
        s = (C.c_int64 * 5)()
        _check(lib().wfst_decoder_get_lattice_stats(self.h, int(channel), s))
        return dict(links_recorded=s[0], walk_links=s[1], walk_tokens=s[2], compaction_scanned=s[3], compaction_moved=s[4])

    def raw_lattice(self, channel, use_final_probs=True):
        """GetRawLattice of a finalized channel (lattice mode).  Returns a dict of numpy arrays, or
        None for the reference's `return false`."""
        ns, na = C.c_int32(0), C.c_int32(0)
        rc = lib().wfst_decoder_get_raw_lattice(self.h, int(channel), int(bool(use_final_probs)), 0, 0, C.byref(ns),
                                                C.byref(na), *([None] * 10))
        if rc != WFST_OK and not (rc == -4 and ns.value > 0):  # -4 with sizes = "give me bigger buffers"
            _check(rc)
        if ns.value == 0:
            return None
        S, A = ns.value, na.value
        fin, fr, gs = (np.zeros(S, np.int32) for _ in range(3))
        co = np.zeros(S, np.float32)
        src, dst, il, ol = (np.zeros(A, np.int32) for _ in range(4))
        gr, ac = np.zeros(A, np.float32), np.zeros(A, np.float32)
        _check(lib().wfst_decoder_get_raw_lattice(self.h, int(channel), int(bool(use_final_probs)), S, A, C.byref(ns),
                                                  C.byref(na), _i32(fin), _i32(fr), _i32(gs), _f32(co), _i32(src), _i32(dst),
                                                  _i32(il), _i32(ol), _f32(gr), _f32(ac)))
        return dict(n_states=S, st_final=fin, st_frame=fr, st_state=gs, st_cost=co, a_src=src, a_dst=dst, a_ilabel=il,
                    a_olabel=ol, a_graph=gr, a_acoustic=ac)

    def prefetch_determinized(self, detached=False):
        """Start the determinization of every finalized channel now, on a side stream (wfst_decoder_prefetch_determinized):
        best_paths() / nbest() run beside it, determinized_lattice(c) finds the work done or waits.  detached=True: init /
        advance / finalize do not wait for it either -- the channels decode their next utterances beside it -- and the lattices
        are fetched with prefetched_lattice(c) (wfst_decoder_prefetch_determinized_detached)."""
        _check((lib().wfst_decoder_prefetch_determinized_detached if detached else lib().wfst_decoder_prefetch_determinized)(self.h))

    def harvest_prefetched(self):
        """Wait for a prefetch in flight and take its lattices over (wfst_decoder_harvest_prefetched)."""
        _check(lib().wfst_decoder_harvest_prefetched(self.h))

    def prefetched_lattice(self, channel):
        """The determinized lattice the last HARVESTED detached prefetch made of `channel`'s utterance (never waits: right behind
        the prefetch call for utterance k it returns utterance k - 1's)."""
        return self._det_fetch(lambda *a: lib().wfst_decoder_get_prefetched_lattice(self.h, int(channel), *a))

    def determinized_lattice(self, channel, use_final_probs=True):
        """GetLattice (GetRawLattice + DeterminizeLatticeWrapper) of a channel: dict of numpy arrays, or None."""
        return self._det_fetch(lambda *a: lib().wfst_decoder_get_determinized_lattice(self.h, int(channel), int(bool(use_final_probs)), *a))

    def prefetched_lattices(self):
        """prefetched_lattice(c) for every channel, in one sweep over one buffer (the per-call work of the binding -- seven pointer
        casts, an allocation, a closure -- is most of what a lattice of a hundred states costs to fetch)."""
        return self._det_fetch_all(lib().wfst_decoder_get_prefetched_lattice, ())

    def determinized_lattices(self, use_final_probs=True):
        """determinized_lattice(c) for every channel (the first call runs the determinizer for all finalized channels)."""
        return self._det_fetch_all(lib().wfst_decoder_get_determinized_lattice, (int(bool(use_final_probs)),))

    def _det_fetch_all(self, f, extra):
        S, A = 1024, 2048
        stride = S + 6 * A
        buf = np.empty((self.n, stride), np.int32)
        base = buf.ctypes.data
        ns, na = C.c_int32(0), C.c_int32(0)
        pns, pna = C.byref(ns), C.byref(na)
        vp = C.c_void_p
        out = [None] * self.n
        for c in range(self.n):
            p = base + 4 * stride * c
            rc = f(self.h, c, *extra, S, A, pns, pna, vp(p), vp(p + 4 * S), vp(p + 4 * (S + A)), vp(p + 4 * (S + 2 * A)), vp(p + 4 * (S + 3 * A)),
                   vp(p + 4 * (S + 4 * A)), vp(p + 4 * (S + 5 * A)))
            if rc != WFST_OK:   # (larger than the common case, or an error: the careful path)
                out[c] = self._det_fetch(lambda *a: f(self.h, c, *extra, *a))
                continue
            s_, a_ = ns.value, na.value
            if s_ == 0:
                continue
            row = buf[c]
            out[c] = dict(n_states=s_, st_final=row[:s_], a_src=row[S:S + a_], a_dst=row[S + A:S + A + a_], a_ilabel=row[S + 2 * A:S + 2 * A + a_],
                          a_olabel=row[S + 3 * A:S + 3 * A + a_], a_graph=row[S + 4 * A:S + 4 * A + a_].view(np.float32),
                          a_acoustic=row[S + 5 * A:S + 5 * A + a_].view(np.float32))
        return out

    def _det_fetch(self, call):
        ns, na = C.c_int32(0), C.c_int32(0)
        S, A = 1024, 2048   # (a determinized lattice is a narrow chain: one call as a rule; a second one with the sizes it returned otherwise)
        for attempt in range(2):
            buf = np.empty(S + 6 * A, np.int32)   # one block per call: final flags | src | dst | ilabel | olabel | graph | acoustic
            p = buf.ctypes.data
            I32, F32 = C.POINTER(C.c_int32), C.POINTER(C.c_float)
            at = lambda k, T: C.cast(p + 4 * (S + k * A), T)
            rc = call(S, A, C.byref(ns), C.byref(na), C.cast(p, I32), at(0, I32), at(1, I32), at(2, I32), at(3, I32), at(4, F32), at(5, F32))
            if rc == -4 and attempt == 0 and (ns.value > S or na.value > A):
                S, A = max(S, ns.value), max(A, na.value)
                continue
            _check(rc)
            break
        if ns.value == 0:
            return None
        s_, a_ = ns.value, na.value
        seg = lambda k: buf[S + k * A: S + k * A + a_]
        return dict(n_states=s_, st_final=buf[:s_], a_src=seg(0), a_dst=seg(1), a_ilabel=seg(2), a_olabel=seg(3),
                    a_graph=seg(4).view(np.float32), a_acoustic=seg(5).view(np.float32))

    def rescored_lattice(self, channel, old_lm, new_lm, use_final_probs=True):
        """GetLattice under --use-second: determinized lattice o old LM (scale -1) o new LM, composed on the device."""
        ns, na = C.c_int32(0), C.c_int32(0)
        rc = lib().wfst_decoder_get_rescored_lattice(self.h, int(channel), int(bool(use_final_probs)), old_lm.h, new_lm.h, 0, 0,
                                                     C.byref(ns), C.byref(na), *([None] * 7))
        if rc != WFST_OK and not (rc == -4 and ns.value > 0):
            _check(rc)
        if ns.value == 0:
            return None
        S, A = ns.value, na.value
        fin = np.zeros(S, np.int32)
        src, dst, il, ol = (np.zeros(A, np.int32) for _ in range(4))
        gr, ac = np.zeros(A, np.float32), np.zeros(A, np.float32)
        _check(lib().wfst_decoder_get_rescored_lattice(self.h, int(channel), int(bool(use_final_probs)), old_lm.h, new_lm.h, S, A,
                                                       C.byref(ns), C.byref(na), _i32(fin), _i32(src), _i32(dst), _i32(il), _i32(ol),
                                                       _f32(gr), _f32(ac)))
        return dict(n_states=S, st_final=fin, a_src=src, a_dst=dst, a_ilabel=il, a_olabel=ol, a_graph=gr, a_acoustic=ac)

    def rescore_lattices(self, old_lm, new_lm, channels=None, use_final_probs=True):
        """The second LM pass of a BATCH of finalized channels (None: all of them) in one launch per stage
        (wfst_decoder_rescore_lattices); rescored_lattice(c, ...) with the same LMs then returns the kept result."""
        ch = None if channels is None else np.ascontiguousarray(channels, np.int32)
        _check(lib().wfst_decoder_rescore_lattices(self.h, _i32(ch) if ch is not None else None, 0 if ch is None else len(ch),
                                                   int(bool(use_final_probs)), old_lm.h, new_lm.h))

    def nbest_paths_batch(self, n, old_lm=None, new_lm=None, channels=None, use_final_probs=True):
        """GetNbest of a BATCH of finalized channels (None: all of them): one launch per stage (wfst_decoder_nbest_paths_batch);
        nbest_paths(c, n, ...) with the same arguments then returns the kept result."""
        ch = None if channels is None else np.ascontiguousarray(channels, np.int32)
        _check(lib().wfst_decoder_nbest_paths_batch(self.h, _i32(ch) if ch is not None else None, 0 if ch is None else len(ch), int(n),
                                                    int(bool(use_final_probs)), old_lm.h if old_lm is not None else None,
                                                    new_lm.h if new_lm is not None else None))

    def nbest_paths(self, channel, n, old_lm=None, new_lm=None, use_final_probs=True):
        """GetNbest as lattices: NShortestPath over the determinized lattice (with LMs: over its second-pass rescoring), on the
        device.  List of paths in ascending cost, each dict(olabel, graph, acoustic: per-arc arrays front to back, the last arc
        the final weight's; tot: the path's cost)."""
        npth, na = C.c_int32(0), C.c_int32(0)
        lm1 = old_lm.h if old_lm is not None else None
        lm2 = new_lm.h if new_lm is not None else None
        K, A = int(n), min(int(n) * 256, 1 << 22)   # (one call in the common case; a second one with the sizes it returned otherwise)
        for attempt in range(2):
            off = np.zeros(K + 1, np.int32)
            tot = np.zeros(K, np.float32)
            ol = np.zeros(A, np.int32)
            gr, ac = np.zeros(A, np.float32), np.zeros(A, np.float32)
            rc = lib().wfst_decoder_get_nbest_paths(self.h, int(channel), int(n), int(bool(use_final_probs)), lm1, lm2, K, A,
                                                    C.byref(npth), C.byref(na), _i32(off), _f32(tot), _i32(ol), _f32(gr), _f32(ac))
            if rc == -4 and attempt == 0 and (npth.value > K or na.value > A):
                K, A = max(K, npth.value), max(A, na.value)
                continue
            _check(rc)
            break
        K = npth.value
        return [dict(olabel=ol[off[i]:off[i + 1]].copy(), graph=gr[off[i]:off[i + 1]].copy(), acoustic=ac[off[i]:off[i + 1]].copy(),
                     tot=float(tot[i])) for i in range(K)]

    def raw_lattices(self, channels=None, use_final_probs=True, threads=0):
        """GetRawLattice of many finalized channels.  The first call fetches the pruned lattices of all
        finalized channels from the device in one sweep; the per-lattice host work (numbering, arc
        order) then runs on `threads` host threads (0 = min(16, cpu count))."""
        import os
        from concurrent.futures import ThreadPoolExecutor

        ch = list(range(self.n)) if channels is None else [int(c) for c in channels]
        if not ch:
            return []
        first = self.raw_lattice(ch[0], use_final_probs)  # fills the host-side cache (single threaded)
        nt = threads or min(16, os.cpu_count() or 1)
        if nt <= 1 or len(ch) == 1:
            return [first] + [self.raw_lattice(c, use_final_probs) for c in ch[1:]]
        with ThreadPoolExecutor(max_workers=nt) as ex:
            rest = list(ex.map(lambda c: self.raw_lattice(c, use_final_probs), ch[1:]))
        return [first] + rest

    def nbest(self, n, channels=None, max_words=256):
        """GetNbest + LatticeToVector of finalized channels (lattice mode): per channel a list of
        dicts {words, tot_score, lm_score}, cheapest first."""
        ch = list(range(self.n)) if channels is None else [int(c) for c in channels]
        cnt = len(ch)
        arr = np.asarray(ch, np.int32)
        npaths = np.zeros(cnt, np.int32)
        nw = np.zeros((cnt, n), np.int32)
        words = np.zeros((cnt, n, max_words), np.int32)
        tot = np.zeros((cnt, n), np.float32)
        lm = np.zeros((cnt, n), np.float32)
        _check(lib().wfst_decoder_get_nbest(self.h, _i32(arr), cnt, int(n), int(max_words), _i32(npaths), _i32(nw), _i32(words),
                                            _f32(tot), _f32(lm)))
        return _LazyList(cnt, lambda i: [dict(words=words[i, k, : min(nw[i, k], max_words)].copy(), n_words=int(nw[i, k]), tot_score=float(tot[i, k]),
                                              lm_score=float(lm[i, k])) for k in range(npaths[i])])

    def path_flags(self):
        """Which kernel paths the decoder runs (wfst_decoder_get_path_flags)."""
        f = (C.c_int32 * 8)()
        _check(lib().wfst_decoder_get_path_flags(self.h, f))
        return dict(zip(("staged", "two_launch", "gc_stride", "degcode", "ll_row", "best_exp", "soft_limit", "channel_groups"), [int(x) for x in f]))

    @property
    def n_groups(self):
        return int(lib().wfst_decoder_channel_groups(self.h))

    def set_profiling(self, on):
        _check(lib().wfst_decoder_set_profiling(self.h, int(bool(on))))

    def profile(self):
        ms = (C.c_double * 3)()
        n = (C.c_int64 * 3)()
        _check(lib().wfst_decoder_get_profile(self.h, ms, n))
        busy = (C.c_double * 3)()
        _check(lib().wfst_decoder_get_profile_busy(self.h, busy))
        return dict(expand_ms=ms[0], expand_launches=n[0], insert_ms=ms[1], insert_launches=n[1],
                    closure_ms=ms[2], closure_launches=n[2], expand_busy_ms=busy[0], insert_busy_ms=busy[1], closure_busy_ms=busy[2])

    def frontier(self, channel, cap=1 << 20):
        st = np.zeros(cap, np.int32)
        co = np.zeros(cap, np.float32)
        n = lib().wfst_decoder_get_frontier(self.h, int(channel), int(cap), _i32(st), _f32(co))
        if n < 0:
            _check(n)
        k = min(n, cap)
        return st[:k].copy(), co[:k].copy()
