"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/wfst_decoder.h
declares, and refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    p = importlib.import_module("asr-decoder_amd")
    p.build.build()
    return p


def declared_functions():
    src = open(os.path.join(ROOT, "include", "wfst_decoder.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wfst_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(pkg):
    assert declared_functions() == sorted(pkg.wfstdec.SYMBOLS)


def test_library_exports_every_declared_symbol(pkg):
    lib = ctypes.CDLL(pkg.wfstdec.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_config_default_matches_reference_defaults(pkg):
    # lattice-faster-decoder-conf.h:35-44
    c = pkg.wfstdec.Config(0, 0, 0, 0, 0, 0, 0, 0)
    pkg.wfstdec.lib().wfst_config_default(ctypes.byref(c))
    assert (c.beam, c.max_active, c.min_active, c.lattice_beam, c.prune_interval) == (16.0, 2147483647, 200, 10.0, 25)
    assert (c.beam_delta, c.hash_ratio) == (0.5, 2.0) and abs(c.prune_scale - 0.1) < 1e-7


def test_lattice_to_vector_is_host_only(pkg):
    import numpy as np

    il = np.array([0, 5, 0, 7], np.int32)
    ol = np.array([0, 0, 9, 3], np.int32)
    g = np.array([0, 0.5, 0.25, 1.0], np.float32)
    a = np.array([0, 2.0, 0, 1.5], np.float32)
    words = np.zeros(4, np.int32)
    tids = np.zeros(4, np.int32)
    nw, nt = ctypes.c_int32(), ctypes.c_int32()
    tot, lm = ctypes.c_float(), ctypes.c_float()
    f = lambda x, t: x.ctypes.data_as(ctypes.POINTER(t))
    rc = pkg.wfstdec.lib().wfst_lattice_to_vector(f(il, ctypes.c_int32), f(ol, ctypes.c_int32), f(g, ctypes.c_float),
                                                  f(a, ctypes.c_float), 4, f(words, ctypes.c_int32), 4, ctypes.byref(nw),
                                                  f(tids, ctypes.c_int32), 4, ctypes.byref(nt), ctypes.byref(tot), ctypes.byref(lm))
    assert rc == 0 and list(words[: nw.value]) == [9, 3] and list(tids[: nt.value]) == [5, 7]
    assert tot.value == 5.25 and lm.value == 1.75


def test_no_gpu_means_loud_failure(pkg):
    if pkg.wfstdec.device_count() > 0:
        pytest.skip("a GPU is visible here")
    s = pkg.synth.make_hclg_like(50, seed=1, n_tid=20, n_words=5)
    with pytest.raises(pkg.wfstdec.WfstError) as ei:
        pkg.wfstdec.Graph.from_arrays(s.start, s.final_state, s.state_info, s.arcs)
    assert ei.value.code == -3


def test_host_mirror_builds_and_cli_prints_usage(pkg):
    import subprocess

    host = os.path.join(ROOT, "asr-decoder_amd", "host")
    subprocess.check_call(["make", "-s", "-C", host])
    p = subprocess.run([os.path.join(host, "wfst-decode")], capture_output=True, text=True)
    assert p.returncode == 1 and "usage: wfst-decode" in p.stderr
    hdr = open(os.path.join(host, "wfst-host.h")).read()
    for name in ("class DecoderItf", "class DecodableInterface", "struct LatticeFasterDecoderConfig", "class Fst",
                 "class Lattice", "bool LatticeToVector", "class GpuLatticeDecoder : public DecoderItf"):
        assert name in hdr


def test_vectorised_lattice_to_vector_equals_c_entry_point(pkg):
    """wfstdec.BatchDecoder.best_paths sums scores with numpy cumsum; it must be bit-identical to
    the sequential float32 accumulation of wfst_lattice_to_vector (the reference's LatticeToVector)."""
    import numpy as np

    rng = np.random.default_rng(3)
    n = 700
    g = rng.uniform(0, 4, n).astype(np.float32)
    a = rng.uniform(0, 9, n).astype(np.float32)
    il = rng.integers(0, 3, n).astype(np.int32)
    ol = rng.integers(0, 2, n).astype(np.int32)
    words = np.zeros(n, np.int32)
    tids = np.zeros(n, np.int32)
    nw, nt = ctypes.c_int32(), ctypes.c_int32()
    tot, lm = ctypes.c_float(), ctypes.c_float()
    f = lambda x, t: x.ctypes.data_as(ctypes.POINTER(t))
    pkg.wfstdec.lib().wfst_lattice_to_vector(f(il, ctypes.c_int32), f(ol, ctypes.c_int32), f(g, ctypes.c_float), f(a, ctypes.c_float),
                                             n, f(words, ctypes.c_int32), n, ctypes.byref(nw), f(tids, ctypes.c_int32), n,
                                             ctypes.byref(nt), ctypes.byref(tot), ctypes.byref(lm))
    t2 = np.cumsum((g + a).astype(np.float32), dtype=np.float32)[-1]
    l2 = np.cumsum(g, dtype=np.float32)[-1]
    assert np.float32(tot.value).tobytes() == np.float32(t2).tobytes()
    assert np.float32(lm.value).tobytes() == np.float32(l2).tobytes()
    assert list(words[: nw.value]) == list(ol[ol != 0]) and list(tids[: nt.value]) == list(il[il != 0])
    # the batch form (what wfstdec.BatchDecoder.best_paths calls): three hop lists of different lengths in [3][cap] arrays
    cap = n + 5
    nh = np.array([n, 0, 123], np.int32)
    pad = lambda x: np.stack([np.concatenate([x, np.full(5, 7, x.dtype)])] * 3)
    IL, OL, GG, AA = pad(il), pad(ol), pad(g), pad(a)
    ts, ls = np.zeros(3, np.float32), np.zeros(3, np.float32)
    cw, ct = np.zeros(3, np.int32), np.zeros(3, np.int32)
    rc = pkg.wfstdec.lib().wfst_lattice_to_vector_batch(f(IL, ctypes.c_int32), f(OL, ctypes.c_int32), f(GG, ctypes.c_float), f(AA, ctypes.c_float),
                                                        f(nh, ctypes.c_int32), 3, cap, f(ts, ctypes.c_float), f(ls, ctypes.c_float),
                                                        f(cw, ctypes.c_int32), f(ct, ctypes.c_int32))
    assert rc == 0 and ts[0].tobytes() == np.float32(t2).tobytes() and ls[0].tobytes() == np.float32(l2).tobytes()
    assert ts[1] == 0 and ls[1] == 0 and [cw[0], ct[0], cw[1], ct[1]] == [int((ol != 0).sum()), int((il != 0).sum()), 0, 0]
    assert ts[2].tobytes() == np.cumsum((g[:123] + a[:123]).astype(np.float32), dtype=np.float32)[-1].tobytes()
    # ... and its label half: every path's words / transition-ids packed path after path (best_paths hands out slices of them)
    W, T = np.zeros(int(cw.sum()), np.int32), np.zeros(int(ct.sum()), np.int32)
    wo, to = np.zeros(4, np.int32), np.zeros(4, np.int32)
    rc = pkg.wfstdec.lib().wfst_lattice_labels_batch(f(IL, ctypes.c_int32), f(OL, ctypes.c_int32), f(nh, ctypes.c_int32), 3, cap,
                                                     f(W, ctypes.c_int32), f(wo, ctypes.c_int32), f(T, ctypes.c_int32), f(to, ctypes.c_int32))
    assert rc == 0 and list(wo) == [0, int(cw[0]), int(cw[0]), int(cw[0] + cw[2])] and list(to) == [0, int(ct[0]), int(ct[0]), int(ct[0] + ct[2])]
    assert list(W[: wo[1]]) == list(ol[ol != 0]) and list(T[: to[1]]) == list(il[il != 0])
    assert list(W[wo[2]:]) == list(ol[:123][ol[:123] != 0]) and list(T[to[2]:]) == list(il[:123][il[:123] != 0])


def test_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/wfst_decoder.h must compile as C99 (no C++ in it), and a
    C program that uses it must link against the library."""
    import subprocess

    src = tmp_path / "use.c"
    src.write_text(
        '#include "wfst_decoder.h"\n'
        "int main(void) {\n"
        "  wfst_config c; wfst_limits l = {0, 0, 0, 0}; wfst_arc a = {0, 0, 0.0f, 0}; wfst_state_info s = {0, 0, 0};\n"
        "  wfst_config_default(&c);\n"
        "  (void)l; (void)a; (void)s;\n"
        "  return (c.beam > 0.0f && wfst_last_error() != 0) ? 0 : 1;\n"
        "}\n")
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "asr-decoder_amd", "lib")
    exe = str(tmp_path / "use")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + inc, str(src), "-o", exe,
                           "-L" + libdir, "-lwfstdec", "-Wl,-rpath," + libdir])
    assert subprocess.run([exe]).returncode == 0


def test_options_defaults_and_range_check(pkg):
    o = pkg.wfstdec.Options()
    assert (o.channel_groups, o.use_hip_graph, o.log2_partitions, o.log2_lds_slots) == (0, 1, -1, 12)
    assert (o.joint_max, o.expand_workgroups, o.insert_workgroups, o.upload_slice_frames, o.tile_tokens, o.debug) == (1536, 0, 768, 48, 256, 0)
    go = pkg.wfstdec.GraphOptions()
    assert (go.row_align_slots, go.flatten_closures) == (8, 1)
    with pytest.raises(TypeError):
        pkg.wfstdec.Options(no_such_field=1)


def test_library_reads_no_environment_variables(pkg):
    """scheduling knobs are wfst_options fields, not process environment"""
    import subprocess

    out = subprocess.run(["nm", "-D", "--undefined-only", pkg.wfstdec.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in out


def test_host_header_kaldi_decodable_branch(pkg, tmp_path):
    """-DWFST_KALDI_DECODABLE: AmInterface is kaldi::DecodableInterface, as under the reference's -DKALDI
    (src/itf/decodable-itf.h:55-62).  Compiled against a fixture with Kaldi's interface (Kaldi itself is
    not in this image): a Kaldi decodable must be accepted by DecoderItf::AdvanceDecoding as it is."""
    import subprocess

    kal = tmp_path / "kaldi" / "itf"
    kal.mkdir(parents=True)
    (kal / "decodable-itf.h").write_text(
        "#pragma once\nnamespace kaldi { typedef float BaseFloat; typedef int int32;\n"
        "class DecodableInterface { public:\n"
        "  virtual BaseFloat LogLikelihood(int32 frame, int32 index) = 0;\n"
        "  virtual bool IsLastFrame(int32 frame) const = 0;\n"
        "  virtual int32 NumFramesReady() const { return -1; }\n"
        "  virtual int32 NumIndices() const = 0;\n"
        "  virtual ~DecodableInterface() {}\n};\n}\n")
    (tmp_path / "t.cc").write_text(
        '#include "wfst-host.h"\n#include <type_traits>\n'
        "static_assert(std::is_same<datemoon::AmInterface, kaldi::DecodableInterface>::value, \"AmInterface\");\n"
        "struct D : kaldi::DecodableInterface { float LogLikelihood(int, int) override { return 0; }\n"
        "  bool IsLastFrame(int) const override { return true; } int NumFramesReady() const override { return 0; }\n"
        "  int NumIndices() const override { return 1; } };\n"
        "void f(datemoon::DecoderItf *d) { D x; d->AdvanceDecoding(&x); }\n")
    host = os.path.join(ROOT, "asr-decoder_amd", "host")
    for extra in (["-DWFST_KALDI_DECODABLE"], ["-DKALDI"]):
        subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Werror", "-I" + host, "-I" + str(tmp_path / "kaldi")] + extra +
                              [str(tmp_path / "t.cc")])
        subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-I" + host, "-I" + str(tmp_path / "kaldi")] + extra +
                              [os.path.join(host, "wfst-host.cc")])
