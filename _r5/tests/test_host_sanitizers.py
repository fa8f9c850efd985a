"""Host-side parsers under AddressSanitizer + UBSan (CPU build only; GPU sanitizers are not available
on this pool): the graph-file reader (flat / OpenFst vector / OpenFst const) and the lattice-file
reader of the C++ mirror, on valid files, every kind of truncation and a few hundred random
corruptions.  A damaged file must be refused or read as garbage VALUES -- never crash, read out of
bounds or allocate without bound."""
import os
import subprocess

import numpy as np
import pytest

from golden_util import GOLDEN_DIR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=512", UBSAN_OPTIONS="halt_on_error=1")


def _mutations(data, rng, n):
    yield data
    for cut in sorted(set([0, 1, 3, 4, 8, 11, 23, 24, 40, 57, 58, 59, 60, 61, 80, len(data) // 3, len(data) // 2, len(data) - 17, len(data) - 1])):
        if 0 <= cut < len(data):
            yield data[:cut]
    for _ in range(n):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 6))):
            pos = int(rng.integers(0, min(len(b), 4096) if rng.random() < 0.7 else len(b)))
            b[pos] = int(rng.integers(0, 256))
        yield bytes(b)


@pytest.fixture(scope="module")
def asan_ingest(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("asan") / "asan_ingest")
    subprocess.check_call(["g++", "-std=c++17"] + SAN + [os.path.join(ROOT, "tests", "asan_ingest_main.cc"),
                                                         os.path.join(ROOT, "asr-decoder_amd", "csrc", "wfst_openfst.cc"), "-o", exe])
    return exe


@pytest.mark.parametrize("kind", ["vector_fst", "const_fst", "ref_flat_from_vector"])
def test_graph_reader_survives_damaged_files(kind, asan_ingest, tmp_path):
    z = np.load(os.path.join(GOLDEN_DIR, "openfst.npz"))
    data = bytes(z[kind])
    rng = np.random.default_rng(5)
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    n_ok = n_refused = 0
    for m in _mutations(data, rng, 150):
        with open(src, "wb") as f:
            f.write(m)
        p = subprocess.run([asan_ingest, src, dst], capture_output=True, text=True, env=ENV, timeout=60)
        assert p.returncode in (0, 12, 13, 16), "sanitizer report or crash (exit %d):\\n%s" % (p.returncode, p.stderr[-3000:])
        n_ok += p.returncode == 0
        n_refused += p.returncode != 0
    assert n_ok >= 1 and n_refused >= 10


@pytest.fixture(scope="module")
def asan_lattice_copy(tmp_path_factory):
    importlib = __import__("importlib")
    importlib.import_module("asr-decoder_amd.build").build()
    exe = str(tmp_path_factory.mktemp("asanl") / "asan_lattice_copy")
    host = os.path.join(ROOT, "asr-decoder_amd", "host")
    lib = os.path.join(ROOT, "asr-decoder_amd", "lib")
    subprocess.check_call(["g++", "-std=c++14", "-pthread"] + SAN + [os.path.join(host, "wfst-lattice-copy.cc"), os.path.join(host, "wfst-host.cc"),
                                                                     "-o", exe, "-L" + lib, "-lwfstdec", "-Wl,-rpath," + lib])
    return exe


def test_lattice_reader_survives_damaged_files(asan_lattice_copy, tmp_path):
    z = np.load(os.path.join(GOLDEN_DIR, "lattice_file.npz"))
    data = bytes(z["data"])
    rng = np.random.default_rng(6)
    src, dst = str(tmp_path / "in.lat"), str(tmp_path / "out.lat")
    n = 0
    for m in _mutations(data, rng, 150):
        with open(src, "wb") as f:
            f.write(m)
        p = subprocess.run([asan_lattice_copy, src, dst], capture_output=True, text=True, env=ENV, timeout=60)
        assert p.returncode in (0, 1), "sanitizer report or crash (exit %d):\\n%s" % (p.returncode, p.stderr[-3000:])
        n += 1
    assert n > 150
