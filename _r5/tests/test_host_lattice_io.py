"""The host mirror's Lattice::Read / Lattice::Write (asr-decoder_amd/host/wfst-host.cc) against a
file the REFERENCE wrote with its own Lattice::Write (tests/golden/lattice_file.npz, reference
newfst/lattice-fst.cc:38-64): reading every lattice and writing it back must reproduce the file
byte for byte.  CPU only (the copy tool needs no GPU)."""
import os
import subprocess

import numpy as np
import pytest

import pyoracle
from golden_util import GOLDEN_DIR, Golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "asr-decoder_amd", "host")


@pytest.fixture(scope="module")
def copy_tool():
    importlib = __import__("importlib")
    importlib.import_module("asr-decoder_amd.build").build()  # libwfstdec.so (cross-compiles without a GPU)
    subprocess.check_call(["make", "-s", "-C", HOST])
    return os.path.join(HOST, "wfst-lattice-copy")


def test_lattice_file_roundtrip_is_byte_identical(copy_tool, tmp_path):
    z = np.load(os.path.join(GOLDEN_DIR, "lattice_file.npz"))
    data = bytes(z["data"])
    src, dst = str(tmp_path / "in.lat"), str(tmp_path / "out.lat")
    with open(src, "wb") as f:
        f.write(data)
    out = subprocess.run([copy_tool, src, dst], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split()[0] == str(int(z["n_lattices"]))
    with open(dst, "rb") as f:
        assert f.read() == data


def test_lattice_file_holds_the_golden_lattices(oracle, tmp_path):
    """The file's three lattices are the cfg-1 finalized lattices of lattice_hclg600 (same generator
    inputs): the numpy parser of the format and the oracle agree with what the reference wrote."""
    z = np.load(os.path.join(GOLDEN_DIR, "lattice_file.npz"))
    lats = pyoracle.parse_lattice_file(bytes(z["data"]))
    g = Golden("lattice_hclg600")
    assert len(lats) == int(z["n_lattices"]) == len(g.utts)
    want = {ui: k for k, cd, md, ui in g.cases()
            if g.meta["cases"][k]["cfg"] == int(z["cfg"]) and md["finalize"] and md["use_final_probs"]}
    for ui, L in enumerate(lats):
        k = want[ui]
        ok, ns, nf, na, start = (int(x) for x in g.z["c%d_counts" % k])
        assert (L.n_states, int(L.st_final.sum()), len(L.a_src), L.start) == (ns, nf, na, start)
        assert np.array_equal(L.arc_multiset(), g.z["c%d_arcs" % k])


def test_truncated_lattice_file_is_rejected(copy_tool, tmp_path):
    z = np.load(os.path.join(GOLDEN_DIR, "lattice_file.npz"))
    data = bytes(z["data"])
    src, dst = str(tmp_path / "in.lat"), str(tmp_path / "out.lat")
    with open(src, "wb") as f:
        f.write(data[: len(data) // 2 + 3])
    out = subprocess.run([copy_tool, src, dst], capture_output=True, text=True, timeout=60)
    assert "error" in out.stderr.lower()
    assert int(out.stdout.split()[0]) < int(z["n_lattices"])
