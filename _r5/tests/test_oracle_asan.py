"""The C oracle (the checker everything else is held to) under AddressSanitizer + UBSan: every golden
case -- best paths, traces, streaming chunks, raw lattices in both modes -- replayed in a subprocess
that loads an instrumented build of oracle/wfst_oracle.c.  CPU only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "oracle"), os.path.join(%(root)r, "tests")]
import pyoracle
pyoracle.OracleDecoder.SO = os.environ["ASAN_ORACLE_SO"]
from golden_util import Golden, check_result
orc = pyoracle.OracleDecoder()
tmp = os.environ["ASAN_TMP"]
n = 0
for name in ("hclg600", "eps_chains", "quirk_parallel_arcs", "no_final", "dead_end"):
    g = Golden(name)
    h = orc.load_graph(g.write_graph(os.path.join(tmp, name + ".bin")))
    for k, cd, md, ui in g.cases():
        trace = md.pop("trace", False)
        check_result(orc.decode(h, pyoracle.Config(**cd), g.utts[ui], g.tid2pdf, trace=trace, **md), g.expected(k), name)
        n += 1
    orc.free_graph(h)
g = Golden("lattice_hclg600")
h = orc.load_graph(g.write_graph(os.path.join(tmp, "lat.bin")))
for k, cd, md, ui in g.cases():
    for order_free in (False, True):
        orc.set_order_free(order_free)
        pyoracle.oracle_raw_lattice(orc, h, pyoracle.Config(**cd), g.utts[ui], g.tid2pdf, **md)
        n += 1
orc.set_order_free(False)
orc.free_graph(h)
print("asan oracle ok", n)
'''


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    so = str(tmp_path / "libwfst_oracle_asan.so")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-ffp-contract=off", "-msse2", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "oracle", "wfst_oracle.c"), "-lm"])
    pre = []
    for lib in ("libasan.so", "libubsan.so"):
        p = subprocess.check_output(["gcc", "-print-file-name=" + lib], text=True).strip()
        if not os.path.isabs(p):
            pytest.skip("sanitizer runtime %s not found" % lib)
        pre.append(p)
    env = dict(os.environ, ASAN_ORACLE_SO=so, ASAN_TMP=str(tmp_path), LD_PRELOAD=":".join(pre), ASAN_OPTIONS="detect_leaks=0")
    p = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT)], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "asan oracle ok" in p.stdout, (p.stdout[-500:], p.stderr[-3000:])
