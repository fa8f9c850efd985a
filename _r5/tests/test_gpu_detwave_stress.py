"""-m gpu: the one-wave-per-lattice determinizer (asr-decoder_amd/csrc/wfst_determinize_wave.h) against the host build of the shared
algorithm (wfst_determinize.h, itself held against the reference's determinizer by tests/test_determinize_host.py), lattice by
lattice, through the development harness tools/det_bench.hip -- built THREE times: with the product's sizes, and with sizes so small
that every fall-back of the wave runs on ordinary lattices (closures that outgrow the LDS buffers and are run again on one lane,
queue entries with more epsilon arcs than a window prices, strings longer than a lane's label buffer, windows whose offers meet).
Every variant must give the host build's lattice: same states, same arcs with bit-identical weights."""
import os
import subprocess

import numpy as np
import pytest

import pyoracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _write_bin(path, L):
    with open(path, "wb") as f:
        np.asarray([L.n_states, len(L.a_src)], np.int32).tofile(f)
        np.asarray(L.st_final, np.int32).tofile(f)
        rec = np.zeros(len(L.a_src), dtype=[("src", "<i4"), ("dst", "<i4"), ("il", "<i4"), ("ol", "<i4"), ("g", "<f4"), ("ac", "<f4")])
        rec["src"], rec["dst"], rec["il"], rec["ol"], rec["g"], rec["ac"] = L.a_src, L.a_dst, L.a_il, L.a_ol, L.a_graph, L.a_ac
        rec.tofile(f)


def test_wave_determinizer_equals_the_host_build_at_every_size(synth, oracle, tmp_path):
    # raw lattices of a few hundred to a few thousand states (the CPU restatement, order-free mode: what the device's decoder leaves)
    files = []
    oracle.set_order_free(True)
    try:
        # (graph sizes / beams of tests/test_determinize_host.py: lattices the subset construction finishes on -- wider lattice beams on
        # these small dense graphs are the adversarial case where it is exponential, for the reference as much as here)
        for gi, (S, T, beam, lb) in enumerate([(6000, 80, 11.0, 4.0), (3000, 60, 12.0, 5.0), (600, 40, 13.0, 7.0)]):
            g = synth.make_hclg_like(S, seed=31 + (3 - gi), n_tid=600, n_words=500)
            m = synth.default_tid2pdf(600)
            gp = str(tmp_path / ("g%d.bin" % gi))
            g.write(gp)
            h = oracle.load_graph(gp)
            cd = dict(beam=beam, max_active=1000000, min_active=0, lattice_beam=lb)
            for u in range(3):
                ll = synth.make_loglikes(g, T, 300, m, seed=200 * (3 - gi) + u, mu=-2.2)[0]
                O = pyoracle.oracle_raw_lattice(oracle, h, pyoracle.Config(**cd), ll, m)
                if O is None or not O.ok or O.n_states < 20:
                    continue
                p = str(tmp_path / ("lat_%d_%d.bin" % (gi, u)))
                _write_bin(p, O)
                files.append(p)
            oracle.free_graph(h)
    finally:
        oracle.set_order_free(False)
    assert len(files) >= 6
    src = os.path.join(ROOT, "tools", "det_bench.hip")
    inc = os.path.join(ROOT, "asr-decoder_amd", "csrc")
    for tag, defs in (("product", []), ("tiny", ["-DDETW_CUR=32", "-DDETW_ARCS=1", "-DDETW_LABS=2"]), ("small", ["-DDETW_CUR=256", "-DDETW_ARCS=2", "-DDETW_LABS=8"])):
        exe = str(tmp_path / ("det_bench_" + tag))
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-I", inc, "-o", exe, src] + defs)
        p = subprocess.run([exe, "--variant", "1", "--reps", "1"] + files, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "all equal to the host build" in p.stdout, (tag, p.stdout[-1500:], p.stderr[-500:])
        assert p.stdout.count(" OK ") == len(files), (tag, p.stdout[-1500:])
