#!/bin/bash
# the drop-in leg's CLI runs by hand (timing experiments): needs /tmp/wfst_bench_graph_2850000.bin (any bench.py run leaves it)
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
python - <<'P'
import importlib, os, struct, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
pkg = importlib.import_module("asr-decoder_amd"); synth = pkg.synth
import argparse
gp = "/tmp/wfst_bench_graph_2850000.bin"
g = synth.Graph.read(gp) if os.path.exists(gp) else None
if g is None:
    g = synth.make_hclg_like(2850000, seed=7, n_tid=6000); g.write(gp)
m = synth.default_tid2pdf(6000)
os.makedirs("/tmp/dp", exist_ok=True)
np.asarray(m, "<i4").tofile("/tmp/dp/tid2pdf.bin")
open("/tmp/dp/decoder.conf", "w").write("--beam=13\n--max-active=1000000\n--min-active=0\n--lattice-beam=7\n--prune-interval=25\n--beam-delta=0.5\n")
with open("/tmp/dp/ll.bin", "wb") as f:
    for i in range(128):
        x = synth.make_loglikes_multi(g, 300, 3000, m, seed=i, n_paths=272, mu=-4.0, sigma=1.0, jitter=0.5, ac_lo=0.5)[0]   # bench.py make_utts
        key = ("utt%04d" % i).encode()
        f.write(struct.pack("<i", len(key)) + key + struct.pack("<ii", 300, 3000)); f.write(np.ascontiguousarray(x, "<f4").tobytes())
P
for args in "$@"; do
  echo "=== $args"
  timeout 90 asr-decoder_amd/host/wfst-decode --tid2pdf=/tmp/dp/tid2pdf.bin --max-frames=302 --max-tokens=65536 --arena-tokens=4170000 $args /tmp/dp/decoder.conf /tmp/wfst_bench_graph_2850000.bin /tmp/dp/ll.bin 2>&1 >/dev/null | grep -E "LOG pool|LOG Time|ERROR|LOG Done|LOG Frames"
done
