#!/bin/bash
# device timeline of the drop-in shape (wfst-decode --threads=64 --pool=64 --chunk=25 --pull): rocprofv3 kernel + memory-copy trace
# of the CLI itself, summarised by tools/dropin_timeline.py.  Through gpurun from the repo root:  bash tools/dropin_trace.sh [cli args]
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
export TMPDIR=/tmp
bash tools/dropin_probe.sh >/dev/null 2>&1   # (writes /tmp/dp/* and the graph; runs nothing without arguments)
ARGS="${*:---threads=64 --pool=64 --chunk=25 --pull --repeat=4}"
OUT="$R/gpurun_out/dropin_trace"; rm -rf "$OUT"; mkdir -p "$OUT"
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT" -o trace -- \
  "$R/asr-decoder_amd/host/wfst-decode" --tid2pdf=/tmp/dp/tid2pdf.bin --max-frames=302 --max-tokens=65536 --arena-tokens=4170000 $ARGS \
  /tmp/dp/decoder.conf /tmp/wfst_bench_graph_2850000.bin /tmp/dp/ll.bin 2> "$OUT/stderr.log" > /dev/null
grep -E "LOG pool|LOG Time|ERROR|LOG Frames" "$OUT/stderr.log"
python tools/dropin_timeline.py "$OUT"
