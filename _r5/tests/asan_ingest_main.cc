// Test helper (CPU only): the graph-file reader of the product (asr-decoder_amd/csrc/wfst_openfst.cc)
// built with -fsanitize=address,undefined and run on valid, truncated and corrupted files.
// usage: asan_ingest IN OUT  -> exit code 0 (converted), 10 + (-rc) for a refused file
#include <cstdio>
#include <string>

#include "../asr-decoder_amd/csrc/wfst_openfst.h"

int main(int argc, char **argv) {
  if (argc != 3) return 2;
  wfst::HostGraph g;
  std::string err;
  int rc = wfst::read_graph_file(argv[1], &g, &err);
  if (rc != WFST_OK) {
    fprintf(stderr, "refused: %s\n", err.c_str());
    return 10 - rc;
  }
  // touch everything the uploader would touch
  long long sum = 0;
  size_t off = 0;
  for (size_t s = 0; s < g.states.size(); ++s) {
    if (off + g.states[s].num_arcs > g.arcs.size()) { fprintf(stderr, "refused: arc counts exceed the arc array\n"); return 16; }
    for (unsigned i = 0; i < g.states[s].num_arcs; ++i) sum += g.arcs[off + i].ilabel + g.arcs[off + i].nextstate;
    off += g.states[s].num_arcs;
  }
  rc = wfst::write_flat_graph(argv[2], g, &err);
  return rc == WFST_OK ? (sum == -1 ? 3 : 0) : 10 - rc;
}
