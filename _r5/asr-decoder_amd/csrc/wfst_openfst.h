// Host-side graph ingestion: the reference's flat format and OpenFst binary files (vector / const,
// StdArc) -> the arrays wfst_graph_from_arrays takes.  No device code, no HIP calls.
#ifndef WFST_OPENFST_H_
#define WFST_OPENFST_H_

#include <string>
#include <vector>

#include "../../include/wfst_decoder.h"

namespace wfst {

struct HostGraph {
  int32_t start = 0, final_state = 0;
  int32_t total_niepsilons = 0, total_noepsilons = 0;
  std::vector<wfst_state_info> states;  // includes the super-final state (last)
  std::vector<wfst_arc> arcs;
};

// Reads `path` in any of the three formats (detected by the OpenFst magic number / fst type).
// Returns WFST_OK, WFST_E_IO (unreadable / truncated) or WFST_E_FORMAT (not a supported FST); the
// message goes to *err.
int read_graph_file(const char *path, HostGraph *g, std::string *err);
// Writes the reference's flat format (what Fst::ReadFst reads).
int write_flat_graph(const char *path, const HostGraph &g, std::string *err);

}  // namespace wfst
#endif
