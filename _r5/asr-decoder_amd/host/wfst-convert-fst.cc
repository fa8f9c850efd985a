// wfst-convert-fst IN OUT -- the reference's `convert_fst det_min_clg.fst use_clg.fst`
// (fst_format_convert_tool/convert_fst.c:5-27, README.txt): read an OpenFst vector or const fst
// (StdArc) and write the decoder's flat graph format.  Host only; a thin shell over
// wfst_graph_convert_file.  (Unlike the reference tool it replaces OUT instead of appending to it,
// and it does not print every arc.)
#include <iostream>

#include "../../include/wfst_decoder.h"

int main(int argc, char **argv) {
  if (argc != 3) {
    std::cerr << "input error!\nplease input " << argv[0] << " fst out\n";
    return 1;
  }
  if (wfst_graph_convert_file(argv[1], argv[2]) != WFST_OK) {
    std::cerr << "convert error: " << wfst_last_error() << "\n";
    return 1;
  }
  return 0;
}
