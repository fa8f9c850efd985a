// wfst-lattice-copy IN OUT: read every lattice of IN (reference on-disk format, Lattice::Read) and
// write it to OUT (Lattice::Write).  Needs no GPU; used to check the host mirror's lattice I/O
// byte for byte against files the reference wrote (tests/test_host_lattice_io.py).
#include <cstdio>
#include <iostream>

#include "wfst-host.h"

int main(int argc, char **argv) {
  if (argc != 3) {
    std::cerr << "usage: wfst-lattice-copy IN OUT\n";
    return 1;
  }
  FILE *in = fopen(argv[1], "rb");
  if (!in) {
    std::cerr << "Open " << argv[1] << " failed.\n";
    return 1;
  }
  remove(argv[2]);
  int n = 0;
  long long states = 0;
  datemoon::Lattice lat;
  while (lat.Read(in)) {
    if (!lat.Write(std::string(argv[2]))) return 1;
    states += lat.NumStates();
    ++n;
  }
  fclose(in);
  std::cout << n << " lattices, " << states << " states\n";
  return 0;
}
