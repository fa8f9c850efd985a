// Graph ingestion.  Three on-disk formats end up as the same arrays:
//
//  * the reference's flat format (Fst::ReadFst, reference newfst/optimize-fst.h:226-280):
//    6 x int32 {start, final_state, total_states, total_arcs, total_niepsilons, total_noepsilons},
//    StateInfo x S, StdArc x A;
//  * an OpenFst VECTOR fst, StdArc -- what the reference's converter takes
//    (fst_format_convert_tool/read_fst.c:11-187 + write_fst.c:5-62): header, then per state
//    {float final weight, int64 number of arcs, arcs};
//  * an OpenFst CONST fst, StdArc -- what the reference's service loads directly
//    (newfst/const-fst.h:118-245 reader, Fst(ConstFst) newfst/optimize-fst.h:82-134): header, then
//    ConstState x S {float final, u32 pos, u32 narcs, u32 niepsilons, u32 noepsilons}, StdArc x A.
//
// Both OpenFst paths apply the reference's super-final construction: one extra state (id = number
// of states) and, for every state with a final weight other than +inf (Zero), an arc
// {ilabel 0, olabel 0, weight = final weight, nextstate = super-final} placed FIRST among the
// state's arcs; niepsilons and noepsilons count it.  Arcs are otherwise kept in file order -- the
// decoder needs them ilabel-sorted (input epsilons first); wfst_graph_from_arrays checks that.
//
// Where the reference readers silently misread, this one refuses: symbol tables embedded in the
// file (header flags HAS_ISYMBOLS / HAS_OSYMBOLS -- the reference ignores the flags and reads the
// table bytes as states) and arc types other than "standard".  A const fst written with
// --fst_align (flag IS_ALIGNED) is read with OpenFst's 16-byte alignment rule, which the reference
// reader does not know.
#include "wfst_openfst.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace wfst {
namespace {

constexpr int32_t kFstMagicNumber = 2125659606;  // openfst: fst/fst.h
constexpr int32_t kHasIsymbols = 1, kHasOsymbols = 2, kIsAligned = 4;
constexpr int kArchAlignment = 16;               // openfst: fst/const-fst.h / fst/util.h

// bytes from the current position to the end of the file: counts read from a header are checked
// against it before anything is allocated (a damaged count must not become a 30 GB request)
long long bytes_left(FILE *fp) {
  const long pos = ftell(fp);
  if (pos < 0 || fseek(fp, 0, SEEK_END) != 0) return -1;
  const long end = ftell(fp);
  if (fseek(fp, pos, SEEK_SET) != 0) return -1;
  return (long long)end - pos;
}

struct Reader {
  FILE *fp;
  bool ok = true;
  explicit Reader(FILE *f) : fp(f) {}
  template <class T>
  T get() {
    T v{};
    if (ok && fread(&v, sizeof(T), 1, fp) != 1) ok = false;
    return v;
  }
  std::string str() {  // int32 length + bytes
    const int32_t n = get<int32_t>();
    if (!ok || n < 0 || n > 4096) { ok = false; return std::string(); }
    std::string s((size_t)n, '\0');
    if (n && fread(&s[0], 1, (size_t)n, fp) != (size_t)n) ok = false;
    return s;
  }
  void align() {  // skip to the next multiple of 16 bytes
    const long pos = ftell(fp);
    const long pad = (kArchAlignment - pos % kArchAlignment) % kArchAlignment;
    if (pad && fseek(fp, pad, SEEK_CUR) != 0) ok = false;
  }
};

struct ConstState { float weight; uint32_t pos, narcs, niepsilons, noepsilons; };
static_assert(sizeof(ConstState) == 20, "OpenFst ConstFst<StdArc, uint32> state record");

bool is_final(float w) { return !(std::isinf(w) && w > 0); }  // read_fst.c:124: final != +inf

int fail(std::string *err, int rc, const std::string &m) {
  if (err) *err = m;
  return rc;
}

int read_flat(FILE *fp, HostGraph *g, std::string *err) {
  int32_t hdr[6];
  if (fread(hdr, 4, 6, fp) != 6) return fail(err, WFST_E_IO, "truncated header");
  const int32_t S = hdr[2], A = hdr[3];
  if (S <= 0 || A < 0) return fail(err, WFST_E_IO, "bad header");
  if ((long long)S * (long long)sizeof(wfst_state_info) + (long long)A * (long long)sizeof(wfst_arc) > bytes_left(fp))
    return fail(err, WFST_E_IO, "truncated graph file");
  g->start = hdr[0];
  g->final_state = hdr[1];
  g->total_niepsilons = hdr[4];
  g->total_noepsilons = hdr[5];
  g->states.resize((size_t)S);
  g->arcs.resize((size_t)A);
  if (fread(g->states.data(), sizeof(wfst_state_info), (size_t)S, fp) != (size_t)S ||
      fread(g->arcs.data(), sizeof(wfst_arc), (size_t)A, fp) != (size_t)A)
    return fail(err, WFST_E_IO, "truncated graph file");
  return WFST_OK;
}

int read_openfst(FILE *fp, HostGraph *g, std::string *err) {
  Reader r(fp);
  (void)r.get<int32_t>();  // magic, already checked
  const std::string fsttype = r.str(), arctype = r.str();
  const int32_t version = r.get<int32_t>(), flags = r.get<int32_t>();
  (void)r.get<uint64_t>();  // properties
  const int64_t start = r.get<int64_t>(), numstates = r.get<int64_t>(), numarcs = r.get<int64_t>();
  (void)version;
  if (!r.ok) return fail(err, WFST_E_IO, "truncated OpenFst header");
  if (arctype != "standard") return fail(err, WFST_E_FORMAT, "OpenFst arc type \"" + arctype + "\": only \"standard\" (StdArc) graphs decode");
  if (flags & (kHasIsymbols | kHasOsymbols))
    return fail(err, WFST_E_FORMAT, "the fst embeds symbol tables; strip them first (fstsymbols --clear_isymbols --clear_osymbols)");
  if (numstates <= 0 || numstates >= 0x7FFFFFF0ll || start < 0 || start >= numstates)
    return fail(err, WFST_E_FORMAT, "OpenFst header: no states or bad start state");
  long long left = bytes_left(fp);  // asked once: seeking would drop the stdio buffer on every call
  if ((fsttype == "vector" && numstates * 12 > left) ||
      (fsttype == "const" && (numstates * (long long)sizeof(ConstState) > left || numarcs < 0 ||
                              numarcs > (left - numstates * (long long)sizeof(ConstState)) / (long long)sizeof(wfst_arc))))
    return fail(err, WFST_E_IO, "truncated OpenFst file (the header announces more states / arcs than the file holds)");
  const int32_t S = (int32_t)numstates, super_final = S;
  g->start = (int32_t)start;
  g->final_state = super_final;
  g->states.assign((size_t)S + 1, wfst_state_info{0, 0, 0});
  g->arcs.clear();
  g->total_niepsilons = g->total_noepsilons = 0;
  if (fsttype == "vector") {
    // read_fst.c:103-176: {float final, int64 narcs, arcs}; input/output epsilons counted here
    for (int32_t s = 0; s < S; ++s) {
      const float fin = r.get<float>();
      const int64_t na = r.get<int64_t>();
      left -= 12;
      if (!r.ok || na < 0 || left < 0 || na > left / (long long)sizeof(wfst_arc))
        return fail(err, WFST_E_IO, "truncated OpenFst vector fst");
      left -= na * (long long)sizeof(wfst_arc);
      wfst_state_info &si = g->states[(size_t)s];
      if (is_final(fin)) {
        g->arcs.push_back(wfst_arc{0, 0, fin, super_final});
        si.num_arcs++; si.niepsilons++; si.noepsilons++;
      }
      const size_t at = g->arcs.size();
      g->arcs.resize(at + (size_t)na);
      if (na && fread(&g->arcs[at], sizeof(wfst_arc), (size_t)na, fp) != (size_t)na)
        return fail(err, WFST_E_IO, "truncated OpenFst vector fst");
      for (size_t i = at; i < at + (size_t)na; ++i) {
        si.niepsilons += g->arcs[i].ilabel == 0;
        si.noepsilons += g->arcs[i].olabel == 0;
      }
      si.num_arcs += (uint32_t)na;
      g->total_niepsilons += (int32_t)si.niepsilons;
      g->total_noepsilons += (int32_t)si.noepsilons;
    }
  } else if (fsttype == "const") {
    // const-fst.h:196-228 + optimize-fst.h:82-134: epsilon counts come from the file
    if (numarcs < 0 || numarcs >= 0x7FFFFFF0ll) return fail(err, WFST_E_FORMAT, "OpenFst header: bad arc count");
    if (flags & kIsAligned) r.align();
    std::vector<ConstState> cs((size_t)S);
    if (!r.ok || fread(cs.data(), sizeof(ConstState), (size_t)S, fp) != (size_t)S)
      return fail(err, WFST_E_IO, "truncated OpenFst const fst (states)");
    if (flags & kIsAligned) r.align();
    std::vector<wfst_arc> in((size_t)numarcs);
    if (!r.ok || (numarcs && fread(in.data(), sizeof(wfst_arc), (size_t)numarcs, fp) != (size_t)numarcs))
      return fail(err, WFST_E_IO, "truncated OpenFst const fst (arcs)");
    g->arcs.reserve((size_t)numarcs + 1024);
    for (int32_t s = 0; s < S; ++s) {
      const ConstState &c = cs[(size_t)s];
      if ((uint64_t)c.pos + c.narcs > (uint64_t)numarcs) return fail(err, WFST_E_FORMAT, "OpenFst const fst: arc range of a state is out of bounds");
      wfst_state_info &si = g->states[(size_t)s];
      si = wfst_state_info{c.narcs, c.niepsilons, c.noepsilons};
      if (is_final(c.weight)) {  // optimize-fst.h:89-92 (weight != Zero)
        g->arcs.push_back(wfst_arc{0, 0, c.weight, super_final});
        si.num_arcs++; si.niepsilons++; si.noepsilons++;
      }
      g->arcs.insert(g->arcs.end(), in.begin() + c.pos, in.begin() + c.pos + c.narcs);
      g->total_niepsilons += (int32_t)si.niepsilons;
      g->total_noepsilons += (int32_t)si.noepsilons;
    }
  } else {
    return fail(err, WFST_E_FORMAT, "OpenFst fst type \"" + fsttype + "\": only \"vector\" and \"const\" are read");
  }
  if (g->arcs.size() >= 0x7FFFFFF0ull) return fail(err, WFST_E_FORMAT, "more than 2^31 arcs");
  return WFST_OK;
}

}  // namespace

int read_graph_file(const char *path, HostGraph *g, std::string *err) {
  FILE *fp = fopen(path, "rb");
  if (!fp) return fail(err, WFST_E_IO, std::string("cannot open ") + path);
  int32_t magic = 0;
  const bool got = fread(&magic, 4, 1, fp) == 1;
  rewind(fp);
  int rc;
  if (!got) rc = fail(err, WFST_E_IO, "truncated header");
  else if (magic == kFstMagicNumber) rc = read_openfst(fp, g, err);
  else rc = read_flat(fp, g, err);
  fclose(fp);
  return rc;
}

int write_flat_graph(const char *path, const HostGraph &g, std::string *err) {
  FILE *fp = fopen(path, "wb");
  if (!fp) return fail(err, WFST_E_IO, std::string("cannot create ") + path);
  const int32_t hdr[6] = {g.start, g.final_state, (int32_t)g.states.size(), (int32_t)g.arcs.size(), g.total_niepsilons,
                          g.total_noepsilons};
  bool ok = fwrite(hdr, 4, 6, fp) == 6 &&
            fwrite(g.states.data(), sizeof(wfst_state_info), g.states.size(), fp) == g.states.size() &&
            (g.arcs.empty() || fwrite(g.arcs.data(), sizeof(wfst_arc), g.arcs.size(), fp) == g.arcs.size());
  ok = (fclose(fp) == 0) && ok;
  return ok ? WFST_OK : fail(err, WFST_E_IO, std::string("write failed: ") + path);
}

}  // namespace wfst
