// TEST HARNESS -- compiles asr-decoder_amd/csrc/wfst_determinize.h for the HOST so that the algorithm the
// device runs (determinize_kernel) can be held against the reference's own determinizer on the CPU, lattice by
// lattice (tests/test_determinize_host.py).  Not part of the product: the library runs the device build only.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../asr-decoder_amd/csrc/wfst_determinize.h"

using namespace wfst;

// the closure's fast buffers (LDS on the device) emulated at a tiny size, so that closures outgrow them (det_host_set_low)
static int g_low_tmp = 0;

extern "C" {
void det_host_set_low(int tmp_lo) { g_low_tmp = tmp_lo; }
// Raw lattice in: states 0..S-1 (0 = start), final flags, arcs {src, dst, ilabel (transition-id), olabel (word),
// graph, acoustic}.  Determinized lattice out, in the wrapper's output form (after its last Invert): arcs
// {src, dst, ilabel 0, olabel word or 0, graph, acoustic}; a final weight is an arc to an extra final state.
// Returns 0, 1 on capacity overflow (counts are needed sizes then), 2 on bad input.
int det_host_run(int S, const int *is_final, int A, const int *a_src, const int *a_dst, const int *a_il, const int *a_ol,
                 const float *a_g, const float *a_ac, int cap_scale, int max_states, int *n_states, int *st_final,
                 int max_arcs, int *n_arcs, int *o_src, int *o_dst, int *o_il, int *o_ol, float *o_g, float *o_ac) {
  if (S <= 0) return 2;
  // Invert + CSR + ArcSort (by the new input label = word)
  std::vector<int32_t> off((size_t)S + 1, 0);
  for (int i = 0; i < A; ++i) {
    if (a_src[i] < 0 || a_src[i] >= S || a_dst[i] < 0 || a_dst[i] >= S) return 2;
    off[(size_t)a_src[i] + 1]++;
  }
  for (int s = 0; s < S; ++s) off[(size_t)s + 1] += off[s];
  std::vector<DetArc> arcs((size_t)A);
  std::vector<int32_t> cur(off.begin(), off.end() - 1);
  for (int i = 0; i < A; ++i) {
    DetArc d;
    d.ilabel = a_ol[i]; d.olabel = a_il[i]; d.w1 = a_g[i]; d.w2 = a_ac[i]; d.to = a_dst[i];
    arcs[(size_t)cur[a_src[i]]++] = d;
  }
  for (int s = 0; s < S; ++s)
    std::stable_sort(arcs.begin() + off[s], arcs.begin() + off[(size_t)s + 1], [](const DetArc &x, const DetArc &y) { return x.ilabel < y.ilabel; });
  DetCaps c;
  const int64_t base = std::max<int64_t>(1024, (int64_t)cap_scale * (A + S));
  c.trie = (int32_t)(8 * base); c.pool = (int32_t)(16 * base); c.states = (int32_t)(4 * base); c.initials = (int32_t)(4 * base);
  c.arcs = (int32_t)(4 * base); c.tmp = (int32_t)std::max<int64_t>(4096, 2 * (int64_t)(A + S));
  std::vector<int32_t> ws((size_t)det_words(c, S));
  std::vector<int32_t> fin(is_final, is_final + S);
  DetWs W;
  memset(&W, 0, sizeof(W));
  W.n_states = S; W.n_arcs = A; W.off = off.data(); W.arcs = arcs.data(); W.is_final = fin.data();
  W.delta = 1.0f / 1024;   // kDelta, DeterminizeLatticeOptions (lattice-determinize-api.h:16-25)
  det_carve(W, ws.data(), c, S);
  std::vector<DetElem> lo_b((size_t)g_low_tmp + 1), lo_c((size_t)g_low_tmp + 1);
  if (g_low_tmp > 0) { W.tb_lo = lo_b.data(); W.tc_lo = lo_c.data(); W.tmp_lo = g_low_tmp; }
  det_init(W, 0, 1);
  const int err = det_run(W);
  // OutputNoolabel (:307-377) + Invert
  int ns = W.os_n, na = 0;
  for (int s = 0; s < W.os_n && s < max_states; ++s) st_final[s] = 0;
  for (int i = 0; i < W.oa_n; ++i) {
    const DetOutArc &t = W.oarcs[i];
    int dst = t.next;
    if (t.next < 0) {
      dst = ns++;
      if (dst < max_states) st_final[dst] = 1;
    }
    if (na < max_arcs) {
      o_src[na] = t.src; o_dst[na] = dst; o_il[na] = 0; o_ol[na] = t.next < 0 ? 0 : t.ilabel; o_g[na] = t.w1; o_ac[na] = t.w2;
    }
    ++na;
  }
  *n_states = ns;
  *n_arcs = na;
  if ((err || getenv("DET_HOST_STATS")) && getenv("DET_HOST_VERBOSE"))
    fprintf(stderr, "det_host: err %d  trie %d/%d pool %d/%d states %d/%d initials %d/%d arcs %d/%d tmp %d (S %d A %d)\n", err, W.tr_n, c.trie,
            W.pool_n, c.pool, W.os_n, c.states, W.ih_n, c.initials, W.oa_n, c.arcs, c.tmp, S, A);
  return err ? 1 : 0;
}
}
