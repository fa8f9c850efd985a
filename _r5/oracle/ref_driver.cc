// TEST INFRASTRUCTURE -- not part of the product path.
//
// Driver that links the *unmodified* reference decoder (datemoon/ASR-decoder,
// compiled from the sources where they lie under /root/reference by
// oracle/Makefile into oracle/_ref/libref_decoder.so) behind a tiny C ABI so
// that tests and tools can (1) validate oracle/wfst_oracle.c bit for bit and
// (2) emit the golden vectors under tests/golden/.
//
// Only the code in this file is ours; everything it calls is the reference:
//   Fst::ReadFst                     src/newfst/optimize-fst.h:208-280
//   OnlineLatticeDecoderMempool      src/my-decoder/online-decoder-mempool-base.h:77
//   InitDecoding/AdvanceDecoding/FinalizeDecoding/GetBestPath
//                                    src/my-decoder/online-decoder-base-inl.h:41,631,830,1072
//   LatticeToVector                  src/newfst/lattice-functions.cc:179-217
//   OnlineLatticeDecoderMempoolBiglm src/my-decoder/online-decoder-mempool-base-biglm.h:570 (biglm, BASELINE configs[3])
//   ArpaLm / Arpa2Fsa / ComposeArpaLm src/newlm/arpa2fsa.{h,cc}, src/newlm/compose-arpalm.{h,cc}
// The decodable below plays the role of Kaldi's DecodableMatrixScaledMapped
// (kaldi-nnet3bin/kaldi-hclg-my-decoder.cc:107) with the scale pre-applied:
// LogLikelihood(f, tid) = M[f][tid2pdf[tid]].
//
// Never shipped, never imported by the product; `oracle/_ref/` is git-ignored.

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "src/my-decoder/online-decoder-mempool-base.h"
#include "src/my-decoder/online-decoder-mempool-base-biglm.h"
#include "src/newfst/const-fst.h"
#include "src/newfst/lattice-determinize-api.h"
#include "src/newfst/lattice-to-nbest.h"
#include "src/newfst/lattice-functions.h"
#include "src/newfst/compose-lat.h"
#include "src/newlm/compose-arpalm.h"

using namespace datemoon;

namespace {

class MatrixDecodable : public DecodableInterface {
 public:
  MatrixDecodable(const float *m, int T, int stride, const int *tid2pdf, int n_tid)
      : m_(m), T_(T), stride_(stride), map_(tid2pdf), n_tid_(n_tid), ready_(T) {}
  float LogLikelihood(int frame, int index) override {
    int col = map_ ? map_[index] : index;
    return m_[(size_t)frame * stride_ + col];
  }
  bool IsLastFrame(int frame) const override { return frame == T_ - 1; }
  int NumFramesReady() const override { return ready_; }
  int NumIndices() const override { return n_tid_; }
  void SetReady(int r) { ready_ = r; }

 private:
  const float *m_;
  int T_, stride_;
  const int *map_;
  int n_tid_, ready_;
};

// Probe subclass: reads protected members, changes no behaviour.
class ProbeDecoder : public OnlineLatticeDecoderMempool {
 public:
  ProbeDecoder(Fst *g, const LatticeFasterDecoderConfig &c) : OnlineLatticeDecoderMempool(g, c) {}
  int CountFrontier(float *best) const {
    int n = 0;
    float b = FLOAT_INF;
    for (const Elem *e = _toks.GetList(); e != NULL; e = e->tail) {
      ++n;
      if (e->val->_tot_cost < b) b = e->val->_tot_cost;
    }
    *best = b;
    return n;
  }
  int DumpFrontier(int *states, float *costs, int max_n) const {
    int n = 0;
    for (const Elem *e = _toks.GetList(); e != NULL; e = e->tail) {
      if (n < max_n) {
        states[n] = e->key;
        costs[n] = e->val->_tot_cost;
      }
      ++n;
    }
    return n;
  }
  int NumToks() const { return _num_toks; }
  int NumLinks() const { return _num_links; }
};

}  // namespace

extern "C" {

struct RefConfig {
  float beam;
  int max_active;
  int min_active;
  float lattice_beam;
  int prune_interval;
  float beam_delta;
  float hash_ratio;
  float prune_scale;
};

void *ref_graph_load(const char *path) {
  Fst *g = new Fst();
  if (!g->ReadFst(path)) {
    delete g;
    return NULL;
  }
  return g;
}

void ref_graph_free(void *g) { delete static_cast<Fst *>(g); }

void ref_graph_info(void *gp, int *start, int *n_states, int *n_arcs) {
  Fst *g = static_cast<Fst *>(gp);
  *start = g->Start();
  *n_states = g->TotState();
  *n_arcs = g->TotArc();
}

// Decode one utterance with the reference decoder.
//   chunk <= 0 : one AdvanceDecoding call over all T frames (offline CLI shape,
//                kaldi-hclg-my-decoder.cc:97-109)
//   chunk  > 0 : NumFramesReady grows by `chunk` per call (streaming shape,
//                kaldi-online-nnet3-my-decoder.cc:32-46)
// frame_ntoks/frame_best (nullable, T+1 entries) are filled only when
// chunk == 1 (frontier after InitDecoding at [0], after frame f at [f+1]).
// dump_frame >= 0 with chunk == 1 dumps that frontier (index as above) into
// dump_states/dump_costs (max dump_cap), count into *dump_n.
// Returns 1 if GetBestPath succeeded, 0 otherwise.
int ref_decode(void *gp, const RefConfig *rc, const float *loglikes, int T, int stride,
               const int *tid2pdf, int n_tid, int chunk, int do_finalize, int use_final_probs,
               int *path_ilabel, int *path_olabel, float *path_graph, float *path_ac,
               int max_path, int *n_path, float *tot_score, float *lm_score, int *words,
               int max_words, int *n_words, int *tids, int max_tids, int *n_tids,
               int *frame_ntoks, float *frame_best, int dump_frame, int *dump_states,
               float *dump_costs, int dump_cap, int *dump_n, int *num_toks_end,
               int *num_links_end) {
  Fst *g = static_cast<Fst *>(gp);
  LatticeFasterDecoderConfig cfg;
  cfg._beam = rc->beam;
  cfg._max_active = rc->max_active;
  cfg._min_active = rc->min_active;
  cfg._lattice_beam = rc->lattice_beam;
  cfg._prune_interval = rc->prune_interval;
  cfg._beam_delta = rc->beam_delta;
  cfg._hash_ratio = rc->hash_ratio;
  cfg._prune_scale = rc->prune_scale;

  ProbeDecoder dec(g, cfg);
  MatrixDecodable decodable(loglikes, T, stride, tid2pdf, n_tid);

  dec.InitDecoding();
  if (chunk == 1 && frame_ntoks) frame_ntoks[0] = dec.CountFrontier(&frame_best[0]);
  if (chunk == 1 && dump_frame == 0 && dump_n)
    *dump_n = dec.DumpFrontier(dump_states, dump_costs, dump_cap);
  if (chunk <= 0) {
    dec.AdvanceDecoding(&decodable);
  } else {
    for (int r = 0; r < T;) {
      r = (r + chunk < T) ? r + chunk : T;
      decodable.SetReady(r);
      dec.AdvanceDecoding(&decodable);
      if (chunk == 1 && frame_ntoks) frame_ntoks[r] = dec.CountFrontier(&frame_best[r]);
      if (chunk == 1 && dump_frame == r && dump_n)
        *dump_n = dec.DumpFrontier(dump_states, dump_costs, dump_cap);
    }
  }
  if (do_finalize) dec.FinalizeDecoding();
  if (num_toks_end) *num_toks_end = dec.NumToks();
  if (num_links_end) *num_links_end = dec.NumLinks();

  *n_path = 0;
  *n_words = 0;
  *n_tids = 0;
  *tot_score = 0;
  *lm_score = 0;
  Lattice best_path;
  if (!dec.GetBestPath(&best_path, use_final_probs != 0)) return 0;

  // Hop-by-hop dump in forward order (same walk as LatticeToVector).
  {
    StateId s = best_path.Start();
    LatticeState *cur = best_path.GetState(s);
    int n = 0;
    while (!cur->IsFinal()) {
      LatticeArc *arc = cur->GetArc(0);
      if (n < max_path) {
        path_ilabel[n] = arc->_input;
        path_olabel[n] = arc->_output;
        path_graph[n] = arc->_w.Value1();
        path_ac[n] = arc->_w.Value2();
      }
      ++n;
      cur = best_path.GetState(arc->_to);
    }
    *n_path = n;
  }
  std::vector<int> w, p;
  float tot = 0, lm = 0;
  if (!LatticeToVector(best_path, w, p, tot, lm)) return 0;
  *tot_score = tot;
  *lm_score = lm;
  *n_words = (int)w.size();
  *n_tids = (int)p.size();
  for (int i = 0; i < (int)w.size() && i < max_words; ++i) words[i] = w[i];
  for (int i = 0; i < (int)p.size() && i < max_tids; ++i) tids[i] = p[i];
  return 1;
}

// CPU baseline leg of bench.py: ONE reference decoder object (as one worker thread of the service
// holds, v2-asr/v2-asr-work-thread.h:66) decodes utterances mats[first], mats[first + step], ...
// (wrapping around n_mats) with the offline CLI's call sequence (kaldi-hclg-my-decoder.cc:97-129:
// InitDecoding, AdvanceDecoding, FinalizeDecoding, GetBestPath, LatticeToVector) until `seconds`
// of wall time have passed; an utterance in flight at the deadline is finished and counted.
// Returns the frames decoded; *elapsed = this thread's own wall time.  bench.py runs one such
// loop per host thread over one shared graph.
long long ref_timed_loop(void *gp, const RefConfig *rc, const float *const *mats, const int *T,
                         int n_mats, int stride, const int *tid2pdf, int n_tid, int first, int step,
                         double seconds, double *elapsed, long long *words_out) {
  Fst *g = static_cast<Fst *>(gp);
  LatticeFasterDecoderConfig cfg;
  cfg._beam = rc->beam;
  cfg._max_active = rc->max_active;
  cfg._min_active = rc->min_active;
  cfg._lattice_beam = rc->lattice_beam;
  cfg._prune_interval = rc->prune_interval;
  cfg._beam_delta = rc->beam_delta;
  cfg._hash_ratio = rc->hash_ratio;
  cfg._prune_scale = rc->prune_scale;
  ProbeDecoder dec(g, cfg);
  long long frames = 0, nwords = 0;
  const auto t0 = std::chrono::steady_clock::now();
  double dt = 0.0;
  for (int i = first % n_mats;; i = (i + step) % n_mats) {
    MatrixDecodable decodable(mats[i], T[i], stride, tid2pdf, n_tid);
    dec.InitDecoding();
    dec.AdvanceDecoding(&decodable);
    dec.FinalizeDecoding();
    Lattice best_path;
    if (dec.GetBestPath(&best_path, true)) {
      std::vector<int> w, p;
      float tot = 0, lm = 0;
      if (LatticeToVector(best_path, w, p, tot, lm)) nwords += (long long)w.size();
    }
    frames += T[i];
    dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (dt >= seconds) break;
  }
  if (elapsed) *elapsed = dt;
  if (words_out) *words_out = nwords;
  return frames;
}

// Decode one utterance (one AdvanceDecoding over all frames + FinalizeDecoding when do_finalize)
// and dump the reference's GetRawLattice(use_final_probs) (base-inl.h:869-975): state s is final
// iff st_final[s]; arcs in (state, arc) order.  Returns 1 if GetRawLattice returned true; counts
// are written even when they exceed the caps (then only the first cap entries are).
int ref_raw_lattice(void *gp, const RefConfig *rc, const float *loglikes, int T, int stride,
                    const int *tid2pdf, int n_tid, int do_finalize, int use_final_probs,
                    int max_states, int *n_states, int *start, int *st_final, int max_arcs,
                    int *n_arcs, int *a_src, int *a_dst, int *a_il, int *a_ol, float *a_graph,
                    float *a_ac) {
  Fst *g = static_cast<Fst *>(gp);
  LatticeFasterDecoderConfig cfg;
  cfg._beam = rc->beam;
  cfg._max_active = rc->max_active;
  cfg._min_active = rc->min_active;
  cfg._lattice_beam = rc->lattice_beam;
  cfg._prune_interval = rc->prune_interval;
  cfg._beam_delta = rc->beam_delta;
  cfg._hash_ratio = rc->hash_ratio;
  cfg._prune_scale = rc->prune_scale;
  ProbeDecoder dec(g, cfg);
  MatrixDecodable decodable(loglikes, T, stride, tid2pdf, n_tid);
  dec.InitDecoding();
  dec.AdvanceDecoding(&decodable);
  if (do_finalize) dec.FinalizeDecoding();
  Lattice lat;
  *n_states = 0;
  *n_arcs = 0;
  *start = -1;
  if (!dec.GetRawLattice(&lat, use_final_probs != 0)) return 0;
  const int S = lat.NumStates();
  *n_states = S;
  *start = lat.Start();
  int na = 0;
  for (int s = 0; s < S; ++s) {
    LatticeState *st = lat.GetState(s);
    if (s < max_states) st_final[s] = st->IsFinal() ? 1 : 0;
    const int k = (int)st->GetArcSize();
    for (int i = 0; i < k; ++i) {
      LatticeArc *a = st->GetArc(i);
      if (na < max_arcs) {
        a_src[na] = s;
        a_dst[na] = a->_to;
        a_il[na] = a->_input;
        a_ol[na] = a->_output;
        a_graph[na] = a->_w.Value1();
        a_ac[na] = a->_w.Value2();
      }
      ++na;
    }
  }
  *n_arcs = na;
  return 1;
}

// OpenFst const fst -> the reference's in-memory graph: ConstFst<StdArc,int>::Read
// (newfst/const-fst.h:173-228) + Fst(ConstFst) (newfst/optimize-fst.h:82-134), dumped as the flat
// arrays {num_arcs, niepsilons, noepsilons} x S and {ilabel, olabel, weight bits, nextstate} x A.
// Returns 1 on success; counts are always written.
int ref_constfst_dump(const char *path, int *start, int *final_state, int max_states, int *n_states,
                      unsigned *state_info, int max_arcs, int *n_arcs, int *arcs) {
  ConstFst<StdArc, int> cf;
  if (!cf.Read(std::string(path))) return 0;
  Fst fst(cf);
  *start = fst.Start();
  const int S = fst.TotState(), A = fst.TotArc();
  *n_states = S;
  *n_arcs = A;
  *final_state = S - 1;
  if (!fst.IsFinal(S - 1)) return 0;
  int na = 0;
  for (int s = 0; s < S; ++s) {
    StdState *st = fst.GetState(s);
    const unsigned n = st->GetArcSize();
    if (s < max_states) {
      state_info[3 * s + 0] = n;
      state_info[3 * s + 1] = (unsigned)fst.NumInputEpsilons(s);
      state_info[3 * s + 2] = (unsigned)fst.NumOutputEpsilons(s);
    }
    for (unsigned i = 0; i < n; ++i, ++na) {
      if (na >= max_arcs) continue;
      StdArc *a = st->GetArc(i);
      arcs[4 * na + 0] = a->_input;
      arcs[4 * na + 1] = a->_output;
      float w = a->_w.Value();
      memcpy(&arcs[4 * na + 2], &w, 4);
      arcs[4 * na + 3] = a->_to;
    }
  }
  return na == A ? 1 : 0;
}

// The reference's on-disk lattice format: decode as above and append GetRawLattice to `path` with
// the reference's own Lattice::Write(std::string&) (newfst/lattice-fst.h:327-342, lattice-fst.cc:38).
// Returns 1 if a lattice was written.
int ref_lattice_write(void *gp, const RefConfig *rc, const float *loglikes, int T, int stride,
                      const int *tid2pdf, int n_tid, const char *path) {
  Fst *g = static_cast<Fst *>(gp);
  LatticeFasterDecoderConfig cfg;
  cfg._beam = rc->beam;
  cfg._max_active = rc->max_active;
  cfg._min_active = rc->min_active;
  cfg._lattice_beam = rc->lattice_beam;
  cfg._prune_interval = rc->prune_interval;
  cfg._beam_delta = rc->beam_delta;
  cfg._hash_ratio = rc->hash_ratio;
  cfg._prune_scale = rc->prune_scale;
  ProbeDecoder dec(g, cfg);
  MatrixDecodable decodable(loglikes, T, stride, tid2pdf, n_tid);
  dec.InitDecoding();
  dec.AdvanceDecoding(&decodable);
  dec.FinalizeDecoding();
  Lattice lat;
  if (!dec.GetRawLattice(&lat, true)) return 0;
  std::string file(path);
  return lat.Write(file) ? 1 : 0;
}

// Read lattice number `index` of `path` with the reference's Lattice::Read(FILE*) and dump it like
// ref_raw_lattice does.  Returns 1 on success.
int ref_lattice_read(const char *path, int index, int max_states, int *n_states, int *start,
                     int *st_final, int max_arcs, int *n_arcs, int *a_src, int *a_dst, int *a_il,
                     int *a_ol, float *a_graph, float *a_ac) {
  FILE *fp = fopen(path, "rb");
  if (!fp) return 0;
  Lattice lat;
  bool ok = true;
  for (int i = 0; i <= index && ok; ++i) ok = lat.Read(fp);
  fclose(fp);
  if (!ok) return 0;
  const int S = lat.NumStates();
  *n_states = S;
  *start = lat.Start();
  int na = 0;
  for (int s = 0; s < S; ++s) {
    LatticeState *st = lat.GetState(s);
    if (s < max_states) st_final[s] = st->IsFinal() ? 1 : 0;
    const int k = (int)st->GetArcSize();
    for (int i = 0; i < k; ++i) {
      LatticeArc *a = st->GetArc(i);
      if (na < max_arcs) {
        a_src[na] = s;
        a_dst[na] = a->_to;
        a_il[na] = a->_input;
        a_ol[na] = a->_output;
        a_graph[na] = a->_w.Value1();
        a_ac[na] = a->_w.Value2();
      }
      ++na;
    }
  }
  *n_arcs = na;
  return 1;
}

// The service's n-best pipeline on lattice number `index` of `path` (reference on-disk format), all
// of it the reference's own code: Lattice::Read, LatticeCheckFormat, DeterminizeLatticeWrapper,
// NShortestPath, ConvertNbestToVector, LatticeToVector (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:
// 78-105,139-150).  Entry i: words[i*max_len ..], n_words[i], scores[2*i] = tot, [2*i+1] = lm.
// Returns the number of paths (<= n), -1 if the lattice cannot be read or fails the format check.
int ref_nbest_from_lattice_file(const char *path, int index, int n, int max_len, int *words,
                                int *n_words, float *scores, int *det_states, int *det_arcs) {
  FILE *fp = fopen(path, "rb");
  if (!fp) return -1;
  Lattice lat;
  bool ok = true;
  for (int i = 0; i <= index && ok; ++i) ok = lat.Read(fp);
  fclose(fp);
  if (!ok) return -1;
  if (!LatticeCheckFormat(&lat)) return -1;
  Lattice det;
  DeterminizeLatticeOptions opts;
  bool debug = false;
  if (!DeterminizeLatticeWrapper(&lat, &det, opts, &debug)) return -1;
  if (!LatticeCheckFormat(&det)) return -1;
  if (det_states) *det_states = det.NumStates();
  if (det_arcs) {
    int na = 0;
    for (int s = 0; s < det.NumStates(); ++s) na += (int)det.GetState(s)->GetArcSize();
    *det_arcs = na;
  }
  Lattice nbest_lat;
  NShortestPath(det, &nbest_lat, (size_t)n);
  std::vector<Lattice> paths;
  ConvertNbestToVector(nbest_lat, &paths);
  int k = 0;
  for (size_t i = 0; i < paths.size() && k < n; ++i) {
    std::vector<int> w, p;
    float tot = 0, lm = 0;
    if (!LatticeToVector(paths[i], w, p, tot, lm)) continue;
    n_words[k] = (int)w.size();
    for (int j = 0; j < (int)w.size() && j < max_len; ++j) words[k * max_len + j] = w[j];
    scores[2 * k] = tot;
    scores[2 * k + 1] = lm;
    ++k;
  }
  return k;
}


// The reference's determinized lattice: lattice number `index` of `path` through Lattice::Read,
// LatticeCheckFormat and DeterminizeLatticeWrapper (newfst/lattice-determinize-api.cc:5-21: Invert,
// ArcSort, LatticeDeterminizer::Determinize, OutputNoolabel, Invert), dumped like ref_raw_lattice.
// Returns 1 on success; counts are written even when they exceed the caps.
int ref_determinize_lattice_file(const char *path, int index, int max_states, int *n_states, int *start,
                                 int *st_final, int max_arcs, int *n_arcs, int *a_src, int *a_dst, int *a_il,
                                 int *a_ol, float *a_graph, float *a_ac) {
  FILE *fp = fopen(path, "rb");
  if (!fp) return 0;
  Lattice lat;
  bool ok = true;
  for (int i = 0; i <= index && ok; ++i) ok = lat.Read(fp);
  fclose(fp);
  if (!ok || !LatticeCheckFormat(&lat)) return 0;
  Lattice det;
  DeterminizeLatticeOptions opts;
  bool debug = false;
  if (!DeterminizeLatticeWrapper(&lat, &det, opts, &debug)) return 0;
  const int S = det.NumStates();
  *n_states = S;
  *start = det.Start();
  int na = 0;
  for (int s = 0; s < S; ++s) {
    LatticeState *st = det.GetState(s);
    if (s < max_states) st_final[s] = st->IsFinal() ? 1 : 0;
    const int k = (int)st->GetArcSize();
    for (int i = 0; i < k; ++i) {
      LatticeArc *a = st->GetArc(i);
      if (na < max_arcs) {
        a_src[na] = s;
        a_dst[na] = a->_to;
        a_il[na] = a->_input;
        a_ol[na] = a->_output;
        a_graph[na] = a->_w.Value1();
        a_ac[na] = a->_w.Value2();
      }
      ++na;
    }
  }
  *n_arcs = na;
  return 1;
}

// The service's GetLattice under --use-second (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:53-78) on lattice `index` of `path`:
// DeterminizeLatticeWrapper, then ComposeLattice with ComposeArpaLm(lm1) and with ComposeArpaLm(lm2) (newfst/compose-lat-inl.h),
// dumped like ref_determinize_lattice_file.  lm1 = old LM (rescaled by -1 at load), lm2 = new LM.
int ref_rescore_lattice_file(const char *path, int index, void *lm1, void *lm2, int max_states, int *n_states, int *start,
                             int *st_final, int max_arcs, int *n_arcs, int *a_src, int *a_dst, int *a_il, int *a_ol, float *a_graph,
                             float *a_ac) {
  FILE *fp = fopen(path, "rb");
  if (!fp) return 0;
  Lattice lat;
  bool ok = true;
  for (int i = 0; i <= index && ok; ++i) ok = lat.Read(fp);
  fclose(fp);
  if (!ok || !LatticeCheckFormat(&lat)) return 0;
  Lattice det, lat1, out;
  DeterminizeLatticeOptions opts;
  bool debug = false;
  if (!DeterminizeLatticeWrapper(&lat, &det, opts, &debug)) return 0;
  ComposeArpaLm c1(static_cast<ArpaLm *>(lm1)), c2(static_cast<ArpaLm *>(lm2));
  ComposeLattice<FsaStateId>(&det, static_cast<LatticeComposeItf<FsaStateId> *>(&c1), &lat1);
  ComposeLattice<FsaStateId>(&lat1, static_cast<LatticeComposeItf<FsaStateId> *>(&c2), &out);
  const int S = out.NumStates();
  *n_states = S;
  *start = out.Start();
  int na = 0;
  for (int s = 0; s < S; ++s) {
    LatticeState *st = out.GetState(s);
    if (s < max_states) st_final[s] = st->IsFinal() ? 1 : 0;
    const int k = (int)st->GetArcSize();
    for (int i = 0; i < k; ++i) {
      LatticeArc *a = st->GetArc(i);
      if (na < max_arcs) {
        a_src[na] = s;
        a_dst[na] = a->_to;
        a_il[na] = a->_input;
        a_ol[na] = a->_output;
        a_graph[na] = a->_w.Value1();
        a_ac[na] = a->_w.Value2();
      }
      ++na;
    }
  }
  *n_arcs = na;
  return 1;
}

// The service's GetNbest as LATTICES (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:97-105): GetLattice -- determinize, and with
// lm1 / lm2 given the two ComposeLattice passes of --use-second -- then NShortestPath and ConvertNbestToVector
// (newfst/lattice-to-nbest.cc:15-199).  Path i = arcs path_off[i] .. path_off[i+1]: the arcs of the i-th linear lattice walked from
// its start state (labels and both costs of every arc, the epsilon arcs Reverse adds included).  Returns the number of paths,
// -1 on a read / format failure; *n_arcs is the total even beyond max_arcs.
int ref_nbest_paths_from_lattice_file(const char *path, int index, int n, void *lm1, void *lm2, int max_paths, int *path_off, int max_arcs,
                                      int *n_arcs, int *a_il, int *a_ol, float *a_graph, float *a_ac) {
  FILE *fp = fopen(path, "rb");
  if (!fp) return -1;
  Lattice lat;
  bool ok = true;
  for (int i = 0; i <= index && ok; ++i) ok = lat.Read(fp);
  fclose(fp);
  if (!ok || !LatticeCheckFormat(&lat)) return -1;
  Lattice det, lat1, olat;
  DeterminizeLatticeOptions opts;
  bool debug = false;
  if (!DeterminizeLatticeWrapper(&lat, &det, opts, &debug)) return -1;
  Lattice *src = &det;
  if (lm1 && lm2) {
    ComposeArpaLm c1(static_cast<ArpaLm *>(lm1)), c2(static_cast<ArpaLm *>(lm2));
    ComposeLattice<FsaStateId>(&det, static_cast<LatticeComposeItf<FsaStateId> *>(&c1), &lat1);
    ComposeLattice<FsaStateId>(&lat1, static_cast<LatticeComposeItf<FsaStateId> *>(&c2), &olat);
    src = &olat;
  }
  Lattice nbest_lat;
  NShortestPath(*src, &nbest_lat, (size_t)n);
  std::vector<Lattice> paths;
  ConvertNbestToVector(nbest_lat, &paths);
  int k = 0, na = 0;
  for (size_t i = 0; i < paths.size() && k < max_paths; ++i) {
    Lattice &P = paths[i];
    if (P.Start() == kNoStateId) continue;
    path_off[k] = na;
    LatticeState *st = P.GetState(P.Start());
    int guard = 0;
    while (!st->IsFinal() && st->GetArcSize() > 0 && guard++ < (1 << 20)) {
      LatticeArc *a = st->GetArc(0);
      if (na < max_arcs) {
        a_il[na] = a->_input;
        a_ol[na] = a->_output;
        a_graph[na] = a->_w.Value1();
        a_ac[na] = a->_w.Value2();
      }
      ++na;
      st = P.GetState(a->_to);
    }
    ++k;
  }
  path_off[k] = na;
  *n_arcs = na;
  return k;
}

// ---------------------------------------------------------------------------------------------
// biglm (BASELINE configs[3]): the reference's LM automaton and its on-the-fly rescoring decoder.
// ---------------------------------------------------------------------------------------------

// Arpa2Fsa::ConvertArpa2Fsa + ArpaLm::Write (newlm/arpa2fsa-bin.cc:10-31): ARPA text + word list ->
// the reference's binary LM file.  Returns 1 on success.
int ref_arpa2fsa(const char *arpafile, const char *wordlist, const char *outfile, int nthread) {
  Arpa2Fsa conv(nthread, arpafile, wordlist);
  if (!conv.ConvertArpa2Fsa()) return 0;
  return conv.Write(outfile) ? 1 : 0;
}

// ArpaLm::Read + Rescale (kaldi-hclg-my-decoder-biglm.cc:55-60 rescales the old LM by -1).
void *ref_lm_load(const char *path, float scale) {
  ArpaLm *lm = new ArpaLm();
  if (!lm->Read(path)) {
    delete lm;
    return NULL;
  }
  lm->Rescale(scale);
  return lm;
}
void ref_lm_free(void *lm) { delete static_cast<ArpaLm *>(lm); }
void ref_lm_info(void *lmp, int *bos, int *eos, int *unk, int *order) {
  ArpaLm *lm = static_cast<ArpaLm *>(lmp);
  *bos = lm->BosSymbol();
  *eos = lm->EosSymbol();
  *unk = lm->UnkSymbol();
  *order = lm->NgramOrder();
}
// ComposeArpaLm (newlm/compose-arpalm.cc:5-70): Start, Final, GetArc with the back-off walk.
int ref_lm_start(void *lmp) { return ComposeArpaLm(static_cast<ArpaLm *>(lmp)).Start(); }
float ref_lm_final(void *lmp, int s) { return ComposeArpaLm(static_cast<ArpaLm *>(lmp)).Final(s); }
void ref_lm_getarc(void *lmp, int s, int word, int *next, float *value1) {
  ComposeArpaLm c(static_cast<ArpaLm *>(lmp));
  FsaStateId ns = 0;
  LatticeWeight w;
  Label ol = 0;
  c.GetArc(s, word, &ns, &w, &ol);
  *next = ns;
  *value1 = w.Value1();
}
// n calls of ref_lm_getarc in one go (states[i], words[i]) -> (next[i], value1[i])
void ref_lm_getarc_many(void *lmp, int n, const int *states, const int *words, int *next, float *value1) {
  ComposeArpaLm c(static_cast<ArpaLm *>(lmp));
  for (int i = 0; i < n; ++i) {
    FsaStateId ns = 0;
    LatticeWeight w;
    Label ol = 0;
    c.GetArc(states[i], words[i], &ns, &w, &ol);
    next[i] = ns;
    value1[i] = w.Value1();
  }
}

namespace {
class ProbeBiglm : public OnlineLatticeDecoderMempoolBiglm {
 public:
  ProbeBiglm(Fst *g, const LatticeFasterDecoderConfig &c, ArpaLm *a, ArpaLm *b) : OnlineLatticeDecoderMempoolBiglm(g, c, a, b) {}
  int CountFrontier(float *best) const {
    int n = 0;
    float b = FLOAT_INF;
    for (const Elem *e = _toks.GetList(); e != NULL; e = e->tail) {
      ++n;
      if (e->val->_tot_cost < b) b = e->val->_tot_cost;
    }
    *best = b;
    return n;
  }
  int NumToks() const { return _num_toks; }
  int NumLinks() const { return _num_links; }
};
}  // namespace

// ref_decode() with the biglm decoder (kaldi-nnet3bin/kaldi-hclg-my-decoder-biglm.cc:80-102):
// lm1 = old LM (already rescaled by -1 at load), lm2 = new LM.
int ref_biglm_decode(void *gp, const RefConfig *rc, void *lm1, void *lm2, const float *loglikes, int T,
                     int stride, const int *tid2pdf, int n_tid, int chunk, int do_finalize,
                     int use_final_probs, int *path_ilabel, int *path_olabel, float *path_graph,
                     float *path_ac, int max_path, int *n_path, float *tot_score, float *lm_score,
                     int *words, int max_words, int *n_words, int *tids, int max_tids, int *n_tids,
                     int *frame_ntoks, float *frame_best, int *num_toks_end, int *num_links_end) {
  Fst *g = static_cast<Fst *>(gp);
  LatticeFasterDecoderConfig cfg;
  cfg._beam = rc->beam;
  cfg._max_active = rc->max_active;
  cfg._min_active = rc->min_active;
  cfg._lattice_beam = rc->lattice_beam;
  cfg._prune_interval = rc->prune_interval;
  cfg._beam_delta = rc->beam_delta;
  cfg._hash_ratio = rc->hash_ratio;
  cfg._prune_scale = rc->prune_scale;
  ProbeBiglm dec(g, cfg, static_cast<ArpaLm *>(lm1), static_cast<ArpaLm *>(lm2));
  MatrixDecodable decodable(loglikes, T, stride, tid2pdf, n_tid);
  dec.InitDecoding();
  if (chunk == 1 && frame_ntoks) frame_ntoks[0] = dec.CountFrontier(&frame_best[0]);
  if (chunk <= 0) {
    dec.AdvanceDecoding(&decodable);
  } else {
    for (int r = 0; r < T;) {
      r = (r + chunk < T) ? r + chunk : T;
      decodable.SetReady(r);
      dec.AdvanceDecoding(&decodable);
      if (chunk == 1 && frame_ntoks) frame_ntoks[r] = dec.CountFrontier(&frame_best[r]);
    }
  }
  if (do_finalize) dec.FinalizeDecoding();
  if (num_toks_end) *num_toks_end = dec.NumToks();
  if (num_links_end) *num_links_end = dec.NumLinks();
  *n_path = 0;
  *n_words = 0;
  *n_tids = 0;
  *tot_score = 0;
  *lm_score = 0;
  Lattice best_path;
  if (!dec.GetBestPath(&best_path, use_final_probs != 0)) return 0;
  {
    StateId s = best_path.Start();
    LatticeState *cur = best_path.GetState(s);
    int n = 0;
    while (!cur->IsFinal()) {
      LatticeArc *arc = cur->GetArc(0);
      if (n < max_path) {
        path_ilabel[n] = arc->_input;
        path_olabel[n] = arc->_output;
        path_graph[n] = arc->_w.Value1();
        path_ac[n] = arc->_w.Value2();
      }
      ++n;
      cur = best_path.GetState(arc->_to);
    }
    *n_path = n;
  }
  std::vector<int> w, p;
  float tot = 0, lm = 0;
  if (!LatticeToVector(best_path, w, p, tot, lm)) return 0;
  *tot_score = tot;
  *lm_score = lm;
  *n_words = (int)w.size();
  *n_tids = (int)p.size();
  for (int i = 0; i < (int)w.size() && i < max_words; ++i) words[i] = w[i];
  for (int i = 0; i < (int)p.size() && i < max_tids; ++i) tids[i] = p[i];
  return 1;
}

// ref_raw_lattice() with the biglm decoder: GetRawLattice(use_final_probs) of OnlineLatticeDecoderMempoolBiglm -- what the
// service takes from a `biglm-hclg` decoder (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:58,81).
int ref_biglm_raw_lattice(void *gp, const RefConfig *rc, void *lm1, void *lm2, const float *loglikes, int T, int stride,
                          const int *tid2pdf, int n_tid, int do_finalize, int use_final_probs,
                          int max_states, int *n_states, int *start, int *st_final, int max_arcs,
                          int *n_arcs, int *a_src, int *a_dst, int *a_il, int *a_ol, float *a_graph,
                          float *a_ac) {
  Fst *g = static_cast<Fst *>(gp);
  LatticeFasterDecoderConfig cfg;
  cfg._beam = rc->beam;
  cfg._max_active = rc->max_active;
  cfg._min_active = rc->min_active;
  cfg._lattice_beam = rc->lattice_beam;
  cfg._prune_interval = rc->prune_interval;
  cfg._beam_delta = rc->beam_delta;
  cfg._hash_ratio = rc->hash_ratio;
  cfg._prune_scale = rc->prune_scale;
  ProbeBiglm dec(g, cfg, static_cast<ArpaLm *>(lm1), static_cast<ArpaLm *>(lm2));
  MatrixDecodable decodable(loglikes, T, stride, tid2pdf, n_tid);
  dec.InitDecoding();
  dec.AdvanceDecoding(&decodable);
  if (do_finalize) dec.FinalizeDecoding();
  Lattice lat;
  *n_states = 0;
  *n_arcs = 0;
  *start = -1;
  if (!dec.GetRawLattice(&lat, use_final_probs != 0)) return 0;
  const int S = lat.NumStates();
  *n_states = S;
  *start = lat.Start();
  int na = 0;
  for (int s = 0; s < S; ++s) {
    LatticeState *st = lat.GetState(s);
    if (s < max_states) st_final[s] = st->IsFinal() ? 1 : 0;
    const int k = (int)st->GetArcSize();
    for (int i = 0; i < k; ++i) {
      LatticeArc *a = st->GetArc(i);
      if (na < max_arcs) {
        a_src[na] = s;
        a_dst[na] = a->_to;
        a_il[na] = a->_input;
        a_ol[na] = a->_output;
        a_graph[na] = a->_w.Value1();
        a_ac[na] = a->_w.Value2();
      }
      ++na;
    }
  }
  *n_arcs = na;
  return 1;
}

// ref_timed_loop() with ONE biglm decoder object (bench.py --biglm, cpu_baseline leg).
long long ref_biglm_timed_loop(void *gp, const RefConfig *rc, void *lm1, void *lm2, const float *const *mats,
                               const int *T, int n_mats, int stride, const int *tid2pdf, int n_tid, int first,
                               int step, double seconds, double *elapsed, long long *words_out) {
  Fst *g = static_cast<Fst *>(gp);
  LatticeFasterDecoderConfig cfg;
  cfg._beam = rc->beam;
  cfg._max_active = rc->max_active;
  cfg._min_active = rc->min_active;
  cfg._lattice_beam = rc->lattice_beam;
  cfg._prune_interval = rc->prune_interval;
  cfg._beam_delta = rc->beam_delta;
  cfg._hash_ratio = rc->hash_ratio;
  cfg._prune_scale = rc->prune_scale;
  ProbeBiglm dec(g, cfg, static_cast<ArpaLm *>(lm1), static_cast<ArpaLm *>(lm2));
  long long frames = 0, nwords = 0;
  const auto t0 = std::chrono::steady_clock::now();
  double dt = 0.0;
  for (int i = first % n_mats;; i = (i + step) % n_mats) {
    MatrixDecodable decodable(mats[i], T[i], stride, tid2pdf, n_tid);
    dec.InitDecoding();
    dec.AdvanceDecoding(&decodable);
    dec.FinalizeDecoding();
    Lattice best_path;
    if (dec.GetBestPath(&best_path, true)) {
      std::vector<int> w, p;
      float tot = 0, lm = 0;
      if (LatticeToVector(best_path, w, p, tot, lm)) nwords += (long long)w.size();
    }
    frames += T[i];
    dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (dt >= seconds) break;
  }
  if (elapsed) *elapsed = dt;
  if (words_out) *words_out = nwords;
  return frames;
}

// ref_timed_loop() for the LATTICE pipeline (bench.py --lattice-links ... --determinize, cpu_baseline of BASELINE configs[4]): one
// decoder object per host thread; per utterance InitDecoding, AdvanceDecoding (forward links + PruneActiveTokens every
// prune_interval frames), FinalizeDecoding, GetBestPath + LatticeToVector, then the service's GetNbest chain
// (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:78-105): GetRawLattice, LatticeCheckFormat, DeterminizeLatticeWrapper
// (newfst/lattice-determinize-api.cc:5-21), NShortestPath(n) (newfst/lattice-to-nbest.cc:15-147), ConvertNbestToVector.
// with_post = 0: decode + best path only (what part of the time the decoder itself takes).
// stage_seconds[4] (this thread's own sums) = {decode incl. finalize + best path, GetRawLattice, determinizer, n-best};
// counts[4] = {lattices determinized, raw states, determinized states, n-best paths}.
long long ref_lattice_timed_loop(void *gp, const RefConfig *rc, const float *const *mats, const int *T, int n_mats, int stride,
                                 const int *tid2pdf, int n_tid, int first, int step, double seconds, int with_post, int nbest,
                                 double *elapsed, double *stage_seconds, long long *counts) {
  Fst *g = static_cast<Fst *>(gp);
  LatticeFasterDecoderConfig cfg;
  cfg._beam = rc->beam;
  cfg._max_active = rc->max_active;
  cfg._min_active = rc->min_active;
  cfg._lattice_beam = rc->lattice_beam;
  cfg._prune_interval = rc->prune_interval;
  cfg._beam_delta = rc->beam_delta;
  cfg._hash_ratio = rc->hash_ratio;
  cfg._prune_scale = rc->prune_scale;
  ProbeDecoder dec(g, cfg);
  long long frames = 0;
  double st[4] = {0, 0, 0, 0};
  long long cn[4] = {0, 0, 0, 0};
  typedef std::chrono::steady_clock clk;
  const auto t0 = clk::now();
  double dt = 0.0;
  for (int i = first % n_mats;; i = (i + step) % n_mats) {
    auto a = clk::now();
    MatrixDecodable decodable(mats[i], T[i], stride, tid2pdf, n_tid);
    dec.InitDecoding();
    dec.AdvanceDecoding(&decodable);
    dec.FinalizeDecoding();
    Lattice best_path;
    if (dec.GetBestPath(&best_path, true)) {
      std::vector<int> w, p;
      float tot = 0, lm = 0;
      LatticeToVector(best_path, w, p, tot, lm);
    }
    auto b = clk::now();
    st[0] += std::chrono::duration<double>(b - a).count();
    if (with_post) {
      Lattice lat, det;
      if (dec.GetRawLattice(&lat, true) && LatticeCheckFormat(&lat)) {
        auto c = clk::now();
        st[1] += std::chrono::duration<double>(c - b).count();
        DeterminizeLatticeOptions opts;
        bool debug = false;
        const bool ok = DeterminizeLatticeWrapper(&lat, &det, opts, &debug);
        auto d = clk::now();
        st[2] += std::chrono::duration<double>(d - c).count();
        if (ok) {
          cn[0] += 1;
          cn[1] += lat.NumStates();
          cn[2] += det.NumStates();
          Lattice nbest_lat;
          NShortestPath(det, &nbest_lat, (size_t)nbest);
          std::vector<Lattice> paths;
          ConvertNbestToVector(nbest_lat, &paths);
          for (size_t k = 0; k < paths.size(); ++k) {
            std::vector<int> w, p;
            float tot = 0, lm = 0;
            if (LatticeToVector(paths[k], w, p, tot, lm)) cn[3] += 1;
          }
          st[3] += std::chrono::duration<double>(clk::now() - d).count();
        }
      }
    }
    frames += T[i];
    dt = std::chrono::duration<double>(clk::now() - t0).count();
    if (dt >= seconds) break;
  }
  if (elapsed) *elapsed = dt;
  for (int k = 0; k < 4; ++k) {
    if (stage_seconds) stage_seconds[k] = st[k];
    if (counts) counts[k] = cn[k];
  }
  return frames;
}

}  // extern "C"
