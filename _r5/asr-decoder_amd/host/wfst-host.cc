// Implementation of wfst-host.h: thin C++ over the C ABI.  No decoding happens on the host.
#include "wfst-host.h"

#include <atomic>
#include <thread>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

namespace datemoon {

namespace {
[[noreturn]] void Fatal(const std::string &what) { throw std::runtime_error(what + ": " + wfst_last_error()); }
void Warn(const std::string &msg) { std::cerr << "WARNING (wfst) " << msg << std::endl; }

// hop list (start->final order) -> the linear Lattice the reference's GetBestPath builds
// (base-inl.h:1080-1091): last state = start, state 0 = final.
void HopsToLattice(const int32_t *il, const int32_t *ol, const float *g, const float *ac, int n, Lattice *ofst) {
  ofst->DeleteStates();
  StateId state = ofst->AddState();
  ofst->SetFinal(state);
  for (int k = n - 1; k >= 0; --k) {
    StateId ns = ofst->AddState();
    ofst->AddArc(ns, LatticeArc(il[k], ol[k], state, LatticeWeight(g[k], ac[k])));
    state = ns;
  }
  ofst->SetStart(state);
}
}  // namespace

// ---- config -------------------------------------------------------------------------------
void LatticeFasterDecoderConfig::ReadConfigFile(const std::string &path) {
  std::ifstream in(path.c_str());
  if (!in) throw std::runtime_error("cannot open config file " + path);
  std::string line;
  while (std::getline(in, line)) {
    size_t h = line.find('#');
    if (h != std::string::npos) line.erase(h);
    size_t b = line.find_first_not_of(" \t\r\n");
    if (b == std::string::npos) continue;
    line = line.substr(b, line.find_last_not_of(" \t\r\n") - b + 1);
    if (line.compare(0, 2, "--") != 0) throw std::runtime_error("bad config line: " + line);
    size_t eq = line.find('=');
    if (eq == std::string::npos) throw std::runtime_error("bad config line (no '='): " + line);
    std::string name = line.substr(2, eq - 2), val = line.substr(eq + 1);
    std::replace(name.begin(), name.end(), '_', '-');
    if (name == "beam") _beam = (float)atof(val.c_str());
    else if (name == "max-active") _max_active = atoi(val.c_str());
    else if (name == "min-active") _min_active = atoi(val.c_str());
    else if (name == "lattice-beam") _lattice_beam = (float)atof(val.c_str());
    else if (name == "prune-interval") _prune_interval = atoi(val.c_str());
    else if (name == "beam-delta") _beam_delta = (float)atof(val.c_str());
    else if (name == "hash-ratio") _hash_ratio = (float)atof(val.c_str());
    else if (name == "determinize-lattice") _determinize_lattice = (val == "true" || val == "1");
    else throw std::runtime_error("unknown decoder option --" + name);
  }
}

void LatticeFasterDecoderConfig::Check() const {
  if (!(_beam > 0.0 && _max_active > 1 && _lattice_beam > 0.0 && _prune_interval > 0 && _beam_delta > 0.0 &&
        _hash_ratio >= 1.0 && _prune_scale > 0.0 && _prune_scale < 1.0))
    throw std::runtime_error("LatticeFasterDecoderConfig::Check failed");
}

wfst_config LatticeFasterDecoderConfig::ToC() const {
  wfst_config c;
  c.beam = _beam;
  c.max_active = _max_active;
  c.min_active = _min_active;
  c.lattice_beam = _lattice_beam;
  c.prune_interval = _prune_interval;
  c.beam_delta = _beam_delta;
  c.hash_ratio = _hash_ratio;
  c.prune_scale = _prune_scale;
  return c;
}

// ---- graph ----------------------------------------------------------------------------------
Fst::~Fst() { wfst_graph_free(_graph); }

bool Fst::ReadFst(const char *file, int device) {
  wfst_graph_free(_graph);
  _graph = nullptr;
  if (wfst_graph_load(file, device, &_graph) != WFST_OK) {
    std::cerr << "ReadFst " << file << " failed: " << wfst_last_error() << std::endl;
    return false;
  }
  wfst_graph_info(_graph, &_start, &_final, &_states, &_arcs, nullptr);
  return true;
}

void Fst::SetTid2Pdf(const std::vector<int32_t> &m) {
  if (!_graph) throw std::runtime_error("SetTid2Pdf before ReadFst");
  if (wfst_graph_set_tid2pdf(_graph, m.data(), (int32_t)m.size() - 1) != WFST_OK) Fatal("wfst_graph_set_tid2pdf");
}

// ---- LatticeToVector --------------------------------------------------------------------------
bool LatticeToVector(Lattice &best_path, std::vector<int> &words, std::vector<int> &phones, float &tot, float &lm) {
  if (best_path.Start() == kNoStateId) return false;
  tot = 0;
  lm = 0;
  LatticeState *cur = best_path.GetState(best_path.Start());
  while (!cur->IsFinal()) {
    LatticeArc *arc = cur->GetArc(0);
    if (arc->_input != 0) phones.push_back(arc->_input);
    if (arc->_output != 0) words.push_back(arc->_output);
    lm += arc->_w.Value1();
    tot += arc->_w.Value1() + arc->_w.Value2();
    cur = best_path.GetState(arc->_to);
  }
  return true;
}

// ---- on-disk lattice (reference format, see wfst-host.h) ----------------------------------------
bool Lattice::Write(FILE *fp) {
  if (!fp) return false;
  const uint64_t n = _states.size();
  const int32_t start = _start;
  if (fwrite(&n, 8, 1, fp) != 1 || fwrite(&start, 4, 1, fp) != 1) {
    std::cerr << "Write lattice state number error." << std::endl;
    return false;
  }
  for (LatticeState &st : _states) {
    const int32_t fin = st.IsFinal() ? 1 : 0;
    const uint64_t na = st.GetArcSize();
    if (fwrite(&fin, 4, 1, fp) != 1 || fwrite(&na, 8, 1, fp) != 1) {
      std::cerr << "Write state error." << std::endl;
      return false;
    }
    for (unsigned i = 0; i < na; ++i) {
      const LatticeArc *a = st.GetArc(i);
      const int32_t lab[2] = {a->_input, a->_output};
      const float w[2] = {a->_w.Value1(), a->_w.Value2()};
      const int32_t to = a->_to;
      if (fwrite(lab, 4, 2, fp) != 2 || fwrite(w, 4, 2, fp) != 2 || fwrite(&to, 4, 1, fp) != 1) {
        std::cerr << "Write state arc error." << std::endl;
        return false;
      }
    }
  }
  return true;
}
bool Lattice::Write(const std::string &file) {
  FILE *fp = fopen(file.c_str(), "ab");
  if (!fp) {
    std::cerr << "Write " << file << " failed." << std::endl;
    return false;
  }
  const bool ok = Write(fp);
  fclose(fp);
  if (!ok) std::cerr << "Write " << file << " failed." << std::endl;
  return ok;
}
bool Lattice::Read(FILE *fp) {
  DeleteStates();
  if (!fp) return false;
  uint64_t n = 0;
  int32_t start = 0;
  if (fread(&n, 8, 1, fp) != 1 || fread(&start, 4, 1, fp) != 1) return false;  // also: clean end of file
  for (uint64_t s = 0; s < n; ++s) {
    int32_t fin = 0;
    uint64_t na = 0;
    if (fread(&fin, 4, 1, fp) != 1 || fread(&na, 8, 1, fp) != 1) {
      std::cerr << "Read state error." << std::endl;
      DeleteStates();
      return false;
    }
    const StateId id = AddState();
    if (fin) SetFinal(id);
    for (uint64_t i = 0; i < na; ++i) {
      int32_t lab[2], to;
      float w[2];
      if (fread(lab, 4, 2, fp) != 2 || fread(w, 4, 2, fp) != 2 || fread(&to, 4, 1, fp) != 1) {
        std::cerr << "Read state arc " << i << " error." << std::endl;
        DeleteStates();
        return false;
      }
      AddArc(id, LatticeArc(lab[0], lab[1], to, LatticeWeight(w[0], w[1])));
    }
  }
  _start = start;
  return true;
}
bool Lattice::Read(const std::string &file) {
  FILE *fp = fopen(file.c_str(), "rb");
  if (!fp) {
    std::cerr << "Open " << file << " failed." << std::endl;
    return false;
  }
  const bool ok = Read(fp);
  fclose(fp);
  if (!ok) std::cerr << "Read " << file << " failed." << std::endl;
  return ok;
}

// ---- language model (biglm) -------------------------------------------------------------------
ArpaLm::~ArpaLm() { wfst_lm_free(_lm); }
bool ArpaLm::Read(const char *file, int device) {
  wfst_lm_free(_lm);
  _lm = nullptr;
  _file = file;
  _device = device;
  if (wfst_lm_load(file, _scale, device, &_lm) != WFST_OK) {  // checks the file now; re-uploaded if Rescale follows
    std::cerr << "Read " << file << " failed: " << wfst_last_error() << std::endl;
    return false;
  }
  int32_t ns, na, nw;
  int64_t bytes;
  wfst_lm_info(_lm, &_bos, &_eos, &ns, &na, &nw, &bytes);
  return true;
}
void ArpaLm::Rescale(float scale) {  // arpa2fsa.cc:264-275: weights *= scale (applied when the automaton is uploaded)
  if (scale == 1.0f) return;
  _scale *= scale;
  wfst_lm_free(_lm);
  _lm = nullptr;
}
const wfst_lm *ArpaLm::Handle() {
  // several worker threads construct their decoders over the same two LMs at once (wfst-decode --inflight): the upload
  // happens once, under the lock, into a local that is published only when complete
  std::lock_guard<std::mutex> lock(_mu);
  if (!_lm) {
    if (_file.empty()) throw std::runtime_error("ArpaLm used before Read()");
    wfst_lm *lm = nullptr;
    if (wfst_lm_load(_file.c_str(), _scale, _device, &lm) != WFST_OK) Fatal("ArpaLm upload");
    _lm = lm;
  }
  return _lm;
}

// ---- single-stream decoder --------------------------------------------------------------------
GpuLatticeDecoder::GpuLatticeDecoder(Fst *graph, const LatticeFasterDecoderConfig &config, const wfst_limits *limits)
    : _dec(nullptr), _stride(0), _rows_ready(0), _inited(false) {
  config.Check();
  wfst_config c = config.ToC();
  if (wfst_decoder_create(graph->Handle(), &c, 1, limits, nullptr, &_dec) != WFST_OK) Fatal("wfst_decoder_create");
}
GpuLatticeDecoder::GpuLatticeDecoder(Fst *graph, const LatticeFasterDecoderConfig &config, ArpaLm *oldlm, ArpaLm *newlm,
                                     const wfst_limits *limits)
    : _dec(nullptr), _stride(0), _rows_ready(0), _inited(false) {
  config.Check();
  wfst_config c = config.ToC();
  if (!oldlm || !newlm) throw std::runtime_error("biglm decoder needs both LMs");
  if (wfst_decoder_create_biglm(graph->Handle(), &c, 1, limits, nullptr, oldlm->Handle(), newlm->Handle(), nullptr, &_dec) != WFST_OK)
    Fatal("wfst_decoder_create_biglm");
}
GpuLatticeDecoder::~GpuLatticeDecoder() { wfst_decoder_free(_dec); }

void GpuLatticeDecoder::InitDecoding() {
  if (wfst_decoder_init(_dec, nullptr, 0) != WFST_OK) Fatal("InitDecoding");
  _rows.clear();
  _rows_ready = 0;
  _stride = 0;
  _inited = true;
}

void GpuLatticeDecoder::Pull(AmInterface *d) {
  const int ready = d->NumFramesReady();
  const int stride = d->NumIndices() + 1;
  if (_stride == 0) _stride = stride;
  if (stride != _stride) throw std::runtime_error("decodable changed NumIndices() within an utterance");
  if (ready <= _rows_ready) return;
  _rows.resize((size_t)ready * _stride);
  if (MatrixDecodable *m = dynamic_cast<MatrixDecodable *>(d)) {
    if (m->Stride() != _stride) throw std::runtime_error("MatrixDecodable::Stride() != NumIndices()+1");
    memcpy(&_rows[(size_t)_rows_ready * _stride], m->HostRows() + (size_t)_rows_ready * _stride,
           (size_t)(ready - _rows_ready) * _stride * sizeof(float));
  } else {
    for (int f = _rows_ready; f < ready; ++f) {
      float *row = &_rows[(size_t)f * _stride];
      row[0] = 0.0f;
      for (int i = 1; i < _stride; ++i) row[i] = d->LogLikelihood(f, i);
    }
  }
  _rows_ready = ready;
}

void GpuLatticeDecoder::AdvanceDecoding(AmInterface *decodable, int32 max_num_frames) {
  if (!_inited) throw std::runtime_error("You must call InitDecoding() before AdvanceDecoding");
  Pull(decodable);
  const float *rows = _rows.data();
  int32_t ready = _rows_ready;
  if (ready == 0) return;
  if (wfst_decoder_advance_host(_dec, nullptr, 0, &rows, &ready, _stride, max_num_frames) != WFST_OK)
    Fatal("AdvanceDecoding");
}

BaseFloat GpuLatticeDecoder::ProcessEmitting(AmInterface *decodable) {
  AdvanceDecoding(decodable, 1);
  return 0.0f;
}

void GpuLatticeDecoder::FinalizeDecoding() {
  if (wfst_decoder_finalize(_dec, nullptr, 0) != WFST_OK) Fatal("FinalizeDecoding");
}

int32 GpuLatticeDecoder::NumFramesDecoded() const { return wfst_decoder_num_frames_decoded(_dec, 0); }

bool GpuLatticeDecoder::Decode(AmInterface *decodable) {
  InitDecoding();
  AdvanceDecoding(decodable);
  FinalizeDecoding();
  Lattice tmp;
  return GetBestPath(&tmp, true);
}

static void WarnIfDegraded(wfst_decoder *dec, int channel);

bool GpuLatticeDecoder::GetBestPath(Lattice *ofst, bool use_final_probs) {
  ofst->DeleteStates();
  int cap = 4 * std::max(1, NumFramesDecoded()) + 64;
  for (int attempt = 0; attempt < 2; ++attempt) {
    std::vector<int32_t> il(cap), ol(cap);
    std::vector<float> g(cap), ac(cap);
    int32_t n = 0;
    int rc = wfst_decoder_get_best_path(_dec, nullptr, 0, use_final_probs ? 1 : 0, cap, il.data(), ol.data(), g.data(),
                                        ac.data(), &n);
    if (rc == WFST_E_CAPACITY && n > cap) { cap = n; continue; }
    if (rc == WFST_E_STATE) throw std::runtime_error(wfst_last_error());  // reference: LOG_ERR
    if (rc != WFST_OK) Fatal("GetBestPath");
    WarnIfDegraded(_dec, 0);
    if (n == 0) { Warn("No final token found."); return false; }
    HopsToLattice(il.data(), ol.data(), g.data(), ac.data(), n, ofst);
    return true;
  }
  return false;
}

// GetRawLattice (base-inl.h:869-975) of one channel through the C ABI.  Served after
// FinalizeDecoding by a decoder created in lattice mode (wfst_limits.lattice_links > 0).
static bool RawLatticeOfChannel(wfst_decoder *dec, int channel, Lattice *ofst, bool use_final_probs) {
  ofst->DeleteStates();
  int32_t ns = 0, na = 0;
  int rc = wfst_decoder_get_raw_lattice(dec, channel, use_final_probs ? 1 : 0, 0, 0, &ns, &na, nullptr, nullptr, nullptr,
                                        nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (rc == WFST_E_STATE) { Warn(wfst_last_error()); return false; }
  if (rc != WFST_OK && !(rc == WFST_E_CAPACITY && ns > 0)) Fatal("GetRawLattice");
  if (ns == 0) {
    if (!use_final_probs)  // base-inl.h:879-884
      Warn("You cannot call FinalizeDecoding() and then call GetRawLattice() with use_final_probs == false");
    return false;
  }
  std::vector<int32_t> fin(ns), src(na), dst(na), il(na), ol(na);
  std::vector<float> g(na), ac(na);
  if (wfst_decoder_get_raw_lattice(dec, channel, 1, ns, na, &ns, &na, fin.data(), nullptr, nullptr, nullptr, src.data(),
                                   dst.data(), il.data(), ol.data(), g.data(), ac.data()) != WFST_OK)
    Fatal("GetRawLattice");
  for (int s = 0; s < ns; ++s) {
    StateId id = ofst->AddState();
    if (fin[s]) ofst->SetFinal(id);
  }
  ofst->SetStart(0);
  for (int k = 0; k < na; ++k) ofst->AddArc(src[k], LatticeArc(il[k], ol[k], dst[k], LatticeWeight(g[k], ac[k])));
  return ofst->NumStates() > 0;
}

// GetLattice (base-inl.h:850-866) of one channel: the determinized lattice, built on the device.
static bool DetLatticeOfChannel(wfst_decoder *dec, int channel, Lattice *ofst, bool use_final_probs) {
  ofst->DeleteStates();
  int32_t ns = 0, na = 0;
  int rc = wfst_decoder_get_determinized_lattice(dec, channel, use_final_probs ? 1 : 0, 0, 0, &ns, &na, nullptr, nullptr, nullptr,
                                                 nullptr, nullptr, nullptr, nullptr);
  if (rc == WFST_E_STATE) { Warn(wfst_last_error()); return false; }
  if (rc != WFST_OK && !(rc == WFST_E_CAPACITY && ns > 0)) Fatal("GetLattice");
  if (ns == 0) return false;
  std::vector<int32_t> fin(ns), src(na), dst(na), il(na), ol(na);
  std::vector<float> g(na), ac(na);
  if (wfst_decoder_get_determinized_lattice(dec, channel, use_final_probs ? 1 : 0, ns, na, &ns, &na, fin.data(), src.data(), dst.data(),
                                            il.data(), ol.data(), g.data(), ac.data()) != WFST_OK)
    Fatal("GetLattice");
  for (int s = 0; s < ns; ++s) {
    StateId id = ofst->AddState();
    if (fin[s]) ofst->SetFinal(id);
  }
  ofst->SetStart(0);
  for (int k = 0; k < na; ++k) ofst->AddArc(src[k], LatticeArc(il[k], ol[k], dst[k], LatticeWeight(g[k], ac[k])));
  return true;
}

// GetLattice under --use-second (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:53-78): determinized lattice o old LM (scale -1) o new LM,
// ComposeLattice twice on the device.
static bool RescoredLatticeOfChannel(wfst_decoder *dec, int channel, Lattice *ofst, ArpaLm *oldlm, ArpaLm *newlm, bool use_final_probs) {
  ofst->DeleteStates();
  if (!oldlm || !newlm) throw std::runtime_error("second-pass GetLattice needs both LMs");
  int32_t ns = 0, na = 0;
  int rc = wfst_decoder_get_rescored_lattice(dec, channel, use_final_probs ? 1 : 0, oldlm->Handle(), newlm->Handle(), 0, 0, &ns, &na, nullptr,
                                             nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (rc == WFST_E_STATE) { Warn(wfst_last_error()); return false; }
  if (rc != WFST_OK && !(rc == WFST_E_CAPACITY && ns > 0)) Fatal("GetLattice (second pass)");
  if (ns == 0) return false;
  std::vector<int32_t> fin(ns), src(na), dst(na), il(na), ol(na);
  std::vector<float> g(na), ac(na);
  if (wfst_decoder_get_rescored_lattice(dec, channel, use_final_probs ? 1 : 0, oldlm->Handle(), newlm->Handle(), ns, na, &ns, &na, fin.data(),
                                        src.data(), dst.data(), il.data(), ol.data(), g.data(), ac.data()) != WFST_OK)
    Fatal("GetLattice (second pass)");
  for (int s = 0; s < ns; ++s) {
    StateId id = ofst->AddState();
    if (fin[s]) ofst->SetFinal(id);
  }
  ofst->SetStart(0);
  for (int k = 0; k < na; ++k) ofst->AddArc(src[k], LatticeArc(il[k], ol[k], dst[k], LatticeWeight(g[k], ac[k])));
  return true;
}

static bool ShortlistOfChannel(wfst_decoder *dec, int channel, std::vector<Lattice> &out, int n) {
  out.clear();
  if (n <= 0) return false;
  const int max_words = 1024;
  int32_t np = 0;
  std::vector<int32_t> nw((size_t)n), words((size_t)n * max_words);
  std::vector<float> tot((size_t)n), lm((size_t)n);
  const int32_t ch = channel;
  int rc = wfst_decoder_get_nbest(dec, &ch, 1, n, max_words, &np, nw.data(), words.data(), tot.data(), lm.data());
  if (rc == WFST_E_STATE) { Warn(wfst_last_error()); return false; }
  if (rc != WFST_OK) Fatal("GetNbest");
  for (int k = 0; k < np; ++k) {
    Lattice lat;
    StateId cur = lat.AddState();
    lat.SetStart(cur);
    const int L = std::min(nw[k], max_words);
    // the path weight rides on the first arc (an <eps> arc when the path has no word)
    for (int j = 0; j < std::max(L, 1); ++j) {
      StateId next = lat.AddState();
      const LatticeWeight w = j == 0 ? LatticeWeight(lm[k], tot[k] - lm[k]) : LatticeWeight(0.0f, 0.0f);
      lat.AddArc(cur, LatticeArc(0, L ? words[(size_t)k * max_words + j] : 0, next, w));
      cur = next;
    }
    lat.SetFinal(cur);
    out.push_back(lat);
  }
  return !out.empty();
}

// GetNbest as the service defines it: NShortestPath over GetLattice's result, every path a linear lattice shaped as
// ConvertNbestToVector leaves it (newfst/lattice-to-nbest.cc:149-199): an <eps> arc of weight One in front (the start state the
// second Reverse adds), the lattice's arcs, the final weight's arc, the super-final state's arc and the first Reverse's <eps>.
static bool NbestOfChannel(wfst_decoder *dec, int channel, std::vector<Lattice> &out, int n, ArpaLm *oldlm, ArpaLm *newlm) {
  out.clear();
  if (n <= 0) return false;
  if ((oldlm == nullptr) != (newlm == nullptr)) throw std::runtime_error("second-pass GetNbest needs both LMs");
  const wfst_lm *l1 = oldlm ? oldlm->Handle() : nullptr, *l2 = newlm ? newlm->Handle() : nullptr;
  int32_t np = 0, na = 0;
  std::vector<int32_t> off((size_t)n + 1), ol((size_t)n * 128);
  std::vector<float> tot((size_t)n), g(ol.size()), ac(ol.size());
  int rc = wfst_decoder_get_nbest_paths(dec, channel, n, 1, l1, l2, n, (int32_t)ol.size(), &np, &na, off.data(), tot.data(), ol.data(), g.data(), ac.data());
  if (rc == WFST_E_CAPACITY && na > (int32_t)ol.size()) {
    ol.resize((size_t)na); g.resize((size_t)na); ac.resize((size_t)na);
    rc = wfst_decoder_get_nbest_paths(dec, channel, n, 1, l1, l2, n, na, &np, &na, off.data(), tot.data(), ol.data(), g.data(), ac.data());
  }
  if (rc == WFST_E_STATE) { Warn(wfst_last_error()); return false; }
  if (rc != WFST_OK) Fatal("GetNbest");
  for (int k = 0; k < np; ++k) {
    Lattice lat;
    StateId cur = lat.AddState();
    lat.SetStart(cur);
    auto add = [&](int word, float w1, float w2) {
      StateId next = lat.AddState();
      lat.AddArc(cur, LatticeArc(0, word, next, LatticeWeight(w1, w2)));
      cur = next;
    };
    add(0, 0.0f, 0.0f);
    for (int j = off[k]; j < off[k + 1]; ++j) add(ol[j], g[j], ac[j]);
    add(0, 0.0f, 0.0f);
    add(0, 0.0f, 0.0f);
    lat.SetFinal(cur);
    out.push_back(lat);
  }
  return !out.empty();
}

bool GpuLatticeDecoder::GetNbest(std::vector<Lattice> &nbest_paths, int n) { return NbestOfChannel(_dec, 0, nbest_paths, n, nullptr, nullptr); }
bool GpuLatticeDecoder::GetNbest(std::vector<Lattice> &nbest_paths, int n, ArpaLm *oldlm, ArpaLm *newlm) {
  return NbestOfChannel(_dec, 0, nbest_paths, n, oldlm, newlm);
}
bool GpuLatticeDecoder::GetNbestShortlist(std::vector<Lattice> &nbest_paths, int n) { return ShortlistOfChannel(_dec, 0, nbest_paths, n); }

bool GpuLatticeDecoder::GetLattice(Lattice *ofst, bool use_final_probs) { return DetLatticeOfChannel(_dec, 0, ofst, use_final_probs); }
bool GpuLatticeDecoder::GetLattice(Lattice *ofst, ArpaLm *oldlm, ArpaLm *newlm, bool use_final_probs) {
  return RescoredLatticeOfChannel(_dec, 0, ofst, oldlm, newlm, use_final_probs);
}

bool GpuLatticeDecoder::GetRawLattice(Lattice *ofst, bool use_final_probs) {
  return RawLatticeOfChannel(_dec, 0, ofst, use_final_probs);
}

// ---- batch decoder ------------------------------------------------------------------------------
void GpuBatchDecoder::GetRawLattices(const std::vector<int> &channels, std::vector<Lattice> *ofsts, std::vector<bool> *ok,
                                     bool use_final_probs, int threads) {
  std::vector<int> ch(channels);
  if (ch.empty())
    for (int c = 0; c < _n; ++c) ch.push_back(c);
  const size_t n = ch.size();
  ofsts->assign(n, Lattice());
  std::vector<char> good(n, 0);
  if (n == 0) { ok->clear(); return; }
  good[0] = RawLatticeOfChannel(_dec, ch[0], &(*ofsts)[0], use_final_probs);  // fetches every finalized channel's lists
  int nt = threads > 0 ? threads : (int)std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
  nt = (int)std::min<size_t>((size_t)nt, n);
  std::atomic<size_t> next(1);
  std::vector<std::string> errors((size_t)nt);
  auto work = [&](int k) {
    try {
      for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1))
        good[i] = RawLatticeOfChannel(_dec, ch[i], &(*ofsts)[i], use_final_probs);
    } catch (const std::exception &e) {
      errors[(size_t)k] = e.what();
    }
  };
  std::vector<std::thread> pool;
  for (int k = 1; k < nt; ++k) pool.emplace_back(work, k);
  work(0);
  for (std::thread &t : pool) t.join();
  for (const std::string &e : errors)
    if (!e.empty()) throw std::runtime_error(e);
  ok->assign(good.begin(), good.end());
}
bool GpuBatchDecoder::GetNbest(int channel, std::vector<Lattice> &nbest_paths, int n) {
  return NbestOfChannel(_dec, channel, nbest_paths, n, nullptr, nullptr);
}
bool GpuBatchDecoder::GetNbest(int channel, std::vector<Lattice> &nbest_paths, int n, ArpaLm *oldlm, ArpaLm *newlm) {
  return NbestOfChannel(_dec, channel, nbest_paths, n, oldlm, newlm);
}
bool GpuBatchDecoder::GetNbestShortlist(int channel, std::vector<Lattice> &nbest_paths, int n) {
  return ShortlistOfChannel(_dec, channel, nbest_paths, n);
}
bool GpuBatchDecoder::GetLattice(int channel, Lattice *ofst, bool use_final_probs) {
  return DetLatticeOfChannel(_dec, channel, ofst, use_final_probs);
}
bool GpuBatchDecoder::GetLattice(int channel, Lattice *ofst, ArpaLm *oldlm, ArpaLm *newlm, bool use_final_probs) {
  return RescoredLatticeOfChannel(_dec, channel, ofst, oldlm, newlm, use_final_probs);
}
bool GpuBatchDecoder::GetRawLattice(int channel, Lattice *ofst, bool use_final_probs) {
  return RawLatticeOfChannel(_dec, channel, ofst, use_final_probs);
}
void GpuBatchDecoder::PrefetchLattices() {
  if (wfst_decoder_prefetch_determinized(_dec) != WFST_OK) Fatal("wfst_decoder_prefetch_determinized");
}
void GpuBatchDecoder::PrefetchLatticesDetached() {
  if (wfst_decoder_prefetch_determinized_detached(_dec) != WFST_OK) Fatal("wfst_decoder_prefetch_determinized_detached");
}
void GpuBatchDecoder::HarvestPrefetchedLattices() {
  if (wfst_decoder_harvest_prefetched(_dec) != WFST_OK) Fatal("wfst_decoder_harvest_prefetched");
}
bool GpuBatchDecoder::GetPrefetchedLattice(int channel, Lattice *ofst) {
  ofst->DeleteStates();
  int32_t ns = 0, na = 0;
  int rc = wfst_decoder_get_prefetched_lattice(_dec, channel, 0, 0, &ns, &na, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (rc == WFST_E_STATE) { Warn(wfst_last_error()); return false; }
  if (rc != WFST_OK && !(rc == WFST_E_CAPACITY && ns > 0)) Fatal("GetPrefetchedLattice");
  if (ns == 0) return false;
  std::vector<int32_t> fin(ns), src(na), dst(na), il(na), ol(na);
  std::vector<float> g(na), ac(na);
  if (wfst_decoder_get_prefetched_lattice(_dec, channel, ns, na, &ns, &na, fin.data(), src.data(), dst.data(), il.data(), ol.data(), g.data(),
                                          ac.data()) != WFST_OK)
    Fatal("GetPrefetchedLattice");
  for (int s = 0; s < ns; ++s) {
    StateId id = ofst->AddState();
    if (fin[s]) ofst->SetFinal(id);
  }
  ofst->SetStart(0);
  for (int k = 0; k < na; ++k) ofst->AddArc(src[k], LatticeArc(il[k], ol[k], dst[k], LatticeWeight(g[k], ac[k])));
  return true;
}
void GpuBatchDecoder::GetLattices(const std::vector<int> &channels, std::vector<Lattice> *ofsts, std::vector<bool> *ok, ArpaLm *oldlm,
                                  ArpaLm *newlm, bool use_final_probs) {
  if (!oldlm || !newlm) throw std::runtime_error("second-pass GetLattice needs both LMs");
  std::vector<int32_t> ch(channels.begin(), channels.end());
  const int rc = wfst_decoder_rescore_lattices(_dec, ch.empty() ? nullptr : ch.data(), (int32_t)ch.size(), use_final_probs ? 1 : 0,
                                               oldlm->Handle(), newlm->Handle());
  if (rc == WFST_E_STATE) Warn(wfst_last_error());   // (a channel that is not finalized: the per-channel calls below serve it)
  else if (rc != WFST_OK) Fatal("GetLattices");
  ofsts->assign(channels.size(), Lattice());
  ok->assign(channels.size(), false);
  for (size_t i = 0; i < channels.size(); ++i) (*ok)[i] = RescoredLatticeOfChannel(_dec, channels[i], &(*ofsts)[i], oldlm, newlm, use_final_probs);
}
void GpuBatchDecoder::GetNbests(const std::vector<int> &channels, std::vector<std::vector<Lattice> > *nbests, std::vector<bool> *ok, int n,
                                ArpaLm *oldlm, ArpaLm *newlm) {
  if ((oldlm == nullptr) != (newlm == nullptr)) throw std::runtime_error("second-pass GetNbest needs both LMs");
  nbests->assign(channels.size(), std::vector<Lattice>());
  ok->assign(channels.size(), false);
  if (n <= 0) return;
  std::vector<int32_t> ch(channels.begin(), channels.end());
  if (n <= 4096) {
    const int rc = wfst_decoder_nbest_paths_batch(_dec, ch.empty() ? nullptr : ch.data(), (int32_t)ch.size(), n, 1,
                                                  oldlm ? oldlm->Handle() : nullptr, newlm ? newlm->Handle() : nullptr);
    if (rc == WFST_E_STATE || rc == WFST_E_CAPACITY) Warn(wfst_last_error());   // (the per-channel calls below serve what the batch could not)
    else if (rc != WFST_OK) Fatal("GetNbests");
  }
  for (size_t i = 0; i < channels.size(); ++i) (*ok)[i] = NbestOfChannel(_dec, channels[i], (*nbests)[i], n, oldlm, newlm);
}
GpuBatchDecoder::GpuBatchDecoder(Fst *graph, const LatticeFasterDecoderConfig &config, int n_channels,
                                 const wfst_limits *limits, void *hip_stream)
    : _dec(nullptr), _n(n_channels) {
  config.Check();
  wfst_config c = config.ToC();
  if (wfst_decoder_create(graph->Handle(), &c, n_channels, limits, hip_stream, &_dec) != WFST_OK)
    Fatal("wfst_decoder_create");
}
GpuBatchDecoder::GpuBatchDecoder(Fst *graph, const LatticeFasterDecoderConfig &config, ArpaLm *oldlm, ArpaLm *newlm,
                                 int n_channels, const wfst_limits *limits, void *hip_stream)
    : _dec(nullptr), _n(n_channels) {
  config.Check();
  wfst_config c = config.ToC();
  if (!oldlm || !newlm) throw std::runtime_error("biglm decoder needs both LMs");
  if (wfst_decoder_create_biglm(graph->Handle(), &c, n_channels, limits, nullptr, oldlm->Handle(), newlm->Handle(), hip_stream,
                                &_dec) != WFST_OK)
    Fatal("wfst_decoder_create_biglm");
}
GpuBatchDecoder::~GpuBatchDecoder() { wfst_decoder_free(_dec); }

void GpuBatchDecoder::InitDecoding(const std::vector<int> &ch) {
  if (wfst_decoder_init(_dec, ch.empty() ? nullptr : ch.data(), (int)ch.size()) != WFST_OK) Fatal("InitDecoding");
}
void GpuBatchDecoder::AdvanceDecoding(const std::vector<int> &ch, const std::vector<const float *> &ll,
                                      const std::vector<int> &ready, int stride, int max_num_frames) {
  if (wfst_decoder_advance(_dec, ch.empty() ? nullptr : ch.data(), (int)ch.size(), ll.data(), ready.data(), stride,
                           max_num_frames) != WFST_OK)
    Fatal("AdvanceDecoding");
}
void GpuBatchDecoder::AdvanceDecodingHost(const std::vector<int> &ch, const std::vector<const float *> &ll,
                                          const std::vector<int> &ready, int stride, int max_num_frames) {
  if (wfst_decoder_advance_host(_dec, ch.empty() ? nullptr : ch.data(), (int)ch.size(), ll.data(), ready.data(),
                                stride, max_num_frames) != WFST_OK)
    Fatal("AdvanceDecoding");
}
void GpuBatchDecoder::FinalizeDecoding(const std::vector<int> &ch) {
  if (wfst_decoder_finalize(_dec, ch.empty() ? nullptr : ch.data(), (int)ch.size()) != WFST_OK)
    Fatal("FinalizeDecoding");
}
int GpuBatchDecoder::NumFramesDecoded(int channel) const { return wfst_decoder_num_frames_decoded(_dec, channel); }

// A best-path decoder does not fail at wfst_limits.max_tokens_per_frame, it goes on from the limit-th cheapest token (the limit
// acts as a max_active): the result may then differ from the reference's at the configured beam.  Said once per utterance and
// channel, where the reference would have said nothing because it has no such limit.
static void WarnIfDegraded(wfst_decoder *dec, int channel) {
  int32_t n = 0;
  if (wfst_decoder_get_degraded_frames(dec, channel, &n) == WFST_OK && n > 0)
    Warn("channel " + std::to_string(channel) + ": " + std::to_string(n) + " frame(s) held more tokens than max_tokens_per_frame; the search "
         "went on from the cheapest of them (a max_active): raise wfst_limits.max_tokens_per_frame for the result at the configured beam");
}

void GpuBatchDecoder::GetBestPaths(const std::vector<int> &channels, std::vector<Lattice> *ofsts,
                                   std::vector<bool> *ok, bool use_final_probs) {
  const int cnt = channels.empty() ? _n : (int)channels.size();
  int maxf = 1;
  for (int i = 0; i < cnt; ++i) maxf = std::max(maxf, NumFramesDecoded(channels.empty() ? i : channels[i]));
  int cap = 4 * maxf + 64;
  ofsts->assign(cnt, Lattice());
  ok->assign(cnt, false);
  for (int attempt = 0; attempt < 2; ++attempt) {
    std::vector<int32_t> il((size_t)cnt * cap), ol((size_t)cnt * cap), n(cnt);
    std::vector<float> g((size_t)cnt * cap), ac((size_t)cnt * cap);
    int rc = wfst_decoder_get_best_path(_dec, channels.empty() ? nullptr : channels.data(), (int)channels.size(),
                                        use_final_probs ? 1 : 0, cap, il.data(), ol.data(), g.data(), ac.data(), n.data());
    if (rc == WFST_E_CAPACITY && *std::max_element(n.begin(), n.end()) > cap) {
      cap = *std::max_element(n.begin(), n.end());
      continue;
    }
    if (rc == WFST_E_STATE) throw std::runtime_error(wfst_last_error());
    if (rc != WFST_OK) Fatal("GetBestPath");
    for (int i = 0; i < cnt; ++i) {
      WarnIfDegraded(_dec, channels.empty() ? i : channels[i]);
      if (n[i] == 0) continue;
      HopsToLattice(&il[(size_t)i * cap], &ol[(size_t)i * cap], &g[(size_t)i * cap], &ac[(size_t)i * cap], n[i],
                    &(*ofsts)[i]);
      (*ok)[i] = true;
    }
    return;
  }
}

bool GpuBatchDecoder::GetBestPath(int channel, Lattice *ofst, bool use_final_probs) {
  std::vector<Lattice> l;
  std::vector<bool> ok;
  GetBestPaths(std::vector<int>(1, channel), &l, &ok, use_final_probs);
  *ofst = l[0];
  return ok[0];
}

}  // namespace datemoon
