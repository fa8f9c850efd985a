// Implementation of wfst-host.h: thin C++ over the C ABI.  No decoding happens on the host.
#include "wfst-host.h"

#include <atomic>
#include <thread>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

namespace datemoon {

namespace {
[[noreturn]] void Fatal(const std::string &what) { throw std::runtime_error(what + ": " + wfst_last_error()); }
// (one stdio call per line: the worker threads of a service share stderr, and a line put together by several << would interleave)
void Warn(const std::string &msg) { const std::string line = "WARNING (wfst) " + msg + "\n"; fputs(line.c_str(), stderr); fflush(stderr); }

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
  _tid2pdf = m;
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

// ---- channel pool: many decoder objects, one batched device decoder ------------------------------
GpuChannelPool::GpuChannelPool(Fst *graph, const LatticeFasterDecoderConfig &config, int n_channels, const wfst_limits *limits, int linger_us)
    : _dec(nullptr), _graph(graph), _n(n_channels), _linger_us(linger_us), _n_leased(0), _stop(false), _stats() {
  config.Check();
  wfst_config c = config.ToC();
  if (n_channels < 1) throw std::runtime_error("GpuChannelPool needs at least one channel");
  if (wfst_decoder_create(graph->Handle(), &c, n_channels, limits, nullptr, &_dec) != WFST_OK) Fatal("wfst_decoder_create");
  Start(linger_us);
}
GpuChannelPool::GpuChannelPool(Fst *graph, const LatticeFasterDecoderConfig &config, ArpaLm *oldlm, ArpaLm *newlm, int n_channels,
                               const wfst_limits *limits, int linger_us)
    : _dec(nullptr), _graph(graph), _n(n_channels), _linger_us(linger_us), _n_leased(0), _stop(false), _stats() {
  config.Check();
  wfst_config c = config.ToC();
  if (n_channels < 1) throw std::runtime_error("GpuChannelPool needs at least one channel");
  if (!oldlm || !newlm) throw std::runtime_error("biglm decoder needs both LMs");
  if (wfst_decoder_create_biglm(graph->Handle(), &c, n_channels, limits, nullptr, oldlm->Handle(), newlm->Handle(), nullptr, &_dec) != WFST_OK)
    Fatal("wfst_decoder_create_biglm");
  Start(linger_us);
}
void GpuChannelPool::Start(int linger_us) {
  _linger_us = std::max(0, linger_us);
  _leased.assign((size_t)_n, 0);
  if (const char *tf = getenv("WFST_POOL_TRACE")) _trace_file = tf;
  _t_origin = _t_first = std::chrono::steady_clock::now();
  _thread = std::thread([this] { Run(); });
}
GpuChannelPool::~GpuChannelPool() {
  {
    std::lock_guard<std::mutex> lk(_mu);
    _stop = true;
  }
  _cv_work.notify_all();
  if (_thread.joinable()) _thread.join();
  wfst_decoder_free(_dec);   // (waits for the device: no row of the slab is on its way any more)
  if (_slab) wfst_host_free(_slab);
  if (!_trace_file.empty()) {
    if (FILE *f = fopen(_trace_file.c_str(), "w")) {
      fprintf(f, "# ms since the pool started: first request, batch closed, pass done | requests init advance finalize best-path calls | ms of each | device busy at close\n");
      for (const TracePass &t : _trace)
        fprintf(f, "%.3f %.3f %.3f | %d %d %d %d %d | %.3f %.3f %.3f %.3f %.3f | %d\n", t.t_first, t.t_closed, t.t_end, t.n[0], t.n[1], t.n[2], t.n[3], t.n[4],
                t.ms[0], t.ms[1], t.ms[2], t.ms[3], t.ms[4], t.busy);
      fclose(f);
    }
  }
}
float *GpuChannelPool::RowSlot(int channel, size_t floats, size_t *slot_floats) {
  std::lock_guard<std::mutex> lk(_slab_mu);
  if (!_slab) {
    if (_slab_pitch != 0) return nullptr;   // (tried before: no page-locked memory of that size)
    _slab_pitch = (std::max<size_t>(floats, (size_t)1 << 16) + 1023) & ~(size_t)1023;
    _slab = (float *)wfst_host_alloc((size_t)_n * _slab_pitch * sizeof(float));
    if (!_slab) return nullptr;
  }
  if (floats > _slab_pitch || channel < 0 || channel >= _n) return nullptr;
  *slot_floats = _slab_pitch;
  return _slab + (size_t)channel * _slab_pitch;
}
GpuChannelPool::Stats GpuChannelPool::GetStats() {
  std::lock_guard<std::mutex> lk(_mu);
  return _stats;
}
int GpuChannelPool::Lease() {
  std::unique_lock<std::mutex> lk(_mu);
  for (;;) {
    for (int c = 0; c < _n; ++c)
      if (!_leased[(size_t)c]) { _leased[(size_t)c] = 1; ++_n_leased; return c; }
    _cv_free.wait(lk);   // (more decoder objects than channels: this one waits for a destructor)
  }
}
int GpuChannelPool::TryLease() {
  std::lock_guard<std::mutex> lk(_mu);
  for (int c = 0; c < _n; ++c)
    if (!_leased[(size_t)c]) { _leased[(size_t)c] = 1; ++_n_leased; return c; }
  return -1;
}
void GpuChannelPool::Release(int c) {
  {
    std::lock_guard<std::mutex> lk(_mu);
    if (c >= 0 && c < _n && _leased[(size_t)c]) { _leased[(size_t)c] = 0; --_n_leased; }
  }
  _cv_free.notify_one();
}
void GpuChannelPool::Submit(Request *r) {
  {
    std::lock_guard<std::mutex> lk(_mu);
    if (_stop) throw std::runtime_error("GpuChannelPool is shutting down");
    r->done = false;
    r->error = nullptr;
    if (_queue.empty() && !_trace_file.empty()) _t_first = std::chrono::steady_clock::now();
    _queue.push_back(r);
    // (the batcher sleeps without a time-out only on an empty queue; while a batch forms it looks again every few tens of
    // microseconds -- it is woken for the first request and for the one that completes the batch, not sixty-four times)
    if (_queue.size() == 1 || (int)_queue.size() + _n_bp_outstanding >= _n_leased) _cv_work.notify_one();
  }
  {
    std::unique_lock<std::mutex> lk(r->m);
    r->cv.wait(lk, [&] { return r->done; });
  }
  if (r->error) std::rethrow_exception(r->error);
}
void GpuChannelPool::Done(Request *r) {
  std::lock_guard<std::mutex> lk(r->m);
  r->done = true;
  r->cv.notify_one();   // (under the request's mutex: its thread cannot leave Submit -- and destroy the request -- before this returns)
}
void GpuChannelPool::Finish(std::vector<Request *> &rs) {
  for (Request *r : rs) r->decoded = wfst_decoder_num_frames_decoded(_dec, r->channel);
  for (Request *r : rs) Done(r);
}
void GpuChannelPool::Run() {
  std::unique_lock<std::mutex> lk(_mu);
  for (;;) {
    const auto t_wait = std::chrono::steady_clock::now();
    // wait for requests; a best-path list on the device is looked after meanwhile (its results are taken as soon as they have landed)
    while (!_stop && _queue.empty()) {
      if (_bp_flight.empty() && _bp_wait.empty()) { _cv_work.wait(lk); continue; }
      lk.unlock();
      const bool progressed = PollBestPaths(false);
      if (!progressed && _bp_flight.empty()) StartBestPaths();
      lk.lock();
      if (!progressed && _queue.empty() && !_stop) _cv_work.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(30));
    }
    if (_queue.empty()) {   // (_stop: what is on the device is taken, what waits is served, then out)
      lk.unlock();
      while (!_bp_flight.empty() || !_bp_wait.empty()) { PollBestPaths(true); StartBestPaths(); }
      return;
    }
    // the other leased channels' requests are on their way more often than not (their threads were released together): a short
    // wait makes one batch of them instead of two.  Threads that wait for a best path (on the device, or in line for it) are not
    // coming: they count as arrived.
    // ... and while the device has a backlog, requests go on joining (two cohorts of threads that alternate merge into one; a
    // frame of 40 channels takes the device as long as a frame of 64): wfst_decoder_calls_in_flight counts the advance calls not
    // yet finished -- with three or more outstanding the batch may grow for nothing, below that it goes (the device is a call or
    // two from running dry, and this batch's rows have to get there first)
    // ... and with three advance calls outstanding the batch waits in any case: a fourth would stand in wfst_decoder_advance_host
    // until the device has caught up (its staging sets are used in rotation), and the batcher with it -- finished utterances' best
    // paths would lie on the device untaken, their threads idle
    auto arrived = [&] { return (int)(_queue.size() + _bp_wait.size() + _bp_flight.size()); };
    {
      const auto t_first = std::chrono::steady_clock::now();
      for (;;) {
        if (_stop) break;
        // (a batch that is at least half full waits ten times as long for the rest: threads released together come back spread over
        // a few hundred microseconds, and cut in two they stay two cohorts -- a device call of half the channels takes as long as one
        // of all)
        const auto waited = std::chrono::steady_clock::now() - t_first;
        const bool lingered = waited >= std::chrono::microseconds(_linger_us) &&
                              (2 * arrived() < _n_leased || waited >= std::chrono::microseconds(10 * (long long)_linger_us));
        if (lingered || arrived() >= _n_leased) {
          lk.unlock();
          const int depth = wfst_decoder_calls_in_flight(_dec);
          if (!_bp_flight.empty()) PollBestPaths(false);
          lk.lock();
          if (depth < 3) break;   // (2: the same, measured)
        }
        // (wait_until on the SYSTEM clock = pthread_cond_timedwait: what ThreadSanitizer's runtime intercepts -- a steady-clock wait is
        // pthread_cond_clockwait, which gcc 11's does not, and the tool then loses the mutex's hand-over; a jump of the wall clock
        // costs one short wait at most, the loop looks at its conditions again either way)
        _cv_work.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(lingered ? 20 : std::max(1, _linger_us)));
      }
    }
    std::vector<Request *> batch;
    batch.swap(_queue);
    _stats.ms_waiting += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_wait).count();
    const auto t_first = _t_first;
    lk.unlock();
    if (!_trace_file.empty()) {
      auto ms = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(t - _t_origin).count(); };
      TracePass tp;
      memset(&tp, 0, sizeof(tp));
      tp.t_first = ms(t_first);
      tp.t_closed = ms(std::chrono::steady_clock::now());
      tp.busy = wfst_decoder_busy(_dec);
      for (Request *r : batch) tp.n[r->kind] += 1;
      _trace.push_back(tp);
    }
    Execute(batch);   // (releases every kind's requesters as soon as that kind is served: Finish)
    if (!_trace_file.empty()) _trace.back().t_end = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - _t_origin).count();
    lk.lock();
    _stats.batches += 1;
    _stats.requests += (long long)batch.size();
  }
}
// One pass over what has arrived.  A channel has at most one request in a batch (its thread waits for it), so the kinds can be
// served in any order: init, advance, finalize, best path, then the one-off calls.  A batched C-ABI call validates every listed
// channel before it enqueues anything: when it refuses the batch, the requests are retried one by one and each gets its own verdict.
void GpuChannelPool::Execute(std::vector<Request *> &batch) {
  std::vector<Request *> by_kind[kKinds];
  for (Request *r : batch) by_kind[r->kind].push_back(r);
  auto listed = [&](std::vector<Request *> &rs, const char *what, auto &&call) {
    if (rs.empty()) return;
    std::vector<int32_t> ch;
    for (Request *r : rs) ch.push_back(r->channel);
    if (call(ch.data(), (int32_t)ch.size()) == WFST_OK) return;
    if (rs.size() == 1) { rs[0]->error = std::make_exception_ptr(std::runtime_error(std::string(what) + ": " + wfst_last_error())); return; }
    for (Request *r : rs) {
      const int32_t c = r->channel;
      if (call(&c, 1) != WFST_OK) r->error = std::make_exception_ptr(std::runtime_error(std::string(what) + ": " + wfst_last_error()));
    }
  };
  double ms[kKinds] = {0, 0, 0, 0, 0};
  // a kind's requesters go on as soon as that kind is served: the threads whose chunks have just been enqueued pull their next
  // chunks while the batcher fetches other channels' best paths (which waits for the device)
  auto clocked = [&](int kind, auto &&f) {
    if (by_kind[kind].empty()) return;
    const auto t0 = std::chrono::steady_clock::now();
    f();
    ms[kind] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (Request *r : by_kind[kind]) r->decoded = wfst_decoder_num_frames_decoded(_dec, r->channel);
    for (Request *r : by_kind[kind]) Done(r);
  };
  clocked(kInit, [&] { listed(by_kind[kInit], "InitDecoding", [&](const int32_t *ch, int32_t n) { return wfst_decoder_init(_dec, ch, n); }); });
  clocked(kAdvance, [&] { ExecuteAdvance(by_kind[kAdvance]); });
  clocked(kFinalize, [&] { listed(by_kind[kFinalize], "FinalizeDecoding", [&](const int32_t *ch, int32_t n) { return wfst_decoder_finalize(_dec, ch, n); }); });
  // best paths: the requests join the waiting list; what is on the device is looked at, the next list is started -- the requesters
  // are released when THEIR list's results have landed (PollBestPaths), not at the end of this pass
  {
    const auto t0 = std::chrono::steady_clock::now();
    for (Request *r : by_kind[kBestPath]) _bp_wait.push_back(r);
    CountBestPaths();
    PollBestPaths(false);
    StartBestPaths();
    ms[kBestPath] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  clocked(kCall, [&] {
    for (Request *r : by_kind[kCall]) {
      try {
        r->call(_dec);
      } catch (...) {
        r->error = std::current_exception();
      }
    }
  });
  {
    std::lock_guard<std::mutex> lk(_mu);
    for (int k = 0; k < kKinds; ++k) _stats.ms_by_kind[k] += ms[k];
  }
  if (!_trace_file.empty() && !_trace.empty())
    for (int k = 0; k < kKinds; ++k) _trace.back().ms[k] = ms[k];
}
void GpuChannelPool::ExecuteAdvance(std::vector<Request *> &requests) {
  // one call per (stride, max_num_frames): in a service every stream has the same model, i.e. one call
  std::vector<Request *> all(requests);   // (consumed below; the caller's list is what it marks done)
  while (!all.empty()) {
    std::vector<Request *> rs, rest;
    for (Request *r : all) (r->stride == all[0]->stride && r->max_num_frames == all[0]->max_num_frames ? rs : rest).push_back(r);
    all.swap(rest);
    // (in channel order: the rows of consecutive channels are equally spaced slots of the pool's slab -- the library uploads such
    // runs as one 2-D copy each)
    std::sort(rs.begin(), rs.end(), [](const Request *x, const Request *y) { return x->channel < y->channel; });
    auto call = [&](std::vector<Request *> &q) {
      std::vector<int32_t> ch, ready;
      std::vector<const float *> rows;
      for (Request *r : q) { ch.push_back(r->channel); ready.push_back(r->ready); rows.push_back(r->rows); }
      return wfst_decoder_advance_host(_dec, ch.data(), (int32_t)ch.size(), rows.data(), ready.data(), q[0]->stride, q[0]->max_num_frames);
    };
    long long frames = 0;
    for (Request *r : rs) frames += std::max(0, r->ready - wfst_decoder_num_frames_decoded(_dec, r->channel));
    {
      std::lock_guard<std::mutex> lk(_mu);
      _stats.advance_calls += 1;
      _stats.advance_requests += (long long)rs.size();
      _stats.frames += frames;
    }
    if (call(rs) == WFST_OK) continue;
    if (rs.size() == 1) { rs[0]->error = std::make_exception_ptr(std::runtime_error(std::string("AdvanceDecoding: ") + wfst_last_error())); continue; }
    for (Request *r : rs) {
      std::vector<Request *> one(1, r);
      if (call(one) != WFST_OK) r->error = std::make_exception_ptr(std::runtime_error(std::string("AdvanceDecoding: ") + wfst_last_error()));
    }
  }
}
void GpuChannelPool::StartBestPaths() {
  if (!_bp_flight.empty() || _bp_wait.empty()) return;
  _bp_ufp = _bp_wait[0]->use_final_probs ? 1 : 0;
  std::vector<Request *> rest;
  for (Request *r : _bp_wait) ((r->use_final_probs ? 1 : 0) == _bp_ufp ? _bp_flight : rest).push_back(r);
  _bp_wait.swap(rest);
  std::vector<int32_t> ch;
  int maxf = 1;
  for (Request *r : _bp_flight) { ch.push_back(r->channel); maxf = std::max(maxf, wfst_decoder_num_frames_decoded(_dec, r->channel)); }
  _bp_cap = 4 * maxf + 64;
  if (wfst_decoder_best_path_enqueue(_dec, ch.data(), (int32_t)ch.size(), _bp_ufp, _bp_cap) != WFST_OK) {
    // refused (one of the channels: GetBestPath before InitDecoding ...): request by request, each its own verdict
    std::vector<Request *> rs;
    rs.swap(_bp_flight);
    CountBestPaths();
    ExecuteBestPath(rs);
    Finish(rs);
  }
}
// true: a list's results were taken (its requesters are released)
bool GpuChannelPool::PollBestPaths(bool block) {
  if (_bp_flight.empty()) return false;
  if (!block && wfst_decoder_best_path_ready(_dec) != 1) return false;
  const int cnt = (int)_bp_flight.size(), cap = _bp_cap;
  std::vector<int32_t> il((size_t)cnt * cap), ol((size_t)cnt * cap), n((size_t)cnt, 0);
  std::vector<float> g((size_t)cnt * cap), ac((size_t)cnt * cap);
  const int rc = wfst_decoder_best_path_fetch(_dec, il.data(), ol.data(), g.data(), ac.data(), n.data());
  std::vector<Request *> rs;
  rs.swap(_bp_flight);
  CountBestPaths();
  if (rc != WFST_OK) {
    // a path longer than the capacity, or one channel's device error: the synchronous path sorts it out request by request
    ExecuteBestPath(rs);
    Finish(rs);
    return true;
  }
  for (int i = 0; i < cnt; ++i) {
    Request *r = rs[(size_t)i];
    r->n_hops = n[(size_t)i];
    const size_t o = (size_t)i * cap;
    r->il.assign(il.begin() + (long)o, il.begin() + (long)o + r->n_hops);
    r->ol.assign(ol.begin() + (long)o, ol.begin() + (long)o + r->n_hops);
    r->g.assign(g.begin() + (long)o, g.begin() + (long)o + r->n_hops);
    r->ac.assign(ac.begin() + (long)o, ac.begin() + (long)o + r->n_hops);
    // (the final result says whether the per-frame token limit bound on the way; partial results do not stop for it)
    int32_t dg = 0;
    if (_bp_ufp && wfst_decoder_get_degraded_frames(_dec, r->channel, &dg) == WFST_OK) r->degraded = dg;
  }
  Finish(rs);
  return true;
}
void GpuChannelPool::ExecuteBestPath(std::vector<Request *> &all) {
  for (int ufp = 0; ufp < 2; ++ufp) {
    std::vector<Request *> rs;
    for (Request *r : all)
      if ((r->use_final_probs ? 1 : 0) == ufp) rs.push_back(r);
    if (rs.empty()) continue;
    const int cnt = (int)rs.size();
    std::vector<int32_t> ch((size_t)cnt);
    int maxf = 1;
    for (int i = 0; i < cnt; ++i) { ch[(size_t)i] = rs[(size_t)i]->channel; maxf = std::max(maxf, wfst_decoder_num_frames_decoded(_dec, ch[(size_t)i])); }
    int cap = 4 * maxf + 64;
    for (int attempt = 0; attempt < 2; ++attempt) {
      std::vector<int32_t> il((size_t)cnt * cap), ol((size_t)cnt * cap), n((size_t)cnt, 0);
      std::vector<float> g((size_t)cnt * cap), ac((size_t)cnt * cap);
      const int rc = wfst_decoder_get_best_path(_dec, ch.data(), cnt, ufp, cap, il.data(), ol.data(), g.data(), ac.data(), n.data());
      if (rc == WFST_E_CAPACITY && *std::max_element(n.begin(), n.end()) > cap && attempt == 0) { cap = *std::max_element(n.begin(), n.end()); continue; }
      if (rc != WFST_OK && cnt > 1 && attempt == 0) {
        // the batch was refused for one of its channels (GetBestPath before InitDecoding, a channel's device error ...): one by one
        for (Request *r : rs) { std::vector<Request *> one(1, r); ExecuteBestPath(one); }
        break;
      }
      for (int i = 0; i < cnt; ++i) {
        Request *r = rs[(size_t)i];
        if (rc != WFST_OK) { r->error = std::make_exception_ptr(std::runtime_error(std::string("GetBestPath: ") + wfst_last_error())); continue; }
        r->n_hops = n[(size_t)i];
        const size_t o = (size_t)i * cap;
        r->il.assign(il.begin() + (long)o, il.begin() + (long)o + r->n_hops);
        r->ol.assign(ol.begin() + (long)o, ol.begin() + (long)o + r->n_hops);
        r->g.assign(g.begin() + (long)o, g.begin() + (long)o + r->n_hops);
        r->ac.assign(ac.begin() + (long)o, ac.begin() + (long)o + r->n_hops);
        // (the final result says whether the per-frame token limit bound on the way; partial results do not stop for it)
        int32_t dg = 0;
        if (ufp && wfst_decoder_get_degraded_frames(_dec, r->channel, &dg) == WFST_OK) r->degraded = dg;
      }
      break;
    }
  }
}

// ---- single-stream decoder --------------------------------------------------------------------
GpuLatticeDecoder::GpuLatticeDecoder(Fst *graph, const LatticeFasterDecoderConfig &config, const wfst_limits *limits)
    : _dec(nullptr), _pool(nullptr), _chan(0), _decoded(0), _rows(nullptr), _rows_cap(0), _rows_pinned(false), _stride(0), _rows_ready(0), _inited(false) {
  config.Check();
  Share(graph, config, nullptr, nullptr, limits);
  if (_pool) return;
  wfst_config c = config.ToC();
  if (wfst_decoder_create(graph->Handle(), &c, 1, limits, nullptr, &_dec) != WFST_OK) Fatal("wfst_decoder_create");
  SetColumns(graph);
}
GpuLatticeDecoder::GpuLatticeDecoder(Fst *graph, const LatticeFasterDecoderConfig &config, ArpaLm *oldlm, ArpaLm *newlm,
                                     const wfst_limits *limits)
    : _dec(nullptr), _pool(nullptr), _chan(0), _decoded(0), _rows(nullptr), _rows_cap(0), _rows_pinned(false), _stride(0), _rows_ready(0), _inited(false) {
  config.Check();
  if (!oldlm || !newlm) throw std::runtime_error("biglm decoder needs both LMs");
  Share(graph, config, oldlm, newlm, limits);
  if (_pool) return;
  wfst_config c = config.ToC();
  if (wfst_decoder_create_biglm(graph->Handle(), &c, 1, limits, nullptr, oldlm->Handle(), newlm->Handle(), nullptr, &_dec) != WFST_OK)
    Fatal("wfst_decoder_create_biglm");
  SetColumns(graph);
}

// ---- ShareDevice: the reference's constructor shape over shared device decoders ------------------------------------------------
namespace {
struct SharedEntry {
  const wfst_graph *graph;
  wfst_config cfg;
  bool has_lim;
  wfst_limits lim;
  ArpaLm *oldlm, *newlm;
  std::shared_ptr<GpuChannelPool> pool;
};
struct SharedRegistry {
  std::mutex mu;
  int n_channels = 0, linger_us = 50;
  std::vector<SharedEntry> entries;
};
// (never destroyed: worker threads may still hold decoder objects when the process leaves main, and the device runtime's own
// teardown order is not ours to rely on)
SharedRegistry &Registry() { static SharedRegistry *r = new SharedRegistry(); return *r; }
}  // namespace

void GpuLatticeDecoder::ShareDevice(int n_channels, int linger_us) {
  SharedRegistry &R = Registry();
  std::lock_guard<std::mutex> lk(R.mu);
  R.n_channels = std::max(0, n_channels);
  R.linger_us = linger_us;
  if (R.n_channels == 0) R.entries.clear();   // (a shared decoder goes when its last object does)
}
void GpuLatticeDecoder::Share(Fst *graph, const LatticeFasterDecoderConfig &config, ArpaLm *oldlm, ArpaLm *newlm, const wfst_limits *limits) {
  SharedRegistry &R = Registry();
  std::lock_guard<std::mutex> lk(R.mu);
  if (R.n_channels <= 0) return;
  const wfst_config c = config.ToC();
  for (SharedEntry &e : R.entries) {
    if (e.graph != graph->Handle() || memcmp(&e.cfg, &c, sizeof(c)) != 0 || e.oldlm != oldlm || e.newlm != newlm) continue;
    if (e.has_lim != (limits != nullptr) || (limits && memcmp(&e.lim, limits, sizeof(*limits)) != 0)) continue;
    const int ch = e.pool->TryLease();
    if (ch < 0) continue;   // (full: the next one, or a new one)
    _shared = e.pool;
    _pool = _shared.get();
    _dec = _pool->_dec;
    _chan = ch;
    SetColumns(graph);
    return;
  }
  SharedEntry e;
  e.graph = graph->Handle();
  e.cfg = c;
  e.has_lim = limits != nullptr;
  memset(&e.lim, 0, sizeof(e.lim));
  if (limits) e.lim = *limits;
  e.oldlm = oldlm; e.newlm = newlm;
  e.pool.reset(oldlm ? new GpuChannelPool(graph, config, oldlm, newlm, R.n_channels, limits, R.linger_us)
                     : new GpuChannelPool(graph, config, R.n_channels, limits, R.linger_us));
  const int ch = e.pool->TryLease();
  _shared = e.pool;
  _pool = _shared.get();
  _dec = _pool->_dec;
  _chan = ch;
  SetColumns(graph);
  R.entries.push_back(e);
}
GpuLatticeDecoder::GpuLatticeDecoder(GpuChannelPool *pool)
    : _dec(nullptr), _pool(pool), _chan(0), _decoded(0), _rows(nullptr), _rows_cap(0), _rows_pinned(false), _stride(0), _rows_ready(0), _inited(false) {
  if (!pool) throw std::runtime_error("GpuLatticeDecoder: NULL pool");
  _dec = pool->_dec;
  SetColumns(pool->_graph);
  _chan = pool->Lease();
}
// the graph reads column tid2pdf[ilabel]: one representative transition-id per pdf is all the decodable is asked for (wfst-host.h,
// Fst::SetTid2Pdf)
void GpuLatticeDecoder::SetColumns(const Fst *graph) {
  _rep.clear();
  if (!graph) return;
  const std::vector<int32_t> &m = graph->Tid2Pdf();
  int32_t n_pdf = 0;
  for (size_t t = 1; t < m.size(); ++t) n_pdf = std::max(n_pdf, m[t] + 1);
  if (n_pdf <= 0) return;
  _rep.assign((size_t)n_pdf, 0);
  for (size_t t = m.size() - 1; t >= 1; --t)
    if (m[t] >= 0) _rep[(size_t)m[t]] = (int32_t)t;   // (the lowest transition-id of the pdf)
}
GpuLatticeDecoder::~GpuLatticeDecoder() {
  // (rows of an utterance abandoned right behind an AdvanceDecoding may still be on their way to the device from the page-locked
  // buffer freed below: the device is waited for first; an object that fetched its result, or never decoded, has nothing in flight)
  if (_rows_pinned && _rows_ready > 0 && _inited) {
    try {
      OnDevice([&] { (void)wfst_decoder_sync(_dec); });
    } catch (...) {
    }
  }
  if (_pool) _pool->Release(_chan);
  else wfst_decoder_free(_dec);
  if (_rows_in_pool) _rows = nullptr;   // (the pool's)
  else if (_rows_pinned) wfst_host_free(_rows);
  else free(_rows);
  _shared.reset();   // (ShareDevice: the shared decoder goes with its last object once the registry has let go of it)
}
// the rows pulled from the decodable live in page-locked memory (they are uploaded chunk by chunk, beside the search over the chunk
// before): grown by doubling, the history kept
void GpuLatticeDecoder::GrowRows(size_t floats) {
  if (floats <= _rows_cap) return;
  if (_pool && !_rows) {
    // the channel's slot of the pool's slab, if the rows fit there (they do when the service has reserved its longest utterance)
    size_t cap = 0;
    if (float *slot = _pool->RowSlot(_chan, floats, &cap)) {
      _rows = slot;
      _rows_cap = cap;
      _rows_pinned = true;
      _rows_in_pool = true;
      return;
    }
  }
  const size_t ncap = std::max<size_t>(floats, std::max<size_t>(2 * _rows_cap, (size_t)1 << 16));
  bool pinned = true;
  float *np = (float *)wfst_host_alloc(ncap * sizeof(float));
  if (!np) {
    pinned = false;
    np = (float *)malloc(ncap * sizeof(float));
    if (!np) throw std::bad_alloc();
  }
  if (_rows && _rows_ready > 0) memcpy(np, _rows, (size_t)_rows_ready * _stride * sizeof(float));
  // (rows of the old page-locked buffer may still be on their way to the device -- wfst_decoder_advance_host returns when they are
  // enqueued: the device is waited for before the buffer goes; a slot of the pool's slab stays where it is)
  if (_rows_in_pool) {
    _rows_in_pool = false;
  } else {
    if (_rows_pinned && _rows_ready > 0 && _inited) OnDevice([&] { (void)wfst_decoder_sync(_dec); });
    if (_rows_pinned) wfst_host_free(_rows);
    else free(_rows);
  }
  _rows = np;
  _rows_cap = ncap;
  _rows_pinned = pinned;
}

// the C-ABI calls of one decoder are not re-entrant: with a pool they run in its batcher thread, one request after the other
template <class F>
void GpuLatticeDecoder::OnDevice(F &&f) {
  if (!_pool) { f(); return; }
  GpuChannelPool::Request r;
  r.kind = GpuChannelPool::kCall;
  r.channel = _chan;
  r.call = [&](wfst_decoder *) { f(); };
  _pool->Submit(&r);
  _decoded = r.decoded;
}

void GpuLatticeDecoder::ReserveRows(int frames, int num_indices) {
  const size_t stride = _rep.empty() ? (size_t)(num_indices + 1) : (((size_t)_rep.size() + 3) & ~(size_t)3);   // (per-pdf rows where the graph maps)
  if (frames > 0 && num_indices > 0) GrowRows((size_t)frames * stride);
}

void GpuLatticeDecoder::InitDecoding() {
  if (_pool) {
    GpuChannelPool::Request r;
    r.kind = GpuChannelPool::kInit;
    r.channel = _chan;
    _pool->Submit(&r);
    _decoded = r.decoded;
  } else if (wfst_decoder_init(_dec, nullptr, 0) != WFST_OK) Fatal("InitDecoding");
  _rows_ready = 0;   // (the buffer stays: the next utterance's rows go over this one's)
  _stride = 0;
  _inited = true;
}

void GpuLatticeDecoder::Pull(AmInterface *d) {
  const int ready = d->NumFramesReady();
  MatrixDecodable *md = dynamic_cast<MatrixDecodable *>(d);
  if (!_rep.empty() && !md) {
    // rows of one column per pdf (padded to a multiple of four columns: the expansion stages such a row in LDS by 16-byte DMAs)
    const int n_pdf = (int)_rep.size(), stride = (n_pdf + 3) & ~3;
    if (_stride == 0) _stride = stride;
    if (stride != _stride) throw std::runtime_error("decodable changed its shape within an utterance");
    if (d->NumIndices() < *std::max_element(_rep.begin(), _rep.end())) throw std::runtime_error("the decodable has fewer indices than the graph's tid2pdf maps");
    if (ready <= _rows_ready) return;
    GrowRows((size_t)ready * _stride);
    for (int f = _rows_ready; f < ready; ++f) {
      float *row = &_rows[(size_t)f * _stride];
      for (int p = 0; p < n_pdf; ++p) row[p] = _rep[(size_t)p] > 0 ? d->LogLikelihood(f, _rep[(size_t)p]) : 0.0f;
      for (int p = n_pdf; p < _stride; ++p) row[p] = 0.0f;
    }
    _rows_ready = ready;
    return;
  }
  const int stride = d->NumIndices() + 1;
  if (_stride == 0) _stride = stride;
  if (stride != _stride) throw std::runtime_error("decodable changed NumIndices() within an utterance");
  if (ready <= _rows_ready) return;
  GrowRows((size_t)ready * _stride);
  if (MatrixDecodable *m = md) {
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
  Pull(decodable);   // (in the caller's thread: with a pool, while the device decodes the batch before)
  const float *rows = _rows;
  int32_t ready = _rows_ready;
  if (ready == 0) return;
  if (_pool) {
    GpuChannelPool::Request r;
    r.kind = GpuChannelPool::kAdvance;
    r.channel = _chan;
    r.rows = rows; r.ready = ready; r.stride = _stride; r.max_num_frames = max_num_frames;
    _pool->Submit(&r);
    _decoded = r.decoded;
    return;
  }
  if (wfst_decoder_advance_host(_dec, nullptr, 0, &rows, &ready, _stride, max_num_frames) != WFST_OK)
    Fatal("AdvanceDecoding");
}

BaseFloat GpuLatticeDecoder::ProcessEmitting(AmInterface *decodable) {
  AdvanceDecoding(decodable, 1);
  return 0.0f;
}

void GpuLatticeDecoder::FinalizeDecoding() {
  if (_pool) {
    GpuChannelPool::Request r;
    r.kind = GpuChannelPool::kFinalize;
    r.channel = _chan;
    _pool->Submit(&r);
    _decoded = r.decoded;
    return;
  }
  if (wfst_decoder_finalize(_dec, nullptr, 0) != WFST_OK) Fatal("FinalizeDecoding");
}

int32 GpuLatticeDecoder::NumFramesDecoded() const { return _pool ? _decoded : wfst_decoder_num_frames_decoded(_dec, 0); }

bool GpuLatticeDecoder::Decode(AmInterface *decodable) {
  InitDecoding();
  AdvanceDecoding(decodable);
  FinalizeDecoding();
  Lattice tmp;
  return GetBestPath(&tmp, true);
}

static void WarnIfDegraded(wfst_decoder *dec, int channel);
static void WarnDegraded(int channel, int n);

bool GpuLatticeDecoder::GetBestPath(Lattice *ofst, bool use_final_probs) {
  ofst->DeleteStates();
  if (_pool) {
    // batched with the other channels' requests (one wfst_decoder_get_best_path for all of them); the lattice is built here, in
    // the caller's thread
    GpuChannelPool::Request r;
    r.kind = GpuChannelPool::kBestPath;
    r.channel = _chan;
    r.use_final_probs = use_final_probs;
    _pool->Submit(&r);
    _decoded = r.decoded;
    if (r.degraded > 0) WarnDegraded(_chan, r.degraded);
    if (r.n_hops == 0) { Warn("No final token found."); return false; }
    HopsToLattice(r.il.data(), r.ol.data(), r.g.data(), r.ac.data(), r.n_hops, ofst);
    return true;
  }
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

bool GpuLatticeDecoder::GetNbest(std::vector<Lattice> &nbest_paths, int n) {
  bool ok = false;
  OnDevice([&] { ok = NbestOfChannel(_dec, _chan, nbest_paths, n, nullptr, nullptr); });
  return ok;
}
bool GpuLatticeDecoder::GetNbest(std::vector<Lattice> &nbest_paths, int n, ArpaLm *oldlm, ArpaLm *newlm) {
  bool ok = false;
  OnDevice([&] { ok = NbestOfChannel(_dec, _chan, nbest_paths, n, oldlm, newlm); });
  return ok;
}
bool GpuLatticeDecoder::GetNbestShortlist(std::vector<Lattice> &nbest_paths, int n) {
  bool ok = false;
  OnDevice([&] { ok = ShortlistOfChannel(_dec, _chan, nbest_paths, n); });
  return ok;
}

bool GpuLatticeDecoder::GetLattice(Lattice *ofst, bool use_final_probs) {
  bool ok = false;
  OnDevice([&] { ok = DetLatticeOfChannel(_dec, _chan, ofst, use_final_probs); });
  return ok;
}
bool GpuLatticeDecoder::GetLattice(Lattice *ofst, ArpaLm *oldlm, ArpaLm *newlm, bool use_final_probs) {
  bool ok = false;
  OnDevice([&] { ok = RescoredLatticeOfChannel(_dec, _chan, ofst, oldlm, newlm, use_final_probs); });
  return ok;
}

bool GpuLatticeDecoder::GetRawLattice(Lattice *ofst, bool use_final_probs) {
  bool ok = false;
  OnDevice([&] { ok = RawLatticeOfChannel(_dec, _chan, ofst, use_final_probs); });
  return ok;
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
static void WarnDegraded(int channel, int n) {
  Warn("channel " + std::to_string(channel) + ": " + std::to_string(n) + " frame(s) held more tokens than max_tokens_per_frame; the search "
       "went on from the cheapest of them (a max_active): raise wfst_limits.max_tokens_per_frame for the result at the configured beam");
}
static void WarnIfDegraded(wfst_decoder *dec, int channel) {
  int32_t n = 0;
  if (wfst_decoder_get_degraded_frames(dec, channel, &n) == WFST_OK && n > 0) WarnDegraded(channel, n);
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
