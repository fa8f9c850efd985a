// C++ host mirror of the reference's interface for the token-passing path, over the C ABI of
// include/wfst_decoder.h.  Same names, argument meaning and error behaviour as the reference so
// that a caller written against datemoon/ASR-decoder compiles against this header instead:
//
//   DecodableInterface / AmInterface     src/itf/decodable-itf.h:65-104
//   DecoderItf                            src/my-decoder/decoder-itf.h:10-25
//   LatticeFasterDecoderConfig            src/my-decoder/lattice-faster-decoder-conf.h:8-68
//   Fst (ReadFst/Start/IsFinal/TotState)  src/newfst/optimize-fst.h:53-307
//   Lattice / LatticeArc / LatticeWeight  src/newfst/lattice-fst.h:15-346, src/newfst/weigth.h:192-262
//   LatticeToVector                       src/newfst/lattice-functions.cc:179-217
//   ArpaLm (Read / Rescale)               src/newlm/arpa2fsa.h:249-441          (biglm)
//   OnlineLatticeDecoderMempoolBiglm      src/my-decoder/online-decoder-mempool-base-biglm.h:21-30,570
//
// GpuLatticeDecoder is the drop-in for OnlineLatticeDecoderMempool (one utterance stream,
// decodable pulled through LogLikelihood()); GpuBatchDecoder is the batch shape the MI355X wants
// (many channels per call, matrices already in HBM).  Nothing here decodes on the CPU.
#ifndef WFST_HOST_H_
#define WFST_HOST_H_

#include <cstdint>
#include <cstdio>
#include <limits>
#include <stdexcept>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/wfst_decoder.h"

namespace datemoon {

typedef float BaseFloat;
typedef int int32;
typedef int StateId;
typedef int Label;
const int kNoStateId = -1;

// ---- boundary A: what the decoder calls ---------------------------------------------------
// Under -DKALDI the reference's AmInterface IS kaldi::DecodableInterface (src/itf/decodable-itf.h:
// 55-62), so that any Kaldi decodable (nnet3 looped, DecodableMatrixScaledMapped, ...) plugs in.
// Same switch here: -DWFST_KALDI_DECODABLE (or the reference's own -DKALDI) with Kaldi's src/ on
// the include path.
#if defined(WFST_KALDI_DECODABLE) || defined(KALDI)
}  // namespace datemoon
#include "itf/decodable-itf.h"
namespace datemoon {
typedef kaldi::DecodableInterface DecodableInterface;
#else
class DecodableInterface {
 public:
  virtual float LogLikelihood(int frame, int index) = 0;  // already scaled; the decoder negates it
  virtual bool IsLastFrame(int frame) const = 0;
  virtual int NumFramesReady() const = 0;
  virtual int NumIndices() const = 0;  // indices are 1-based: 1..NumIndices()
  virtual ~DecodableInterface() {}
};
#endif
typedef DecodableInterface AmInterface;

// Optional fast path: a decodable that can hand over its rows in one piece (no per-element
// virtual calls).  Row f must hold LogLikelihood(f, i) at column i, i in [0, NumIndices()].
class MatrixDecodable : public DecodableInterface {
 public:
  virtual const float *HostRows() const = 0;  // row-major [NumFramesReady()][Stride()]
  virtual int Stride() const = 0;
};

// ---- config ---------------------------------------------------------------------------------
struct LatticeFasterDecoderConfig {
  float _beam;
  int _max_active;
  int _min_active;
  float _lattice_beam;
  int _prune_interval;
  bool _determinize_lattice;
  float _beam_delta;
  float _hash_ratio;
  float _prune_scale;
  LatticeFasterDecoderConfig()
      : _beam(16.0f), _max_active(std::numeric_limits<int>::max()), _min_active(200), _lattice_beam(10.0f),
        _prune_interval(25), _determinize_lattice(true), _beam_delta(0.5f), _hash_ratio(2.0f), _prune_scale(0.1f) {}
  // "--name=value" lines (beam, max-active, min-active, lattice-beam, prune-interval, beam-delta,
  // hash-ratio), the option names the reference registers (conf.h:46-61).  Unknown names throw.
  void ReadConfigFile(const std::string &path);
  void Check() const;  // same conditions as the reference's asserts (conf.h:62-67); throws
  wfst_config ToC() const;
};

// ---- graph ----------------------------------------------------------------------------------
class Fst {
 public:
  Fst() : _graph(nullptr) {}
  ~Fst();
  bool ReadFst(const char *file, int device = 0);  // false (with a message on stderr) on failure
  bool Init(const char *file, const char *) { return ReadFst(file); }
  void SetTid2Pdf(const std::vector<int32_t> &tid2pdf);  // entry 0 unused
  StateId Start() const { return _start; }
  bool IsFinal(StateId id) const { return id == _final; }
  StateId TotState() const { return _states; }
  int TotArc() const { return _arcs; }
  const wfst_graph *Handle() const { return _graph; }

 private:
  Fst(const Fst &);
  Fst &operator=(const Fst &);
  wfst_graph *_graph;
  int32_t _start = 0, _final = 0, _states = 0, _arcs = 0;
};

// ---- language model (biglm) -----------------------------------------------------------------
// The reference's ArpaLm as the biglm caller uses it (kaldi-nnet3bin/kaldi-hclg-my-decoder-biglm.cc:
// 55-60): `lm1.Read(file); lm2.Read(file); lm1.Rescale(-1.0);`.  Read checks the file; the automaton
// goes to HBM, with the scale applied, when a decoder first asks for it.
class ArpaLm {
 public:
  ArpaLm() : _scale(1.0f), _device(0), _lm(nullptr) {}
  ~ArpaLm();
  bool Read(const char *file, int device = 0);  // false (with a message on stderr) on failure
  void Rescale(float scale);
  int BosSymbol() const { return _bos; }
  int EosSymbol() const { return _eos; }
  const wfst_lm *Handle();

 private:
  ArpaLm(const ArpaLm &);
  ArpaLm &operator=(const ArpaLm &);
  std::string _file;
  float _scale;
  int _device;
  wfst_lm *_lm;
  std::mutex _mu;   // Handle() may be called by several worker threads at once (one decoder per thread over shared LMs)
  int32_t _bos = -1, _eos = -1;
};

// ---- lattice (linear best path is all this path produces) ---------------------------------------
struct LatticeWeight {
  float _value1, _value2;  // graph cost, acoustic cost
  LatticeWeight() : _value1(0), _value2(0) {}
  LatticeWeight(float a, float b) : _value1(a), _value2(b) {}
  float Value1() const { return _value1; }
  float Value2() const { return _value2; }
  static LatticeWeight One() { return LatticeWeight(0.0f, 0.0f); }
};
struct LatticeArc {
  Label _input, _output;
  LatticeWeight _w;
  StateId _to;
  LatticeArc() : _input(0), _output(0), _to(0) {}
  LatticeArc(Label i, Label o, StateId to, LatticeWeight w) : _input(i), _output(o), _w(w), _to(to) {}
};
class LatticeState {
 public:
  LatticeState() : _final(false) {}
  bool IsFinal() const { return _final; }
  void SetFinal() { _final = true; }
  void AddArc(const LatticeArc &a) { _arcs.push_back(a); }
  LatticeArc *GetArc(unsigned i) { return i < _arcs.size() ? &_arcs[i] : nullptr; }
  unsigned GetArcSize() const { return (unsigned)_arcs.size(); }

 private:
  std::vector<LatticeArc> _arcs;
  bool _final;
};
class Lattice {
 public:
  Lattice() : _start(kNoStateId) {}
  void DeleteStates() { _states.clear(); _start = kNoStateId; }
  StateId AddState() { _states.push_back(LatticeState()); return (StateId)_states.size() - 1; }
  void SetStart(StateId s) { _start = s; }
  void SetFinal(StateId s) { _states[s].SetFinal(); }
  void AddArc(StateId s, const LatticeArc &a) { _states[s].AddArc(a); }
  StateId Start() const { return _start; }
  StateId NumStates() const { return (StateId)_states.size(); }
  LatticeState *GetState(StateId s) { return &_states[s]; }
  // The reference's on-disk lattice (newfst/lattice-fst.cc:38-101, lattice-fst.h:124-172,
  // arc.h:38-86, weigth.h:229-258), little-endian, LP64: u64 number of states, i32 start state,
  // then per state {i32 final, u64 number of arcs, arcs x {i32 ilabel, i32 olabel, f32 graph cost,
  // f32 acoustic cost, i32 nextstate}}.  Write(file) APPENDS, as the reference does ("ab"), so one
  // file holds the lattices of consecutive utterances; Read(FILE*) reads the next one.
  bool Write(FILE *fp);
  bool Write(const std::string &file);
  bool Read(FILE *fp);
  bool Read(const std::string &file);

 private:
  std::vector<LatticeState> _states;
  StateId _start;
};

bool LatticeToVector(Lattice &best_path, std::vector<int> &best_words_arr, std::vector<int> &best_phones_arr,
                     float &best_tot_score, float &best_lm_score);

// ---- boundary B: what callers use -------------------------------------------------------------
class DecoderItf {
 public:
  virtual ~DecoderItf() {}
  virtual void InitDecoding() = 0;
  virtual void AdvanceDecoding(AmInterface *decodable, int32 max_num_frames = -1) = 0;
  virtual void FinalizeDecoding() = 0;
  virtual int32 NumFramesDecoded() const = 0;
  virtual BaseFloat ProcessEmitting(AmInterface *decodable) = 0;
  virtual void ProcessNonemitting(BaseFloat cost_cutoff) = 0;
  virtual bool Decode(AmInterface *decodable) = 0;
  virtual bool GetBestPath(Lattice *ofst, bool use_final_probs = true) = 0;
  virtual bool GetRawLattice(Lattice *ofst, bool use_final_probs = true) = 0;
};

// One utterance stream on channel 0 of a private 1-channel device decoder.  Scores are pulled
// through LogLikelihood(f, i) for the frames that became ready since the last call (or taken in
// one piece from a MatrixDecodable) and shipped to the GPU; the search runs there.
// Fatal conditions throw std::runtime_error (the reference's LOG_ERR does, util/log-message.cc:
// 122-145); soft ones print a warning and return false, as in the reference.
class GpuLatticeDecoder : public DecoderItf {
 public:
  GpuLatticeDecoder(Fst *graph, const LatticeFasterDecoderConfig &config, const wfst_limits *limits = nullptr);
  // OnlineLatticeDecoderMempoolBiglm(fst, config, oldlm, newlm) (biglm.h:21-30): on-the-fly LM rescoring
  GpuLatticeDecoder(Fst *graph, const LatticeFasterDecoderConfig &config, ArpaLm *oldlm, ArpaLm *newlm,
                    const wfst_limits *limits = nullptr);
  ~GpuLatticeDecoder() override;
  void InitDecoding() override;
  void AdvanceDecoding(AmInterface *decodable, int32 max_num_frames = -1) override;
  void FinalizeDecoding() override;
  int32 NumFramesDecoded() const override;
  // The device frame step fuses ProcessEmitting and ProcessNonemitting: ProcessEmitting decodes
  // exactly one frame (emitting arcs + epsilon closure) and returns the cutoff it used for the
  // closure; ProcessNonemitting is then a no-op.
  BaseFloat ProcessEmitting(AmInterface *decodable) override;
  void ProcessNonemitting(BaseFloat) override {}
  // InitDecoding + all ready frames + FinalizeDecoding.  (The reference's Decode() reads one frame
  // past the end, base-inl.h:615; that landmine is not reproduced.)
  bool Decode(AmInterface *decodable) override;
  bool GetBestPath(Lattice *ofst, bool use_final_probs = true) override;
  // after FinalizeDecoding, decoder created with wfst_limits.lattice_links > 0 (else: warning + false)
  bool GetRawLattice(Lattice *ofst, bool use_final_probs = true) override;
  // GetLattice (online-decoder-base.h:182, base-inl.h:850-866): GetRawLattice + DeterminizeLatticeWrapper, both on
  // the device; arcs carry ilabel 0 / olabel = word, final states have no arcs (the reference's output convention)
  bool GetLattice(Lattice *ofst, bool use_final_probs = true);
  // the same with the service's second LM pass (--use-second, kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:53-78): the determinized
  // lattice composed with the old LM (rescaled by -1) and with the new one -- ComposeLattice twice (newfst/compose-lat-inl.h), on the device
  bool GetLattice(Lattice *ofst, ArpaLm *oldlm, ArpaLm *newlm, bool use_final_probs = true);
  // OnlineClgLatticeFastDecoder::GetNbest (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:97-105): GetLattice + NShortestPath +
  // ConvertNbestToVector (newfst/lattice-to-nbest.cc), on the device: the n (<= 4096) cheapest paths of the determinized lattice,
  // each a linear Lattice with that lattice's own arcs on it (ilabel 0, olabel = word, both costs per arc) between the epsilon
  // arcs the reference's Reverse / AddSuperFinalState leave -- arc for arc what the reference returns.  With LMs: over the
  // second-pass lattice (--use-second).  Same conditions as GetRawLattice.
  bool GetNbest(std::vector<Lattice> &nbest_paths, int n);
  bool GetNbest(std::vector<Lattice> &nbest_paths, int n, ArpaLm *oldlm, ArpaLm *newlm);
  // the short list computed on the raw lattice without determinizing it (n <= 16, every channel of a batch in one launch): the same
  // word sequences and totals; each path's FIRST arc carries its whole weight (graph = lm_score, acoustic = tot - lm), so that
  // LatticeToVector gives words, tot_score and lm_score
  bool GetNbestShortlist(std::vector<Lattice> &nbest_paths, int n);

 private:
  void Pull(AmInterface *decodable);
  wfst_decoder *_dec;
  std::vector<float> _rows;  // host history [frames][stride]
  int _stride, _rows_ready;
  bool _inited;
};

// Batch shape: n_channels utterances per call, device-resident matrices.
class GpuBatchDecoder {
 public:
  GpuBatchDecoder(Fst *graph, const LatticeFasterDecoderConfig &config, int n_channels,
                  const wfst_limits *limits = nullptr, void *hip_stream = nullptr);
  GpuBatchDecoder(Fst *graph, const LatticeFasterDecoderConfig &config, ArpaLm *oldlm, ArpaLm *newlm, int n_channels,
                  const wfst_limits *limits = nullptr, void *hip_stream = nullptr);  // biglm
  ~GpuBatchDecoder();
  void InitDecoding(const std::vector<int> &channels = std::vector<int>());
  void AdvanceDecoding(const std::vector<int> &channels, const std::vector<const float *> &device_loglikes,
                       const std::vector<int> &num_frames_ready, int stride, int max_num_frames = -1);
  void AdvanceDecodingHost(const std::vector<int> &channels, const std::vector<const float *> &host_loglikes,
                           const std::vector<int> &num_frames_ready, int stride, int max_num_frames = -1);
  void FinalizeDecoding(const std::vector<int> &channels = std::vector<int>());
  int NumFramesDecoded(int channel) const;
  bool GetBestPath(int channel, Lattice *ofst, bool use_final_probs = true);
  bool GetRawLattice(int channel, Lattice *ofst, bool use_final_probs = true);
  // the raw lattices of many channels: one device fetch, then the per-lattice host work on
  // `threads` host threads (0: up to 16)
  void GetRawLattices(const std::vector<int> &channels, std::vector<Lattice> *ofsts, std::vector<bool> *ok,
                      bool use_final_probs = true, int threads = 0);
  bool GetNbest(int channel, std::vector<Lattice> &nbest_paths, int n);
  bool GetNbest(int channel, std::vector<Lattice> &nbest_paths, int n, ArpaLm *oldlm, ArpaLm *newlm);
  bool GetNbestShortlist(int channel, std::vector<Lattice> &nbest_paths, int n);   // (n <= 16; see GpuLatticeDecoder)
  // GetLattice ahead of its request: the finalized channels go to the determinizer now, on a side stream; GetBestPaths / GetNbest
  // run beside it and the first GetLattice finds the work done or waits (wfst_decoder_prefetch_determinized)
  void PrefetchLattices();
  // ... detached: the channels go on to their next utterances beside the determinizer (wfst_decoder_prefetch_determinized_detached);
  // the lattices of the utterances finalized at that call are fetched with GetPrefetchedLattice once harvested -- by the next
  // PrefetchLatticesDetached, or by HarvestPrefetchedLattices (which waits)
  void PrefetchLatticesDetached();
  void HarvestPrefetchedLattices();
  bool GetPrefetchedLattice(int channel, Lattice *ofst);
  // GetLattice of one channel; the first call after FinalizeDecoding determinizes every finalized channel in one launch
  bool GetLattice(int channel, Lattice *ofst, bool use_final_probs = true);
  bool GetLattice(int channel, Lattice *ofst, ArpaLm *oldlm, ArpaLm *newlm, bool use_final_probs = true);   // with the second LM pass
  void GetBestPaths(const std::vector<int> &channels, std::vector<Lattice> *ofsts, std::vector<bool> *ok,
                    bool use_final_probs = true);
  // The service's post-processing for MANY finalized channels at once -- GetLattice under --use-second and GetNbest, which the
  // reference runs per utterance on one worker thread each (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:50-105,
  // v2-asr/v2-asr-work-thread.h:66): every stage is one launch for the whole list (wfst_decoder_rescore_lattices /
  // wfst_decoder_nbest_paths_batch), the results come back once.  (*ok)[i] = what the per-channel call would return.
  void GetLattices(const std::vector<int> &channels, std::vector<Lattice> *ofsts, std::vector<bool> *ok, ArpaLm *oldlm, ArpaLm *newlm,
                   bool use_final_probs = true);
  void GetNbests(const std::vector<int> &channels, std::vector<std::vector<Lattice> > *nbests, std::vector<bool> *ok, int n,
                 ArpaLm *oldlm = nullptr, ArpaLm *newlm = nullptr);
  wfst_decoder *Handle() { return _dec; }

 private:
  wfst_decoder *_dec;
  int _n;
};

// the reference's name for the biglm decoder (biglm.h:570): `OnlineLatticeDecoderMempoolBiglm decode(&fst, opt, &lm1, &lm2);`
typedef GpuLatticeDecoder OnlineLatticeDecoderMempoolBiglm;

}  // namespace datemoon
#endif
