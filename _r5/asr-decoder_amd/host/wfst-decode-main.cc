// wfst-decode: offline batch decode of precomputed log-likelihood matrices on an MI355X.
// Same job and call sequence as the reference CLI kaldi-nnet3bin/kaldi-hclg-my-decoder.cc
// (graph + decoder config + per-utterance matrices -> word ids, scores, real-time factor), with
// plain files instead of Kaldi tables:
//
//   wfst-decode [--tid2pdf=FILE] [--batch=N] [--single-stream] [--lattice-out=FILE [--lattice-links=N] [--determinize]]
//               [--lm-old=FILE --lm-new=FILE]
//   --lm-old/--lm-new  biglm (kaldi-hclg-my-decoder-biglm.cc): rescore on the fly with new LM - old LM; the files
//                  are the reference's binary LMs (arpa2fsa-bin); the old one is rescaled by -1 as the reference CLI does
//               CONFIG GRAPH LOGLIKES [WORDS_OUT]
//   --second-lm-old/--second-lm-new  the service's --use-second: GetLattice (--determinize) and GetNbest (--nbest) run the second LM pass
//                  (ComposeLattice with the old LM rescaled by -1, then with the new one) on the determinized lattice, on the device
//   --nbest-lattice-out  also write every n-best path as the linear lattice GetNbest returns (NShortestPath + ConvertNbestToVector)
//   --lattice-out  also write GetRawLattice of every utterance, in utterance order, in the
//                  reference's on-disk lattice format (Lattice::Write, newfst/lattice-fst.cc:38-64;
//                  lattice mode: N forward links kept per utterance).  An utterance without a
//                  lattice is written as an empty one (0 states, start -1).
//   --determinize  the lattices written are GetLattice's (determinized on the device, base-inl.h:850-866) instead of
//                  GetRawLattice's
//   --chunk=N      single-stream only: the streaming caller's shape (OnlineClgLatticeFastDecoder::ProcessData,
//                  kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:10-48): NumFramesReady() grows by N frames per
//                  AdvanceDecoding call; after every call the partial result GetBestPath(use_final_probs = false) is
//                  printed as "KEY@frames word-ids..." (GetBestPathTxt(..., false), :122-137); with --nbest also the
//                  partial n-best (lattice mode serves GetNbest mid-utterance)
//   --inflight=K   batch shape only: K batches in flight, each on its own GpuBatchDecoder (own HIP
//                  stream) driven by its own host thread -- the reference service's model of one
//                  decoder object per thread (v2-asrbin/v2-asr-service.cc:95-105); the GPU overlaps
//                  independent batches.  Output order is unchanged.
//   --devices=a,b,...  batch shape only: the node's GPUs: the graph (and the LMs) are uploaded once per listed device, each
//                  device gets `inflight` GpuBatchDecoders of its own, each driven by its own host thread; batch b goes to device
//                  b mod n -- utterances share nothing but the read-only graph, so there is no exchange between devices, and the
//                  results are merged in input order on the host (the reference's model of N worker threads over one shared
//                  graph, v2-asrbin/v2-asr-service.cc:95-105, with one graph replica per device).  A device may be listed twice.
//   --nbest=N      also print the N-best word sequences of every utterance (the service's
//                  GetNbestTxt, kaldi-online-nnet3-my-decoder.cc:139-150) as "KEY-k w1 w2 ..." to
//                  stdout and "LOG KEY-k tot_score .. lm_score .." to stderr (lattice mode)
//   --lattice-text same lattices as text: "KEY", one line "src dst ilabel olabel graph_cost
//                  acoustic_cost" per arc, one line "state" per final state, then an empty line
//
//   CONFIG    text file of --beam=.. --max-active=.. lines (reference option names)
//   GRAPH     flat graph in the reference format (Fst::ReadFst)
//   LOGLIKES  binary: repeated { int32 key_len, key bytes, int32 frames, int32 cols,
//             float32[frames*cols] }, cols = pdfs (with --tid2pdf) or NumIndices()+1
//   tid2pdf   binary int32 array, entry 0 unused
//
// Output lines "key word-ids..." like the reference's words_writer (:126-129); the log at the end
// prints the reference's "real-time factor assuming 100 frames/sec" (:189-192).
#include <atomic>
#include <chrono>
#include <thread>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>

#include "wfst-host.h"

using namespace datemoon;

namespace {
struct Utt {
  std::string key;
  int frames, cols;
  std::vector<float> m;
};
bool ReadUtt(std::ifstream &in, Utt *u) {
  int32_t kl;
  if (!in.read((char *)&kl, 4)) return false;
  u->key.resize(kl);
  in.read(&u->key[0], kl);
  in.read((char *)&u->frames, 4);
  in.read((char *)&u->cols, 4);
  u->m.resize((size_t)u->frames * u->cols);
  in.read((char *)u->m.data(), u->m.size() * 4);
  return (bool)in;
}
// DecodableInterface over a host matrix: the shape every reference caller has.
class HostMatrixDecodable : public MatrixDecodable {
 public:
  explicit HostMatrixDecodable(const Utt &u) : _u(u), _ready(u.frames) {}
  float LogLikelihood(int f, int i) override { return _u.m[(size_t)f * _u.cols + i]; }
  bool IsLastFrame(int f) const override { return f == _u.frames - 1; }
  int NumFramesReady() const override { return _ready; }
  void SetFramesReady(int n) { _ready = std::min(n, _u.frames); }  // streaming: frames that have "arrived"
  int NumIndices() const override { return _u.cols - 1; }
  const float *HostRows() const override { return _u.m.data(); }
  int Stride() const override { return _u.cols; }

 private:
  const Utt &_u;
  int _ready;
};
}  // namespace

int main(int argc, char **argv) {
  try {
    std::string tid2pdf_file, lm_old_file, lm_new_file, second_old_file, second_new_file, nbest_lattice_file;
    int batch = 128;
    bool single = false, determinize = false;
    std::string lattice_file, lattice_text;
    long long lattice_links = 1ll << 22;
    int nbest = 0, inflight = 1, chunk = 0;
    std::vector<int> devices(1, 0);
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
      std::string a = argv[i];
      if (a.compare(0, 10, "--tid2pdf=") == 0) tid2pdf_file = a.substr(10);
      else if (a.compare(0, 8, "--batch=") == 0) batch = atoi(a.c_str() + 8);
      else if (a == "--single-stream") single = true;
      else if (a == "--determinize") determinize = true;
      else if (a.compare(0, 14, "--lattice-out=") == 0) lattice_file = a.substr(14);
      else if (a.compare(0, 15, "--lattice-text=") == 0) lattice_text = a.substr(15);
      else if (a.compare(0, 16, "--lattice-links=") == 0) lattice_links = atoll(a.c_str() + 16);
      else if (a.compare(0, 8, "--nbest=") == 0) nbest = atoi(a.c_str() + 8);
      else if (a.compare(0, 11, "--inflight=") == 0) inflight = std::max(1, atoi(a.c_str() + 11));
      else if (a.compare(0, 8, "--chunk=") == 0) chunk = std::max(0, atoi(a.c_str() + 8));
      else if (a.compare(0, 10, "--devices=") == 0) {
        devices.clear();
        for (size_t p0 = 10; p0 <= a.size();) {
          const size_t p1 = std::min(a.find(',', p0), a.size());
          if (p1 > p0) devices.push_back(atoi(a.substr(p0, p1 - p0).c_str()));
          p0 = p1 + 1;
        }
        if (devices.empty()) { std::cerr << "--devices needs a list of device ordinals\n"; return 1; }
      }
      else if (a.compare(0, 9, "--lm-old=") == 0) lm_old_file = a.substr(9);
      else if (a.compare(0, 9, "--lm-new=") == 0) lm_new_file = a.substr(9);
      else if (a.compare(0, 16, "--second-lm-old=") == 0) second_old_file = a.substr(16);
      else if (a.compare(0, 16, "--second-lm-new=") == 0) second_new_file = a.substr(16);
      else if (a.compare(0, 20, "--nbest-lattice-out=") == 0) nbest_lattice_file = a.substr(20);
      else pos.push_back(a);
    }
    if (pos.size() < 3) {
      std::cerr << "usage: wfst-decode [--tid2pdf=FILE] [--batch=N] [--single-stream [--chunk=N]] [--inflight=K] [--devices=a,b,...] [--nbest=N] [--lattice-out=FILE] [--determinize] "
                   "[--lattice-text=FILE] [--lattice-links=N] [--lm-old=FILE --lm-new=FILE] [--second-lm-old=FILE --second-lm-new=FILE] [--nbest-lattice-out=FILE] CONFIG GRAPH LOGLIKES [WORDS_OUT]\n";
      return 1;
    }
    LatticeFasterDecoderConfig opt;
    opt.ReadConfigFile(pos[0]);
    if (single && devices.size() > 1) { std::cerr << "--devices lists several devices: batch shape only\n"; return 1; }
    // one graph replica per listed device (fsts[0] also serves --single-stream)
    std::vector<std::unique_ptr<Fst> > fsts;
    for (size_t di = 0; di < devices.size(); ++di) {
      fsts.emplace_back(new Fst());
      if (!fsts.back()->ReadFst(pos[1].c_str(), devices[di])) return 1;
    }
    Fst &fst = *fsts[0];
    if (!tid2pdf_file.empty()) {
      std::ifstream t(tid2pdf_file.c_str(), std::ios::binary | std::ios::ate);
      if (!t) { std::cerr << "cannot open " << tid2pdf_file << "\n"; return 1; }
      std::vector<int32_t> m((size_t)t.tellg() / 4);
      t.seekg(0);
      t.read((char *)m.data(), m.size() * 4);
      for (auto &f : fsts) f->SetTid2Pdf(m);
    }
    std::ifstream in(pos[2].c_str(), std::ios::binary);
    if (!in) { std::cerr << "cannot open " << pos[2] << "\n"; return 1; }
    std::ofstream fout;
    if (pos.size() > 3) fout.open(pos[3].c_str());
    std::ostream &out = pos.size() > 3 ? (std::ostream &)fout : std::cout;

    std::ofstream lat_out;
    if (!lattice_text.empty()) {
      lat_out.open(lattice_text.c_str());
      lat_out.precision(9);
    }
    if (!lattice_file.empty()) remove(lattice_file.c_str());  // Lattice::Write(file) appends
    if (!nbest_lattice_file.empty()) remove(nbest_lattice_file.c_str());
    const bool want_lattice = !lattice_file.empty() || !lattice_text.empty() || nbest > 0;
    auto emit_nbest = [&](const Utt &u, std::vector<Lattice> &paths) {
      for (size_t k = 0; k < paths.size(); ++k) {
        std::vector<int> words, phones;
        float tot = 0, lm = 0;
        if (!nbest_lattice_file.empty() && !paths[k].Write(nbest_lattice_file)) throw std::runtime_error("cannot write " + nbest_lattice_file);
        if (!LatticeToVector(paths[k], words, phones, tot, lm)) continue;
        out << u.key << '-' << (k + 1);
        for (int w : words) out << ' ' << w;
        out << '\n';
        std::cerr << "LOG " << u.key << '-' << (k + 1) << " tot_score " << tot << " lm_score " << lm << "\n";
      }
    };
    // biglm (kaldi-nnet3bin/kaldi-hclg-my-decoder-biglm.cc:55-60,80): both LM files, the old one rescaled by -1
    const bool biglm = !lm_old_file.empty() || !lm_new_file.empty();
    struct LmPair { ArpaLm a, b; };
    std::vector<std::unique_ptr<LmPair> > lms, slms;   // [device index]: the search's LMs, the second pass's
    auto load_pairs = [&](const std::string &f_old, const std::string &f_new, std::vector<std::unique_ptr<LmPair> > *v) -> bool {
      for (size_t di = 0; di < devices.size(); ++di) {
        v->emplace_back(new LmPair());
        if (!v->back()->a.Read(f_old.c_str(), devices[di]) || !v->back()->b.Read(f_new.c_str(), devices[di])) return false;
        v->back()->a.Rescale(-1.0);
        v->back()->a.Handle();   // both automata go to HBM here, once, before any worker thread asks for them
        v->back()->b.Handle();
      }
      return true;
    };
    if (biglm) {
      if (lm_old_file.empty() || lm_new_file.empty()) { std::cerr << "--lm-old and --lm-new go together\n"; return 1; }
      if (!load_pairs(lm_old_file, lm_new_file, &lms)) return 1;
    }
    ArpaLm *lm1p = biglm ? &lms[0]->a : nullptr, *lm2p = biglm ? &lms[0]->b : nullptr;
    // the service's second pass (--use-second, kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:53-78): GetLattice / GetNbest compose the
    // determinized lattice with the old LM (rescaled by -1) and with the new one
    const bool second = !second_old_file.empty() || !second_new_file.empty();
    if (second) {
      if (second_old_file.empty() || second_new_file.empty()) { std::cerr << "--second-lm-old and --second-lm-new go together\n"; return 1; }
      if (!load_pairs(second_old_file, second_new_file, &slms)) return 1;
    }
    ArpaLm *slm1p = second ? &slms[0]->a : nullptr, *slm2p = second ? &slms[0]->b : nullptr;
    // exact n-best (NShortestPath on the determinized lattice) where its lattices are asked for, a second pass runs or the list is long
    const bool exact_nbest = !nbest_lattice_file.empty() || second || nbest > 16;
    wfst_limits limits = {0, 0, 0, 0, 0};  // zeros = the library defaults
    limits.lattice_links = want_lattice ? lattice_links : 0;
    auto emit_lattice = [&](const Utt &u, Lattice &lat, bool ok) {
      if (!ok) lat.DeleteStates();
      if (!lattice_file.empty() && !lat.Write(lattice_file)) throw std::runtime_error("cannot write " + lattice_file);
      if (!lat_out.is_open()) return;
      lat_out << u.key << '\n';
      if (ok)
        for (StateId s = 0; s < lat.NumStates(); ++s) {
          LatticeState *st = lat.GetState(s);
          for (size_t i = 0; i < st->GetArcSize(); ++i) {
            LatticeArc *a = st->GetArc(i);
            lat_out << s << ' ' << a->_to << ' ' << a->_input << ' ' << a->_output << ' ' << a->_w.Value1() << ' '
                    << a->_w.Value2() << '\n';
          }
          if (st->IsFinal()) lat_out << s << '\n';
        }
      lat_out << '\n';
    };

    std::vector<Utt> utts;
    for (Utt u; ReadUtt(in, &u);) utts.push_back(u);
    int num_success = 0, num_fail = 0;
    long long frame_count = 0;
    double tot_like = 0;
    auto t0 = std::chrono::steady_clock::now();
    auto emit = [&](const Utt &u, Lattice &best, bool ok) {
      std::vector<int> words, phones;
      float tot = 0, lm = 0;
      if (!ok || !LatticeToVector(best, words, phones, tot, lm)) {
        std::cerr << "WARNING Did not successfully decode utterance " << u.key << ", len = " << u.frames << "\n";
        ++num_fail;
        return;
      }
      out << u.key;
      for (int w : words) out << ' ' << w;
      out << '\n';
      std::cerr << "LOG " << u.key << " tot_score " << tot << " lm_score " << lm << " over " << u.frames << " frames.\n";
      tot_like += -tot;
      frame_count += u.frames;
      ++num_success;
    };
    if (single) {  // the reference's shape: one decoder object, one utterance at a time
      std::unique_ptr<GpuLatticeDecoder> decode_p(biglm ? new OnlineLatticeDecoderMempoolBiglm(&fst, opt, lm1p, lm2p, &limits)
                                                        : new GpuLatticeDecoder(&fst, opt, &limits));
      GpuLatticeDecoder &decode = *decode_p;
      for (const Utt &u : utts) {
        HostMatrixDecodable decodable(u);
        decode.InitDecoding();
        if (chunk > 0) {  // the service's loop: data arrives, AdvanceDecoding, partial result
          for (int ready = chunk; ; ready += chunk) {
            decodable.SetFramesReady(ready);
            decode.AdvanceDecoding(&decodable);
            if (ready >= u.frames) break;
            Lattice part;
            std::vector<int> w, ph;
            float tot = 0, lm = 0;
            out << u.key << "@" << decode.NumFramesDecoded();
            if (decode.GetBestPath(&part, false) && LatticeToVector(part, w, ph, tot, lm))
              for (size_t k = 0; k < w.size(); ++k) out << " " << w[k];
            out << "\n";
            if (nbest > 0) {
              // (partial lists come from the raw lattice: its unpruned last frames make the mid-utterance lattice expensive to
              // determinize, for the reference as much as here)
              std::vector<Lattice> paths;
              decode.GetNbestShortlist(paths, std::min(nbest, 16));
              for (size_t k = 0; k < paths.size(); ++k) {
                std::vector<int> nw, nph;
                float t2 = 0, l2 = 0;
                LatticeToVector(paths[k], nw, nph, t2, l2);
                out << u.key << "@" << decode.NumFramesDecoded() << "-" << (k + 1);
                for (size_t q = 0; q < nw.size(); ++q) out << " " << nw[q];
                out << "\n";
              }
            }
          }
        } else {
          decode.AdvanceDecoding(&decodable);
        }
        decode.FinalizeDecoding();
        Lattice best;
        bool ok = decode.GetBestPath(&best);
        emit(u, best, ok);
        if (want_lattice) {
          Lattice lat;
          bool lok = determinize ? (second ? decode.GetLattice(&lat, slm1p, slm2p) : decode.GetLattice(&lat)) : decode.GetRawLattice(&lat);
          emit_lattice(u, lat, lok);
        }
        if (nbest > 0) {
          std::vector<Lattice> paths;
          if (!exact_nbest) decode.GetNbestShortlist(paths, nbest);
          else if (second) decode.GetNbest(paths, nbest, slm1p, slm2p);
          else decode.GetNbest(paths, nbest);
          emit_nbest(u, paths);
        }
      }
    } else {  // the MI355X shape: `batch` utterances per pass, `inflight` passes at a time
      struct BatchOut {
        std::vector<Lattice> best, lats;
        std::vector<bool> ok, lat_ok;
        std::vector<std::vector<Lattice> > nbest;
      };
      const size_t n_batches = (utts.size() + batch - 1) / batch;
      std::vector<BatchOut> outs(n_batches);
      std::atomic<size_t> next(0);
      const int n_dev = (int)devices.size(), n_workers = inflight * n_dev;
      std::vector<std::string> errors((size_t)n_workers);
      std::vector<std::atomic<size_t> > next_of_dev((size_t)n_dev);
      for (auto &x : next_of_dev) x = 0;
      auto worker = [&](int k) {
        try {
          // worker k drives a decoder on device k mod n_dev, over that device's graph replica (and LMs)
          const int di = k % n_dev;
          Fst *wf = fsts[(size_t)di].get();
          ArpaLm *lm1 = biglm ? &lms[(size_t)di]->a : nullptr, *lm2 = biglm ? &lms[(size_t)di]->b : nullptr;
          ArpaLm *slm1 = second ? &slms[(size_t)di]->a : nullptr, *slm2 = second ? &slms[(size_t)di]->b : nullptr;
          std::unique_ptr<GpuBatchDecoder> decode_p(biglm ? new GpuBatchDecoder(wf, opt, lm1, lm2, batch, &limits)
                                                          : new GpuBatchDecoder(wf, opt, batch, &limits));  // its own stream
          GpuBatchDecoder &decode = *decode_p;
          for (;;) {
            // one device: the next batch nobody has taken; several: batch b belongs to device b mod n_dev (its workers share them)
            const size_t b = n_dev == 1 ? next.fetch_add(1) : (size_t)di + (size_t)n_dev * next_of_dev[(size_t)di].fetch_add(1);
            if (b >= n_batches) return;
            const size_t b0 = b * (size_t)batch;
            const int n = (int)std::min<size_t>(batch, utts.size() - b0);
            std::vector<int> ch(n), ready(n);
            std::vector<const float *> rows(n);
            const int stride = utts[b0].cols;
            for (int i = 0; i < n; ++i) {
              ch[i] = i;
              ready[i] = utts[b0 + i].frames;
              rows[i] = utts[b0 + i].m.data();
              if (utts[b0 + i].cols != stride) throw std::runtime_error("all matrices of a batch must have the same width");
            }
            BatchOut &o = outs[b];
            decode.InitDecoding(ch);
            decode.AdvanceDecodingHost(ch, rows, ready, stride);
            decode.FinalizeDecoding(ch);
            if (want_lattice && determinize && !second) decode.PrefetchLattices();   // the determinizer runs beside the best paths
            decode.GetBestPaths(ch, &o.best, &o.ok);
            if (want_lattice && determinize) {
              o.lats.assign(n, Lattice());
              o.lat_ok.assign(n, false);
              for (int i = 0; i < n; ++i) o.lat_ok[i] = second ? decode.GetLattice(i, &o.lats[i], slm1, slm2) : decode.GetLattice(i, &o.lats[i]);
            } else if (want_lattice) {
              decode.GetRawLattices(ch, &o.lats, &o.lat_ok);
            }
            if (nbest > 0) {
              o.nbest.resize(n);
              // (short lists for the whole batch come from one launch on the raw lattices; longer ones are NShortestPath per channel)
              for (int i = 0; i < n; ++i) {
                if (!exact_nbest) decode.GetNbestShortlist(i, o.nbest[i], nbest);
                else if (second) decode.GetNbest(i, o.nbest[i], nbest, slm1, slm2);
                else decode.GetNbest(i, o.nbest[i], nbest);
              }
            }
          }
        } catch (const std::exception &e) {
          errors[(size_t)k] = e.what();
        }
      };
      std::vector<std::thread> threads;
      for (int k = 1; k < n_workers; ++k) threads.emplace_back(worker, k);
      worker(0);
      for (std::thread &t : threads) t.join();
      for (const std::string &e : errors)
        if (!e.empty()) throw std::runtime_error(e);
      for (size_t b = 0; b < n_batches; ++b) {  // results in input order
        const size_t b0 = b * (size_t)batch;
        BatchOut &o = outs[b];
        const int n = (int)o.best.size();
        for (int i = 0; i < n; ++i) emit(utts[b0 + i], o.best[i], o.ok[i]);
        for (int i = 0; i < n && want_lattice; ++i) emit_lattice(utts[b0 + i], o.lats[i], o.lat_ok[i]);
        for (int i = 0; i < n && nbest > 0; ++i) emit_nbest(utts[b0 + i], o.nbest[i]);
      }
    }
    double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << "LOG Time taken " << elapsed << "s: real-time factor assuming 100 frames/sec is "
              << (frame_count ? elapsed * 100.0 / frame_count : 0.0) << "\n";
    std::cerr << "LOG Done " << num_success << " utterances, failed for " << num_fail << "\n";
    std::cerr << "LOG Overall log-likelihood per frame is " << (frame_count ? tot_like / frame_count : 0.0) << " over "
              << frame_count << " frames.\n";
    return num_success != 0 ? 0 : 1;
  } catch (const std::exception &e) {
    std::cerr << "ERROR " << e.what() << "\n";
    return 2;
  }
}
