// Driver of the GpuChannelPool host logic over the C-ABI test double (fake_wfstdec.cc), built with -fsanitize=thread:
// N threads x one GpuLatticeDecoder(pool) each x several ragged utterances fed in chunks through LogLikelihood pulls.
// Every utterance's result must carry ITS frames and the checksum of ITS rows (no row lost, duplicated or sent to another channel),
// misuse must come back as an exception to the misusing thread only, and the batcher must have batched.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../asr-decoder_amd/host/wfst-host.h"

using namespace datemoon;
extern "C" long long fake_calls(wfst_decoder *d, int k);

namespace {
struct Utt { int frames, cols; std::vector<float> m; };
class Pull : public DecodableInterface {
 public:
  explicit Pull(const Utt &u) : _u(u), _ready(0) {}
  float LogLikelihood(int f, int i) override { return _u.m[(size_t)f * _u.cols + i]; }
  bool IsLastFrame(int f) const override { return f == _u.frames - 1; }
  int NumFramesReady() const override { return _ready; }
  int NumIndices() const override { return _u.cols - 1; }
  void SetReady(int n) { _ready = n < _u.frames ? n : _u.frames; }
 private:
  const Utt &_u;
  int _ready;
};
double checksum(const Utt &u) {
  double s = 0;
  for (int f = 0; f < u.frames; ++f)
    for (int k = 1; k < u.cols; ++k) s += (double)u.m[(size_t)f * u.cols + k] * (double)((f % 7) + 1);
  return s;
}
}  // namespace

int main(int argc, char **argv) {
  const int n_threads = argc > 1 ? atoi(argv[1]) : 16, n_utts = argc > 2 ? atoi(argv[2]) : 96, chunk = 5;
  std::vector<Utt> utts((size_t)n_utts);
  unsigned seed = 12345;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (seed >> 8) & 0xFFFF; };
  for (Utt &u : utts) {
    u.frames = 1 + (int)(rnd() % 40);
    u.cols = 9;
    u.m.resize((size_t)u.frames * u.cols);
    for (float &x : u.m) x = (float)(rnd() % 1000) / 37.0f;
  }
  LatticeFasterDecoderConfig cfg;
  Fst fst;   // (never read: the test double ignores the graph)
  // "share": the reference's constructor shape over SHARED decoders -- ShareDevice(half the threads): the objects fill one shared
  // decoder and open a second
  const bool share = argc > 3 && std::string(argv[3]) == "share";
  if (share) GpuLatticeDecoder::ShareDevice(n_threads > 1 ? (n_threads + 1) / 2 : 1, 20);
  GpuChannelPool pool(&fst, cfg, n_threads, nullptr, /*linger_us=*/20);
  std::atomic<size_t> next(0);
  std::atomic<int> bad(0), misuse_caught(0);
  auto worker = [&](int k) {
    std::unique_ptr<GpuLatticeDecoder> dp(share ? new GpuLatticeDecoder(&fst, cfg) : new GpuLatticeDecoder(&pool));
    GpuLatticeDecoder &dec = *dp;
    DecoderItf &d = dec;
    if (k == 0) {   // misuse: AdvanceDecoding on a channel whose utterance is finalized -- this thread's exception, nobody else's
      Utt &u = utts[0];
      Pull p(u);
      p.SetReady(u.frames);
      d.InitDecoding();
      d.AdvanceDecoding(&p);
      d.FinalizeDecoding();
      try { d.AdvanceDecoding(&p); bad++; } catch (const std::runtime_error &) { misuse_caught++; }
    }
    for (;;) {
      const size_t ui = next.fetch_add(1);
      if (ui >= utts.size()) return;
      const Utt &u = utts[ui];
      Pull p(u);
      d.InitDecoding();
      for (int ready = chunk;; ready += chunk) {
        p.SetReady(ready);
        d.AdvanceDecoding(&p);
        if (d.NumFramesDecoded() != (ready < u.frames ? ready : u.frames)) bad++;
        if (ready >= u.frames) break;
        if ((ui + ready) % 3 == 0) {   // a partial result now and then
          Lattice part;
          if (!d.GetBestPath(&part, false)) bad++;
        }
      }
      d.FinalizeDecoding();
      Lattice best;
      if (!d.GetBestPath(&best)) { bad++; continue; }
      std::vector<int> words, phones;
      float tot = 0, lm = 0;
      LatticeToVector(best, words, phones, tot, lm);
      const int hops = u.frames / 16 + 1;
      if ((int)phones.size() != hops || phones[0] != u.frames) { bad++; continue; }   // ilabel = frames handed over
      const double want = checksum(u);
      if (std::fabs((double)lm - want) > 1e-3 * (1.0 + std::fabs(want))) bad++;     // graph cost of hop 0 = the rows' checksum
    }
  };
  std::vector<std::thread> th;
  for (int k = 1; k < n_threads; ++k) th.emplace_back(worker, k);
  worker(0);
  for (std::thread &t : th) t.join();
  if (share) {   // (the shared decoders' own statistics are not reachable from here: the results above are the check)
    GpuLatticeDecoder::ShareDevice(0);
    printf("bad %d misuse_caught %d frames 0/0 advance_calls 0 advance_requests 0 batches 0 fake_advance_calls 0\n", bad.load(), misuse_caught.load());
    return (bad.load() != 0 || misuse_caught.load() != 1) ? 1 : 0;
  }
  const GpuChannelPool::Stats st = pool.GetStats();
  long long frames = 0;
  for (const Utt &u : utts) frames += u.frames;
  frames += utts[0].frames;   // (the misuse probe decoded utterance 0 once more)
  printf("bad %d misuse_caught %d frames %lld/%lld advance_calls %lld advance_requests %lld batches %lld fake_advance_calls %lld\n", bad.load(), misuse_caught.load(),
         st.frames, frames, st.advance_calls, st.advance_requests, st.batches, fake_calls(pool.Handle(), 1));
  if (bad.load() != 0 || misuse_caught.load() != 1 || st.frames != frames) return 1;
  if (st.advance_requests < 2 * st.advance_calls && n_threads >= 8) return 2;   // (it batched: at least two requests per call on average)
  return 0;
}
