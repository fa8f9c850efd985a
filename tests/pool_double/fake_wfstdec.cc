// A TEST DOUBLE of the C ABI (include/wfst_decoder.h) for the host-side logic of GpuChannelPool under ThreadSanitizer: no device,
// no decoding -- a channel counts the rows it was handed and "decodes" a checksum of them, so that a driver can tell whether every
// row of every utterance reached its own channel exactly once, in order.  It also polices the pool's contract: the C ABI's calls on
// one decoder are NOT re-entrant, so any two calls that overlap abort the process.  Only the entry points the pool's batched path
// uses are defined; the test links with --unresolved-symbols=ignore-all.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/wfst_decoder.h"

struct wfst_decoder {
  int n = 0;
  std::vector<int> state;        // 0 fresh, 1 initialised, 2 finalized
  std::vector<int> rows;         // rows handed over since init
  std::vector<double> sum;       // checksum of them
  std::atomic<int> inside{0};
  std::atomic<long long> busy_until_ns{0};
  long long calls[4] = {0, 0, 0, 0};
  // the two halves of a list's best path: "on the device" for 120 us
  std::vector<int32_t> bp_list;
  int bp_ufp = 1, bp_cap = 0;
  long long bp_ready_ns = 0;
};

namespace {
thread_local std::string g_err;
long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Guard {   // two overlapping calls on one decoder = the pool broke its contract
  wfst_decoder *d;
  explicit Guard(wfst_decoder *x) : d(x) {
    if (d->inside.fetch_add(1) != 0) { fprintf(stderr, "FAKE ABI: re-entrant call on one decoder\n"); abort(); }
  }
  ~Guard() { d->inside.fetch_sub(1); }
};
int fail(int rc, const char *m) { g_err = m; return rc; }
}  // namespace

extern "C" {
const char *wfst_last_error(void) { return g_err.c_str(); }
int wfst_decoder_create(const wfst_graph *, const wfst_config *, int32_t n, const wfst_limits *, void *, wfst_decoder **out) {
  wfst_decoder *d = new wfst_decoder();
  d->n = n;
  d->state.assign(n, 0); d->rows.assign(n, 0); d->sum.assign(n, 0.0);
  *out = d;
  return WFST_OK;
}
void wfst_decoder_free(wfst_decoder *d) { delete d; }
int wfst_decoder_init(wfst_decoder *d, const int32_t *ch, int32_t n) {
  Guard g(d);
  d->calls[0]++;
  for (int i = 0; i < (ch ? n : d->n); ++i) {
    const int c = ch ? ch[i] : i;
    if (c < 0 || c >= d->n) return fail(WFST_E_ARG, "channel out of range");
    d->state[c] = 1; d->rows[c] = 0; d->sum[c] = 0.0;
  }
  return WFST_OK;
}
int wfst_decoder_advance_host(wfst_decoder *d, const int32_t *ch, int32_t n, const float *const *rows, const int32_t *ready, int32_t stride, int32_t) {
  Guard g(d);
  d->calls[1]++;
  for (int i = 0; i < n; ++i) {   // validate everything before "enqueueing" anything, as the real call does
    const int c = ch[i];
    if (c < 0 || c >= d->n) return fail(WFST_E_ARG, "channel out of range");
    if (d->state[c] != 1) return fail(WFST_E_STATE, d->state[c] == 0 ? "AdvanceDecoding before InitDecoding" : "AdvanceDecoding after FinalizeDecoding");
    if (ready[i] < d->rows[c]) return fail(WFST_E_ARG, "NumFramesReady decreased");
  }
  for (int i = 0; i < n; ++i) {
    const int c = ch[i];
    for (int f = d->rows[c]; f < ready[i]; ++f)
      for (int k = 1; k < stride; ++k) d->sum[c] += (double)rows[i][(size_t)f * stride + k] * (double)((f % 7) + 1);
    d->rows[c] = ready[i];
  }
  d->busy_until_ns = std::max(d->busy_until_ns.load(), now_ns()) + 100000;   // the "device" takes 100 us per advance call, one after the other
  std::this_thread::sleep_for(std::chrono::microseconds(30));
  return WFST_OK;
}
int wfst_decoder_finalize(wfst_decoder *d, const int32_t *ch, int32_t n) {
  Guard g(d);
  d->calls[2]++;
  for (int i = 0; i < n; ++i) {
    if (d->state[ch[i]] == 0) return fail(WFST_E_STATE, "FinalizeDecoding before InitDecoding");
    d->state[ch[i]] = 2;
  }
  return WFST_OK;
}
int wfst_decoder_sync(wfst_decoder *d) {
  Guard g(d);
  return WFST_OK;
}
int wfst_decoder_busy(wfst_decoder *d) {
  Guard g(d);
  return now_ns() < d->busy_until_ns.load() ? 1 : 0;
}
int wfst_decoder_calls_in_flight(wfst_decoder *d) {
  Guard g(d);
  const long long left = d->busy_until_ns.load() - now_ns();
  return left <= 0 ? 0 : (int)std::min<long long>(4, (left + 99999) / 100000);   // (the double's "device" runs 100 us per call)
}
int wfst_decoder_num_frames_decoded(wfst_decoder *d, int32_t c) { return d->rows[c]; }
int wfst_decoder_get_degraded_frames(wfst_decoder *d, int32_t, int32_t *n) { Guard g(d); *n = 0; return WFST_OK; }
// one hop per 16 frames + 1: ilabel = frames, olabel = channel, graph cost = the checksum, acoustic = 0
int wfst_decoder_get_best_path(wfst_decoder *d, const int32_t *ch, int32_t n, int32_t use_final, int32_t cap, int32_t *il, int32_t *ol, float *g, float *ac,
                               int32_t *n_hops) {
  Guard gd(d);
  d->calls[3]++;
  int rc = WFST_OK;
  for (int i = 0; i < n; ++i) {
    const int c = ch[i];
    if (d->state[c] == 0) return fail(WFST_E_STATE, "GetBestPath before InitDecoding");
    if (d->state[c] == 2 && !use_final) return fail(WFST_E_STATE, "finalized");
    const int hops = d->rows[c] / 16 + 1;
    n_hops[i] = hops;
    if (hops > cap) { rc = fail(WFST_E_CAPACITY, "cap"); continue; }
    for (int k = 0; k < hops; ++k) {
      il[(size_t)i * cap + k] = d->rows[c]; ol[(size_t)i * cap + k] = c + 1;
      g[(size_t)i * cap + k] = k == 0 ? (float)d->sum[c] : 0.0f; ac[(size_t)i * cap + k] = 0.0f;
    }
  }
  return rc;
}
int wfst_decoder_best_path_enqueue(wfst_decoder *d, const int32_t *ch, int32_t n, int32_t use_final, int32_t cap) {
  Guard gd(d);
  if (!d->bp_list.empty()) return fail(WFST_E_STATE, "a best-path request is outstanding");
  for (int i = 0; i < n; ++i) {
    if (d->state[ch[i]] == 0) return fail(WFST_E_STATE, "GetBestPath before InitDecoding");
    if (d->state[ch[i]] == 2 && !use_final) return fail(WFST_E_STATE, "finalized");
  }
  d->bp_list.assign(ch, ch + n);
  d->bp_ufp = use_final; d->bp_cap = cap;
  d->bp_ready_ns = now_ns() + 120000;
  return WFST_OK;
}
int wfst_decoder_best_path_ready(wfst_decoder *d) {
  Guard gd(d);
  if (d->bp_list.empty()) return fail(WFST_E_STATE, "nothing outstanding");
  return now_ns() >= d->bp_ready_ns ? 1 : 0;
}
int wfst_decoder_best_path_fetch(wfst_decoder *d, int32_t *il, int32_t *ol, float *g, float *ac, int32_t *n_hops) {
  while (now_ns() < d->bp_ready_ns) std::this_thread::sleep_for(std::chrono::microseconds(10));
  std::vector<int32_t> list;
  list.swap(d->bp_list);
  return wfst_decoder_get_best_path(d, list.data(), (int32_t)list.size(), d->bp_ufp, d->bp_cap, il, ol, g, ac, n_hops);
}
void wfst_graph_free(wfst_graph *) {}
void wfst_lm_free(wfst_lm *) {}
void *wfst_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }   // (plain memory stands in for page-locked: the pool's row slab and the objects' own buffers take the same paths)
void wfst_host_free(void *p) { free(p); }
long long fake_calls(wfst_decoder *d, int k) { return d->calls[k]; }
}
