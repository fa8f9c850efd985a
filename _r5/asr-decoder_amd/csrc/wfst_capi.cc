// C ABI (include/wfst_decoder.h) of the MI355X-native batched WFST decoder: graph upload,
// per-batch device state, the frame loop that enqueues the HIP kernels.  Host C++; compiled by
// hipcc together with wfst_kernels.hip into libwfstdec.so.  No CPU decoding path exists here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <array>
#include <vector>

#include "../../include/wfst_decoder.h"
#include "wfst_device.h"
#include "wfst_openfst.h"

using namespace wfst;
namespace wfst { int insert_kernel_set_lds(int bytes); }
static const int32_t kHeaderLabel = -2;  // ilabel_host marker of a header slot

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail(WFST_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));            \
  } while (0)

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    release();
    const hipError_t e = hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) { p = nullptr; return e; }   // (n stays 0: a failed buffer is never taken for a large enough one)
    n = count;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  size_t bytes() const { return n * sizeof(T); }
};

}  // namespace

struct wfst_graph {
  int device = 0;
  int32_t start = 0, final_state = 0, n_states = 0, n_arcs = 0;
  int32_t max_col = 0;  // largest log-likelihood column any arc reads
  int32_t max_olabel = 0, min_olabel = 0;  // over all arcs (biglm: the LMs must know every word)
  std::vector<int32_t> ilabel_host;  // original ilabels (re-mapped when tid2pdf changes)
  std::vector<uint16_t> code_host;   // degree code of the target state of every arc slot (fused rows; wfst_device.h)
  int32_t packed = 0;                // the arcs' first word = column | code << kColBits (ilabels below 2^20)
  DevBuf<int4> arcs;  // interleaved rows: header + arcs per state
  std::vector<int32_t> pos_host;  // row position of each original state id (sorted)
  int32_t orig_start = 0, orig_final = 0;
  DevBuf<int32_t> arc_ilabel, arc_olabel, arc_src, eps_target_state;
  DevBuf<int4> eps_flat, pseudo;
  DevBuf<float> pseudo_w;  // weights of every fused closure path, root to leaf (kPseudoDepthMax per path)
  int32_t fused = 0;  // epsilon closures folded into the rows as pseudo arcs (wfst_device.h)
  uint32_t start_eps = 0;
  int32_t n_eps_targets = 0;
  GraphDev view() const {
    GraphDev g;
    g.arcs = arcs.p;
    g.arc_ilabel = arc_ilabel.p;
    g.arc_olabel = arc_olabel.p;
    g.arc_src = arc_src.p;
    g.eps_target_state = eps_target_state.p;
    g.eps_flat = eps_flat.p;
    g.pseudo = pseudo.p;
    g.pseudo_w = pseudo_w.p;
    g.fused = fused;
    g.col_mask = packed ? kColMask : 0x7FFFFFFF;
    g.degcode = (packed && fused) ? 1 : 0;
    g.start_eps = start_eps;
    g.n_eps_targets = n_eps_targets;
    g.start = start;
    g.final_state = final_state;
    g.n_states = n_states;
    g.n_arcs = n_arcs;
    return g;
  }
  ~wfst_graph() {
    arcs.release();
    arc_ilabel.release();
    arc_olabel.release();
    arc_src.release();
    eps_target_state.release();
    eps_flat.release();
    pseudo.release();
    pseudo_w.release();
  }
};

struct wfst_lm {
  int device = 0;
  int32_t bos = 0, eos = 0, unk = 0, n_states = 0, n_arcs = 0, start = 0, start_arcs = 0;
  DevBuf<int4> st;
  DevBuf<int32_t> words;
  DevBuf<int2> wt;
  DevBuf<int4> hash;
  uint32_t hmask = 0;
  LmDev view() const {
    LmDev L;
    L.st = st.p;
    L.words = words.p;
    L.wt = wt.p;
    L.n_states = n_states;
    L.n_arcs = n_arcs;
    L.bos = bos;
    L.eos = eos;
    L.start = start;
    L.start_arcs = start_arcs;
    L.hash = hash.p;
    L.hmask = hmask;
    return L;
  }
  ~wfst_lm() {
    st.release();
    words.release();
    wt.release();
    hash.release();
  }
};

struct wfst_decoder {
  const wfst_graph *graph = nullptr;
  int device = 0;
  wfst_config cfg;
  wfst_limits lim = {0, 0, 0, 0, 0, 0, 0, 0};   // as resolved by create
  int32_t n_channels = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  DecoderDev D;
  DevBuf<ChanCtl> ctl;
  DevBuf<int4> tok;
  DevBuf<int32_t> emit_cnt, prune_par;
  DevBuf<int32_t> frame_off, bucket_cnt, eps_toki, eps_occ_list, eps_won_list, target, chan_list;
  DevBuf<int4> bucket, worklist, links, lat_toks;
  DevBuf<unsigned long long> pair_keys, eps_keys;  // biglm
  DevBuf<int32_t> tok_lm, bucket_lm, pair_list;
  DevBuf<int32_t> link_off, link_mid;
  DevBuf<uint2> extra;
  DevBuf<int32_t> remap;
  DevBuf<LatArc> lat_arcs;
  DevBuf<unsigned long long> lat_stats;
  DevBuf<FrameCtl> fctl;
  DevBuf<unsigned long long> dbg_t;
  DevBuf<int32_t> items, item_pref, degraded;
  int32_t det_only = -1;   // determinize_alone: the one channel wfst_decoder_get_determinized_lattice shall determinize, afresh
  DevBuf<TileDesc> tiles;
  int insert_wgs = 768;
  std::vector<int> gpar;  // step parity per group (persists across advance calls)
  int expand_wgs = 2048;
  DevBuf<float> cutoff_hist;
  DevBuf<unsigned long long> eps_vals;
  DevBuf<const float *> ll_base;
  // host mirrors (the frame loop is deterministic, so the host knows these without a read-back)
  std::vector<int32_t> h_decoded, h_target, h_state;  // state: 0 = never inited, 1 = decoding, 2 = finalized
  std::vector<const float *> h_ll_base;
  // pinned staging
  int32_t *p_target = nullptr, *p_chan = nullptr;
  const float **p_ll = nullptr;
  ChanCtl *p_ctl = nullptr;
  // best-path output buffers (device), grown on demand
  DevBuf<int32_t> bp_chain;
  DevBuf<NbEntry> nb_list;  // n-best scratch, allocated by the first wfst_decoder_get_nbest
  DevBuf<int32_t> nb_scratch, nb_out_i;
  DevBuf<float> nb_out_f;
  NbestDev nb = {};
  // determinization (wfst_decoder_get_determinized_lattice): workspace allocated by the first call
  DevBuf<int32_t> det_ws, det_result;
  DevBuf<int4> det_out_a;
  DevBuf<float2> det_out_w;
  DetDev det = {};
  // second-pass LM composition (wfst_decoder_get_rescored_lattice): workspace allocated by the first call
  DevBuf<int32_t> cmp_ws, cmp_result, cmp_fin;
  DevBuf<int32_t> np_ws, np_out, np_off, np_arcs;   // n cheapest paths of a determinized / rescored lattice (wfst_decoder_get_nbest_paths)
  DevBuf<float> np_tot;
  DevBuf<NbPathEntry> np_lists;
  DevBuf<int4> cmp_out_a;
  DevBuf<float2> cmp_out_w;
  CmpDev cmp = {};
  int32_t cmp_slots = 0;              // lattices one compose launch takes
  // results of the batched post-processing (wfst_decoder_rescore_lattices / wfst_decoder_nbest_paths_batch), served by the per-channel
  // fetch calls while they still belong to the channel's state and to the same request
  struct PostKey { const wfst_lm *o = nullptr, *n = nullptr; int32_t use_final = 0, n_paths = 0, decoded = -1; bool valid = false; };
  struct RescLattice { PostKey key; int32_t n_states = 0; std::vector<int4> a; std::vector<float2> w; std::vector<int32_t> fin; };
  struct NbPaths { PostKey key; std::vector<int32_t> off, olabel; std::vector<float> tot, graph, ac; };
  std::vector<RescLattice> resc_cache;
  std::vector<NbPaths> nbp_cache;
  // what the determinizer's / the composition's workspace slots hold since the last batched call (one chunk): a batch of n-best
  // requests right behind the batch of second passes of the same channels starts from those lattices
  std::vector<int32_t> post_dev_list, post_dev_decoded, post_dev_dres, post_dev_cres;
  const wfst_lm *post_dev_o = nullptr, *post_dev_n = nullptr;
  struct DetLattice { int32_t n_states = 0, n_proper = 0, err = 0; std::vector<int4> a; std::vector<float2> w; };
  std::vector<DetLattice> det_cache;
  std::vector<char> det_cached;
  std::vector<int32_t> det_live_nd;   // NumFramesDecoded() the cached lattice of a LIVE channel belongs to (-1: none)
  std::vector<char> det_live_final = std::vector<char>();
  int32_t det_slots = 0;              // lattices one determinize launch takes (workspace slots)
  // a batch's determinized lattices packed back to back on the device (det_pack_kernel) and their pinned landing place on the host
  DevBuf<int4> det_pack_a;
  DevBuf<float2> det_pack_w;
  void *det_pack_pin = nullptr;
  size_t det_pack_pin_bytes = 0;
  // wfst_decoder_prefetch_determinized: a determinize launch in flight on a side stream (its channels, its result words)
  int stagger_us = 0;   // lattice decoders with several channel groups: group g starts its frame chain g x this many microseconds late
  bool pf_pending = false;
  bool pf_detached = false;           // the pending prefetch runs detached (wfst_decoder_prefetch_determinized_detached)
  std::vector<DetLattice> pf_cache;   // detached: the lattices of the last harvested prefetch, whatever the channels have gone on to
  std::vector<char> pf_have;
  std::vector<int32_t> fin_epoch, pf_epoch;   // FinalizeDecoding calls per channel; ... as of the pending prefetch
  std::vector<int32_t> pf_list, pf_res;
  int32_t *pf_pin = nullptr;          // [2][n_channels * 4] pinned: the channel list going up, the launch's result words coming down (a copy from or
                                      // to pageable memory would hold the calling thread until the launch is over)
  DevBuf<int32_t> pf_dev;
  hipStream_t det_stream = nullptr;   // (a decoder without channel groups; otherwise the second group's stream, idle between advances)
  hipEvent_t pf_ev_start = nullptr, pf_ev_done = nullptr;
  // host-fed log-likelihood history (advance_host)
  hipStream_t copy_stream = nullptr;  // host -> device uploads of advance_host
  // pruned lattices fetched from the device (lattice mode): filled for ALL finalized channels by the
  // first wfst_decoder_get_raw_lattice after a FinalizeDecoding, dropped by init / finalize
  std::vector<std::vector<int4> > lat_cache_tok;
  std::vector<std::vector<LatArc> > lat_cache_arc;
  std::vector<char> lat_cached;
  char *lat_pin = nullptr;  // pinned staging for that fetch
  DevBuf<int32_t> bp_all;   // GetBestPath: {n_hops | ilabel | olabel | graph | acoustic} of the listed channels, one D2H copy
  char *bp_pin = nullptr;   // its pinned staging
  size_t bp_pin_bytes = 0;
  size_t lat_pin_bytes = 0;
  std::vector<int32_t> lat_cache_nd;
  std::vector<float *> hist_dev;
  std::vector<size_t> hist_rows_cap;
  std::vector<int32_t> hist_rows;
  int32_t hist_stride = 0;
  int tiles_per_channel = 16;
  int upload_slice = 48;  // wfst_options.upload_slice_frames
  // the frame loop of one advance call, captured once per (frames per group, stride) and replayed
  bool use_graph = true;
  std::map<std::vector<int>, hipGraphExec_t> graphs;
  // channel groups: each runs its frame loop on its own stream (fork/join around the main stream)
  int n_groups = 1;
  std::vector<hipStream_t> gstreams;
  std::vector<hipEvent_t> gevents;  // [0] fork, [1..] join per group
  // optional kernel timing (wfst_decoder_set_profiling)
  bool profiling = false;
  std::vector<hipEvent_t> ev_pool;
  std::vector<std::pair<int, int>> ev_pairs[4];  // [kernel class] -> (start, stop) event indices
  std::vector<std::array<int, 4>> ev_log;        // the same launches in order: {kind, channel group, start, stop} (WFST_PROFILE_DUMP of a WFST_AB_SWITCHES build)
  size_t ev_used = 0;
  int ev_get() {
    if (ev_used == ev_pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return -1;
      ev_pool.push_back(e);
    }
    return (int)ev_used++;
  }

  ~wfst_decoder() {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    if (pf_pending && pf_ev_done) (void)hipEventSynchronize(pf_ev_done);   // (a prefetching determinizer still reads the decoder's buffers)
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (det_stream) (void)hipStreamDestroy(det_stream);
    if (pf_ev_start) (void)hipEventDestroy(pf_ev_start);
    if (pf_ev_done) (void)hipEventDestroy(pf_ev_done);
    if (lat_pin) (void)hipHostFree(lat_pin);
    if (bp_pin) (void)hipHostFree(bp_pin);
    for (float *p : hist_dev)
      if (p) (void)hipFree(p);
    for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    for (auto &kv : graphs) (void)hipGraphExecDestroy(kv.second);
    for (hipStream_t st : gstreams) if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : gevents) (void)hipEventDestroy(ev);
    if (p_target) (void)hipHostFree(p_target);
    if (p_chan) (void)hipHostFree(p_chan);
    if (p_ll) (void)hipHostFree((void *)p_ll);
    if (p_ctl) (void)hipHostFree(p_ctl);
    if (pf_pin) (void)hipHostFree(pf_pin);
    if (det_pack_pin) (void)hipHostFree(det_pack_pin);
    pair_keys.release(); pair_list.release(); eps_keys.release(); tok_lm.release(); bucket_lm.release(); remap.release();
    det_ws.release(); det_result.release(); det_out_a.release(); det_out_w.release(); det_pack_a.release(); det_pack_w.release(); pf_dev.release();
    cmp_ws.release(); cmp_result.release(); cmp_fin.release(); cmp_out_a.release(); cmp_out_w.release();
    np_ws.release(); np_out.release(); np_off.release(); np_arcs.release(); np_tot.release(); np_lists.release();
    ctl.release(); tok.release(); frame_off.release(); bucket_cnt.release(); emit_cnt.release(); prune_par.release();
    eps_toki.release(); eps_occ_list.release(); eps_won_list.release(); worklist.release(); target.release(); chan_list.release();
    bucket.release(); links.release(); lat_toks.release(); lat_arcs.release(); lat_stats.release(); link_off.release(); link_mid.release(); extra.release(); fctl.release(); dbg_t.release(); items.release(); item_pref.release(); degraded.release(); tiles.release(); cutoff_hist.release(); eps_vals.release(); ll_base.release(); nb_list.release(); nb_scratch.release(); nb_out_i.release(); nb_out_f.release(); bp_chain.release(); bp_all.release();
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }
};

extern "C" {

void wfst_config_default(wfst_config *c) {  // lattice-faster-decoder-conf.h:35-44
  c->beam = 16.0f;
  c->max_active = 2147483647;
  c->min_active = 200;
  c->lattice_beam = 10.0f;
  c->prune_interval = 25;
  c->beam_delta = 0.5f;
  c->hash_ratio = 2.0f;
  c->prune_scale = 0.1f;
}

void wfst_options_default(wfst_options *o) {
  o->channel_groups = 0;
  o->use_hip_graph = 1;
  o->log2_partitions = -1;  // by the decoder's kind (wfst_decoder_create_ex): 5 -- 32 partitions: measured best at batch 128 for best-path decoders (16: insert slower, 64: more bucket atomics) --, 6 for lattice decoders
  o->log2_lds_slots = 12;
  o->joint_max = 1536;
  o->expand_workgroups = 0;   // by the decoder's kind: 2048, lattice decoders 3072
  o->insert_workgroups = 768;
  o->upload_slice_frames = 48;
  o->tile_tokens = 256;
  o->debug = 0;
}

void wfst_graph_options_default(wfst_graph_options *o) {
  o->row_align_slots = 8;
  o->flatten_closures = 1;
  o->fuse_closures = 1;
}

const char *wfst_last_error(void) { return g_err.c_str(); }

int wfst_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

/* ---------------------------------------------------------------- graph */

// Device row layout ("ext" index space): state s owns the slots [pos(s), pos(s) + 1 + num_arcs(s))
// of ONE int4 array, pos(s) = arc_begin(s) + s.  Slot pos(s) is the state's header
// {(n_emit << 12) | n_eps, original state id, own next_eps word, 0}; its arcs follow, epsilon arcs
// first.  A state is identified on the device by pos(s), so finding a token's arcs is ONE gather
// whose cache lines also hold the arcs themselves.  Labels and sources live in cold arrays with
// the same indexing.
static int upload_columns(wfst_graph *g, const int32_t *tid2pdf, int32_t n_tid, std::vector<int4> *ext) {
  const int64_t N = (int64_t)g->ilabel_host.size();
  int32_t max_col = 0;
  std::vector<int4> cur;
  if (!ext) {  // set_tid2pdf: read-modify-write of the column word only
    cur.resize((size_t)N);
    HIP_TRY(hipMemcpy(cur.data(), g->arcs.p, (size_t)N * sizeof(int4), hipMemcpyDeviceToHost));
    ext = &cur;
  }
  for (int64_t a = 0; a < N; ++a) {
    const int32_t il = g->ilabel_host[a];
    if (il == kHeaderLabel) continue;
    int32_t col = -1;
    if (il != 0) {
      if (tid2pdf) {
        if (il < 0 || il > n_tid) return fail(WFST_E_ARG, "arc ilabel outside tid2pdf range");
        col = tid2pdf[il];
      } else {
        col = il;
      }
      if (col < 0) return fail(WFST_E_ARG, "negative log-likelihood column");
      if (g->packed && col > kColMask) return fail(WFST_E_ARG, "log-likelihood column beyond 2^20 on a graph whose ilabels stay below it");
      max_col = std::max(max_col, col);
      if (g->packed) col |= (int32_t)((uint32_t)g->code_host[(size_t)a] << kColBits);
    }
    (*ext)[a].x = col;
  }
  HIP_TRY(hipMemcpy(g->arcs.p, ext->data(), (size_t)N * sizeof(int4), hipMemcpyHostToDevice));
  g->max_col = max_col;
  return WFST_OK;
}

int wfst_graph_from_arrays(int32_t start, int32_t final_state, int32_t n_states, int32_t n_arcs,
                           const wfst_state_info *states, const wfst_arc *arcs, int device,
                           wfst_graph **out) {
  return wfst_graph_from_arrays_ex(start, final_state, n_states, n_arcs, states, arcs, device, nullptr, out);
}

int wfst_graph_from_arrays_ex(int32_t start, int32_t final_state, int32_t n_states, int32_t n_arcs,
                              const wfst_state_info *states, const wfst_arc *arcs, int device,
                              const wfst_graph_options *gopt, wfst_graph **out) {
  if (!out) return fail(WFST_E_ARG, "out is NULL");
  wfst_graph_options GO;
  wfst_graph_options_default(&GO);
  if (gopt) GO = *gopt;
  if (GO.row_align_slots < 1 || GO.row_align_slots > 64) return fail(WFST_E_ARG, "row_align_slots must be 1..64");
  *out = nullptr;
  if (n_states <= 0 || n_arcs < 0 || !states || (n_arcs > 0 && !arcs))
    return fail(WFST_E_ARG, "empty graph or NULL arrays");
  if (start < 0 || start >= n_states) return fail(WFST_E_ARG, "start state out of range");
  int ndev = wfst_device_count();
  if (ndev <= 0) return fail(WFST_E_DEVICE, "no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(WFST_E_ARG, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  if ((int64_t)n_arcs + n_states >= (int64_t)kNoArc)
    return fail(WFST_E_FORMAT, "graphs with states + arcs >= 2^29 are not supported");
  // A row (header + arcs, 16-byte slots) that fits in k lines is placed so that it touches only k lines:
  // expanding a token is a random gather, priced per LINE -- 128 bytes on MI355X, whatever part of it is
  // used (tools/ubench_gather_granularity.hip: 16-, 32-, 64- and 128-byte aligned random reads all run at
  // the same ~50 G reads/s; a 128-byte read straddling two lines at 2/3 of that) -- and an unaligned 3-arc
  // row straddles two.  Costs ~15 % padding slots.  row_align_slots = 1 packs the rows tightly.
  const int64_t line_slots = GO.row_align_slots;

  // pass 0: validate, arc offsets, epsilon targets
  std::vector<int64_t> aoff((size_t)n_states + 1, 0);
  std::vector<uint8_t> is_target((size_t)n_states, 0);
  int64_t off = 0;
  for (int32_t s = 0; s < n_states; ++s) {
    const uint32_t na = states[s].num_arcs, ne = states[s].niepsilons;
    if (ne > na || off + na > n_arcs) return fail(WFST_E_FORMAT, "state arc counts inconsistent with total_arcs");
    if (ne > kEpsMask) return fail(WFST_E_FORMAT, "state with more than 4095 input-epsilon arcs");
    if (na - ne >= (1u << (32 - kEpsBits))) return fail(WFST_E_FORMAT, "state with 2^20 or more emitting arcs");
    aoff[(size_t)s] = off;
    for (uint32_t i = 0; i < na; ++i) {
      const wfst_arc &a = arcs[off + i];
      if ((i < ne) != (a.ilabel == 0))
        return fail(WFST_E_FORMAT, "input-epsilon arcs must precede a state's other arcs (reference flat format)");
      if (a.nextstate < 0 || a.nextstate >= n_states) return fail(WFST_E_FORMAT, "arc nextstate out of range");
      if (i < ne) is_target[a.nextstate] = 1;
    }
    off += na;
  }
  aoff[(size_t)n_states] = off;
  if (off != n_arcs) return fail(WFST_E_FORMAT, "sum of num_arcs != total_arcs");

  // pass 0b: FUSED epsilon closures (wfst_device.h "pseudo arcs").  The whole epsilon closure of every
  // state with epsilon arcs out as a list of paths {target, last arc, parent path, weight of the last
  // arc}, parents first.  Fusable iff every closure is small and shallow (no epsilon cycle) and no
  // epsilon arc has a negative weight (then a path's cost is never below its prefix's, and one test of
  // the arrival against the cutoff stands for ProcessNonemitting's test at every hop, base-inl.h:391,415).
  struct ClPath { int32_t target, src, arc_i, parent, depth; float w; };
  constexpr int kClPathCap = 48, kClDepthCap = kPseudoDepthMax;
  std::vector<int32_t> cl_first((size_t)n_states, 0), cl_cnt((size_t)n_states, 0);
  std::vector<ClPath> cl;
  bool fused = GO.fuse_closures != 0;
  for (int32_t s = 0; s < n_states && fused; ++s) {
    if (!states[s].niepsilons) continue;
    const size_t first = cl.size();
    cl_first[(size_t)s] = (int32_t)first;
    // breadth first: the state's own epsilon arcs, then those of every path's end state in list order
    for (int64_t k = -1; fused && (k < 0 || (size_t)k < cl.size() - first); ++k) {
      const int32_t u = k < 0 ? s : cl[first + (size_t)k].target;
      const int32_t dep = k < 0 ? 1 : cl[first + (size_t)k].depth + 1;
      for (uint32_t i = 0; i < states[u].niepsilons; ++i) {
        const wfst_arc &a = arcs[aoff[(size_t)u] + i];
        if (!(a.weight >= 0.0f) || dep > kClDepthCap || cl.size() - first >= (size_t)kClPathCap) { fused = false; break; }
        cl.push_back(ClPath{a.nextstate, u, (int32_t)i, (int32_t)k, dep, a.weight});
      }
    }
    cl_cnt[(size_t)s] = (int32_t)(cl.size() - first);
  }
  std::vector<int32_t> n_pseudo((size_t)n_states, 0);
  if (fused) {
    for (int32_t s = 0; s < n_states && fused; ++s) {
      int64_t np = 0;
      for (uint32_t i = states[s].niepsilons; i < states[s].num_arcs; ++i) np += cl_cnt[(size_t)arcs[aoff[(size_t)s] + i].nextstate];
      if (np >= (1 << 20)) fused = false;
      n_pseudo[(size_t)s] = (int32_t)np;
    }
  }
  if (!fused) {
    cl.clear();
    std::fill(n_pseudo.begin(), n_pseudo.end(), 0);
    std::fill(cl_cnt.begin(), cl_cnt.end(), 0);
  }

  // pass 1: positions
  std::vector<int32_t> pos((size_t)n_states);
  int64_t next_slot = 0;
  for (int32_t s = 0; s < n_states; ++s) {
    const int64_t sz = 1 + (int64_t)states[s].num_arcs + 2 * (int64_t)n_pseudo[(size_t)s], in_line = next_slot % line_slots;
    if ((in_line + sz + line_slots - 1) / line_slots > (sz + line_slots - 1) / line_slots)
      next_slot += line_slots - in_line;
    pos[s] = (int32_t)next_slot;
    next_slot += sz;
    if (next_slot >= (int64_t)kNoArc) return fail(WFST_E_FORMAT, "graph too large for 29-bit row indices");
  }
  const int64_t N = next_slot;  // slots of rows[] (headers + arcs + pseudo arcs + padding)
  // next_eps word per state: bit 31 = has outgoing epsilon arcs, bits 30..0 = 1 + ordinal among
  // the epsilon-target states (the index of the state's slot in every channel's epsilon table)
  std::vector<uint32_t> next_eps((size_t)n_states, 0u);
  std::vector<int32_t> h_targets;
  for (int32_t st = 0; st < n_states; ++st) {
    if (states[st].niepsilons) next_eps[st] |= kFlagOutEps;
    if (is_target[st]) {
      h_targets.push_back(pos[st]);
      next_eps[st] |= (uint32_t)h_targets.size();
    }
  }
  // degree code of every state (wfst_device.h), carried by the arcs that lead to it; graphs with ilabels of 2^20 and more
  // keep the plain column word
  bool packed = true;
  for (int64_t i = 0; i < n_arcs && packed; ++i) packed = arcs[i].ilabel >= 0 && arcs[i].ilabel <= kColMask;
  auto code_of = [&](int32_t st) -> uint32_t {
    return pack_code(states[st].niepsilons, states[st].num_arcs - states[st].niepsilons, (uint32_t)n_pseudo[(size_t)st]);
  };
  std::vector<uint16_t> h_code((size_t)N, (uint16_t)kCodeUnknown);
  // pass 2: rows
  std::vector<int4> ext((size_t)N, make_int4(0, -1, 0, 0));
  std::vector<int32_t> h_src((size_t)N, 0), h_il((size_t)N, kHeaderLabel), h_ol((size_t)N, 0);
  int32_t max_ol = 0, min_ol = 0;
  off = 0;
  for (int32_t s = 0; s < n_states; ++s) {
    const uint32_t na = states[s].num_arcs, ne = states[s].niepsilons;
    int4 h;
    h.x = (int32_t)(((na - ne) << kEpsBits) | ne);
    h.y = s;
    h.z = n_pseudo[(size_t)s];  // pseudo arcs behind the emitting arcs (fused closures)
    h.w = 0;
    ext[(size_t)pos[s]] = h;
    {  // pseudo arcs: one per (emitting arc, path of its target's closure)
      size_t q = (size_t)pos[s] + 1 + na;
      for (uint32_t i = ne; i < na && fused; ++i) {
        const wfst_arc &a = arcs[off + i];
        for (int32_t p = 0; p < cl_cnt[(size_t)a.nextstate]; ++p, q += 2) {
          const ClPath &cp = cl[(size_t)cl_first[(size_t)a.nextstate] + p];
          int4 v;  // two slots per pseudo arc, loaded together: no dependent look-up for a one-hop path
          v.x = -1;
          v.y = cl_first[(size_t)a.nextstate] + p;  // its path in pseudo[] (walked for paths of several hops)
          memcpy(&v.z, &a.weight, 4);               // weight of the emitting arc
          v.w = pos[cp.target];
          ext[q] = v;
          h_code[q] = (uint16_t)code_of(cp.target);  // the token of the path's end state
          h_src[q] = pos[s];
          h_il[q] = a.ilabel;                        // same log-likelihood column as the emitting arc
          h_ol[q] = 0;
          int4 b;
          b.x = (int32_t)((uint32_t)(pos[cp.src] + 1 + cp.arc_i) | flags_of(next_eps[(size_t)cp.target]));  // last arc | flags of the end state
          memcpy(&b.y, &cp.w, 4);                    // weight of the last arc
          b.z = cp.depth;
          b.w = 0;
          if (cp.depth >= 2) memcpy(&b.w, &cl[(size_t)cl_first[(size_t)a.nextstate] + cp.parent].w, 4);  // weight of the hop before
          ext[q + 1] = b;                            // (h_il stays kHeaderLabel: no log-likelihood column here)
        }
      }
    }
    for (uint32_t i = 0; i < na; ++i) {
      const wfst_arc &a = arcs[off + i];
      const size_t q = (size_t)pos[s] + 1 + i;
      int4 v;
      v.x = -1;
      v.y = (int32_t)next_eps[a.nextstate];
      memcpy(&v.z, &a.weight, 4);
      v.w = pos[a.nextstate];
      ext[q] = v;
      h_code[q] = (uint16_t)code_of(a.nextstate);
      h_src[q] = (int32_t)((uint32_t)pos[s] | (i < ne ? 0x80000000u : 0u));
      h_il[q] = a.ilabel;
      h_ol[q] = a.olabel;
      max_ol = std::max(max_ol, a.olabel);
      min_ol = std::min(min_ol, a.olabel);
    }
    off += na;
  }

  // pass 3: flattened epsilon closures (wfst_device.h "eps_flat"): breadth-first over a state's
  // epsilon arcs; a closure with more than kFlatMax paths (or an epsilon cycle) stays iterative
  std::vector<int4> h_flat;
  if (GO.flatten_closures) {
    struct Node { int32_t state, parent; };
    for (int32_t s = 0; s < n_states; ++s) {
      if (!states[s].niepsilons) continue;
      int4 ent[kFlatMax];
      Node queue[kFlatMax + 1];
      int n_ent = 0, qh = 0, qt = 0;
      bool complete = true;
      queue[qt++] = Node{s, -1};
      while (qh < qt && complete) {
        const Node u = queue[qh++];
        for (uint32_t i = 0; i < states[u.state].niepsilons; ++i) {
          if (n_ent == kFlatMax) { complete = false; break; }
          const wfst_arc &a = arcs[aoff[u.state] + i];
          const int32_t v = a.nextstate;
          int4 e;
          e.x = (int32_t)(next_eps[v] & 0x7FFFFFFFu) - 1;
          e.y = pos[u.state] + 1 + (int32_t)i;
          e.z = (u.parent + 1) | (states[v].niepsilons ? 8 : 0);
          memcpy(&e.w, &a.weight, 4);
          ent[n_ent] = e;
          if (states[v].niepsilons) queue[qt++] = Node{v, n_ent};  // qt <= n_ent + 1 <= kFlatMax
          ++n_ent;
        }
      }
      if (!complete || n_ent == 0) continue;
      ext[(size_t)pos[s]].w = (int32_t)(((uint32_t)h_flat.size() << 3) | (uint32_t)n_ent);
      h_flat.insert(h_flat.end(), ent, ent + n_ent);
    }
    if (h_flat.size() >= (1u << 28)) return fail(WFST_E_FORMAT, "too many flattened epsilon closures");
  }

  // the paths themselves: {last arc (row index) | flags of the target state, parent path or -1, weight bits, depth}
  std::vector<int4> h_pseudo(cl.size());
  for (size_t k = 0; k < cl.size(); ++k) {
    const ClPath &cp = cl[k];
    int4 v;
    v.x = (int32_t)((uint32_t)(pos[cp.src] + 1 + cp.arc_i) | flags_of(next_eps[(size_t)cp.target]));
    v.y = -1;  // parent: set below (indices are relative to the closure's first entry)
    memcpy(&v.z, &cp.w, 4);
    v.w = cp.depth;
    h_pseudo[k] = v;
  }
  {
    for (int32_t s = 0; s < n_states; ++s)
      for (int32_t p = 0; p < cl_cnt[(size_t)s]; ++p) {
        const size_t k = (size_t)cl_first[(size_t)s] + p;
        h_pseudo[k].y = cl[k].parent < 0 ? -1 : cl_first[(size_t)s] + cl[k].parent;
      }
  }

  // ... and their weights in path order, for the expansion of a path of three hops or more (the sums must run root to leaf)
  std::vector<float> h_pseudo_w(std::max<size_t>(1, cl.size()) * (size_t)kPseudoDepthMax, 0.0f);
  for (int32_t st = 0; st < n_states; ++st)
    for (int32_t pth = 0; pth < cl_cnt[(size_t)st]; ++pth) {
      const size_t k = (size_t)cl_first[(size_t)st] + pth;
      int32_t q = pth, dep = cl[k].depth;
      while (q >= 0 && dep > 0) {   // leaf to root
        const ClPath &cp = cl[(size_t)cl_first[(size_t)st] + q];
        h_pseudo_w[k * kPseudoDepthMax + (size_t)(dep - 1)] = cp.w;
        q = cp.parent;
        --dep;
      }
    }

  wfst_graph *g = new wfst_graph();
  g->device = device;
  g->fused = fused ? 1 : 0;
  g->start = pos[start];
  g->final_state = (final_state >= 0 && final_state < n_states) ? pos[final_state] : -1;
  g->orig_start = start;
  g->orig_final = final_state;
  g->n_states = n_states;
  g->n_arcs = n_arcs;
  g->max_olabel = max_ol;
  g->min_olabel = min_ol;
  g->ilabel_host.swap(h_il);
  g->code_host.swap(h_code);
  g->packed = packed ? 1 : 0;
  g->pos_host.swap(pos);
  g->start_eps = next_eps[start];
  g->n_eps_targets = (int32_t)h_targets.size();
  hipError_t e;
  if ((e = g->arcs.alloc((size_t)N + 8)) != hipSuccess ||   // (+8: kernels that ask for a row's first arcs WITH its header read a few slots past a short row)
      (e = g->arc_ilabel.alloc((size_t)N)) != hipSuccess ||
      (e = g->arc_olabel.alloc((size_t)N)) != hipSuccess || (e = g->arc_src.alloc((size_t)N)) != hipSuccess ||
      (e = g->eps_target_state.alloc(h_targets.size())) != hipSuccess ||
      (e = g->eps_flat.alloc(std::max<size_t>(1, h_flat.size()))) != hipSuccess ||
      (e = g->pseudo.alloc(std::max<size_t>(1, h_pseudo.size()))) != hipSuccess ||
      (e = g->pseudo_w.alloc(h_pseudo_w.size())) != hipSuccess) {
    delete g;
    return fail(WFST_E_DEVICE, std::string("hipMalloc(graph): ") + hipGetErrorString(e));
  }
  int rc = WFST_OK;
  if (hipMemcpy(g->arc_src.p, h_src.data(), h_src.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(g->arc_olabel.p, h_ol.data(), h_ol.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
      (!h_targets.empty() && hipMemcpy(g->eps_target_state.p, h_targets.data(), h_targets.size() * 4, hipMemcpyHostToDevice) != hipSuccess) ||
      (!h_flat.empty() && hipMemcpy(g->eps_flat.p, h_flat.data(), h_flat.size() * sizeof(int4), hipMemcpyHostToDevice) != hipSuccess) ||
      (!h_pseudo.empty() && hipMemcpy(g->pseudo.p, h_pseudo.data(), h_pseudo.size() * sizeof(int4), hipMemcpyHostToDevice) != hipSuccess) ||
      hipMemcpy(g->pseudo_w.p, h_pseudo_w.data(), h_pseudo_w.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(g->arc_ilabel.p, g->ilabel_host.data(), g->ilabel_host.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
    rc = fail(WFST_E_DEVICE, "hipMemcpy(graph) failed");
  if (rc == WFST_OK) rc = upload_columns(g, nullptr, 0, &ext);
  if (rc != WFST_OK) {
    delete g;
    return rc;
  }
  *out = g;
  return WFST_OK;
}

int wfst_graph_load(const char *path, int device, wfst_graph **out) {
  return wfst_graph_load_ex(path, device, nullptr, out);
}

int wfst_graph_load_ex(const char *path, int device, const wfst_graph_options *gopt, wfst_graph **out) {
  if (!path || !out) return fail(WFST_E_ARG, "NULL argument");
  *out = nullptr;
  wfst::HostGraph hg;
  std::string err;
  const int rc = wfst::read_graph_file(path, &hg, &err);
  if (rc != WFST_OK) return fail(rc, err);
  return wfst_graph_from_arrays_ex(hg.start, hg.final_state, (int32_t)hg.states.size(), (int32_t)hg.arcs.size(),
                                   hg.states.data(), hg.arcs.data(), device, gopt, out);
}

int wfst_graph_convert_file(const char *in_path, const char *flat_out_path) {
  if (!in_path || !flat_out_path) return fail(WFST_E_ARG, "NULL argument");
  wfst::HostGraph hg;
  std::string err;
  int rc = wfst::read_graph_file(in_path, &hg, &err);
  if (rc != WFST_OK) return fail(rc, err);
  rc = wfst::write_flat_graph(flat_out_path, hg, &err);
  return rc == WFST_OK ? WFST_OK : fail(rc, err);
}

int wfst_graph_set_tid2pdf(wfst_graph *g, const int32_t *tid2pdf, int32_t n_tid) {
  if (!g) return fail(WFST_E_ARG, "NULL graph");
  HIP_TRY(hipSetDevice(g->device));
  return upload_columns(g, tid2pdf, n_tid, nullptr);
}

int wfst_graph_info(const wfst_graph *g, int32_t *start, int32_t *final_state, int32_t *n_states,
                    int32_t *n_arcs, int64_t *device_bytes) {
  if (!g) return fail(WFST_E_ARG, "NULL graph");
  if (start) *start = g->orig_start;
  if (final_state) *final_state = g->orig_final;
  if (n_states) *n_states = g->n_states;
  if (n_arcs) *n_arcs = g->n_arcs;
  if (device_bytes)
    *device_bytes = (int64_t)(g->arcs.bytes() + g->arc_ilabel.bytes() +
                              g->arc_olabel.bytes() + g->arc_src.bytes() + g->eps_target_state.bytes() + g->eps_flat.bytes() +
                              g->pseudo.bytes() + g->pseudo_w.bytes());
  return WFST_OK;
}

void wfst_graph_free(wfst_graph *g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  delete g;
}

/* ------------------------------------------------------------------- LM */

int wfst_lm_from_arrays(int32_t bos, int32_t eos, int32_t unk, int32_t n_states, const wfst_lm_state *states,
                        int32_t n_arcs, const wfst_lm_arc *arcs, float scale, int device, wfst_lm **out) {
  if (!out) return fail(WFST_E_ARG, "out is NULL");
  *out = nullptr;
  if (n_states <= 0 || n_arcs < 0 || !states || (n_arcs > 0 && !arcs)) return fail(WFST_E_ARG, "empty LM or NULL arrays");
  int ndev = wfst_device_count();
  if (ndev <= 0) return fail(WFST_E_DEVICE, "no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(WFST_E_ARG, "device index out of range");
  // the checks the reference leaves out (it indexes and follows whatever the file holds)
  std::vector<int4> st((size_t)n_states);
  int64_t off = 0;
  for (int32_t s = 0; s < n_states; ++s) {
    const wfst_lm_state &x = states[s];
    if (x.arc_num < 0 || off + x.arc_num > n_arcs) return fail(WFST_E_FORMAT, "LM state arc counts inconsistent with the arc total");
    if (x.backoff_id < 0 || x.backoff_id >= n_states) return fail(WFST_E_FORMAT, "LM back-off state out of range");
    int4 v;
    v.x = (int32_t)off;
    v.y = x.arc_num;
    const float bw = scale != 1.0f ? x.backoff_prob * scale : x.backoff_prob;  // Fsa::Rescale, arpa2fsa.cc:264-275
    memcpy(&v.z, &bw, 4);
    v.w = x.backoff_id;
    st[(size_t)s] = v;
    for (int32_t i = 0; i < x.arc_num; ++i) {
      const wfst_lm_arc &a = arcs[off + i];
      if (a.tostateid < 0 || a.tostateid >= n_states) return fail(WFST_E_FORMAT, "LM arc destination out of range");
      if (s == 0 ? a.wordid != i : (i > 0 && arcs[off + i - 1].wordid >= a.wordid))
        return fail(WFST_E_FORMAT, s == 0 ? "LM state 0 must hold arc k for word id k (arpa2fsa.cc:253-254)"
                                          : "LM arcs of a state must be sorted by word id (arpa2fsa.h:194-210)");
    }
    off += x.arc_num;
  }
  if (off != n_arcs) return fail(WFST_E_FORMAT, "sum of LM arc counts != total arcs");
  for (int32_t s = 0; s < n_states; ++s) {  // every back-off chain reaches the empty history
    int32_t t = s, hops = 0;
    while (t != 0 && hops <= 64) { t = states[t].backoff_id; ++hops; }
    if (t != 0) return fail(WFST_E_FORMAT, "LM back-off chain does not end in state 0");
  }
  if (bos < 0 || bos >= states[0].arc_num || eos < 0 || eos >= states[0].arc_num)
    return fail(WFST_E_FORMAT, "LM <s> / </s> ids have no arc from the empty-history state");
  std::vector<int32_t> words((size_t)std::max(n_arcs, 1));
  std::vector<int2> wt((size_t)std::max(n_arcs, 1));
  for (int32_t a = 0; a < n_arcs; ++a) {
    words[(size_t)a] = arcs[a].wordid;
    const float w = scale != 1.0f ? arcs[a].weight * scale : arcs[a].weight;
    int2 v;
    memcpy(&v.x, &w, 4);
    v.y = arcs[a].tostateid;
    wt[(size_t)a] = v;
  }
  // the (state, word) table of every state but the empty history (LmDev::hash)
  size_t hsize = 1024;
  while (hsize < 4 * (size_t)std::max(1, n_arcs - states[0].arc_num)) hsize <<= 1;   // (at most a quarter full: short probe sequences, lm_step)
  std::vector<int4> htab(hsize, make_int4(-1, -1, 0, 0));
  {
    int64_t o = 0;
    for (int32_t s = 0; s < n_states; ++s) {
      for (int32_t i = 0; i < states[s].arc_num && s != 0; ++i) {
        const wfst_lm_arc &a = arcs[o + i];
        uint32_t slot = lm_hash(s, a.wordid) & (uint32_t)(hsize - 1);
        while (htab[slot].x >= 0) slot = (slot + 1) & (uint32_t)(hsize - 1);
        htab[slot] = make_int4(s, a.wordid, wt[(size_t)(o + i)].x, a.tostateid);
      }
      o += states[s].arc_num;
    }
  }
  HIP_TRY(hipSetDevice(device));
  wfst_lm *lm = new wfst_lm();
  lm->device = device;
  lm->bos = bos;
  lm->eos = eos;
  lm->unk = unk;
  lm->n_states = n_states;
  lm->n_arcs = n_arcs;
  lm->start_arcs = states[0].arc_num;
  lm->hmask = (uint32_t)(hsize - 1);
  lm->start = arcs[bos].tostateid;  // ComposeArpaLm::Start: the arc of state 0 for <s> (compose-arpalm.cc:5-13)
  hipError_t e;
  if ((e = lm->st.alloc(st.size())) != hipSuccess || (e = lm->words.alloc(words.size())) != hipSuccess ||
      (e = lm->wt.alloc(wt.size())) != hipSuccess ||
      (e = hipMemcpy(lm->st.p, st.data(), st.size() * sizeof(int4), hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(lm->words.p, words.data(), words.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(lm->wt.p, wt.data(), wt.size() * sizeof(int2), hipMemcpyHostToDevice)) != hipSuccess ||
      (e = lm->hash.alloc(htab.size())) != hipSuccess ||
      (e = hipMemcpy(lm->hash.p, htab.data(), htab.size() * sizeof(int4), hipMemcpyHostToDevice)) != hipSuccess) {
    delete lm;
    return fail(WFST_E_DEVICE, std::string("LM upload: ") + hipGetErrorString(e));
  }
  *out = lm;
  return WFST_OK;
}

int wfst_lm_load(const char *path, float scale, int device, wfst_lm **out) {
  if (!path || !out) return fail(WFST_E_ARG, "NULL argument");
  *out = nullptr;
  FILE *fp = fopen(path, "rb");
  if (!fp) return fail(WFST_E_IO, std::string("cannot open LM file ") + path);
  fseek(fp, 0, SEEK_END);
  const long fsize = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  int32_t hdr[3], n_states = 0, n_arcs = 0;
  uint64_t orders = 0;
  std::vector<wfst_lm_state> st;
  std::vector<wfst_lm_arc> ar;
  int rc = WFST_OK;
  // sizes are checked against the file before anything is allocated
  if (fread(hdr, 4, 3, fp) != 3 || fread(&orders, 8, 1, fp) != 1 || orders > 64 ||
      fseek(fp, (long)(4 * orders), SEEK_CUR) != 0 || fread(&n_states, 4, 1, fp) != 1)
    rc = fail(WFST_E_IO, "truncated LM file (header)");
  else if (n_states <= 0 || (int64_t)n_states * 12 > fsize)
    rc = fail(WFST_E_FORMAT, "LM state count inconsistent with the file size");
  else {
    st.resize((size_t)n_states);
    if (fread(st.data(), sizeof(wfst_lm_state), st.size(), fp) != st.size() || fread(&n_arcs, 4, 1, fp) != 1)
      rc = fail(WFST_E_IO, "truncated LM file (states)");
    else if (n_arcs < 0 || (int64_t)n_arcs * 12 > fsize)
      rc = fail(WFST_E_FORMAT, "LM arc count inconsistent with the file size");
    else {
      ar.resize((size_t)n_arcs);
      if (fread(ar.data(), sizeof(wfst_lm_arc), ar.size(), fp) != ar.size()) rc = fail(WFST_E_IO, "truncated LM file (arcs)");
    }
  }
  fclose(fp);
  if (rc != WFST_OK) return rc;
  return wfst_lm_from_arrays(hdr[0], hdr[1], hdr[2], n_states, st.data(), n_arcs, ar.data(), scale, device, out);
}

int wfst_lm_info(const wfst_lm *lm, int32_t *bos, int32_t *eos, int32_t *n_states, int32_t *n_arcs, int32_t *n_words,
                 int64_t *device_bytes) {
  if (!lm) return fail(WFST_E_ARG, "NULL LM");
  if (bos) *bos = lm->bos;
  if (eos) *eos = lm->eos;
  if (n_states) *n_states = lm->n_states;
  if (n_arcs) *n_arcs = lm->n_arcs;
  if (n_words) *n_words = lm->start_arcs;
  if (device_bytes) *device_bytes = (int64_t)(lm->st.bytes() + lm->words.bytes() + lm->wt.bytes() + lm->hash.bytes());
  return WFST_OK;
}

void wfst_lm_free(wfst_lm *lm) {
  if (!lm) return;
  (void)hipSetDevice(lm->device);
  delete lm;
}

/* -------------------------------------------------------------- decoder */

// wfst_options.debug carries, besides the kernel phase timers (32 / 64 / 128) and the lattice decoders' comparison mode (0x1000),
// A/B switches of timing experiments (0x2000 no two-launch frames, 0x4000 no seed tiles, 0x10000 gathered log-likelihoods, 0x20000
// the insert launch looks for the best token, 0x40000 the back-pruning's raw frames on one workgroup per channel): honoured only by a library built with -DWFST_AB_SWITCHES (tools/ab_bench.sh),
// ignored by the product build.
#ifdef WFST_AB_SWITCHES
static constexpr bool kAbSwitches = true;
#else
static constexpr bool kAbSwitches = false;
#endif

static int check_config(const wfst_config *c) {  // LatticeFasterDecoderConfig::Check, conf.h:62-67
  if (!(c->beam > 0.0f && c->max_active > 1 && c->lattice_beam > 0.0f && c->prune_interval > 0 &&
        c->beam_delta > 0.0f && c->hash_ratio >= 1.0f && c->prune_scale > 0.0f && c->prune_scale < 1.0f))
    return fail(WFST_E_ARG, "invalid decoder config (LatticeFasterDecoderConfig::Check)");
  if (c->min_active < 0) return fail(WFST_E_ARG, "min_active < 0");
  return WFST_OK;
}

int wfst_decoder_create(const wfst_graph *g, const wfst_config *cfg, int32_t n_channels,
                        const wfst_limits *limits, void *hip_stream, wfst_decoder **out) {
  return wfst_decoder_create_ex(g, cfg, n_channels, limits, nullptr, hip_stream, out);
}

int wfst_decoder_create_ex(const wfst_graph *g, const wfst_config *cfg, int32_t n_channels,
                           const wfst_limits *limits, const wfst_options *options, void *hip_stream,
                           wfst_decoder **out) {
  return wfst_decoder_create_biglm(g, cfg, n_channels, limits, options, nullptr, nullptr, hip_stream, out);
}

int wfst_decoder_create_biglm(const wfst_graph *g, const wfst_config *cfg, int32_t n_channels,
                              const wfst_limits *limits, const wfst_options *options, const wfst_lm *old_lm,
                              const wfst_lm *new_lm, void *hip_stream, wfst_decoder **out) {
  if (!out) return fail(WFST_E_ARG, "out is NULL");
  *out = nullptr;
  if (!g || !cfg || n_channels <= 0) return fail(WFST_E_ARG, "NULL graph/config or n_channels <= 0");
  // insert work items carry the channel in 15 bits (wfst_kernels.hip plan_channel)
  if (n_channels > 32767) return fail(WFST_E_ARG, "at most 32767 channels per decoder");
  int rc = check_config(cfg);
  if (rc != WFST_OK) return rc;
  wfst_options O;
  wfst_options_default(&O);
  if (options) O = *options;
  if (O.channel_groups < 0 || O.channel_groups > 8 || O.log2_partitions < -1 || O.log2_partitions > 6 ||
      O.log2_lds_slots < 8 || O.log2_lds_slots > 13 || O.joint_max < 1 || O.expand_workgroups < 0 ||
      O.insert_workgroups < 1 || O.upload_slice_frames < 0 || O.tile_tokens < 64 || O.tile_tokens > 256)
    return fail(WFST_E_ARG, "wfst_options field out of range");
  HIP_TRY(hipSetDevice(g->device));
  wfst_limits L = {0, 0, 0, 0, 0, 0, 0, 0};
  if (limits) L = *limits;
  if (L.det_raw_states < 0 || L.det_raw_arcs < 0 || L.det_workspace_bytes < 0) return fail(WFST_E_ARG, "negative determinizer limit");
  const bool big = old_lm != nullptr || new_lm != nullptr;
  if (big) {
    if (!old_lm || !new_lm) return fail(WFST_E_ARG, "biglm needs both LMs");
    if (old_lm->device != g->device || new_lm->device != g->device) return fail(WFST_E_ARG, "the LMs must be on the graph's device");
    const int32_t nw = std::min(old_lm->start_arcs, new_lm->start_arcs);
    if (g->min_olabel < 0 || g->max_olabel >= nw)
      return fail(WFST_E_FORMAT, "the graph has output label " + std::to_string(g->max_olabel) + " but the LMs' empty-history state only has arcs for word ids below " + std::to_string(nw));
    if (L.lm_pairs <= 0) L.lm_pairs = 262144;
    if (L.lm_pairs > (1ll << 28)) return fail(WFST_E_ARG, "lm_pairs too large");
  }
  if (L.max_frames <= 0) L.max_frames = 4096;
  // (default: 65536 -- best-path decoders degrade at it, they do not fail: DecoderDev::soft_limit --, or four times a finite max_active -- that many tokens are expanded per frame, their arrivals are more -- up to 262144)
  if (L.max_tokens_per_frame <= 0)
    L.max_tokens_per_frame = cfg->max_active < (1 << 28) ? (int32_t)std::min<int64_t>(262144, std::max<int64_t>(32768, 4ll * cfg->max_active)) : 65536;
  if (L.arena_tokens <= 0)  // room for max_frames frames at 1/64 of the per-frame token limit (include/wfst_decoder.h)
    L.arena_tokens = std::min<int64_t>(0x7FFFFFF0ll, std::max<int64_t>(4194304, (int64_t)L.max_frames * std::max<int64_t>(256, L.max_tokens_per_frame / 64)));
  if (L.arena_tokens > 0x7FFFFFF0ll) return fail(WFST_E_ARG, "arena_tokens must fit int32");
  if (L.lattice_links < 0 || L.lattice_links > 0x7FFFFFF0ll) return fail(WFST_E_ARG, "lattice_links must fit int32");
  // launch shapes left to the library (wfst_options_default): a lattice decoder's wide beams fill a heavy channel's 32 buckets
  // beyond an insert workgroup's table (sub-passes over the bucket: the tail of the launch) -- 64 partitions and a larger expansion
  // grid measured 4 % faster on the beam-15 leg, the same at beam 13; best-path decoders keep 32 / 2048 (bucket atomics)
  if (O.log2_partitions < 0) O.log2_partitions = (L.lattice_links > 0 && !big) ? 6 : 5;
  if (O.expand_workgroups == 0) O.expand_workgroups = (L.lattice_links > 0 && !big) ? 3072 : 2048;
  // hash partitions per channel: each insert workgroup owns an LDS table of lds_slots entries;
  // a bucket too full for it is handled in sub-passes, so these are speed knobs, not limits
  const int64_t M = L.max_tokens_per_frame;
  int log2lds = O.log2_lds_slots, lds_slots = 1 << log2lds, log2part = O.log2_partitions;
  while (log2part > 0 && (int64_t)lds_slots << (log2part - 1) >= 4 * M) --log2part;  // tiny limits: fewer parts
  const int n_part = 1 << log2part;
  const int64_t bucket_cap = std::max<int64_t>(2048, 8 * M / n_part);

  wfst_decoder *d = new wfst_decoder();
  d->graph = g;
  d->device = g->device;
  d->cfg = *cfg;
  d->lim = L;
  d->n_channels = n_channels;
  if (hip_stream) {
    d->stream = (hipStream_t)hip_stream;
  } else {
    if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) {
      delete d;
      return fail(WFST_E_DEVICE, "hipStreamCreate failed");
    }
    d->own_stream = true;
  }
  size_t pair_cap = 0;
  const size_t B = (size_t)n_channels;
  size_t ecap = (size_t)std::max(1, g->n_eps_targets);
  if (big) {  // hashed epsilon table (tokens of one frame on epsilon-target states) and LM pair table: powers of two
    ecap = 1024;
    while (ecap < 4 * (size_t)L.max_tokens_per_frame) ecap <<= 1;
    pair_cap = 1024;
    while (pair_cap < (size_t)L.lm_pairs + (size_t)L.lm_pairs / 3 + 1) pair_cap <<= 1;  // stays below 3/4 full
  }
  int64_t lat_arc_cap = 0, lat_tok_cap = 0;
  const size_t fo = (size_t)L.max_frames + 2;
  hipError_t e = hipSuccess;
  auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
  A(d->ctl.alloc(B));
  A(d->tok.alloc(B * (size_t)L.arena_tokens));
  A(d->frame_off.alloc(B * fo));
  A(d->cutoff_hist.alloc(B * fo));
  A(d->bucket.alloc(B * (size_t)n_part * (size_t)bucket_cap));
  A(d->bucket_cnt.alloc(B * (size_t)n_part));
  A(d->emit_cnt.alloc(B * 32));
  A(d->prune_par.alloc(B * kPruneParInts));
  A(d->eps_vals.alloc(B * ecap));
  A(d->eps_toki.alloc(B * ecap));
  A(d->eps_occ_list.alloc(B * (size_t)L.max_tokens_per_frame));
  A(d->eps_won_list.alloc(B * (size_t)L.max_tokens_per_frame));
  A(d->worklist.alloc(B * 2 * (size_t)L.max_tokens_per_frame));
  if (big) {
    A(d->pair_keys.alloc(B * pair_cap));
    A(d->pair_list.alloc(B * pair_cap));
    A(d->eps_keys.alloc(B * ecap));
    A(d->tok_lm.alloc(B * (size_t)L.arena_tokens));
    A(d->bucket_lm.alloc(B * (size_t)n_part * (size_t)bucket_cap));
  }
  // new indices while the arena is compacted: the back-pruning of lattice mode, the token collection of best-path mode
  A(d->remap.alloc(B * (size_t)L.arena_tokens));
  if (L.lattice_links > 0) {
    A(d->links.alloc(B * (size_t)L.lattice_links));
    A(d->link_off.alloc(B * ((size_t)L.max_frames + 3)));
    A(d->link_mid.alloc(B * ((size_t)L.max_frames + 3)));
    A(d->extra.alloc(B * (size_t)L.arena_tokens));
    // GetRawLattice may be asked for at any time (base-inl.h:869-975): everything alive -- the pruned
    // history and the raw frames since the last PruneActiveTokens pass -- must fit the resolved lists
    lat_arc_cap = L.lattice_links;
    lat_tok_cap = L.arena_tokens;
    A(d->lat_arcs.alloc(B * (size_t)lat_arc_cap));
    A(d->lat_toks.alloc(B * (size_t)lat_tok_cap));
    A(d->lat_stats.alloc(B * 4));
  }
  // tiles of 128 tokens at least (prep_frame); a frame of a soft-limit decoder may hold several times the per-frame limit (every
  // candidate the buckets took can be a token: 8 x the limit): room for every channel at twice the limit and then some
  const size_t tile_cap = B * ((size_t)L.max_tokens_per_frame / 64 + 2) + (size_t)L.max_tokens_per_frame / 16;
  A(d->fctl.alloc(8));
  A(d->dbg_t.alloc(128));
  A(d->tiles.alloc(8 * tile_cap));
  const size_t item_cap = 2 * B * (size_t)n_part;  // two lists (heavy from the front, light from the back), each sized for the worst case
  A(d->items.alloc(8 * item_cap));
  A(d->item_pref.alloc(8 * item_cap * 64));   // the record prefix of every listed item, beside it (plan_channel)
  A(d->degraded.alloc(B));
  A(d->target.alloc(B));
  A(d->chan_list.alloc(B));
  A(d->ll_base.alloc(B));
  A(hipHostMalloc((void **)&d->p_target, B * 4));
  A(hipHostMalloc((void **)&d->p_chan, B * 4));
  A(hipHostMalloc((void **)&d->p_ll, B * sizeof(float *)));
  A(hipHostMalloc((void **)&d->p_ctl, B * sizeof(ChanCtl)));
  if (e == hipSuccess) A(hipMemsetAsync(d->ctl.p, 0, d->ctl.bytes(), d->stream));
  if (e == hipSuccess) A(hipMemsetAsync(d->bucket_cnt.p, 0, d->bucket_cnt.bytes(), d->stream));
  if (e == hipSuccess) A(hipMemsetAsync(d->emit_cnt.p, 0, d->emit_cnt.bytes(), d->stream));
  if (e == hipSuccess) A(hipMemsetAsync(d->prune_par.p, 0, d->prune_par.bytes(), d->stream));
  if (e == hipSuccess) A(hipMemsetAsync(d->fctl.p, 0, d->fctl.bytes(), d->stream));
  if (e == hipSuccess) A(hipMemsetAsync(d->dbg_t.p, 0, d->dbg_t.bytes(), d->stream));
  if (e == hipSuccess) A(hipMemsetAsync(d->eps_vals.p, 0xFF, d->eps_vals.bytes(), d->stream));
  if (e == hipSuccess && big) A(hipMemsetAsync(d->eps_keys.p, 0xFF, d->eps_keys.bytes(), d->stream));
  if (e == hipSuccess && big) A(hipMemsetAsync(d->pair_keys.p, 0xFF, d->pair_keys.bytes(), d->stream));
  {
    const int per_slot = (L.lattice_links > 0 && big) ? 20 : (L.lattice_links > 0 || big) ? 16 : 12;
    if (e == hipSuccess && lds_slots * per_slot > 65536) A((hipError_t)insert_kernel_set_lds(lds_slots * per_slot));
  }

  if (e == hipSuccess) A(hipMemsetAsync(d->degraded.p, 0, d->degraded.bytes(), d->stream));
  if (e == hipSuccess) A(hipMemsetAsync(d->target.p, 0, d->target.bytes(), d->stream));
  if (e == hipSuccess) A(hipMemsetAsync(d->ll_base.p, 0, d->ll_base.bytes(), d->stream));
  if (e == hipSuccess) A(hipStreamSynchronize(d->stream));
  if (e != hipSuccess) {
    delete d;
    return fail(WFST_E_DEVICE, std::string("decoder allocation failed: ") + hipGetErrorString(e));
  }
  DecoderDev &D = d->D;
  D.g = g->view();
  D.ctl = d->ctl.p;
  D.tok = d->tok.p;
  D.frame_off = d->frame_off.p;
  D.cutoff_hist = d->cutoff_hist.p;
  D.bucket = d->bucket.p;
  D.bucket_cnt = d->bucket_cnt.p;
  D.emit_cnt = d->emit_cnt.p;
  D.prune_par = d->prune_par.p;
  D.eps_vals = d->eps_vals.p;
  D.eps_toki = d->eps_toki.p;
  D.eps_occ_list = d->eps_occ_list.p;
  D.eps_won_list = d->eps_won_list.p;
  D.worklist = d->worklist.p;
  D.links = d->links.p;
  D.link_off = d->link_off.p;
  D.link_mid = d->link_mid.p;
  D.extra = d->extra.p;
  D.remap = d->remap.p;
  D.lat_arcs = d->lat_arcs.p;
  D.lat_toks = d->lat_toks.p;
  D.lat_stats = d->lat_stats.p;
  D.lat_arc_cap = (int32_t)lat_arc_cap;
  D.lat_tok_cap = (int32_t)lat_tok_cap;
  D.link_cap = L.lattice_links;
  D.lattice = L.lattice_links > 0 ? 1 : 0;
  D.big = big ? 1 : 0;
  // (lattice decoders use the fused rows too: the epsilon arrivals come through the insert launch and the closure kernel
  // only lists the epsilon links; debug 0x1000 keeps the iterated closure pass, for comparison)
  D.fused = (g->fused && !big && (L.lattice_links == 0 || !(O.debug & 0x1000))) ? 1 : 0;
  D.link_delta = (D.fused && L.lattice_links > 0) ? 1 : 0;
  // degree codes in the tokens (wfst_device.h): fused rows, a packed graph, and an arena whose indices leave 9 bits of a
  // backpointer free (up to 2^22 tokens: the default 4194304)
  D.tok_idx_bits = 31;
  D.degcode = 0;
  if (D.fused && g->packed && L.lattice_links == 0) {   // (lattice mode keeps the closure flags in those record bits)
    int bits = 1;
    while ((1ll << bits) < L.arena_tokens) ++bits;
    if (31 - bits >= kCodeRestBits) { D.tok_idx_bits = bits; D.degcode = 1; }
  }
  if (big) {
    D.lm_old = old_lm->view();
    D.lm_new = new_lm->view();
  } else {
    memset(&D.lm_old, 0, sizeof(D.lm_old));
    memset(&D.lm_new, 0, sizeof(D.lm_new));
  }
  // two launches per frame (wfst_device.h): fused best-path decoders whose max_active can never bind (it is at least the
  // per-frame token limit) and whose min_active is 0; every gc_stride-th frame keeps the closure launch (token collection
  // check), gc_stride + 1 frames at the per-frame limit fitting the arena's collection reserve (gc_base_mark)
  D.best_row = (D.fused && !D.lattice && !big) ? 1 : 0;
  D.two_launch = 0;
  D.gc_stride = 1;
  // expand_kernel_staged (the tile's arcs staged in LDS by gather DMA): every decoder on the fused rows.  Where max_active (or the
  // per-frame limit) binds, most of a frame's tokens lie above the cutoff: the frame boundary then cuts COMPACTING tiles
  // (wfst_kernels.hip kStSuper), whose dead tokens do not cost a tile's round trips -- round 3 kept the round-2 expansion
  // (two candidates per thread in registers, rounds of 512) for those decoders; it is gone.  (Timing-experiment bits of
  // wfst_options.debug are honoured by WFST_AB_SWITCHES builds only.)
  const int ab_bits = kAbSwitches ? O.debug : 0;
  d->stagger_us = (ab_bits & 0x80000) ? ((ab_bits >> 20) & 0xFF) * 100 : 0;   // (0x80000 + a count of 100 us in bits 20..27, A/B)
  D.prune_raw_min = (O.debug & 0x800) ? 0 : 800000;
  if (kAbSwitches) { if (const char *e = getenv("WFST_PRUNE_RAW_MIN")) D.prune_raw_min = atoi(e); }   // (timing experiments)
   // (below: the one-workgroup walk in LDS is done sooner -- beam 13 of the bench; 0x800: the tests' switch)
  // closure launches of a lattice decoder on the fused rows: workgroups per channel (they share the frame's epsilon links; a heavy
  // channel's frame is a dozen sweeps for one workgroup).  wfst_options.debug 0x100 / 0x200 / 0x300: 1 / 2 / 8 of them, for the tests
  { static const int kSlabs[4] = {4, 1, 2, 8}; D.closure_slabs = kSlabs[(O.debug >> 8) & 3]; }
  D.prune_raw = (ab_bits & 0x40000) ? 0 : 1;   // (0x40000, A/B: the raw frames of a back-pruning pass on one workgroup per channel, as until round 4)
  D.staged = (D.fused && !big) ? 1 : 0;
  D.st_tile_tokens = O.tile_tokens & ~7;
  // the expansion finds the frame's best token itself (its cheapest candidate): the staged kernel of best_row decoders (0x20000: A/B)
  D.best_exp = (D.staged && D.best_row && !(ab_bits & 0x20000)) ? 1 : 0;
  D.seed_tiles = (D.best_row && !(ab_bits & 0x4000)) ? 1 : 0;
  // fused best-path decoders: max_tokens_per_frame is a max_active, not a capacity (the frame keeps every token the arena takes and
  // GetCutoff tightens to the limit-th cheapest; wfst_decoder_get_degraded_frames counts the frames on which it did)
  D.soft_limit = D.best_row;
  // the token collection's reserve (wfst_kernels.hip gc_base_mark): an eighth of the arena, two frames at the per-frame limit at least;
  // a two-launch decoder checks the mark every gc_stride-th frame only (a third launch on that frame), so it takes up to a quarter
  // of the arena where that buys a longer stride (17 frames at the limit cover the longest, 16); never more than half
  {
    const int64_t M = std::max<int64_t>(1, L.max_tokens_per_frame), A = L.arena_tokens;
    int64_t reserve = std::max<int64_t>(2 * M, A / 8);
    const bool two = D.best_row && !(ab_bits & 0x2000);
    if (two) reserve = std::max<int64_t>(reserve, std::min<int64_t>(17 * M, A / 4));
    if (reserve >= A / 2) reserve = A / 2;
    D.gc_reserve = reserve;
    if (two) {
      const int64_t stride = reserve / M - 1;
      if (stride >= 1) { D.two_launch = 1; D.gc_stride = (int32_t)std::min<int64_t>(stride, 16); }
    }
  }
  D.pair_keys = d->pair_keys.p;
  D.pair_list = d->pair_list.p;
  D.pair_cap = (int32_t)pair_cap;
  D.tok_lm = d->tok_lm.p;
  D.bucket_lm = d->bucket_lm.p;
  D.eps_keys = d->eps_keys.p;
  D.fctl = d->fctl.p;
  D.dbg_t = d->dbg_t.p;
  D.tiles = d->tiles.p;
  D.tile_cap = (int32_t)tile_cap;
  D.items = d->items.p;
  D.item_pref = d->item_pref.p;
  D.degraded = d->degraded.p;
  D.item_cap = (int32_t)item_cap;
  D.ll_base = d->ll_base.p;
  D.n_channels = n_channels;
  D.stride = 0;
  D.n_part = n_part;
  D.log2part = log2part;
  D.lds_slots = lds_slots;
  D.log2lds = log2lds;
  D.bucket_cap = (int32_t)bucket_cap;
  D.joint_max = O.joint_max;
  D.ecap = (int32_t)ecap;
  D.max_tok = L.max_tokens_per_frame;
  D.wl_cap = L.max_tokens_per_frame;
  D.max_frames = L.max_frames;
  D.arena_cap = L.arena_tokens;
  D.beam = cfg->beam;
  D.lattice_beam = cfg->lattice_beam;
  D.beam_delta = cfg->beam_delta;
  D.prune_scale = cfg->prune_scale;
  D.max_active = cfg->max_active;
  D.min_active = cfg->min_active;
  D.prune_interval = cfg->prune_interval;
  D.dbg = O.debug;
  d->h_decoded.assign(B, 0);
  d->h_target.assign(B, 0);
  d->h_state.assign(B, 0);
  d->h_ll_base.assign(B, nullptr);
  d->hist_dev.assign(B, nullptr);
  d->hist_rows_cap.assign(B, 0);
  d->hist_rows.assign(B, 0);
  d->use_graph = O.use_hip_graph != 0;
  d->upload_slice = O.upload_slice_frames;
  // automatic: two groups from 64 channels up -- one group's expansion overlaps the other's insert / closure step
  // (measured on the bench workload: equal at 64 channels, +5 % at 128, +7 % at 256; three or more streams share
  // hardware queues and lose)
  // (biglm decoders: three -- their launches are chains of LM look-ups on small frontiers, a third stream still finds idle
  // CUs: 36.5 vs 40.7 ms per step at 128 channels; a fourth shares a hardware queue and loses; lattice decoders: three as well --
  // the back-pruning steps and the per-frame link work of one group leave room beside two others: 54.2 -> 52.3 ms per step at
  // beam 13, 183.6 -> 170.4 at beam 15)
  // (round 4: best-path decoders too -- with the shorter launches of this round a third group's frame chain finds room beside two
  // others: 18.65-19.1 -> 18.33 ms per step at 128 channels; four groups: 27.8, a fifth stream shares a hardware queue)
  const bool three = n_channels >= 96;
  // (round 4: FOUR groups pay for biglm and lattice decoders of 128 channels -- their frames are chains of small launches -- where the
  // process runs with eight hardware queues (GPU_MAX_HW_QUEUES=8 in its environment before the HIP runtime starts): the runtime's
  // default of four makes the fifth stream of a process share a queue, and two streams on one queue take turns (biglm 63 ms per
  // step); with eight: biglm 30.2 -> 27.7 ms, lattice mode 50.6 -> 49.6 / 225 -> 223; five groups: 60 ms -- more than four queues
  // busy at once is the wall; best-path decoders: 18.7 against 18.4 with three.  The library reads no environment: a host that sets
  // the runtime up that way asks for wfst_options.channel_groups = 4 -- bench.py does)
  d->n_groups = std::min(O.channel_groups > 0 ? O.channel_groups : three ? 3 : (n_channels >= 64 ? 2 : 1), n_channels);
  if (d->n_groups > 1) {
    d->gstreams.resize(d->n_groups);
    d->gevents.resize(d->n_groups + 1);
    hipError_t ge = hipSuccess;
    d->gstreams[0] = nullptr;  // group 0 runs on the decoder's own stream
    for (size_t g = 1; g < d->gstreams.size(); ++g) if (ge == hipSuccess) ge = hipStreamCreateWithFlags(&d->gstreams[g], hipStreamNonBlocking);
    for (auto &ev : d->gevents) if (ge == hipSuccess) ge = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (ge != hipSuccess) { delete d; return fail(WFST_E_DEVICE, "stream/event creation failed"); }
  }
  d->expand_wgs = O.expand_workgroups;
  d->insert_wgs = O.insert_workgroups;
  d->gpar.assign(8, 0);
  *out = d;
  return WFST_OK;
}

void wfst_decoder_free(wfst_decoder *d) {
  if (d && (d->D.dbg & 224)) {  // debug phase timers (100 MHz ticks)
    unsigned long long t[128];
    (void)hipSetDevice(d->device);
    (void)hipDeviceSynchronize();
    if (hipMemcpy(t, d->dbg_t.p, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess) {
      const char *names[] = {"closure:setup", "closure:rounds", "closure:commit", "closure:clear", "closure:finalize", "closure:prep",
                             "insert:init", "insert:pass1", "insert:alloc", "insert:pass2", "insert:tail",
                             "expand:tile-load+scan", "expand:candidates (staged: arcs landed)", "expand:bound+count (staged: loglikes landed)", "expand:bucket-atomics (staged: priced)",
                             "expand:write (staged: ranked + bucket atomics)", "expand:stats+ticket (staged: written)"};
      for (int k = 0; k < 17; ++k)
        if (t[3 * k + 2])
          fprintf(stderr, "[wfst dbg] %-18s n=%llu mean=%.2f us max=%.2f us\n", names[k], t[3 * k + 2],
                  0.01 * t[3 * k] / t[3 * k + 2], 0.01 * t[3 * k + 1]);
      {
        const char *more[] = {"insert:head (launch/ticket -> item known)", "insert:countdown+ticket", "insert:frame boundary"};
        for (int k = 22; k < 25; ++k)
          if (t[3 * k + 2]) fprintf(stderr, "[wfst dbg] %-18s n=%llu mean=%.2f us max=%.2f us\n", more[k - 22], t[3 * k + 2], 0.01 * t[3 * k] / t[3 * k + 2], 0.01 * t[3 * k + 1]);
      }
      if (t[62] && (d->D.dbg & 128))
        fprintf(stderr, "[wfst dbg] %-18s n=%llu mean=%.2f us max=%.2f us\n", "expand:descriptor+tokens landed", t[62], 0.01 * t[60] / t[62], 0.01 * t[61]);
      if (t[52])
        fprintf(stderr, "[wfst dbg] prune passes=%llu walk mean=%.1f us compaction mean=%.1f us frames walked mean=%.1f\n", t[52],
                0.01 * t[51] / t[52], 0.01 * t[54] / t[52], (double)t[53] / t[52]);
      if (t[52])
        fprintf(stderr, "[wfst dbg] per pass: walk raw frames %.1f us, old frames %.1f us; frames in LDS %.1f, in HBM %.1f; compaction: token flags %.1f us, token move %.1f us, link flags %.1f us, link move %.1f us; raw tokens %.0f, raw links %.0f, survivors %.0f\n",
                0.01 * t[40] / t[52], 0.01 * t[41] / t[52], (double)t[42] / t[52], (double)t[43] / t[52], 0.01 * t[44] / t[52], 0.01 * t[45] / t[52], 0.01 * t[46] / t[52], 0.01 * t[47] / t[52],
                (double)t[48] / t[52], (double)t[49] / t[52], (double)t[50] / t[52]);
      if (t[52]) {
        fprintf(stderr, "[wfst dbg] walk length per channel and pass, 0.4 ms bins: raw frames priced in the walk");
        for (int b = 0; b < 11; ++b) fprintf(stderr, " %llu", t[80 + b]);
        fprintf(stderr, " | behind the raw launch");
        for (int b = 0; b < 11; ++b) fprintf(stderr, " %llu", t[117 + b]);
        fprintf(stderr, "\n");
      }
      if (t[116])
        fprintf(stderr, "[wfst dbg] best path: per channel frontier scan %.1f us, backpointer walk %.1f us (%.1f hops), %.2f epsilon-won tokens resolved by a frame scan %.1f us, hop pass %.1f us\n",
                0.01 * t[110] / t[116], 0.01 * t[111] / t[116], (double)t[114] / t[116], (double)t[115] / t[116], 0.01 * t[112] / t[116], 0.01 * t[113] / t[116]);
      fprintf(stderr, "[wfst dbg] closure rounds total=%llu launches*chan=%llu max_seeds=%llu max_rounds=%llu first round mean=%.2f us max=%.2f us\n",
              t[56], t[57], t[58], t[59], t[57] ? 0.01 * t[60] / t[57] : 0.0, 0.01 * t[61]);
    }
  }
  delete d;
}

// Resolve a channel list: returns the device pointer to use (nullptr = all channels) and count.
static int stage_channels(wfst_decoder *d, const int32_t *channels, int32_t n, const int32_t **dev, int32_t *cnt) {
  if (!channels) {
    *dev = nullptr;
    *cnt = d->n_channels;
    return WFST_OK;
  }
  if (n <= 0 || n > d->n_channels) return fail(WFST_E_ARG, "bad channel count");
  std::vector<char> seen((size_t)d->n_channels, 0);
  for (int i = 0; i < n; ++i) {
    if (channels[i] < 0 || channels[i] >= d->n_channels) return fail(WFST_E_ARG, "channel index out of range");
    if (seen[channels[i]]) return fail(WFST_E_ARG, "duplicate channel in list");
    seen[channels[i]] = 1;
  }
  HIP_TRY(hipStreamSynchronize(d->stream));  // staging buffer reuse
  memcpy(d->p_chan, channels, (size_t)n * 4);
  HIP_TRY(hipMemcpyAsync(d->chan_list.p, d->p_chan, (size_t)n * 4, hipMemcpyHostToDevice, d->stream));
  *dev = d->chan_list.p;
  *cnt = n;
  return WFST_OK;
}

static int finish_prefetch(wfst_decoder *d);

int wfst_decoder_init(wfst_decoder *d, const int32_t *channels, int32_t n) {
  if (!d) return fail(WFST_E_ARG, "NULL decoder");
  HIP_TRY(hipSetDevice(d->device));
  if (!d->pf_detached) { const int rcp = finish_prefetch(d); if (rcp != WFST_OK) return rcp; }   // (the determinizer may be reading these channels' lattices; a detached one has read them)
  const int32_t *dev;
  int32_t cnt;
  int rc = stage_channels(d, channels, n, &dev, &cnt);
  if (rc != WFST_OK) return rc;
  launch_init(d->D, dev, cnt, d->stream);
  HIP_TRY(hipGetLastError());
  for (int i = 0; i < cnt; ++i) {
    const int c = channels ? channels[i] : i;
    d->h_decoded[c] = 0;
    d->h_target[c] = 0;
    d->h_state[c] = 1;
    d->h_ll_base[c] = nullptr;
    d->hist_rows[c] = 0;
    if (!d->lat_cached.empty()) d->lat_cached[c] = 0;
    if (!d->det_cached.empty()) { d->det_cached[c] = 0; d->det_live_nd[c] = -1; }
    if (!d->resc_cache.empty()) { d->resc_cache[(size_t)c].key.valid = false; d->nbp_cache[(size_t)c].key.valid = false; }
    d->post_dev_list.clear();
  }
  return WFST_OK;
}

static int advance_device(wfst_decoder *d, const int32_t *channels, int32_t n, const float *const *loglikes,
                          const int32_t *n_frames_ready, int32_t stride, int32_t max_num_frames) {
  if (!loglikes || !n_frames_ready) return fail(WFST_E_ARG, "NULL loglikes / n_frames_ready");
  const int32_t cnt = channels ? n : d->n_channels;
  if (cnt <= 0 || cnt > d->n_channels) return fail(WFST_E_ARG, "bad channel count");
  if (!d->pf_detached) { const int rcp = finish_prefetch(d); if (rcp != WFST_OK) return rcp; }   // (a prefetching determinizer borrows the second group's stream; a detached one has its own)
  if (stride <= d->graph->max_col)
    return fail(WFST_E_ARG, "stride too small: the graph reads log-likelihood column " + std::to_string(d->graph->max_col));
  if (d->D.stride != 0 && d->D.stride != stride) {
    // a different row stride only matters for channels still holding frames; keep it simple
    for (int c = 0; c < d->n_channels; ++c)
      if (d->h_state[c] != 0 && d->h_decoded[c] > 0 && d->h_ll_base[c] != nullptr) {
        bool listed = false;
        for (int i = 0; i < cnt; ++i) listed |= ((channels ? channels[i] : i) == c);
        if (!listed) return fail(WFST_E_ARG, "all live channels of a decoder must use one stride");
      }
  }
  int steps = 0;
  for (int i = 0; i < cnt; ++i) {
    const int c = channels ? channels[i] : i;
    if (c < 0 || c >= d->n_channels) return fail(WFST_E_ARG, "channel index out of range");
    if (d->h_state[c] == 0) return fail(WFST_E_STATE, "AdvanceDecoding before InitDecoding");
    if (d->h_state[c] == 2) return fail(WFST_E_STATE, "AdvanceDecoding after FinalizeDecoding");
    if (n_frames_ready[i] < d->h_decoded[c]) return fail(WFST_E_ARG, "NumFramesReady decreased");  // base-inl.h:641
    if (!loglikes[i] && n_frames_ready[i] > 0) return fail(WFST_E_ARG, "NULL log-likelihood matrix");
    int target = n_frames_ready[i];
    if (max_num_frames >= 0) target = std::min(target, d->h_decoded[c] + max_num_frames);  // base-inl.h:643-648
    if (target > d->D.max_frames) return fail(WFST_E_CAPACITY, "utterance longer than wfst_limits.max_frames");
    steps = std::max(steps, target - d->h_decoded[c]);
  }
  // The device's row pointers follow the caller's even when no frame is decoded by this call: the
  // matrix may have moved (advance_host regrows its history buffer), and GetBestPath / the lattice
  // pruning read acoustic costs of PAST frames through ll_base[c].
  bool moved = false;
  for (int i = 0; i < cnt; ++i) {
    const int c = channels ? channels[i] : i;
    if (loglikes[i] && loglikes[i] != d->h_ll_base[c]) moved = true;
  }
  if (steps == 0 && !moved) return WFST_OK;
  HIP_TRY(hipStreamSynchronize(d->stream));  // pinned staging reuse
  for (int i = 0; i < cnt; ++i) {
    const int c = channels ? channels[i] : i;
    int target = n_frames_ready[i];
    if (max_num_frames >= 0) target = std::min(target, d->h_decoded[c] + max_num_frames);
    d->h_target[c] = target;
    if (loglikes[i]) d->h_ll_base[c] = loglikes[i];
  }
  for (int c = 0; c < d->n_channels; ++c) {
    d->p_target[c] = d->h_target[c];
    d->p_ll[c] = d->h_ll_base[c];
  }
  HIP_TRY(hipMemcpyAsync(d->target.p, d->p_target, (size_t)d->n_channels * 4, hipMemcpyHostToDevice, d->stream));
  HIP_TRY(hipMemcpyAsync((void *)d->ll_base.p, (const void *)d->p_ll, (size_t)d->n_channels * sizeof(float *),
                         hipMemcpyHostToDevice, d->stream));
  d->D.stride = stride;
  {
    // expand_kernel_staged_row (wfst_kernels.hip): the frame's log-likelihood row of a tile's channel staged in LDS by 16-byte DMAs
    bool ok = d->D.staged && !(kAbSwitches && (d->D.dbg & 0x10000)) && (stride & 3) == 0 && stride <= 3072;
    for (int c = 0; ok && c < d->n_channels; ++c)
      if (d->h_ll_base[c] && ((uintptr_t)d->h_ll_base[c] & 15u)) ok = false;
#ifdef WFST_FORCE_GATHER   // (A/B builds: the gather form of the staged expansion whatever the stride)
    ok = false;
#endif
    d->D.ll_row = ok ? 1 : 0;
  }
  if (steps == 0) return WFST_OK;
  // frames to decode per channel group
  const int G = d->n_groups, per = (d->n_channels + G - 1) / G;
  std::vector<int> gsteps(G, 0);
  for (int c = 0; c < d->n_channels; ++c)
    gsteps[c / per] = std::max(gsteps[c / per], d->h_target[c] - d->h_decoded[c]);
  int timed_group = 0, timed_kind = -1;   // (of the launch being enqueued: for the timeline dump)
  auto timed = [&](int cls, hipStream_t st, auto &&launch) {
    if (!d->profiling) { launch(); return; }
    const int a = d->ev_get(), b = d->ev_get();
    if (a >= 0 && b >= 0) (void)hipEventRecord(d->ev_pool[a], st);
    launch();
    if (a >= 0 && b >= 0) {
      (void)hipEventRecord(d->ev_pool[b], st);
      d->ev_pairs[cls].push_back({a, b});
      if (kAbSwitches) d->ev_log.push_back({timed_kind >= 0 ? timed_kind : cls, timed_group, a, b});
    }
  };
  // a back-pruning step: four launches (wfst_kernels.hip launch_lattice_prune_step); timed one by one for the timeline dump
  auto prune_step_launches = [&](int off, int cnt, int g, int par, hipStream_t st) {
    if (!(kAbSwitches && d->profiling)) { timed(2, st, [&] { launch_lattice_prune_step(d->D, off, cnt, d->target.p, g, par, st, -1); }); return; }
    for (int stage = 0; stage < 4; ++stage) {
      timed_kind = 10 + stage;
      timed(2, st, [&] { launch_lattice_prune_step(d->D, off, cnt, d->target.p, g, par, st, stage); });
    }
    timed_kind = -1;
  };
  const std::vector<int> gpar0(d->gpar);  // parity each group starts this call with
  // lattice mode: does step s of group g bring a channel to a multiple of prune_interval, with frames left to decode?
  auto prune_step = [&](int g, int s) -> bool {
    if (!d->D.lattice) return false;
    const int off = g * per, hi = std::min(d->n_channels, off + per);
    for (int c = off; c < hi; ++c) {
      const int nd = d->h_decoded[c] + s + 1;
      if (d->h_state[c] == 1 && nd > 0 && nd <= d->h_target[c] - 1 && nd % d->cfg.prune_interval == 0) return true;   // (s = -1: where the channel stands)
    }
    return false;
  };
  // the frame loop of one channel group on stream st: steps [s_lo, s_hi) of the call's gsteps[g]
  auto enqueue_group = [&](int g, hipStream_t st, int s_lo, int s_hi) {
    const int off = g * per, cnt = std::min(per, d->n_channels - off);
    if (cnt <= 0 || gsteps[g] == 0) return;
    timed_group = g;
    int par = gpar0[g] ^ (s_lo & 1);
    if (s_lo == 0) {
      // GetCutoff + tile list only -- behind PruneActiveTokens where the call before this one stopped at a multiple of prune_interval
      if (prune_step(g, -1)) prune_step_launches(off, cnt, g, par, st);
      else timed(2, st, [&] { launch_closure(d->D, off, cnt, d->target.p, 1, g, par, st, 0); });
    }
    for (int s = s_lo; s < s_hi; ++s) {
      // two launches per frame where the decoder allows (wfst_device.h two_launch): the insert launch closes the frame and
      // prepares the next; every gc_stride-th frame is a classic one (its closure launch checks the token arena)
      const bool classic = !d->D.two_launch || (s % d->D.gc_stride) == d->D.gc_stride - 1;
      const bool more = s + 1 < gsteps[g];
      timed(0, st, [&] { launch_expand(d->D, g, par, d->expand_wgs, st); });
      timed(1, st, [&] { launch_insert(d->D, off, cnt, d->target.p, classic ? 0 : more ? 1 : 2, g, par, d->insert_wgs, st); });
      // lattice mode: PruneActiveTokens (base-inl.h:660-661) on the steps at which a channel of the group reaches a multiple of
      // prune_interval and goes on decoding -- a launch of its own, which also prepares the next frame
      const bool prune = more && prune_step(g, s);
      if (classic) timed(2, st, [&] { launch_closure(d->D, off, cnt, d->target.p, more && !prune, g, par ^ 1, st, 1); });
      if (prune) prune_step_launches(off, cnt, g, par ^ 1, st);
      par ^= 1;
    }
  };
  // launch-bound inner loop -> hipGraph: per-frame state lives in ChanCtl on the device, so the
  // captured kernels and their arguments are identical for every call with the same frame counts.
  // One graph PER CHANNEL GROUP, each launched on its own stream: independent graph launches overlap
  // on the GPU (parallel branches inside one graph were measured to run serially).
  // A graph launch costs the calling thread ~0.5 us per node BEFORE the device sees the first of them (a 300-frame group: 0.35 ms;
  // three groups one after the other: the last one's frame chain -- the step's critical path -- started a millisecond late).  A long
  // call is therefore launched as a short HEAD graph (kHeadSteps frames) and the REST: every group's head is on the device
  // within microseconds, the rests are submitted while the heads run.  part 0: the whole call as one graph (short calls),
  // 1: the head, 2: the rest.
  constexpr int kHeadSteps = 32;
  auto graphed = [&](int g) { return d->use_graph && !d->profiling && gsteps[g] >= 4; };
  auto split = [&](int g) { return graphed(g) && gsteps[g] >= 4 * kHeadSteps; };
  auto run_group = [&](int g, hipStream_t st, int part) -> int {
    if (gsteps[g] == 0) return WFST_OK;
    if (!graphed(g)) { if (part != 2) enqueue_group(g, st, 0, gsteps[g]); return WFST_OK; }
    if (!split(g) && part == 2) return WFST_OK;
    const int s_lo = (split(g) && part == 2) ? kHeadSteps : 0, s_hi = (split(g) && part != 2) ? kHeadSteps : gsteps[g];
    std::vector<int> key = {g, gsteps[g], (int)stride, gpar0[g], (int)d->D.ll_row, s_lo, s_hi};
    for (int s = -1; s < gsteps[g]; ++s)
      if (prune_step(g, s)) key.push_back(s);   // (the launch sequence differs with the steps that prune)
    auto it = d->graphs.find(key);
    if (it == d->graphs.end()) {
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      enqueue_group(g, st, s_lo, s_hi);
      HIP_TRY(hipStreamEndCapture(st, &graph));
      HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      HIP_TRY(hipGraphDestroy(graph));
      if (d->graphs.size() >= 64) {  // bounded cache
        HIP_TRY(hipDeviceSynchronize());
        for (auto &kv : d->graphs) (void)hipGraphExecDestroy(kv.second);
        d->graphs.clear();
      }
      it = d->graphs.emplace(key, exec).first;
    }
    HIP_TRY(hipGraphLaunch(it->second, st));
    return WFST_OK;
  };
  if (G == 1) {
    for (int part = 1; part <= 2; ++part) {
      const int rc = run_group(0, d->stream, part);
      if (rc != WFST_OK) return rc;
    }
  } else {
    // group 0 runs on the decoder's own stream, the others fork from it and join it again (every stream in use
    // takes one of the few hardware queues; streams beyond those share a queue and serialise)
    HIP_TRY(hipEventRecord(d->gevents[0], d->stream));  // targets / row pointers are uploaded
    for (int part = 1; part <= 2; ++part) {   // every group's head first, then the rests
      for (int g = 1; g < G; ++g) {
        if (gsteps[g] == 0) continue;
        if (part == 1) HIP_TRY(hipStreamWaitEvent(d->gstreams[g], d->gevents[0], 0));
        if (part == 1 && d->stagger_us > 0) launch_delay(g * d->stagger_us, d->gstreams[g]);
        const int rc = run_group(g, d->gstreams[g], part);
        if (rc != WFST_OK) return rc;
        if (part == 2) HIP_TRY(hipEventRecord(d->gevents[1 + g], d->gstreams[g]));
      }
      const int rc = run_group(0, d->stream, part);
      if (rc != WFST_OK) return rc;
    }
    for (int g = 1; g < G; ++g)
      if (gsteps[g] != 0) HIP_TRY(hipStreamWaitEvent(d->stream, d->gevents[1 + g], 0));
  }
  HIP_TRY(hipGetLastError());
  for (int g = 0; g < G; ++g) d->gpar[g] = gpar0[g] ^ (gsteps[g] & 1);
  for (int c = 0; c < d->n_channels; ++c) d->h_decoded[c] = std::max(d->h_decoded[c], d->h_target[c]);
  return WFST_OK;
}

int wfst_decoder_advance(wfst_decoder *d, const int32_t *channels, int32_t n, const float *const *loglikes,
                         const int32_t *n_frames_ready, int32_t stride, int32_t max_num_frames) {
  if (!d) return fail(WFST_E_ARG, "NULL decoder");
  HIP_TRY(hipSetDevice(d->device));
  return advance_device(d, channels, n, loglikes, n_frames_ready, stride, max_num_frames);
}

int wfst_decoder_advance_host(wfst_decoder *d, const int32_t *channels, int32_t n,
                              const float *const *loglikes_host, const int32_t *n_frames_ready,
                              int32_t stride, int32_t max_num_frames) {
  if (!d) return fail(WFST_E_ARG, "NULL decoder");
  if (!loglikes_host || !n_frames_ready) return fail(WFST_E_ARG, "NULL loglikes / n_frames_ready");
  HIP_TRY(hipSetDevice(d->device));
  const int32_t cnt = channels ? n : d->n_channels;
  if (cnt <= 0 || cnt > d->n_channels) return fail(WFST_E_ARG, "bad channel count");
  if (d->hist_stride != 0 && d->hist_stride != stride) {
    for (int c = 0; c < d->n_channels; ++c)
      if (d->hist_rows[c] > 0) return fail(WFST_E_ARG, "stride changed while channels hold frames");
  }
  d->hist_stride = stride;
  std::vector<const float *> dev_ptrs((size_t)cnt);
  for (int i = 0; i < cnt; ++i) {
    const int c = channels ? channels[i] : i;
    if (c < 0 || c >= d->n_channels) return fail(WFST_E_ARG, "channel index out of range");
    const int32_t have = d->hist_rows[c], want = n_frames_ready[i];
    if (want < have) return fail(WFST_E_ARG, "NumFramesReady decreased");
    if ((size_t)want > d->hist_rows_cap[c]) {
      size_t ncap = std::max<size_t>((size_t)want, std::max<size_t>(d->hist_rows_cap[c] * 2, 256));
      float *np = nullptr;
      HIP_TRY(hipMalloc((void **)&np, ncap * (size_t)stride * 4));
      if (have > 0) {
        HIP_TRY(hipStreamSynchronize(d->stream));
        HIP_TRY(hipMemcpy(np, d->hist_dev[c], (size_t)have * stride * 4, hipMemcpyDeviceToDevice));
      }
      if (d->hist_dev[c]) {
        HIP_TRY(hipStreamSynchronize(d->stream));
        HIP_TRY(hipFree(d->hist_dev[c]));
      }
      d->hist_dev[c] = np;
      d->hist_rows_cap[c] = ncap;
    }
    if (want > have && !loglikes_host[i]) return fail(WFST_E_ARG, "NULL log-likelihood matrix");
    dev_ptrs[i] = d->hist_dev[c];
  }
  // Upload and decode in slices of kSlice frames: the host copies slice k+1 (pageable memory: the
  // copy call returns when the caller's buffer is consumed) while the GPU decodes slice k, so the
  // PCIe time of a long hand-over hides behind the search instead of preceding it.
  if (!d->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
  const int kSlice = std::max(1, d->upload_slice);
  int longest = 0;
  for (int i = 0; i < cnt; ++i) longest = std::max(longest, n_frames_ready[i] - d->hist_rows[channels ? channels[i] : i]);
  const bool sliced = max_num_frames < 0 && d->upload_slice > 0 && longest > 2 * kSlice;
  std::vector<int32_t> ready((size_t)cnt);
  for (int done = 0; done < std::max(longest, 1); done += sliced ? kSlice : std::max(longest, 1)) {
    const int upto = sliced ? done + kSlice : longest;
    for (int i = 0; i < cnt; ++i) {
      const int c = channels ? channels[i] : i;
      const int32_t have = d->hist_rows[c];
      const int32_t want = std::min<int32_t>(n_frames_ready[i], have + std::max(0, upto - done));
      if (want > have) {
        HIP_TRY(hipMemcpyAsync(d->hist_dev[c] + (size_t)have * stride, loglikes_host[i] + (size_t)have * stride,
                               (size_t)(want - have) * stride * 4, hipMemcpyHostToDevice, d->copy_stream));
        d->hist_rows[c] = want;
      }
      ready[i] = d->hist_rows[c];
    }
    HIP_TRY(hipStreamSynchronize(d->copy_stream));  // rows are in HBM (and the caller's buffers consumed)
    const int rc = advance_device(d, channels, n, dev_ptrs.data(), ready.data(), stride, max_num_frames);
    if (rc != WFST_OK) return rc;
  }
  return WFST_OK;
}

int wfst_decoder_finalize(wfst_decoder *d, const int32_t *channels, int32_t n) {
  if (!d) return fail(WFST_E_ARG, "NULL decoder");
  HIP_TRY(hipSetDevice(d->device));
  if (!d->pf_detached) { const int rcp = finish_prefetch(d); if (rcp != WFST_OK) return rcp; }
  const int32_t *dev;
  int32_t cnt;
  int rc = stage_channels(d, channels, n, &dev, &cnt);
  if (rc != WFST_OK) return rc;
  for (int i = 0; i < cnt; ++i) {
    const int c = channels ? channels[i] : i;
    if (d->h_state[c] == 0) return fail(WFST_E_STATE, "FinalizeDecoding before InitDecoding");
  }
  launch_set_finalized(d->D, dev, cnt, d->stream);
  if (d->D.lattice) launch_lattice_prune(d->D, dev, cnt, d->stream);  // PruneForwardLinksFinal + backward pruning
  HIP_TRY(hipGetLastError());
  for (int i = 0; i < cnt; ++i) {
    const int c = channels ? channels[i] : i;
    d->h_state[c] = 2;
    if (d->fin_epoch.empty()) d->fin_epoch.assign((size_t)d->n_channels, 0);
    ++d->fin_epoch[(size_t)c];
    if (!d->lat_cached.empty()) d->lat_cached[c] = 0;
    if (!d->det_cached.empty()) { d->det_cached[c] = 0; d->det_live_nd[c] = -1; }
    if (!d->resc_cache.empty()) { d->resc_cache[(size_t)c].key.valid = false; d->nbp_cache[(size_t)c].key.valid = false; }
    d->post_dev_list.clear();
  }
  return WFST_OK;
}

static int read_ctl(wfst_decoder *d) {
  HIP_TRY(hipMemcpyAsync(d->p_ctl, d->ctl.p, d->ctl.bytes(), hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  return WFST_OK;
}

// what a channel's error word says (ChanCtl::error), as the call that finds it reports it
static constexpr int32_t kDetErrCtl = 0x10000;   // DetLattice::err: the channel's utterance ended in a device error (its error word in the low bits)
static int fail_ctl_error(int c, int e) {
  std::string m = "channel " + std::to_string(c) + " exceeded a device capacity:";
  if (e & kErrTableFull) m += " hash table (max_tokens_per_frame)";
  if (e & kErrArenaFull) m += " token arena (arena_tokens)";
  if (e & kErrFrontierFull) m += " frontier (max_tokens_per_frame)";
  if (e & kErrWorklistFull) m += " epsilon worklist (max_tokens_per_frame)";
  if (e & kErrFramesFull) m += " frames (max_frames)";
  if (e & kErrBucketFull) m += " candidate bucket (max_tokens_per_frame)";
  if (e & kErrLinksFull) m += " forward links (lattice_links)";
  if (e & kErrPairsFull) m += " LM pair states (lm_pairs)";
  if (e & kErrInternal) return fail(WFST_E_DEVICE, "channel " + std::to_string(c) + ": internal invariant violated on the device (a forward link without its token, or a backpointer without its predecessor)");
  return fail(WFST_E_CAPACITY, m);
}

static int check_ctl_errors(wfst_decoder *d) {
  for (int c = 0; c < d->n_channels; ++c)
    if (d->p_ctl[c].error) return fail_ctl_error(c, d->p_ctl[c].error);
  return WFST_OK;
}

int wfst_decoder_sync(wfst_decoder *d) {
  if (!d) return fail(WFST_E_ARG, "NULL decoder");
  HIP_TRY(hipSetDevice(d->device));
  int rc = read_ctl(d);
  if (rc != WFST_OK) return rc;
  return check_ctl_errors(d);
}

int wfst_decoder_num_frames_decoded(wfst_decoder *d, int32_t channel) {
  if (!d || channel < 0 || channel >= d->n_channels) return fail(WFST_E_ARG, "bad decoder/channel");
  return d->h_decoded[channel];
}

int wfst_decoder_get_best_path(wfst_decoder *d, const int32_t *channels, int32_t n, int32_t use_final_probs,
                               int32_t cap, int32_t *ilabel, int32_t *olabel, float *graph_cost,
                               float *acoustic_cost, int32_t *n_hops) {
  if (!d || !ilabel || !olabel || !graph_cost || !acoustic_cost || !n_hops || cap <= 0)
    return fail(WFST_E_ARG, "NULL output or cap <= 0");
  HIP_TRY(hipSetDevice(d->device));
  const int32_t *dev;
  int32_t cnt;
  int rc = stage_channels(d, channels, n, &dev, &cnt);
  if (rc != WFST_OK) return rc;
  for (int i = 0; i < cnt; ++i) {
    const int c = channels ? channels[i] : i;
    if (d->h_state[c] == 0) return fail(WFST_E_STATE, "GetBestPath before InitDecoding");
    if (d->h_state[c] == 2 && !use_final_probs)  // base-inl.h:1100-1102 (LOG_ERR)
      return fail(WFST_E_STATE, "You cannot call FinalizeDecoding() and then GetBestPath with use_final_probs == false");
  }
  const size_t need = (size_t)cnt * (size_t)cap;
  // one device block {n_hops[cnt] (padded to 4 words) | ilabel | olabel | graph | acoustic} -> one copy into pinned
  // host memory -> the caller's arrays (five copies into pageable memory cost five staging round trips)
  const size_t head = ((size_t)cnt + 3) & ~(size_t)3, words = head + 4 * need;
  if (d->bp_all.n < words || d->bp_chain.n < need) {
    HIP_TRY(hipStreamSynchronize(d->stream));
    HIP_TRY(d->bp_all.alloc(words));
    HIP_TRY(d->bp_chain.alloc(need));
  }
  if (d->bp_pin_bytes < words * 4) {
    if (d->bp_pin) (void)hipHostFree(d->bp_pin);
    d->bp_pin = nullptr;
    d->bp_pin_bytes = 0;
    HIP_TRY(hipHostMalloc((void **)&d->bp_pin, words * 4, hipHostMallocDefault));
    d->bp_pin_bytes = words * 4;
  }
  int32_t *dn = d->bp_all.p, *dil = dn + head, *dol = dil + need;
  float *dg = reinterpret_cast<float *>(dol + need), *dac = dg + need;
  launch_best_path(d->D, dev, cnt, use_final_probs ? 1 : 0, cap, dil, dol, dg, dac, dn, d->bp_chain.p, d->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(d->bp_pin, d->bp_all.p, words * 4, hipMemcpyDeviceToHost, d->stream));
  rc = read_ctl(d);   // synchronises the stream
  if (rc != WFST_OK) return rc;
  {
    const int32_t *hp = reinterpret_cast<const int32_t *>(d->bp_pin);
    memcpy(n_hops, hp, (size_t)cnt * 4);
    memcpy(ilabel, hp + head, need * 4);
    memcpy(olabel, hp + head + need, need * 4);
    memcpy(graph_cost, hp + head + 2 * need, need * 4);
    memcpy(acoustic_cost, hp + head + 3 * need, need * 4);
  }
  rc = check_ctl_errors(d);
  if (rc != WFST_OK) return rc;
  for (int i = 0; i < cnt; ++i)
    if (n_hops[i] > cap) return fail(WFST_E_CAPACITY, "best path longer than cap hops; n_hops holds the needed size");
  return WFST_OK;
}

int wfst_decoder_get_nbest(wfst_decoder *d, const int32_t *channels, int32_t n_channels, int32_t n, int32_t max_words,
                           int32_t *n_paths, int32_t *n_words, int32_t *words, float *tot_score, float *lm_score) {
  if (!d || !n_paths || !n_words || !words || !tot_score || !lm_score) return fail(WFST_E_ARG, "NULL argument");
  if (n <= 0 || n > 16 || max_words <= 0) return fail(WFST_E_ARG, "n must be 1..16 and max_words > 0");
  if (!d->D.lattice) return fail(WFST_E_STATE, "GetNbest needs a decoder created with wfst_limits.lattice_links > 0");
  HIP_TRY(hipSetDevice(d->device));
  const int32_t *dev;
  int32_t cnt;
  int rc = stage_channels(d, channels, n_channels, &dev, &cnt);
  if (rc != WFST_OK) return rc;
  bool any_live = false;
  for (int i = 0; i < cnt; ++i) {
    const int st = d->h_state[channels ? channels[i] : i];
    if (st == 0) return fail(WFST_E_STATE, "GetNbest before InitDecoding");
    any_live |= st == 1;
  }
  // mid-utterance (the service's partial n-best, v2-asr/v2-asr-task.h:319): resolve what is alive now -- for the LIVE
  // channels of the list only (the finalized ones were resolved by FinalizeDecoding)
  if (any_live) {
    std::vector<int32_t> live_list;
    for (int i = 0; i < cnt; ++i) {
      const int c = channels ? channels[i] : i;
      if (d->h_state[c] == 1) live_list.push_back(c);
    }
    if ((int32_t)live_list.size() == cnt) {
      launch_lattice_emit(d->D, dev, cnt, 1, d->stream);
    } else {
      const int32_t *ldev;
      int32_t lcnt;
      rc = stage_channels(d, live_list.data(), (int32_t)live_list.size(), &ldev, &lcnt);
      if (rc != WFST_OK) return rc;
      launch_lattice_emit(d->D, ldev, lcnt, 1, d->stream);
      rc = stage_channels(d, channels, n_channels, &dev, &cnt);   // (waits for the emit launch before the list is reused)
      if (rc != WFST_OK) return rc;
    }
  }
  NbestDev &N = d->nb;
  if (!d->nb_list.p) {  // first use: per-channel k-best lists and index scratch
    // k-best lists: 16 entries x 24 bytes per lattice state; up to 262144 states per lattice, less
    // when that would take more than ~4 GB over all channels (never below 32768)
    const int64_t budget = (int64_t)(4ll << 30) / ((int64_t)d->n_channels * 16 * (int64_t)sizeof(NbEntry));
    N.tok_cap = (int32_t)std::min<int64_t>(d->D.lat_tok_cap, std::min<int64_t>(262144, std::max<int64_t>(32768, budget)));
    N.arc_cap = (int32_t)std::min<int64_t>(d->D.lat_arc_cap, 4ll * N.tok_cap);
    N.scratch_ints = (3ll * N.tok_cap + 1 + 3ll * (d->D.max_frames + 2) + 4ll * N.arc_cap + 3) & ~3ll;   // (nbest_kernel: in-arc records of 16 bytes first, a multiple of 16 bytes per channel)
    HIP_TRY(hipStreamSynchronize(d->stream));
    HIP_TRY(d->nb_list.alloc((size_t)d->n_channels * (size_t)N.tok_cap * 16));
    HIP_TRY(d->nb_scratch.alloc((size_t)d->n_channels * (size_t)N.scratch_ints));
    N.list = d->nb_list.p;
    N.scratch = d->nb_scratch.p;
  }
  N.K = 16;
  N.n = n;
  N.max_words = max_words;
  const size_t per_i = 1 + (size_t)n + (size_t)n * max_words, per_f = 2 * (size_t)n;
  if (d->nb_out_i.n < (size_t)cnt * per_i || d->nb_out_f.n < (size_t)cnt * per_f) {
    HIP_TRY(hipStreamSynchronize(d->stream));
    HIP_TRY(d->nb_out_i.alloc((size_t)d->n_channels * per_i));
    HIP_TRY(d->nb_out_f.alloc((size_t)d->n_channels * per_f));
  }
  N.out_n = d->nb_out_i.p;
  N.out_nwords = N.out_n + cnt;
  N.out_words = N.out_nwords + (size_t)cnt * n;
  N.out_tot = d->nb_out_f.p;
  N.out_lm = N.out_tot + (size_t)cnt * n;
  launch_nbest(d->D, N, dev, cnt, d->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(n_paths, N.out_n, (size_t)cnt * 4, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipMemcpyAsync(n_words, N.out_nwords, (size_t)cnt * n * 4, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipMemcpyAsync(words, N.out_words, (size_t)cnt * n * max_words * 4, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipMemcpyAsync(tot_score, N.out_tot, (size_t)cnt * n * 4, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipMemcpyAsync(lm_score, N.out_lm, (size_t)cnt * n * 4, hipMemcpyDeviceToHost, d->stream));
  rc = read_ctl(d);  // synchronises the stream
  if (rc != WFST_OK) return rc;
  rc = check_ctl_errors(d);
  if (rc != WFST_OK) return rc;
  for (int i = 0; i < cnt; ++i)
    if (n_paths[i] < 0) return fail(WFST_E_CAPACITY, ("lattice too large for the n-best search (more than " + std::to_string(N.tok_cap) + " states or " + std::to_string(N.arc_cap) + " arcs)"));
  return WFST_OK;
}

int wfst_lattice_to_vector(const int32_t *ilabel, const int32_t *olabel, const float *graph_cost,
                           const float *acoustic_cost, int32_t n_hops, int32_t *words, int32_t max_words,
                           int32_t *n_words, int32_t *tids, int32_t max_tids, int32_t *n_tids,
                           float *tot_score, float *lm_score) {
  // newfst/lattice-functions.cc:179-217; n_hops == 0 is its `Start() == kNoStateId` -> false
  if (n_hops < 0 || !n_words || !n_tids || !tot_score || !lm_score) return fail(WFST_E_ARG, "bad argument");
  float tot = 0, lm = 0;
  int nw = 0, nt = 0;
  for (int k = 0; k < n_hops; ++k) {
    if (ilabel[k] != 0) { if (tids && nt < max_tids) tids[nt] = ilabel[k]; ++nt; }
    if (olabel[k] != 0) { if (words && nw < max_words) words[nw] = olabel[k]; ++nw; }
    lm += graph_cost[k];
    tot += graph_cost[k] + acoustic_cost[k];
  }
  *n_words = nw;
  *n_tids = nt;
  *tot_score = tot;
  *lm_score = lm;
  return WFST_OK;
}

int wfst_lattice_to_vector_batch(const int32_t *ilabel, const int32_t *olabel, const float *graph_cost,
                                 const float *acoustic_cost, const int32_t *n_hops, int32_t n_paths, int32_t cap,
                                 float *tot_score, float *lm_score, int32_t *n_words, int32_t *n_tids) {
  if (!ilabel || !olabel || !graph_cost || !acoustic_cost || !n_hops || !tot_score || !lm_score || n_paths < 0 || cap <= 0)
    return fail(WFST_E_ARG, "bad argument");
  for (int32_t p = 0; p < n_paths; ++p) {
    const size_t o = (size_t)p * (size_t)cap;
    const int32_t n = std::min(std::max(n_hops[p], 0), cap);
    float tot = 0, lm = 0;   // newfst/lattice-functions.cc:179-217: sequential float sums in forward order
    int32_t nw = 0, nt = 0;
    for (int32_t k = 0; k < n; ++k) {
      nt += ilabel[o + k] != 0;
      nw += olabel[o + k] != 0;
      lm += graph_cost[o + k];
      tot += graph_cost[o + k] + acoustic_cost[o + k];
    }
    tot_score[p] = tot;
    lm_score[p] = lm;
    if (n_words) n_words[p] = nw;
    if (n_tids) n_tids[p] = nt;
  }
  return WFST_OK;
}

int wfst_lattice_labels_batch(const int32_t *ilabel, const int32_t *olabel, const int32_t *n_hops, int32_t n_paths, int32_t cap,
                              int32_t *words, int32_t *word_off, int32_t *tids, int32_t *tid_off) {
  if (!ilabel || !olabel || !n_hops || !words || !word_off || !tids || !tid_off || n_paths < 0 || cap <= 0)
    return fail(WFST_E_ARG, "bad argument");
  int32_t nw = 0, nt = 0;
  for (int32_t p = 0; p < n_paths; ++p) {
    const size_t o = (size_t)p * (size_t)cap;
    const int32_t n = std::min(std::max(n_hops[p], 0), cap);
    word_off[p] = nw;
    tid_off[p] = nt;
    for (int32_t k = 0; k < n; ++k) {   // newfst/lattice-functions.cc:195-206: the nonzero labels in hop order
      if (ilabel[o + k] != 0) tids[nt++] = ilabel[o + k];
      if (olabel[o + k] != 0) words[nw++] = olabel[o + k];
    }
  }
  word_off[n_paths] = nw;
  tid_off[n_paths] = nt;
  return WFST_OK;
}

int wfst_decoder_set_profiling(wfst_decoder *d, int32_t enable) {
  if (!d) return fail(WFST_E_ARG, "NULL decoder");
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipStreamSynchronize(d->stream));
  d->profiling = enable != 0;
  d->ev_used = 0;
  for (auto &v : d->ev_pairs) v.clear();
  d->ev_log.clear();
  return WFST_OK;
}

int wfst_decoder_get_profile(wfst_decoder *d, double ms[3], int64_t launches[3]) {
  if (!d || !ms || !launches) return fail(WFST_E_ARG, "bad argument");
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipStreamSynchronize(d->stream));
  for (int k = 0; k < 3; ++k) {
    double tot = 0;
    for (auto &pr : d->ev_pairs[k]) {
      float t = 0;
      HIP_TRY(hipEventElapsedTime(&t, d->ev_pool[pr.first], d->ev_pool[pr.second]));
      tot += t;
    }
    ms[k] = tot;
    launches[k] = (int64_t)d->ev_pairs[k].size();
  }
  return WFST_OK;
}

int wfst_decoder_channel_groups(wfst_decoder *d) { return d ? d->n_groups : fail(WFST_E_ARG, "NULL decoder"); }

int wfst_decoder_get_path_flags(wfst_decoder *d, int32_t flags[8]) {
  if (!d || !flags) return fail(WFST_E_ARG, "bad argument");
  const DecoderDev &D = d->D;
  flags[0] = D.staged; flags[1] = D.two_launch; flags[2] = D.gc_stride; flags[3] = D.degcode;
  flags[4] = D.ll_row; flags[5] = D.best_exp; flags[6] = D.soft_limit; flags[7] = d->n_groups;
  return WFST_OK;
}

int wfst_decoder_get_profile_busy(wfst_decoder *d, double busy_ms[3]) {
  if (!d || !busy_ms) return fail(WFST_E_ARG, "bad argument");
  HIP_TRY(hipSetDevice(d->device));
  for (hipStream_t st : d->gstreams) if (st) HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipStreamSynchronize(d->stream));
  // union of the launches' [start, stop] intervals of each kernel class: with several channel groups the launches
  // of different groups run concurrently, so the sum of their durations counts shared time twice
  // time base: the EARLIEST recorded start event (with several channel groups the first pair listed is not it)
  int base = -1;
  for (int k = 0; k < 3; ++k)
    for (auto &pr : d->ev_pairs[k]) {
      if (base < 0) { base = pr.first; continue; }
      float dt = 0;
      HIP_TRY(hipEventElapsedTime(&dt, d->ev_pool[base], d->ev_pool[pr.first]));
      if (dt < 0) base = pr.first;
    }
  for (int k = 0; k < 3; ++k) {
    std::vector<std::pair<float, float>> iv;
    iv.reserve(d->ev_pairs[k].size());
    for (auto &pr : d->ev_pairs[k]) {
      float a = 0, b = 0;
      HIP_TRY(hipEventElapsedTime(&a, d->ev_pool[base], d->ev_pool[pr.first]));
      HIP_TRY(hipEventElapsedTime(&b, d->ev_pool[base], d->ev_pool[pr.second]));
      iv.push_back({a, b});
    }
    std::sort(iv.begin(), iv.end());
    double tot = 0;
    float lo = 0, hi = 0;
    bool open = false;
    for (auto &x : iv) {
      if (open && x.first <= hi) { hi = std::max(hi, x.second); continue; }
      if (open) tot += hi - lo;
      lo = x.first; hi = x.second; open = true;
    }
    if (open) tot += hi - lo;
    busy_ms[k] = tot;
  }
  if (kAbSwitches) {   // timing experiments: the profiled step's launches as a timeline (kind, channel group, start ms, stop ms)
    if (const char *path = getenv("WFST_PROFILE_DUMP")) {
      if (FILE *f = fopen(path, "w")) {
        for (auto &e : d->ev_log) {
          float a = 0, b = 0;
          (void)hipEventElapsedTime(&a, d->ev_pool[base], d->ev_pool[e[2]]);
          (void)hipEventElapsedTime(&b, d->ev_pool[base], d->ev_pool[e[3]]);
          fprintf(f, "%d,%d,%.4f,%.4f\n", e[0], e[1], a, b);
        }
        fclose(f);
      }
    }
  }
  return WFST_OK;
}

int wfst_decoder_get_raw_lattice(wfst_decoder *d, int32_t channel, int32_t use_final_probs, int32_t cap_states,
                                 int32_t cap_arcs, int32_t *n_states, int32_t *n_arcs, int32_t *st_final,
                                 int32_t *st_frame, int32_t *st_state, float *st_cost, int32_t *a_src,
                                 int32_t *a_dst, int32_t *a_ilabel, int32_t *a_olabel, float *a_graph,
                                 float *a_acoustic) {
  if (!d || channel < 0 || channel >= d->n_channels || !n_states || !n_arcs) return fail(WFST_E_ARG, "bad argument");
  if (!d->D.lattice) return fail(WFST_E_STATE, "GetRawLattice needs a decoder created with wfst_limits.lattice_links > 0");
  if (d->h_state[channel] == 0) return fail(WFST_E_STATE, "GetRawLattice before InitDecoding");
  const bool live = d->h_state[channel] == 1;  // mid-utterance: everything alive now (base-inl.h:869-975)
  HIP_TRY(hipSetDevice(d->device));
  *n_states = 0;
  *n_arcs = 0;
  if (!live && !use_final_probs) return WFST_OK;  // base-inl.h:879-884: finalized && !use_final_probs -> false
  if (d->lat_cached.empty()) {
    d->lat_cached.assign((size_t)d->n_channels, 0);
    d->lat_cache_nd.assign((size_t)d->n_channels, 0);
    d->lat_cache_tok.resize((size_t)d->n_channels);
    d->lat_cache_arc.resize((size_t)d->n_channels);
  }
  if (live) {
    // resolve what is alive right now (the pruned history + the raw frames since the last PruneActiveTokens
    // pass) and fetch this channel's two lists; never cached: the next frame changes them
    const int32_t *dev;
    int32_t cnt;
    int rc = stage_channels(d, &channel, 1, &dev, &cnt);
    if (rc != WFST_OK) return rc;
    launch_lattice_emit(d->D, dev, 1, use_final_probs ? 1 : 0, d->stream);
    HIP_TRY(hipGetLastError());
    rc = read_ctl(d);
    if (rc != WFST_OK) return rc;
    rc = check_ctl_errors(d);
    if (rc != WFST_OK) return rc;
    const ChanCtl &cc = d->p_ctl[channel];
    d->lat_cache_nd[channel] = cc.n_decoded;
    d->lat_cache_tok[channel].resize((size_t)cc.lat_toks);
    d->lat_cache_arc[channel].resize((size_t)cc.lat_arcs);
    if (cc.lat_toks)
      HIP_TRY(hipMemcpyAsync(d->lat_cache_tok[channel].data(), d->lat_toks.p + (size_t)channel * (size_t)d->D.lat_tok_cap,
                             (size_t)cc.lat_toks * sizeof(int4), hipMemcpyDeviceToHost, d->stream));
    if (cc.lat_arcs)
      HIP_TRY(hipMemcpyAsync(d->lat_cache_arc[channel].data(), d->lat_arcs.p + (size_t)channel * (size_t)d->D.lat_arc_cap,
                             (size_t)cc.lat_arcs * sizeof(LatArc), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    d->lat_cached[channel] = 0;
  } else if (!d->lat_cached[channel]) {
    // The pruned lattices were left resolved by FinalizeDecoding.  One control-block read, then
    // the two small lists of EVERY finalized channel in one sweep of copies and one synchronisation:
    // a caller that walks all channels of a batch pays the device round trips once.
    int rc = read_ctl(d);
    if (rc != WFST_OK) return rc;
    rc = check_ctl_errors(d);
    if (rc != WFST_OK) return rc;
    // through one pinned staging buffer: the copies are enqueued back to back and waited for once
    size_t need = 0;
    for (int c = 0; c < d->n_channels; ++c) {
      if (d->h_state[c] != 2 || d->lat_cached[c]) continue;
      need += (size_t)d->p_ctl[c].lat_toks * sizeof(int4) + (size_t)d->p_ctl[c].lat_arcs * sizeof(LatArc);
    }
    if (need > d->lat_pin_bytes) {
      if (d->lat_pin) (void)hipHostFree(d->lat_pin);
      d->lat_pin = nullptr;
      d->lat_pin_bytes = 0;
      HIP_TRY(hipHostMalloc((void **)&d->lat_pin, need + need / 4, hipHostMallocDefault));
      d->lat_pin_bytes = need + need / 4;
    }
    size_t off = 0;
    for (int c = 0; c < d->n_channels; ++c) {
      if (d->h_state[c] != 2 || d->lat_cached[c]) continue;
      const ChanCtl &cc = d->p_ctl[c];
      const size_t tb = (size_t)cc.lat_toks * sizeof(int4), ab = (size_t)cc.lat_arcs * sizeof(LatArc);
      if (tb) HIP_TRY(hipMemcpyAsync(d->lat_pin + off, d->lat_toks.p + (size_t)c * (size_t)d->D.lat_tok_cap, tb, hipMemcpyDeviceToHost, d->stream));
      if (ab) HIP_TRY(hipMemcpyAsync(d->lat_pin + off + tb, d->lat_arcs.p + (size_t)c * (size_t)d->D.lat_arc_cap, ab, hipMemcpyDeviceToHost, d->stream));
      off += tb + ab;
    }
    HIP_TRY(hipStreamSynchronize(d->stream));
    off = 0;
    for (int c = 0; c < d->n_channels; ++c) {
      if (d->h_state[c] != 2 || d->lat_cached[c]) continue;
      const ChanCtl &cc = d->p_ctl[c];
      const size_t tb = (size_t)cc.lat_toks * sizeof(int4), ab = (size_t)cc.lat_arcs * sizeof(LatArc);
      d->lat_cache_nd[c] = cc.n_decoded;
      d->lat_cache_tok[c].resize((size_t)cc.lat_toks);
      d->lat_cache_arc[c].resize((size_t)cc.lat_arcs);
      if (tb) memcpy(d->lat_cache_tok[c].data(), d->lat_pin + off, tb);
      if (ab) memcpy(d->lat_cache_arc[c].data(), d->lat_pin + off + tb, ab);
      off += tb + ab;
      d->lat_cached[c] = 1;
    }
  }
  const int nd = d->lat_cache_nd[channel];
  if (nd <= 0) return WFST_OK;
  std::vector<int4> tk = d->lat_cache_tok[channel];   // sorted below: work on a copy
  const std::vector<LatArc> &ar = d->lat_cache_arc[channel];
  const int n_tok = (int)tk.size(), n_arc = (int)ar.size();
  // tokens sorted by arena index = by frame, creation order inside a frame
  std::sort(tk.begin(), tk.end(), [](const int4 &a, const int4 &b) { return a.x < b.x; });
  auto find_tok = [&](int32_t arena_idx) -> int {
    auto it = std::lower_bound(tk.begin(), tk.end(), arena_idx, [](const int4 &a, int32_t v) { return a.x < v; });
    return (it != tk.end() && it->x == arena_idx) ? (int)(it - tk.begin()) : -1;
  };
  // the reference returns false when a frame has no token left (base-inl.h:906-911)
  {
    std::vector<char> seen((size_t)nd + 1, 0);
    for (const int4 &t : tk) seen[t.w & 0x3FFFFFFF] = 1;
    for (int f = 0; f <= nd; ++f)
      if (!seen[f]) return WFST_OK;
  }
  *n_states = n_tok;
  *n_arcs = n_arc;
  if (n_tok > cap_states || n_arc > cap_arcs) return fail(WFST_E_CAPACITY, "lattice larger than the given capacities");  // size probe
  std::vector<int32_t> src((size_t)n_arc), dst((size_t)n_arc);
  for (int i = 0; i < n_arc; ++i) {
    src[i] = find_tok(ar[i].src_tok);
    dst[i] = find_tok(ar[i].dst_tok);
    if (src[i] < 0 || dst[i] < 0) return fail(WFST_E_DEVICE, "internal: lattice arc without its tokens");
  }
  // topological numbering inside each frame: depth along the surviving epsilon links
  std::vector<int32_t> depth((size_t)n_tok, 0);
  for (int round = 0; round < 1 << 20; ++round) {
    bool changed = false;
    for (int i = 0; i < n_arc; ++i)
      if (ar[i].is_eps && depth[dst[i]] < depth[src[i]] + 1) { depth[dst[i]] = depth[src[i]] + 1; changed = true; }
    if (!changed) break;
  }
  std::vector<int32_t> order((size_t)n_tok), new_id((size_t)n_tok);
  for (int i = 0; i < n_tok; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    const int fa = tk[a].w & 0x3FFFFFFF, fb = tk[b].w & 0x3FFFFFFF;
    return fa != fb ? fa < fb : depth[a] < depth[b];
  });
  for (int i = 0; i < n_tok; ++i) new_id[order[i]] = i;
  for (int i = 0; i < n_tok; ++i) {
    const int s = new_id[i];
    if (st_final) st_final[s] = (tk[i].w >> 30) & 1;
    if (st_frame) st_frame[s] = tk[i].w & 0x3FFFFFFF;
    if (st_state) st_state[s] = tk[i].y;
    if (st_cost) memcpy(&st_cost[s], &tk[i].z, 4);
  }
  std::vector<int32_t> idx((size_t)n_arc);
  for (int i = 0; i < n_arc; ++i) idx[i] = i;
  // arcs sorted by source state; destination / labels / cost keep the order deterministic (the
  // device appends them in whatever order the waves retire)
  std::sort(idx.begin(), idx.end(), [&](int a, int b) {
    const int sa = new_id[src[a]], sb = new_id[src[b]];
    if (sa != sb) return sa < sb;
    const int da = new_id[dst[a]], db = new_id[dst[b]];
    if (da != db) return da < db;
    if (ar[a].ilabel != ar[b].ilabel) return ar[a].ilabel < ar[b].ilabel;
    if (ar[a].olabel != ar[b].olabel) return ar[a].olabel < ar[b].olabel;
    return ar[a].graph < ar[b].graph;
  });
  for (int k = 0; k < n_arc; ++k) {
    const int i = idx[k];
    if (a_src) a_src[k] = new_id[src[i]];
    if (a_dst) a_dst[k] = new_id[dst[i]];
    if (a_ilabel) a_ilabel[k] = ar[i].ilabel;
    if (a_olabel) a_olabel[k] = ar[i].olabel;
    if (a_graph) a_graph[k] = ar[i].graph;
    if (a_acoustic) a_acoustic[k] = ar[i].acoustic;
  }
  return WFST_OK;
}

// first use of the determinizer: its workspace (wfst_limits.det_raw_states / det_raw_arcs / det_workspace_bytes), for det_slots
// lattices at a time -- every channel of the decoder where the budget allows
static int ensure_det_workspace(wfst_decoder *d) {
  DetDev &X = d->det;
  if (d->det_ws.p) return WFST_OK;
    // first use: the determinizer's workspace (wfst_limits.det_raw_states / det_raw_arcs / det_workspace_bytes), for
  // det_slots lattices at a time -- every channel of the decoder where the budget allows
  X.raw_states_cap = (int32_t)std::min<int64_t>(d->D.lat_tok_cap, d->lim.det_raw_states > 0 ? d->lim.det_raw_states : 65536);
  X.raw_arcs_cap = (int32_t)std::min<int64_t>(d->D.lat_arc_cap, d->lim.det_raw_arcs > 0 ? d->lim.det_raw_arcs : 2ll * X.raw_states_cap);
  const int32_t base = std::max(4096, X.raw_states_cap);
  X.caps.trie = 8 * base; X.caps.pool = 16 * base; X.caps.states = 2 * base; X.caps.initials = 2 * base;
  X.caps.arcs = 4 * base; X.caps.tmp = std::max(8192, 2 * (X.raw_arcs_cap + X.raw_states_cap));
  X.out_cap = X.caps.arcs;
  X.words_per_channel = 3 * (int64_t)X.raw_states_cap + 1 + 5 * (int64_t)X.raw_arcs_cap + det_words(X.caps, X.raw_states_cap) + 16;
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  const int64_t per = X.words_per_channel * 4 + (int64_t)X.out_cap * (int64_t)(sizeof(int4) + sizeof(float2));
  const int64_t budget = d->lim.det_workspace_bytes > 0 ? d->lim.det_workspace_bytes : (int64_t)(total_b / 8);
  d->det_slots = (int32_t)std::max<int64_t>(1, std::min<int64_t>(d->n_channels, budget / per));
  HIP_TRY(hipStreamSynchronize(d->stream));
  HIP_TRY(d->det_ws.alloc((size_t)d->det_slots * (size_t)X.words_per_channel));
  HIP_TRY(d->det_result.alloc((size_t)d->det_slots * 4));
  HIP_TRY(d->det_out_a.alloc((size_t)d->det_slots * (size_t)X.out_cap));
  HIP_TRY(d->det_out_w.alloc((size_t)d->det_slots * (size_t)X.out_cap));
  X.ws = d->det_ws.p;
  X.result = d->det_result.p;
  X.out_a = d->det_out_a.p;
  X.out_w = d->det_out_w.p;
  d->det_cache.resize((size_t)d->n_channels);
  d->det_cached.assign((size_t)d->n_channels, 0);
  d->det_live_nd.assign((size_t)d->n_channels, -1);
  d->det_live_final.assign((size_t)d->n_channels, 0);
  return WFST_OK;
}

// The results of a determinize launch over `list` (workspace slot i = list[i]; res = the launch's result words, on their way or
// here already) into the host cache: waits for the decoder's stream, checks the channels' error words, fetches the arcs.
static int harvest_determinized(wfst_decoder *d, const std::vector<int32_t> &list, const std::vector<int32_t> &res, bool live, int32_t use_final_probs,
                                bool detached = false) {
  DetDev &X = d->det;
  if (!detached) {
    // A device error of one of THESE channels' utterances is kept with that channel's lattice (DetLattice::err, kDetErrCtl | the error
    // word) and reported when that lattice is asked for; a harvest never fails for it -- it also runs in front of InitDecoding /
    // AdvanceDecoding / FinalizeDecoding of other channels (finish_prefetch), which have nothing to do with it -- and the other
    // channels' lattices of the same launch are kept.
    const int rc = read_ctl(d);  // synchronises the stream
    if (rc != WFST_OK) return rc;
  } else {
    // (a detached prefetch: the channels may be in the middle of their next utterances -- the errors of the utterances these
    // lattices belong to were reported when their best paths were fetched)
    if (d->pf_cache.empty()) d->pf_cache.resize((size_t)d->n_channels);
    d->pf_have.assign((size_t)d->n_channels, 0);   // what an EARLIER prefetch left belongs to utterances two steps back: gone, not served as this one's
  }
  // a batch: the lattices packed back to back on the device, two copies for all of them (two per lattice were 256 copy calls for
  // 128 utterances); a single lattice, or a batch beyond the packing buffers: straight from its slot
  size_t total = 0;
  for (int i = 0; i < (int)list.size(); ++i)
    if (!res[4 * i + 2]) total += (size_t)std::min(res[4 * i + 1], X.out_cap);
  constexpr size_t kPackCap = (size_t)1 << 20;
  const bool packed = list.size() > 1 && total > 0 && total <= kPackCap;
  if (packed) {
    if (!d->det_pack_a.p) {
      HIP_TRY(d->det_pack_a.alloc(kPackCap));
      HIP_TRY(d->det_pack_w.alloc(kPackCap));
    }
    const size_t need = total * (sizeof(int4) + sizeof(float2));
    if (d->det_pack_pin_bytes < need) {
      if (d->det_pack_pin) (void)hipHostFree(d->det_pack_pin);
      d->det_pack_pin = nullptr;
      d->det_pack_pin_bytes = 0;
      HIP_TRY(hipHostMalloc(&d->det_pack_pin, need + need / 2, hipHostMallocDefault));
      d->det_pack_pin_bytes = need + need / 2;
    }
    launch_det_pack(X, (int)list.size(), d->det_pack_a.p, d->det_pack_w.p, (int64_t)kPackCap, d->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(d->det_pack_pin, d->det_pack_a.p, total * sizeof(int4), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipMemcpyAsync((char *)d->det_pack_pin + total * sizeof(int4), d->det_pack_w.p, total * sizeof(float2), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
  }
  size_t off = 0;
  for (int i = 0; i < (int)list.size(); ++i) {
    wfst_decoder::DetLattice &L = detached ? d->pf_cache[(size_t)list[i]] : d->det_cache[(size_t)list[i]];
    L.n_states = 0; L.n_proper = 0; L.a.clear(); L.w.clear();
    L.err = res[4 * i + 2];   // reported when THIS channel's lattice is asked for
    if (!detached && d->p_ctl[list[i]].error) L.err = kDetErrCtl | d->p_ctl[list[i]].error;
    if (detached) d->pf_have[(size_t)list[i]] = 1;
    else {
      d->det_cached[(size_t)list[i]] = live ? 0 : 1;
      if (live) { d->det_live_nd[(size_t)list[i]] = d->h_decoded[list[i]]; d->det_live_final[(size_t)list[i]] = use_final_probs ? 1 : 0; }
    }
    if (L.err) continue;
    L.n_states = res[4 * i];
    L.n_proper = res[4 * i + 3];
    const size_t na = (size_t)std::min(res[4 * i + 1], X.out_cap);
    L.a.resize(na);
    L.w.resize(na);
    if (!na) continue;
    if (packed) {
      memcpy(L.a.data(), (const int4 *)d->det_pack_pin + off, na * sizeof(int4));
      memcpy(L.w.data(), (const float2 *)((const char *)d->det_pack_pin + total * sizeof(int4)) + off, na * sizeof(float2));
      off += na;
    } else {
      HIP_TRY(hipMemcpyAsync(L.a.data(), X.out_a + (size_t)i * X.out_cap, na * sizeof(int4), hipMemcpyDeviceToHost, d->stream));
      HIP_TRY(hipMemcpyAsync(L.w.data(), X.out_w + (size_t)i * X.out_cap, na * sizeof(float2), hipMemcpyDeviceToHost, d->stream));
    }
  }
  HIP_TRY(hipStreamSynchronize(d->stream));
  bool all_current = !live;   // every channel of the list still holds the utterance these lattices were made of
  if (detached) {
    // a channel that still holds the very utterance (finalized, not finalized again since): GetLattice finds the work done too
    for (int i = 0; i < (int)list.size(); ++i) {
      const int c = list[i];
      if (d->h_state[c] == 2 && d->pf_epoch[(size_t)i] == d->fin_epoch[(size_t)c]) { d->det_cache[(size_t)c] = d->pf_cache[(size_t)c]; d->det_cached[(size_t)c] = 1; }
      else all_current = false;
    }
  }
  for (int i = 0; i < (int)list.size(); ++i)
    all_current = all_current && d->h_state[list[i]] == 2 && res[4 * i + 2] == 0 && (detached || !d->p_ctl[list[i]].error);
  if (all_current) {
    // the workspace slots hold these lattices: a batched second pass / n-best right behind starts from them (postprocess_batch)
    d->post_dev_list = list;
    d->post_dev_decoded.resize(list.size());
    for (size_t i = 0; i < list.size(); ++i) d->post_dev_decoded[i] = d->h_decoded[list[i]];
    d->post_dev_dres = res;
    d->post_dev_cres.clear();
    d->post_dev_o = nullptr; d->post_dev_n = nullptr;
  }
  return WFST_OK;
}

// GetLattice ahead of its request (include/wfst_decoder.h): the finalized channels not determinized yet go to the determinizer
// NOW, on a side stream -- one lane per lattice, a launch as long as its largest lattice, beside which the decoder's own stream
// serves best paths and n-best lists (both only read the raw lattices; the arena-index -> lattice-state map the n-best search
// and the determinizer each write is the same map).  The first wfst_decoder_get_determinized_lattice finds the work done or waits.
static int prefetch_determinized(wfst_decoder *d, bool detached);
int wfst_decoder_prefetch_determinized(wfst_decoder *d) { return prefetch_determinized(d, false); }
// ... DETACHED: the determinizer's first phase -- everything that reads the channels' state: control blocks, resolved lists, the
// arena-index scratch; a fraction of a millisecond -- runs on the decoder's stream, the subset construction (tens of milliseconds on
// one lane per lattice) on a stream of its own, on the workspace alone: wfst_decoder_init / _advance / _finalize do NOT wait for it,
// the channels go on to their next utterances beside it.  The lattices are kept per channel (wfst_decoder_get_prefetched_lattice)
// until the next detached prefetch is harvested.
int wfst_decoder_prefetch_determinized_detached(wfst_decoder *d) { return prefetch_determinized(d, true); }

static int prefetch_determinized(wfst_decoder *d, bool detached) {
  if (!d) return fail(WFST_E_ARG, "NULL decoder");
  if (!d->D.lattice) return fail(WFST_E_STATE, "GetLattice needs a decoder created with wfst_limits.lattice_links > 0");
  HIP_TRY(hipSetDevice(d->device));
  int rc = finish_prefetch(d);
  if (rc != WFST_OK) return rc;
  rc = ensure_det_workspace(d);
  if (rc != WFST_OK) return rc;
  std::vector<int32_t> list;
  for (int c = 0; c < d->n_channels && (int32_t)list.size() < d->det_slots; ++c)   // (one launch's worth; the rest on request)
    if (d->h_state[c] == 2 && !d->det_cached[c]) list.push_back(c);
  if (list.empty()) return WFST_OK;
  hipStream_t side = (d->n_groups > 1 && !detached) ? d->gstreams[1] : d->det_stream;   // (detached: a stream the frame loop never uses)
  if (!side) {
    HIP_TRY(hipStreamCreateWithFlags(&d->det_stream, hipStreamNonBlocking));
    side = d->det_stream;
  }
  if (d->fin_epoch.empty()) d->fin_epoch.assign((size_t)d->n_channels, 0);
  d->pf_epoch.resize(list.size());
  for (size_t i = 0; i < list.size(); ++i) d->pf_epoch[i] = d->fin_epoch[(size_t)list[i]];
  if (!d->pf_ev_start) {
    HIP_TRY(hipEventCreateWithFlags(&d->pf_ev_start, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&d->pf_ev_done, hipEventDisableTiming));
  }
  if (!d->pf_dev.p) {
    HIP_TRY(hipStreamSynchronize(d->stream));
    HIP_TRY(d->pf_dev.alloc((size_t)d->n_channels));
    HIP_TRY(hipHostMalloc((void **)&d->pf_pin, (size_t)d->n_channels * 8 * 4, hipHostMallocDefault));
  }
  d->pf_list = list;
  int32_t *pin_list = d->pf_pin, *pin_res = d->pf_pin + (size_t)d->n_channels * 4;
  for (size_t i = 0; i < list.size(); ++i) pin_list[i] = list[i];
  d->post_dev_list.clear();   // (the slots are about to hold other lattices than the last batched call's)
  HIP_TRY(hipMemcpyAsync(d->pf_dev.p, pin_list, list.size() * 4, hipMemcpyHostToDevice, d->stream));
  if (detached) launch_determinize(d->D, d->det, d->pf_dev.p, (int32_t)list.size(), d->stream, 1);   // the CSR: in stream order before anything the channels do next
  HIP_TRY(hipEventRecord(d->pf_ev_start, d->stream));   // FinalizeDecoding's pruning and listing are done, the channel list is up
  HIP_TRY(hipStreamWaitEvent(side, d->pf_ev_start, 0));
  launch_determinize(d->D, d->det, d->pf_dev.p, (int32_t)list.size(), side, detached ? 2 : 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(pin_res, d->det.result, list.size() * 4 * 4, hipMemcpyDeviceToHost, side));
  HIP_TRY(hipEventRecord(d->pf_ev_done, side));
  d->pf_pending = true;
  d->pf_detached = detached;
  return WFST_OK;
}

static int finish_prefetch(wfst_decoder *d) {
  if (!d->pf_pending) return WFST_OK;
  HIP_TRY(hipEventSynchronize(d->pf_ev_done));
  d->pf_pending = false;   // (the launch is over: from here on its lattices are either taken over below or lost with the error returned)
  const bool detached = d->pf_detached;
  d->pf_detached = false;
  d->pf_res.assign(d->pf_pin + (size_t)d->n_channels * 4, d->pf_pin + (size_t)d->n_channels * 4 + d->pf_list.size() * 4);
  // (a channel initialised or finalized anew since the launch: hooks in front of those calls came here first)
  return harvest_determinized(d, d->pf_list, d->pf_res, false, 1, detached);
}

// Waits for a prefetch in flight and takes its lattices over (what the next prefetch, or any other use of the determinizer, does
// by itself).
int wfst_decoder_harvest_prefetched(wfst_decoder *d) {
  if (!d) return fail(WFST_E_ARG, "NULL decoder");
  HIP_TRY(hipSetDevice(d->device));
  return finish_prefetch(d);
}

// The determinized lattice of `channel` as the last HARVESTED detached prefetch left it -- the lattice of the utterance the
// channel had finalized when that wfst_decoder_prefetch_determinized_detached was called, whatever the channel has gone on to
// since.  (A prefetch still in flight is not waited for: the call before it is what this returns -- a service fetches utterance
// k - 1's lattices right after it has started utterance k's.)  Same outputs as wfst_decoder_get_determinized_lattice;
// WFST_E_STATE if no harvested detached prefetch has covered the channel.
int wfst_decoder_get_prefetched_lattice(wfst_decoder *d, int32_t channel, int32_t cap_states, int32_t cap_arcs, int32_t *n_states, int32_t *n_arcs,
                                        int32_t *st_final, int32_t *a_src, int32_t *a_dst, int32_t *a_ilabel, int32_t *a_olabel, float *a_graph,
                                        float *a_acoustic) {
  if (!d || channel < 0 || channel >= d->n_channels || !n_states || !n_arcs) return fail(WFST_E_ARG, "bad argument");
  HIP_TRY(hipSetDevice(d->device));
  *n_states = 0;
  *n_arcs = 0;
  if (d->pf_have.empty() || !d->pf_have[(size_t)channel]) return fail(WFST_E_STATE, "no harvested detached prefetch has covered this channel");
  const wfst_decoder::DetLattice &L = d->pf_cache[(size_t)channel];
  if (L.err & kDetErrCtl) return fail_ctl_error(channel, L.err & ~kDetErrCtl);
  if (L.err == 2) return fail(WFST_E_CAPACITY, "channel " + std::to_string(channel) + ": raw lattice larger than the determinizer takes");
  if (L.err) return fail(WFST_E_CAPACITY, "channel " + std::to_string(channel) + ": the subset construction outgrew its workspace (lattice not determinizable within bounds)");
  *n_states = L.n_states;
  *n_arcs = (int32_t)L.a.size();
  if (L.n_states > cap_states || (int32_t)L.a.size() > cap_arcs) return fail(WFST_E_CAPACITY, "lattice larger than the given capacities");
  for (int32_t s = 0; s < L.n_states; ++s)
    if (st_final) st_final[s] = s >= L.n_proper ? 1 : 0;
  for (size_t k = 0; k < L.a.size(); ++k) {
    if (a_src) a_src[k] = L.a[k].x;
    if (a_dst) a_dst[k] = L.a[k].y;
    if (a_ilabel) a_ilabel[k] = 0;
    if (a_olabel) a_olabel[k] = L.a[k].z;
    if (a_graph) a_graph[k] = L.w[k].x;
    if (a_acoustic) a_acoustic[k] = L.w[k].y;
  }
  return WFST_OK;
}

int wfst_decoder_get_determinized_lattice(wfst_decoder *d, int32_t channel, int32_t use_final_probs, int32_t cap_states,
                                          int32_t cap_arcs, int32_t *n_states, int32_t *n_arcs, int32_t *st_final,
                                          int32_t *a_src, int32_t *a_dst, int32_t *a_ilabel, int32_t *a_olabel,
                                          float *a_graph, float *a_acoustic) {
  if (!d || channel < 0 || channel >= d->n_channels || !n_states || !n_arcs) return fail(WFST_E_ARG, "bad argument");
  if (!d->D.lattice) return fail(WFST_E_STATE, "GetLattice needs a decoder created with wfst_limits.lattice_links > 0");
  if (d->h_state[channel] == 0) return fail(WFST_E_STATE, "GetLattice before InitDecoding");
  const bool live = d->h_state[channel] == 1;
  HIP_TRY(hipSetDevice(d->device));
  *n_states = 0;
  *n_arcs = 0;
  if (!live && !use_final_probs) return WFST_OK;  // as GetRawLattice (base-inl.h:879-884)
  DetDev &X = d->det;
  {
    const int rcw = ensure_det_workspace(d);
    if (rcw != WFST_OK) return rcw;
    const int rcp = finish_prefetch(d);   // (the workspace slots are the prefetch's until it is harvested)
    if (rcp != WFST_OK) return rcp;
  }
  // a live channel's result is kept for as long as the channel has not moved on (the size query and the fetch of one request
  // are two calls: the second reuses the first's work)
  const bool live_hit = live && d->det_live_nd[(size_t)channel] == d->h_decoded[channel] &&
                        d->det_live_final[(size_t)channel] == (use_final_probs ? 1 : 0);
  const bool only = d->det_only == channel;   // (determinize_alone: this channel alone, into workspace slot 0, not from a cache)
  if (only || (live && !live_hit) || (!live && !d->det_cached[channel])) {
    // which channels: mid-utterance just this one (its lists are resolved first); after FinalizeDecoding every
    // finalized channel not determinized yet (their lists were resolved by FinalizeDecoding), det_slots per launch
    std::vector<int32_t> all;
    if (live || only) all.push_back(channel);
    else
      for (int c = 0; c < d->n_channels; ++c)
        if (d->h_state[c] == 2 && !d->det_cached[c]) all.push_back(c);
    for (size_t first = 0; first < all.size(); first += (size_t)d->det_slots) {
      const std::vector<int32_t> list(all.begin() + (long)first, all.begin() + (long)std::min(all.size(), first + (size_t)d->det_slots));
      const int32_t *dev;
      int32_t cnt;
      int rc = stage_channels(d, list.data(), (int32_t)list.size(), &dev, &cnt);
      if (rc != WFST_OK) return rc;
      if (live) launch_lattice_emit(d->D, dev, cnt, use_final_probs ? 1 : 0, d->stream);
      d->post_dev_list.clear();   // (the slots are about to hold other lattices than the last batched call's)
      launch_determinize(d->D, X, dev, cnt, d->stream);
      HIP_TRY(hipGetLastError());
      std::vector<int32_t> res((size_t)cnt * 4);
      HIP_TRY(hipMemcpyAsync(res.data(), X.result, res.size() * 4, hipMemcpyDeviceToHost, d->stream));
      rc = harvest_determinized(d, list, res, live, use_final_probs);
      if (rc != WFST_OK) return rc;
    }
  }
  const wfst_decoder::DetLattice &L = d->det_cache[(size_t)channel];
  if (L.err & kDetErrCtl) return fail_ctl_error(channel, L.err & ~kDetErrCtl);
  if (L.err == 2)
    return fail(WFST_E_CAPACITY, "channel " + std::to_string(channel) + ": raw lattice larger than the determinizer takes (" +
                                     std::to_string(X.raw_states_cap) + " states / " + std::to_string(X.raw_arcs_cap) + " arcs)");
  if (L.err)
    return fail(WFST_E_CAPACITY, "channel " + std::to_string(channel) + ": the subset construction outgrew its workspace (lattice not determinizable within bounds)");
  // the raw lattice's own "no lattice" cases (a frame without tokens, nothing decoded) give an empty result here too
  *n_states = L.n_states;
  *n_arcs = (int32_t)L.a.size();
  if (L.n_states > cap_states || (int32_t)L.a.size() > cap_arcs) return fail(WFST_E_CAPACITY, "lattice larger than the given capacities");
  for (int32_t s = 0; s < L.n_states; ++s)
    if (st_final) st_final[s] = s >= L.n_proper ? 1 : 0;
  for (size_t k = 0; k < L.a.size(); ++k) {
    if (a_src) a_src[k] = L.a[k].x;
    if (a_dst) a_dst[k] = L.a[k].y;
    if (a_ilabel) a_ilabel[k] = 0;
    if (a_olabel) a_olabel[k] = L.a[k].z;
    if (a_graph) a_graph[k] = L.w[k].x;
    if (a_acoustic) a_acoustic[k] = L.w[k].y;
  }
  return WFST_OK;
}

// The determinized lattice of ONE channel into workspace slot 0 (a cached host copy of an earlier batch determinization does not
// hold the device copy any more): the other finalized channels are hidden from the batch sweep for the call.
static int determinize_alone(wfst_decoder *d, int32_t channel, int32_t use_final_probs, int32_t *ns, int32_t *na) {
  d->post_dev_list.clear();   // (slot 0 is about to hold another lattice)
  // (det_only: the call below determinizes exactly this channel, afresh, whatever is cached and whichever other channels are
  // finalized -- also when it is the decoder's first determinizer use and the caches do not exist yet)
  d->det_only = channel;
  int rc = wfst_decoder_get_determinized_lattice(d, channel, use_final_probs, 0, 0, ns, na, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  d->det_only = -1;
  if (rc != WFST_OK && !(rc == WFST_E_CAPACITY && *ns > 0)) return rc;
  return WFST_OK;
}

// ComposeLattice with the old LM and with the new one over the determinized lattices of workspace slots [0, cnt), on the device, in ONE
// launch (a workgroup per lattice); res[4 * i ..] = {states, arcs, status, -} of slot i
static int compose_slots(wfst_decoder *d, const wfst_lm *old_lm, const wfst_lm *new_lm, int32_t cnt, int32_t *res) {
  CmpDev &Y = d->cmp;
  if (d->cmp_slots < cnt) {
    Y.pair_cap = 65536;
    Y.arc_cap = 262144;
    Y.ws_ints = 11ll * Y.arc_cap + 13ll * Y.pair_cap + 64;
    HIP_TRY(hipStreamSynchronize(d->stream));
    HIP_TRY(d->cmp_ws.alloc((size_t)cnt * (size_t)Y.ws_ints));
    HIP_TRY(d->cmp_result.alloc((size_t)cnt * 4));
    HIP_TRY(d->cmp_fin.alloc((size_t)cnt * (size_t)Y.pair_cap));
    HIP_TRY(d->cmp_out_a.alloc((size_t)cnt * (size_t)Y.arc_cap));
    HIP_TRY(d->cmp_out_w.alloc((size_t)cnt * (size_t)Y.arc_cap));
    Y.ws = d->cmp_ws.p;
    Y.result = d->cmp_result.p;
    Y.out_fin = d->cmp_fin.p;
    Y.out_a = d->cmp_out_a.p;
    Y.out_w = d->cmp_out_w.p;
    d->cmp_slots = cnt;
  }
  launch_compose2(d->det, Y, old_lm->view(), new_lm->view(), cnt, d->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(res, Y.result, (size_t)cnt * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  return WFST_OK;
}
static int compose_slot0(wfst_decoder *d, const wfst_lm *old_lm, const wfst_lm *new_lm, int32_t res[4]) {
  int rc = compose_slots(d, old_lm, new_lm, 1, res);
  if (rc != WFST_OK) return rc;
  if (res[2] != 0) return fail(WFST_E_CAPACITY, "the composed lattice outgrew the composition workspace (" + std::to_string(d->cmp.pair_cap) + " states / " + std::to_string(d->cmp.arc_cap) + " arcs)");
  return WFST_OK;
}

// ---- the service's post-processing as a BATCH (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:50-105: GetLattice, its second LM pass
// under --use-second, GetNbest -- the reference runs them per utterance, one worker thread each, concurrently,
// v2-asr/v2-asr-work-thread.h:66): the determinized lattices of a list of FINALIZED channels, ComposeLattice x 2 and NShortestPath
// for all of them in one launch each (a workgroup per lattice), the results fetched once and kept for the per-channel fetch calls.
static int postprocess_batch(wfst_decoder *d, const int32_t *channels, int32_t n, int32_t use_final_probs, const wfst_lm *old_lm,
                             const wfst_lm *new_lm, int32_t n_paths) {
  if (!d || (old_lm == nullptr) != (new_lm == nullptr)) return fail(WFST_E_ARG, "bad argument (both LMs or neither)");
  if (!d->D.lattice) return fail(WFST_E_STATE, "GetLattice / GetNbest need a decoder created with wfst_limits.lattice_links > 0");
  if (old_lm && (old_lm->device != d->device || new_lm->device != d->device)) return fail(WFST_E_ARG, "the LMs must be on the decoder's device");
  if (n_paths < 0 || n_paths > 4096) return fail(WFST_E_ARG, "1 <= n <= 4096 paths");
  HIP_TRY(hipSetDevice(d->device));
  std::vector<int32_t> all;
  if (channels) {
    if (n <= 0 || n > d->n_channels) return fail(WFST_E_ARG, "bad channel count");
    std::vector<char> seen((size_t)d->n_channels, 0);
    for (int i = 0; i < n; ++i) {
      const int c = channels[i];
      if (c < 0 || c >= d->n_channels || seen[(size_t)c]) return fail(WFST_E_ARG, "channel index out of range or listed twice");
      seen[(size_t)c] = 1;
      if (d->h_state[c] != 2) return fail(WFST_E_STATE, "the batched post-processing takes finalized channels (mid-utterance: the per-channel calls)");
      all.push_back(c);
    }
  } else {
    for (int c = 0; c < d->n_channels; ++c)
      if (d->h_state[c] == 2) all.push_back(c);
  }
  if (d->resc_cache.empty()) { d->resc_cache.resize((size_t)d->n_channels); d->nbp_cache.resize((size_t)d->n_channels); }
  wfst_decoder::PostKey key;
  key.o = old_lm; key.n = new_lm; key.use_final = use_final_probs ? 1 : 0; key.n_paths = n_paths; key.valid = true;
  auto store_empty = [&](int c) {
    key.decoded = d->h_decoded[c];
    if (n_paths) { wfst_decoder::NbPaths &R = d->nbp_cache[(size_t)c]; R = wfst_decoder::NbPaths(); R.key = key; R.off.assign(1, 0); }
    else { wfst_decoder::RescLattice &R = d->resc_cache[(size_t)c]; R = wfst_decoder::RescLattice(); R.key = key; }
  };
  if (!use_final_probs) {   // a finalized channel without final-probs has no lattice (GetRawLattice, base-inl.h:879-884)
    for (int c : all) store_empty(c);
    return WFST_OK;
  }
  if (all.empty()) return WFST_OK;
  int rc = ensure_det_workspace(d);
  if (rc != WFST_OK) return rc;
  rc = finish_prefetch(d);
  if (rc != WFST_OK) return rc;
  DetDev &X = d->det;
  const size_t chunk = (size_t)std::max(1, std::min(d->det_slots, 128));
  for (size_t first = 0; first < all.size(); first += chunk) {
    const std::vector<int32_t> list(all.begin() + (long)first, all.begin() + (long)std::min(all.size(), first + chunk));
    int32_t cnt = (int32_t)list.size();
    std::vector<int32_t> decoded((size_t)cnt);
    for (int i = 0; i < cnt; ++i) decoded[(size_t)i] = d->h_decoded[list[(size_t)i]];
    // (the determinizer's slots still hold these very lattices -- a prefetch harvested just before, or the batch of second passes
    // just before this batch of n-best requests?  The composition's too, under the same LMs?)
    const bool det_held = list == d->post_dev_list && decoded == d->post_dev_decoded;
    const bool held = det_held && (!old_lm || (d->post_dev_o == old_lm && d->post_dev_n == new_lm && d->post_dev_cres.size() == (size_t)cnt * 4));
    std::vector<int32_t> dres, cres;
    if (held) {
      dres = d->post_dev_dres;
      if (old_lm) cres = d->post_dev_cres;
    } else {
      if (det_held) {
        dres = d->post_dev_dres;   // GetLattice is done (wfst_decoder_prefetch_determinized ran it beside the best paths): the second pass starts here
      } else {
        d->post_dev_list.clear();
        const int32_t *dev;
        rc = stage_channels(d, list.data(), (int32_t)list.size(), &dev, &cnt);
        if (rc != WFST_OK) return rc;
        // GetLattice: the determinized lattices of the chunk, list[i] in workspace slot i
        launch_determinize(d->D, X, dev, cnt, d->stream);
        HIP_TRY(hipGetLastError());
        dres.resize((size_t)cnt * 4);
        HIP_TRY(hipMemcpyAsync(dres.data(), X.result, dres.size() * 4, hipMemcpyDeviceToHost, d->stream));
        rc = read_ctl(d);  // synchronises the stream
        if (rc != WFST_OK) return rc;
        rc = check_ctl_errors(d);
        if (rc != WFST_OK) return rc;
      }
      for (int i = 0; i < cnt; ++i) {
        if (dres[(size_t)4 * i + 2] == 2)
          return fail(WFST_E_CAPACITY, "channel " + std::to_string(list[(size_t)i]) + ": raw lattice larger than the determinizer takes");
        if (dres[(size_t)4 * i + 2])
          return fail(WFST_E_CAPACITY, "channel " + std::to_string(list[(size_t)i]) + ": the subset construction outgrew its workspace");
      }
      // ... the second LM pass: ComposeLattice with the old LM and with the new one
      if (old_lm) {
        cres.resize((size_t)cnt * 4);
        rc = compose_slots(d, old_lm, new_lm, cnt, cres.data());
        if (rc != WFST_OK) return rc;
        for (int i = 0; i < cnt; ++i)
          if (cres[(size_t)4 * i + 2] == 1)
            return fail(WFST_E_CAPACITY, "channel " + std::to_string(list[(size_t)i]) + ": the composed lattice outgrew the composition workspace");
      }
      d->post_dev_list = list; d->post_dev_decoded = decoded; d->post_dev_dres = dres; d->post_dev_cres = cres;
      d->post_dev_o = old_lm; d->post_dev_n = new_lm;
    }
    const std::vector<int32_t> &lres = old_lm ? cres : dres;   // {states, arcs, ...} of the lattices the paths / the fetch are taken from
    const int4 *la = old_lm ? d->cmp.out_a : X.out_a;
    const float2 *lw = old_lm ? d->cmp.out_w : X.out_w;
    const int64_t lstride = old_lm ? d->cmp.arc_cap : X.out_cap;
    // the lattices' arcs (the paths report labels and costs of their arcs; the lattice fetch returns them)
    std::vector<std::vector<int4>> ha((size_t)cnt);
    std::vector<std::vector<float2>> hw((size_t)cnt);
    std::vector<std::vector<int32_t>> hfin((size_t)cnt);
    for (int i = 0; i < cnt; ++i) {
      const size_t na = (size_t)std::max(0, lres[(size_t)4 * i + 1]), nsi = (size_t)std::max(0, lres[(size_t)4 * i]);
      ha[(size_t)i].resize(na);
      hw[(size_t)i].resize(na);
      if (na) {
        HIP_TRY(hipMemcpyAsync(ha[(size_t)i].data(), la + (size_t)i * (size_t)lstride, na * sizeof(int4), hipMemcpyDeviceToHost, d->stream));
        HIP_TRY(hipMemcpyAsync(hw[(size_t)i].data(), lw + (size_t)i * (size_t)lstride, na * sizeof(float2), hipMemcpyDeviceToHost, d->stream));
      }
      if (old_lm && !n_paths && nsi) {
        hfin[(size_t)i].resize(nsi);
        HIP_TRY(hipMemcpyAsync(hfin[(size_t)i].data(), d->cmp.out_fin + (size_t)i * (size_t)d->cmp.pair_cap, nsi * 4, hipMemcpyDeviceToHost, d->stream));
      }
    }
    if (!n_paths) {
      HIP_TRY(hipStreamSynchronize(d->stream));
      for (int i = 0; i < cnt; ++i) {
        const int c = list[(size_t)i];
        wfst_decoder::RescLattice &R = d->resc_cache[(size_t)c];
        key.decoded = d->h_decoded[c];
        R.key = key;
        R.n_states = std::max(0, lres[(size_t)4 * i]);
        R.a.swap(ha[(size_t)i]);
        R.w.swap(hw[(size_t)i]);
        R.fin.swap(hfin[(size_t)i]);
      }
      continue;
    }
    // ... NShortestPath: one workgroup per lattice
    int32_t ns_max = 1, na_max = 1;
    for (int i = 0; i < cnt; ++i) { ns_max = std::max(ns_max, lres[(size_t)4 * i]); na_max = std::max(na_max, lres[(size_t)4 * i + 1]); }
    NbPathsDev P = {};
    P.a = la; P.w = lw; P.res = old_lm ? d->cmp.result : X.result; P.fin = old_lm ? d->cmp.out_fin : nullptr;
    P.in_stride = lstride; P.fin_stride = old_lm ? d->cmp.pair_cap : 0;
    const int64_t nmax = std::max(ns_max, na_max);
    P.ws_ints = 7ll * ns_max + 4ll * nmax + na_max + 16;
    P.list_cap = std::min<int64_t>((int64_t)ns_max * n_paths + 1, 1ll << 24);
    const int64_t out_cap = std::min<int64_t>((int64_t)n_paths * ns_max, 1ll << 22);
    // (every buffer grows on its OWN size: the batch call and the single-channel call share them with different shapes.  The path
    // workspace is the per-lattice worst case times the batch: where the device cannot give that much the call reports a capacity,
    // not a device error -- GpuBatchDecoder::GetNbests then asks channel by channel, which needs one lattice's worth)
    auto grow = [&](auto &buf, int64_t need) -> bool { return (int64_t)buf.n >= need || buf.alloc((size_t)need) == hipSuccess; };
    if (!grow(d->np_ws, P.ws_ints * cnt) || !grow(d->np_lists, P.list_cap * cnt) || !grow(d->np_arcs, out_cap * cnt) ||
        !grow(d->np_off, (int64_t)(n_paths + 1) * cnt) || !grow(d->np_tot, (int64_t)n_paths * cnt) || !grow(d->np_out, 4ll * cnt)) {
      (void)hipGetLastError();
      return fail(WFST_E_CAPACITY, "n-best: no device memory for the path workspace of " + std::to_string(cnt) + " lattices at once (" +
                                       std::to_string(n_paths) + " paths each): ask channel by channel (wfst_decoder_get_nbest_paths)");
    }
    P.n = n_paths;
    P.ws = d->np_ws.p; P.lists = d->np_lists.p;
    P.out = d->np_out.p; P.out_off = d->np_off.p; P.out_tot = d->np_tot.p;
    P.out_arcs = d->np_arcs.p; P.out_cap = (int32_t)out_cap;
    launch_nbest_paths(P, cnt, d->stream);
    HIP_TRY(hipGetLastError());
    std::vector<int32_t> pout((size_t)cnt * 4), poff((size_t)cnt * (size_t)(n_paths + 1));
    std::vector<float> ptot((size_t)cnt * (size_t)n_paths);
    HIP_TRY(hipMemcpyAsync(pout.data(), P.out, pout.size() * 4, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipMemcpyAsync(poff.data(), P.out_off, poff.size() * 4, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipMemcpyAsync(ptot.data(), P.out_tot, ptot.size() * 4, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    for (int i = 0; i < cnt; ++i) {
      const int32_t *o = &pout[(size_t)4 * i];
      if (lres[(size_t)4 * i] <= 0) continue;   // (no lattice: no paths)
      if (o[2] == 3) return fail(WFST_E_DEVICE, "n-best: the lattice has a cycle");
      if (o[2] != 0) return fail(WFST_E_CAPACITY, "n-best: channel " + std::to_string(list[(size_t)i]) + " outgrew the batch's path workspace (ask for it alone: wfst_decoder_get_nbest_paths)");
    }
    for (int i = 0; i < cnt; ++i) {
      const int c = list[(size_t)i];
      wfst_decoder::NbPaths &R = d->nbp_cache[(size_t)c];
      R = wfst_decoder::NbPaths();
      key.decoded = d->h_decoded[c];
      R.key = key;
      const int32_t found = lres[(size_t)4 * i] > 0 ? pout[(size_t)4 * i] : 0, total = lres[(size_t)4 * i] > 0 ? pout[(size_t)4 * i + 1] : 0;
      R.off.assign(1, 0);
      if (found) R.off.assign(poff.begin() + (long)i * (n_paths + 1), poff.begin() + (long)i * (n_paths + 1) + found + 1);
      R.tot.assign(ptot.begin() + (long)i * n_paths, ptot.begin() + (long)i * n_paths + found);
      std::vector<int32_t> arcs((size_t)total);
      if (total) HIP_TRY(hipMemcpy(arcs.data(), P.out_arcs + (size_t)i * (size_t)P.out_cap, (size_t)total * 4, hipMemcpyDeviceToHost));
      R.olabel.resize((size_t)total); R.graph.resize((size_t)total); R.ac.resize((size_t)total);
      for (int32_t k = 0; k < total; ++k) {
        const size_t aidx = (size_t)arcs[(size_t)k];
        R.olabel[(size_t)k] = ha[(size_t)i][aidx].z;
        R.graph[(size_t)k] = hw[(size_t)i][aidx].x;
        R.ac[(size_t)k] = hw[(size_t)i][aidx].y;
      }
    }
  }
  return WFST_OK;
}

int wfst_decoder_rescore_lattices(wfst_decoder *d, const int32_t *channels, int32_t n, int32_t use_final_probs, const wfst_lm *old_lm,
                                  const wfst_lm *new_lm) {
  if (!old_lm || !new_lm) return fail(WFST_E_ARG, "the second pass needs both LMs");
  return postprocess_batch(d, channels, n, use_final_probs, old_lm, new_lm, 0);
}

int wfst_decoder_nbest_paths_batch(wfst_decoder *d, const int32_t *channels, int32_t n, int32_t n_paths, int32_t use_final_probs,
                                   const wfst_lm *old_lm, const wfst_lm *new_lm) {
  if (n_paths < 1) return fail(WFST_E_ARG, "1 <= n <= 4096 paths");
  return postprocess_batch(d, channels, n, use_final_probs, old_lm, new_lm, n_paths);
}

int wfst_decoder_get_rescored_lattice(wfst_decoder *d, int32_t channel, int32_t use_final_probs, const wfst_lm *old_lm, const wfst_lm *new_lm,
                                      int32_t cap_states, int32_t cap_arcs, int32_t *n_states, int32_t *n_arcs, int32_t *st_final,
                                      int32_t *a_src, int32_t *a_dst, int32_t *a_ilabel, int32_t *a_olabel, float *a_graph, float *a_acoustic) {
  if (!d || channel < 0 || channel >= d->n_channels || !n_states || !n_arcs || !old_lm || !new_lm) return fail(WFST_E_ARG, "bad argument");
  if (old_lm->device != d->device || new_lm->device != d->device) return fail(WFST_E_ARG, "the LMs must be on the decoder's device");
  // GetLattice of the service under --use-second: the determinized lattice first (its device copy stays in workspace slot 0 when this
  // channel is determinized alone) ...
  int32_t ns = 0, na = 0;
  *n_states = 0;
  *n_arcs = 0;
  if (d->h_state[channel] == 0) return fail(WFST_E_STATE, "GetLattice before InitDecoding");
  if (!d->D.lattice) return fail(WFST_E_STATE, "GetLattice needs a decoder created with wfst_limits.lattice_links > 0");
  if (!d->resc_cache.empty()) {   // a result of wfst_decoder_rescore_lattices for this very request
    const wfst_decoder::RescLattice &R = d->resc_cache[(size_t)channel];
    if (R.key.valid && R.key.o == old_lm && R.key.n == new_lm && R.key.use_final == (use_final_probs ? 1 : 0) && R.key.decoded == d->h_decoded[channel] && d->h_state[channel] == 2) {
      *n_states = R.n_states;
      *n_arcs = (int32_t)R.a.size();
      if (R.n_states > cap_states || (int32_t)R.a.size() > cap_arcs) return fail(WFST_E_CAPACITY, "lattice larger than the given capacities");
      for (int32_t q = 0; q < R.n_states; ++q)
        if (st_final) st_final[q] = R.fin[(size_t)q];
      for (size_t q = 0; q < R.a.size(); ++q) {
        if (a_src) a_src[q] = R.a[q].x;
        if (a_dst) a_dst[q] = R.a[q].y;
        if (a_ilabel) a_ilabel[q] = 0;
        if (a_olabel) a_olabel[q] = R.a[q].z;
        if (a_graph) a_graph[q] = R.w[q].x;
        if (a_acoustic) a_acoustic[q] = R.w[q].y;
      }
      return WFST_OK;
    }
  }
  int rc = determinize_alone(d, channel, use_final_probs, &ns, &na);
  if (rc != WFST_OK) return rc;
  if (ns == 0) return WFST_OK;   // no lattice (as wfst_decoder_get_raw_lattice)
  HIP_TRY(hipSetDevice(d->device));
  // ... then ComposeLattice with the old LM and with the new one, on the device
  int32_t res[4];
  rc = compose_slot0(d, old_lm, new_lm, res);
  if (rc != WFST_OK) return rc;
  CmpDev &Y = d->cmp;
  *n_states = res[0];
  *n_arcs = res[1];
  if (res[0] > cap_states || res[1] > cap_arcs) return fail(WFST_E_CAPACITY, "lattice larger than the given capacities");
  std::vector<int4> oa((size_t)res[1]);
  std::vector<float2> ow((size_t)res[1]);
  std::vector<int32_t> fin((size_t)res[0]);
  if (res[1]) {
    HIP_TRY(hipMemcpyAsync(oa.data(), Y.out_a, oa.size() * sizeof(int4), hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipMemcpyAsync(ow.data(), Y.out_w, ow.size() * sizeof(float2), hipMemcpyDeviceToHost, d->stream));
  }
  if (res[0]) HIP_TRY(hipMemcpyAsync(fin.data(), Y.out_fin, fin.size() * 4, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  for (int32_t s = 0; s < res[0]; ++s)
    if (st_final) st_final[s] = fin[(size_t)s];
  for (int32_t k = 0; k < res[1]; ++k) {
    if (a_src) a_src[k] = oa[(size_t)k].x;
    if (a_dst) a_dst[k] = oa[(size_t)k].y;
    if (a_ilabel) a_ilabel[k] = 0;
    if (a_olabel) a_olabel[k] = oa[(size_t)k].z;
    if (a_graph) a_graph[k] = ow[(size_t)k].x;
    if (a_acoustic) a_acoustic[k] = ow[(size_t)k].y;
  }
  return WFST_OK;
}

int wfst_decoder_get_nbest_paths(wfst_decoder *d, int32_t channel, int32_t n, int32_t use_final_probs, const wfst_lm *old_lm,
                                 const wfst_lm *new_lm, int32_t cap_paths, int32_t cap_arcs, int32_t *n_paths, int32_t *total_arcs,
                                 int32_t *path_off, float *path_tot, int32_t *a_olabel, float *a_graph, float *a_acoustic) {
  if (!d || channel < 0 || channel >= d->n_channels || !n_paths || !total_arcs || n < 1 || n > 4096 || (old_lm == nullptr) != (new_lm == nullptr))
    return fail(WFST_E_ARG, "bad argument (1 <= n <= 4096; both LMs or neither)");
  if (old_lm && (old_lm->device != d->device || new_lm->device != d->device)) return fail(WFST_E_ARG, "the LMs must be on the decoder's device");
  *n_paths = 0;
  *total_arcs = 0;
  if (d->h_state[channel] == 0) return fail(WFST_E_STATE, "GetNbest before InitDecoding");
  if (!d->D.lattice) return fail(WFST_E_STATE, "GetNbest needs a decoder created with wfst_limits.lattice_links > 0");
  if (!d->nbp_cache.empty()) {   // a result of wfst_decoder_nbest_paths_batch for this very request
    const wfst_decoder::NbPaths &R = d->nbp_cache[(size_t)channel];
    if (R.key.valid && R.key.o == old_lm && R.key.n == new_lm && R.key.use_final == (use_final_probs ? 1 : 0) && R.key.n_paths == n &&
        R.key.decoded == d->h_decoded[channel] && d->h_state[channel] == 2) {
      const int32_t found = (int32_t)R.tot.size(), total = (int32_t)R.olabel.size();
      *n_paths = found;
      *total_arcs = total;
      if (found > cap_paths || total > cap_arcs) return fail(WFST_E_CAPACITY, "n-best larger than the given capacities");
      for (int32_t q = 0; q <= found; ++q)
        if (path_off) path_off[q] = R.off[(size_t)q];
      for (int32_t q = 0; q < found; ++q)
        if (path_tot) path_tot[q] = R.tot[(size_t)q];
      for (int32_t q = 0; q < total; ++q) {
        if (a_olabel) a_olabel[q] = R.olabel[(size_t)q];
        if (a_graph) a_graph[q] = R.graph[(size_t)q];
        if (a_acoustic) a_acoustic[q] = R.ac[(size_t)q];
      }
      return WFST_OK;
    }
  }
  // GetLattice (the determinized lattice, into workspace slot 0; with LMs its second-pass rescoring) ...
  int32_t ns = 0, na = 0;
  int rc = determinize_alone(d, channel, use_final_probs, &ns, &na);
  if (rc != WFST_OK) return rc;
  if (ns == 0) return WFST_OK;
  HIP_TRY(hipSetDevice(d->device));
  NbPathsDev P = {};
  std::vector<int4> oa;
  std::vector<float2> ow;
  if (old_lm) {
    int32_t res[4];
    rc = compose_slot0(d, old_lm, new_lm, res);
    if (rc != WFST_OK) return rc;
    ns = res[0];
    na = res[1];
    if (ns == 0) return WFST_OK;   // (nothing reaches a final state of both LMs)
    P.a = d->cmp.out_a; P.w = d->cmp.out_w; P.res = d->cmp.result; P.fin = d->cmp.out_fin;
    oa.resize((size_t)na);
    ow.resize((size_t)na);
    if (na) {
      HIP_TRY(hipMemcpyAsync(oa.data(), d->cmp.out_a, oa.size() * sizeof(int4), hipMemcpyDeviceToHost, d->stream));
      HIP_TRY(hipMemcpyAsync(ow.data(), d->cmp.out_w, ow.size() * sizeof(float2), hipMemcpyDeviceToHost, d->stream));
    }
  } else {
    P.a = d->det.out_a; P.w = d->det.out_w; P.res = d->det.result; P.fin = nullptr;
  }
  // ... then NShortestPath on the device
  const int64_t nmax = std::max(ns, na);
  const int64_t ws_ints = 7ll * ns + 4ll * nmax + na + 16;
  const int64_t list_cap = std::min<int64_t>((int64_t)ns * n + 1, 1ll << 26);   // (a gigabyte of partial paths at most)
  const int64_t out_cap = std::min<int64_t>((int64_t)n * ns, 1ll << 24);
  HIP_TRY(hipStreamSynchronize(d->stream));
  if ((int64_t)d->np_ws.n < ws_ints) HIP_TRY(d->np_ws.alloc((size_t)ws_ints));
  if ((int64_t)d->np_lists.n < list_cap) HIP_TRY(d->np_lists.alloc((size_t)list_cap));
  if ((int64_t)d->np_arcs.n < out_cap) HIP_TRY(d->np_arcs.alloc((size_t)out_cap));
  if ((int64_t)d->np_off.n < n + 1) HIP_TRY(d->np_off.alloc((size_t)n + 1));
  if ((int64_t)d->np_tot.n < n) HIP_TRY(d->np_tot.alloc((size_t)n));   // (its own size: the batch call sizes the two differently)
  if ((int64_t)d->np_out.n < 4) HIP_TRY(d->np_out.alloc(4));
  P.n = n;
  P.ws = d->np_ws.p; P.ws_ints = (int64_t)d->np_ws.n;
  P.lists = d->np_lists.p; P.list_cap = (int64_t)d->np_lists.n;
  P.out = d->np_out.p; P.out_off = d->np_off.p; P.out_tot = d->np_tot.p;
  P.out_arcs = d->np_arcs.p; P.out_cap = (int32_t)std::min<int64_t>((int64_t)d->np_arcs.n, 0x7fffffff);
  launch_nbest_paths(P, 1, d->stream);
  HIP_TRY(hipGetLastError());
  int32_t out[4];
  HIP_TRY(hipMemcpyAsync(out, P.out, sizeof(out), hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  if (out[2] == 3) return fail(WFST_E_DEVICE, "n-best: the lattice has a cycle");
  if (out[2] != 0) return fail(WFST_E_CAPACITY, "n-best: " + std::to_string(n) + " paths over " + std::to_string(ns) + " states outgrew the path workspace");
  *n_paths = out[0];
  *total_arcs = out[1];
  if (out[0] > cap_paths || out[1] > cap_arcs) return fail(WFST_E_CAPACITY, "n-best larger than the given capacities");
  std::vector<int32_t> off((size_t)out[0] + 1), arcs((size_t)out[1]);
  std::vector<float> tot((size_t)out[0]);
  HIP_TRY(hipMemcpyAsync(off.data(), P.out_off, off.size() * 4, hipMemcpyDeviceToHost, d->stream));
  if (out[0]) HIP_TRY(hipMemcpyAsync(tot.data(), P.out_tot, tot.size() * 4, hipMemcpyDeviceToHost, d->stream));
  if (out[1]) HIP_TRY(hipMemcpyAsync(arcs.data(), P.out_arcs, arcs.size() * 4, hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  const wfst_decoder::DetLattice &L = d->det_cache[(size_t)channel];
  for (int32_t p = 0; p <= out[0]; ++p)
    if (path_off) path_off[p] = off[(size_t)p];
  for (int32_t p = 0; p < out[0]; ++p)
    if (path_tot) path_tot[p] = tot[(size_t)p];
  for (int32_t k = 0; k < out[1]; ++k) {
    const size_t a = (size_t)arcs[(size_t)k];
    const int4 A = old_lm ? oa[a] : L.a[a];
    const float2 W = old_lm ? ow[a] : L.w[a];
    if (a_olabel) a_olabel[k] = A.z;
    if (a_graph) a_graph[k] = W.x;
    if (a_acoustic) a_acoustic[k] = W.y;
  }
  return WFST_OK;
}

int wfst_decoder_get_stats(wfst_decoder *d, int32_t channel, int64_t stats[8]) {
  if (!d || !stats || channel < 0 || channel >= d->n_channels) return fail(WFST_E_ARG, "bad argument");
  HIP_TRY(hipSetDevice(d->device));
  int rc = read_ctl(d);
  if (rc != WFST_OK) return rc;
  const ChanCtl &c = d->p_ctl[channel];
  stats[0] = c.n_decoded;
  stats[1] = (int64_t)c.cnt_N;
  stats[2] = (int64_t)c.cnt_E;
  stats[3] = (int64_t)c.cnt_Z;
  stats[4] = (int64_t)c.cnt_tok;
  stats[5] = c.peak_tokens;
  stats[6] = (int64_t)c.cnt_rec;
  stats[7] = d->D.lattice ? c.link_count : c.lat_toks;  // lattice mode: forward links recorded; best-path mode: token collections run
  return WFST_OK;
}

int wfst_decoder_get_lattice_stats(wfst_decoder *d, int32_t channel, int64_t stats[5]) {
  if (!d || !stats || channel < 0 || channel >= d->n_channels) return fail(WFST_E_ARG, "bad argument");
  if (!d->D.lattice) return fail(WFST_E_STATE, "lattice statistics need a decoder created with wfst_limits.lattice_links > 0");
  HIP_TRY(hipSetDevice(d->device));
  unsigned long long v[4];
  HIP_TRY(hipMemcpyAsync(v, d->lat_stats.p + (size_t)channel * 4, sizeof(v), hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  stats[0] = (int64_t)v[0];
  stats[1] = (int64_t)v[1];
  stats[2] = (int64_t)v[2];
  stats[3] = (int64_t)(v[3] & 0xFFFFFFFFull);
  stats[4] = (int64_t)(v[3] >> 32);
  return WFST_OK;
}

int wfst_decoder_get_degraded_frames(wfst_decoder *d, int32_t channel, int32_t *n_frames) {
  if (!d || !n_frames || channel < 0 || channel >= d->n_channels) return fail(WFST_E_ARG, "bad argument");
  HIP_TRY(hipSetDevice(d->device));
  HIP_TRY(hipMemcpyAsync(n_frames, d->degraded.p + channel, sizeof(int32_t), hipMemcpyDeviceToHost, d->stream));
  HIP_TRY(hipStreamSynchronize(d->stream));
  return WFST_OK;
}

int wfst_decoder_get_frontier(wfst_decoder *d, int32_t channel, int32_t cap, int32_t *states, float *costs) {
  if (!d || channel < 0 || channel >= d->n_channels || cap < 0) return fail(WFST_E_ARG, "bad argument");
  HIP_TRY(hipSetDevice(d->device));
  int rc = read_ctl(d);
  if (rc != WFST_OK) return rc;
  const ChanCtl &c = d->p_ctl[channel];
  const int n = c.front_count, k = std::min(n, cap);
  if (k > 0) {
    std::vector<int4> t((size_t)k);
    HIP_TRY(hipMemcpy(t.data(), d->tok.p + (size_t)channel * (size_t)d->D.arena_cap + c.front_begin,
                      (size_t)k * sizeof(int4), hipMemcpyDeviceToHost));
    const std::vector<int32_t> &pos = d->graph->pos_host;
    for (int i = 0; i < k; ++i) {
      states[i] = (int32_t)(std::lower_bound(pos.begin(), pos.end(), t[i].x) - pos.begin());  // row -> state id
      memcpy(&costs[i], &t[i].y, 4);
    }
  }
  return n;
}

}  // extern "C"
