// Device-side data layout of the batched WFST token-passing decoder (gfx950 / MI355X).
//
// Everything a kernel touches is described here; wfst_capi.cc owns the allocations and
// wfst_kernels.hip the code.  Reference structures replaced (paths relative to the reference's
// src/): Fst state/arc arrays (newfst/optimize-fst.h:60-61), HashList<StateId,Token*>
// (util/hash-list.h), StdToken + backpointer (my-decoder/online-decoder-base.h:52-84).
#ifndef WFST_DEVICE_H_
#define WFST_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wfst {

// ---- graph in HBM: CSR -------------------------------------------------------------------
// state_info[s] = {arc_begin, (n_emit << 12) | n_eps}: one 8-byte load gives both arc ranges of
//   a state: epsilon arcs [arc_begin, arc_begin+n_eps), emitting arcs the n_emit after them.
// arcs[a]       = {ll_col, olabel, weight bits, nextstate}: 16-byte AoS, one dwordx4 per lane.
//   ll_col is the log-likelihood column of the arc's ilabel (tid2pdf applied at upload), -1 for
//   an input-epsilon arc.
// arc_ilabel[a] = original ilabel (transition-id) for output.
// arc_src[a]    = source state of arc a (resolves a winning arc to its source token).
struct GraphDev {
  const uint2 *state_info;
  const int4 *arcs;
  const int32_t *arc_ilabel;
  const int32_t *arc_src;
  int32_t start, final_state, n_states, n_arcs;
};

constexpr int kEpsBits = 12;
constexpr uint32_t kEpsMask = (1u << kEpsBits) - 1;
constexpr int32_t kEmptyKey = -1;
constexpr unsigned long long kEmptyVal = ~0ull;
constexpr uint32_t kNoArc = 0xFFFFFFFFu;

// error bits (ChanCtl::error)
constexpr int kErrTableFull = 1, kErrArenaFull = 2, kErrFrontierFull = 4, kErrWorklistFull = 8,
              kErrFramesFull = 16;

// ---- per-channel control block (one 128-byte line each) ----------------------------------
struct __attribute__((aligned(128))) ChanCtl {
  int32_t n_decoded;     // NumFramesDecoded()
  int32_t target;        // decode frames while n_decoded < target (set per advance call)
  int32_t front_begin;   // arena index of the current frontier's first token
  int32_t front_count;   // tokens in the current frontier
  int32_t cur_tab;       // hash table (0/1) holding the current frontier's states
  int32_t active;        // this frame step processes the channel
  uint32_t bound;        // orderable next_cutoff, tightened during expansion (atomicMin)
  float cur_cutoff;      // GetCutoff() result for the frame being expanded
  float adaptive_beam;
  int32_t n_occ[2];      // occupied-slot list length per hash table
  int32_t error;         // sticky kErr* bits
  int32_t finalized;
  int32_t peak_tokens;
  int32_t pad0[2];
  unsigned long long cnt_N, cnt_E, cnt_Z, cnt_tok, cnt_slots;  // work counters since init
  unsigned long long pad1[3];
};
static_assert(sizeof(ChanCtl) == 128, "ChanCtl must be one 128-byte line");

// ---- decoder (batch of channels) ---------------------------------------------------------
// Per channel c (all slabs are [n_channels][...] and 128-byte aligned per channel):
//   tok[c][arena_cap]        int4 {state, cost bits, prev token (arena index, -1 root), arc}
//   frame_off[c][max_frames+2]  arena offset of each frame's frontier (frame f = tokens
//                               [frame_off[f], frame_off[f+1]))
//   cutoff_hist[c][max_frames+2] cutoff used by the epsilon closure of frame f ([0] = beam)
//   keys/vals/toki[c][2][cap]   open-addressed next-state hash, ping-pong per frame:
//                               key = state, val = (orderable cost << 32 | arc), toki = arena index
//   occ[c][2][cap]              slots occupied in each table (cleared by walking this list)
//   front_slot[c][max_tok]      hash slot of each token of the frontier being built
//   worklist[c][2][wl_cap]      epsilon-closure frontiers (double buffered)
struct DecoderDev {
  GraphDev g;
  ChanCtl *ctl;
  int4 *tok;
  int32_t *frame_off;
  float *cutoff_hist;
  int32_t *keys;
  unsigned long long *vals;
  int32_t *toki;
  int32_t *occ;
  int32_t *front_slot;
  int32_t *worklist;
  const float *const *ll_base;  // [n_channels] device pointers to row 0 of each utterance matrix
  int32_t n_channels;
  int32_t stride;               // floats per log-likelihood row
  int32_t cap, log2cap;         // hash slots per table (power of two)
  int32_t max_tok;              // frontier capacity
  int32_t wl_cap;
  int32_t max_frames;
  int64_t arena_cap;
  // config (LatticeFasterDecoderConfig)
  float beam, lattice_beam, beam_delta;
  int32_t max_active, min_active, prune_interval;
};

// launch wrappers (wfst_kernels.hip)
void launch_init(const DecoderDev &D, const int32_t *chan_list_dev, int n, hipStream_t s);
void launch_boundary(const DecoderDev &D, const int32_t *target_dev, int do_finalize, int do_prep,
                     hipStream_t s);
void launch_set_finalized(const DecoderDev &D, const int32_t *chan_list_dev, int n, hipStream_t s);
void launch_expand(const DecoderDev &D, int tiles_per_channel, hipStream_t s);
void launch_best_path(const DecoderDev &D, const int32_t *chan_list_dev, int n, int use_final,
                      int cap, int32_t *ilabel, int32_t *olabel, float *graph, float *ac,
                      int32_t *n_hops, hipStream_t s);

}  // namespace wfst
#endif
