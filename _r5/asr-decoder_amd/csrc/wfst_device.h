// Device-side data layout of the batched WFST token-passing decoder (gfx950 / MI355X).
//
// Everything a kernel touches is described here; wfst_capi.cc owns the allocations and
// wfst_kernels.hip the code.  Reference structures replaced (paths relative to the reference's
// src/): Fst state/arc arrays (newfst/optimize-fst.h:60-61), HashList<StateId,Token*>
// (util/hash-list.h), StdToken + backpointer (my-decoder/online-decoder-base.h:52-84).
#ifndef WFST_DEVICE_H_
#define WFST_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace wfst {

// ---- graph in HBM: CSR with the row header in front of the row ---------------------------
// rows[]: ONE int4 array in "ext" index space.  State s owns slots [pos(s), pos(s)+1+num_arcs(s)),
//   pos(s) increasing in s (rows are packed, with padding slots where a row would otherwise
//   straddle one more 64-byte line than it needs), and IS identified by pos(s) everywhere on the device.
//   rows[pos(s)]      header {(n_emit << 12) | n_eps, original state id, pseudo arcs of s (fused closures),
//                     (first entry in eps_flat[] << 3) | entries, entries = 0: not flattened}
//   rows[pos(s)+1+i]  arc i {ll_col, next_eps(nextstate), weight bits, pos(nextstate)}; epsilon arcs
//                     first.  ll_col = log-likelihood column of the ilabel (tid2pdf applied at
//                     upload), -1 for an input-epsilon arc.
//   One gather finds a token's arcs, and its cache lines hold the arcs themselves (a separate
//   8-byte state table cost one more 64-byte fabric request per expanded token).
//   next_eps word: bit 31 = the state has outgoing epsilon arcs; bits 30..0 = 1 + its ordinal among
//   the graph's epsilon-TARGET states (0 = no epsilon arc enters it).
// eps_flat[]: the WHOLE epsilon closure of a state as a list, for states whose closure has at most
//   kFlatMax paths (nearly all): entry {epsilon-target ordinal of the path's end, last arc (row
//   index), (parent entry + 1) | 8 if the end state has epsilon arcs out, weight bits of the last arc},
//   parents before children.  The closure kernel prices such a state's closure in ONE round
//   (cost of an entry = cost of its parent + weight, in path order, every prefix below the cutoff)
//   instead of one round per epsilon hop.
// FUSED closures (GraphDev::fused, best-path decoders): the epsilon closure folded into the expansion.
//   Behind a state's emitting arcs sit its PSEUDO ARCS, header.z of them: one per (emitting arc i ->
//   s', path p of s''s whole epsilon closure) = {ll_col of arc i, index of p in pseudo[], weight bits of
//   arc i, row of p's end state}.  pseudo[p] = {last arc of the path (row index) | flags of its end
//   state, parent path or -1, weight bits of the last arc, hops}.  Expanding a pseudo arc prices
//   ProcessEmitting's arrival at s' and ProcessNonemitting's walk from it in one go:
//   ((cur + ac) + w_i) + w_1 + ... + w_k, summed in path order; the candidate it yields is an epsilon
//   arrival (kEpsRec: loses an exact cost tie to an emitting arc; its token's predecessor is found at
//   traceback, kPrevUnresolved).  FindOrAddToken is a minimum and float addition is monotone, so
//   taking the minimum over the arrivals of EVERY candidate at s' equals the reference's walk from the
//   final cost of the token at s'.  Needs: no epsilon cycle, closures of <= 48 paths and <=
//   kPseudoDepthMax hops, no negative epsilon weight (a path then never costs less than its prefix, so
//   one cutoff test on the arrival stands for the test at every hop).  Lattice decoders use the fused rows too
//   (the epsilon arrivals come through the insert launch; one flat pass then lists the frame's epsilon links);
//   biglm decoders, and graphs that do not qualify, run the separate closure pass instead and ignore the pseudo arcs.
// arc_ilabel[], arc_olabel[], arc_src[] (source row | bit 31 for an epsilon arc): cold arrays in
//   the same index space.  eps_target_state[k] = row of epsilon-target ordinal k.
struct GraphDev {
  const int4 *arcs;  // rows[]
  const int32_t *arc_ilabel;
  const int32_t *arc_olabel;
  const int32_t *arc_src;
  const int32_t *eps_target_state;
  const int4 *eps_flat;
  const int4 *pseudo;
  const float *pseudo_w;   // [paths][kPseudoDepthMax]: a path's epsilon weights root to leaf (paths of three hops or more read them)
  int32_t fused;
  int32_t col_mask;     // log-likelihood column of an arc = its first word & col_mask (kColMask where the degree codes ride above it)
  int32_t degcode;      // the arcs of the fused rows carry the degree code of their target state (below)
  int32_t start, final_state, n_states, n_arcs;
  uint32_t start_eps;   // next_eps word of the start state
  int32_t n_eps_targets;
};

// ---- LM automaton in HBM (biglm, BASELINE configs[3]): the reference's Fsa (newlm/arpa2fsa.h:216-247) ----
// st[s]    {first arc, arc count, back-off weight bits, back-off state}
// words[a] word id of arc a (the arcs of a state are word-id sorted: binary search, arpa2fsa.h:194-210;
//          state 0, the empty history, is indexed directly by word id, arpa2fsa.cc:253-254)
// wt[a]    {weight bits, destination state}
// Weights are natural-log probabilities, already rescaled (the old LM by -1).
struct LmDev {
  const int4 *st;
  const int32_t *words;
  const int2 *wt;
  int32_t n_states, n_arcs;
  int32_t bos, eos;
  int32_t start;       // ComposeArpaLm::Start(): the state after <s> (compose-arpalm.cc:5-13)
  int32_t start_arcs;  // arcs of state 0 = word ids it can be asked for
  // arcs of every state but the empty history as ONE open-addressed table {state, word, weight bits, destination} (empty: state
  // -1): Fsa::GetArc's binary search over a state's word-sorted arcs (arpa2fsa.h:194-210) is three to five dependent loads,
  // a probe of this table is one -- and an LM step is a chain of such look-ups (back-off by back-off, LM by LM) that a whole
  // wavefront waits for.  Same arcs, same answers.
  const int4 *hash;
  uint32_t hmask;      // table size - 1 (a power of two, at most a quarter full)
};
__host__ __device__ inline uint32_t lm_hash(int32_t state, int32_t word) {
  uint32_t h = (uint32_t)state * 0x9E3779B1u ^ ((uint32_t)word + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15;
  return h * 0x2C1B3C6Du;
}

#if defined(__HIPCC__)
// Fsa::GetArc (newlm/arpa2fsa.cc:244-262): the arc of LM state `id` for `word`, false if the state
// has none (the caller backs off).  State 0 (empty history) is indexed by word id directly
// (SearchStartArc, arpa2fsa.h:211-214; wfst_decoder_create_biglm checks the graph's labels against
// its arc count); the others by binary search over their word-id sorted arcs (SearchArc, :194-210).
__device__ __forceinline__ bool fsa_getarc(const LmDev &L, int id, int word, float *w, int *to) {
  if (id == 0) {   // (state 0's arcs start the arc array: wfst_lm_from_arrays)
    const int2 x = L.wt[word];
    *w = __int_as_float(x.x);
    *to = x.y;
    return true;
  }
  // every other state: one probe of the LM's (state, word) table instead of a binary search over the state's arcs
  uint32_t slot = lm_hash(id, word) & L.hmask;
  for (;;) {
    const int4 e = L.hash[slot];
    if (e.x == id && e.y == word) { *w = __int_as_float(e.z); *to = e.w; return true; }
    if (e.x < 0) return false;
    slot = (slot + 1) & L.hmask;
  }
}
// ComposeArpaLm::GetArc (newlm/compose-arpalm.cc:52-70): back off until the word is found; the cost
// is minus the sum of the back-off weights and the arc weight, summed in that order.
__device__ __forceinline__ void lm_getarc(const LmDev &L, int s, int word, int *next, float *value1) {
  float weight = 0.0f, w_arc = 0.0f;
  int to = 0;
  while (!fsa_getarc(L, s, word, &w_arc, &to)) {
    const int4 st = L.st[s];
    w_arc = __int_as_float(st.z);
    s = st.w;
    weight += w_arc;
  }
  weight += w_arc;
  *value1 = -1 * weight;
  *next = to;
}
// ComposeArpaLm::Final (compose-arpalm.cc:15-29)
__device__ __forceinline__ float lm_final_cost(const LmDev &L, int s) {
  int next;
  float v;
  lm_getarc(L, s, L.eos, &next, &v);
  return v;
}
#endif

constexpr int kEpsBits = 12;
constexpr int kFlatMax = 4;  // paths of a flattened epsilon closure (3 bits in the header word)
constexpr uint32_t kEpsMask = (1u << kEpsBits) - 1;
constexpr uint32_t kFlagOutEps = 0x80000000u;     // state has outgoing input-epsilon arcs
constexpr uint32_t kFlagEpsTarget = 0x40000000u;  // some input-epsilon arc enters the state
constexpr uint32_t kFlagMask = kFlagOutEps | kFlagEpsTarget;
constexpr uint32_t kEpsWon = 0x80000000u;         // in a packed eps-table value: won by an epsilon arc
constexpr uint32_t kEpsOutBit = 0x40000000u;      // in a packed eps-table value: the state has epsilon arcs out
// token/record flag bits from an arc's next_eps word
// DEGREE CODE (fused best-path decoders): what a token needs to find its arcs WITHOUT reading its row header first --
// the state's epsilon arcs (2 bits, <= 3), emitting arcs (4 bits, <= 15) and pseudo arcs (5 bits, <= 31); kCodeUnknown
// where a count does not fit (the expansion then reads the header, as it always did).  On the graph side it rides in the
// arc's first word above the log-likelihood column (graphs whose ilabels stay below 2^20); in a candidate record -- and so,
// verbatim, in the token the insert kernel writes -- in bits nothing else uses: its two low bits in bits 30..31 of the arc
// word (the closure pass's flags, idle without a closure pass), the rest above the source-token index (arenas of up to
// 2^22 tokens leave 9 bits), or, for an epsilon arrival, in the unresolved-backpointer sentinel: z = kPrevUnresolved - rest.
// The row-header loads it saves are a fifth of the expansion (10 of 51 us per launch, measured by replay).
constexpr int kColBits = 20;
constexpr int32_t kColMask = (1 << kColBits) - 1;
constexpr uint32_t kCodeUnknown = 0x7FFu;
constexpr int kCodeRestBits = 9;   // code >> 2
__host__ __device__ inline uint32_t pack_code(uint32_t n_eps, uint32_t n_emit, uint32_t n_pseudo) {
  if (n_eps > 3u || n_emit > 15u || n_pseudo > 31u) return kCodeUnknown;
  const uint32_t c = n_eps | (n_emit << 2) | (n_pseudo << 6);
  return c == kCodeUnknown ? kCodeUnknown : c;   // (3, 15, 31) itself reads as unknown: the header path is always right
}

__host__ __device__ inline uint32_t flags_of(uint32_t next_eps) {
  return (next_eps & kFlagOutEps) | ((next_eps & 0x7FFFFFFFu) ? kFlagEpsTarget : 0u);
}
constexpr uint32_t kEpsRec = 0x20000000u;          // in a candidate record / token: an epsilon arrival (fused closures)
constexpr uint32_t kArcMask = ~(kFlagMask | kEpsRec);  // arc (row) indices are < 2^29
constexpr uint32_t kNoArc = kArcMask;              // "no arc" (root token), flags kept beside it
constexpr int kPseudoDepthMax = 8;                 // hops of a fused closure path
constexpr int32_t kEmptyKey = -1;
constexpr int32_t kPrevUnresolved = -3;  // token won by an epsilon arc: backpointer found at traceback
constexpr unsigned long long kEmptyVal = ~0ull;

// error bits (ChanCtl::error)
constexpr int kErrTableFull = 1, kErrArenaFull = 2, kErrFrontierFull = 4, kErrWorklistFull = 8,
              kErrFramesFull = 16, kErrBucketFull = 32, kErrLinksFull = 64, kErrPairsFull = 128,
              kErrInternal = 256;   // an invariant of the kernels did not hold (never expected; reported, not papered over)

// ---- per-channel control block (one 128-byte line each) ----------------------------------
struct __attribute__((aligned(128))) ChanCtl {
  int32_t n_decoded;     // NumFramesDecoded()
  int32_t front_begin;   // arena index of the current frontier's first token
  int32_t front_count;   // tokens in the current frontier
  int32_t active;        // this frame step processes the channel
  uint32_t bound;        // orderable next_cutoff, tightened during expansion (atomicMin)
  float cur_cutoff;      // GetCutoff() result for the frame being expanded
  float adaptive_beam;
  int32_t peak_tokens;
  unsigned long long best_next;  // min (orderable cost << 32 | arena index) over the new frame (best_row decoders: | graph row; staged
                                 // best_row decoders: set by the EXPANSION -- the cheapest candidate is the cheapest token)
  int32_t eps_occ;       // occupied slots of the epsilon table   } one 8-byte word: the insert
  int32_t wl_n;          // epsilon-closure seeds queued          } workgroups bump both at once
  int32_t error;         // sticky kErr* bits
  int32_t finalized;
  // one 8-byte word: two-launch decoders allocate an item's tokens (add to the low half) and count the item out (subtract from
  // the high half) with 64-bit atomics on it, so the workgroup that counts the LAST item out learns the frame's token count
  // from the same answer (kFrameErrBit: an item of this frame could not write its tokens)
  int32_t new_count;     // tokens of the frame being built (atomicAdd by the insert workgroups)
  int32_t items_left;    // fused best-path decoders: insert work items of the frame not finished yet (the workgroup that
                         // finishes the last one closes the frame and prepares the next: frame_boundary_fused)
  unsigned long long cnt_N, cnt_E, cnt_Z, cnt_tok, cnt_rec;  // work counters since init
  int32_t link_count;    // lattice mode: forward links recorded so far (atomicAdd)
  int32_t lat_arcs;      // lattice mode, after lattice_prune_kernel: surviving links in lat_arcs[]
  int32_t lat_toks;      //   "   surviving tokens in lat_toks[]
  int32_t tiles_left;    // expansion tiles of the frame not finished yet (the last one plans the insert items)
  union {
    int32_t pair_count;  // biglm: LM pair states interned since InitDecoding (atomicAdd)
    int32_t stores_left; // two-launch decoders: insert items of the frame whose token stores have not all landed yet (counted down,
                         // behind every wave's drain, AFTER the item has counted itself out of items_left: the frame boundary waits
                         // for it only where GetCutoff has to look at the frame's tokens)
  };
  int32_t pruned_upto;   // lattice mode: NumFramesDecoded() at the last back-pruning pass (frames below hold extras)
};
static_assert(offsetof(ChanCtl, new_count) % 8 == 0 && offsetof(ChanCtl, items_left) == offsetof(ChanCtl, new_count) + 4, "the {new_count, items_left} word");
constexpr int32_t kRiskyBit = 1 << 29;   // in items_left (plan_channel): the frame's boundary may have to read the frame's tokens
constexpr unsigned long long kFrameErrBit = 1ull << 62;   // in the {new_count, items_left} word (items_left stays below 2^16)
static_assert(sizeof(ChanCtl) == 128, "ChanCtl must be one 128-byte line");

// One 256/512-token tile of a channel's frontier, listed by prep_frame for the expansion: everything
// a workgroup needs to start, in one 32-byte load (instead of tile -> channel -> control block ->
// log-likelihood pointer, three dependent loads).
struct __attribute__((aligned(32))) TileDesc {
  int32_t chan;
  int32_t tok_begin;    // arena index of the tile's first token
  int32_t tok_count;    // tokens in the tile (<= tile size)
  float cutoff;         // GetCutoff() of the frame
  float adaptive_beam;
  int32_t pad;
  const float *llrow;   // log-likelihood row of the frame being decoded
};
static_assert(sizeof(TileDesc) == 32, "TileDesc is one 32-byte load");

// one arc of the pruned lattice (lattice mode)
struct __attribute__((aligned(32))) LatArc {
  int32_t src_tok, dst_tok;  // arena indices
  int32_t ilabel, olabel;
  float graph, acoustic;
  int32_t src_frame, is_eps;
};
static_assert(sizeof(LatArc) == 32, "LatArc is two 16-byte stores");

// per channel-group frame counters, double buffered by step parity
struct FrameCtl {
  int32_t total_tiles[2];  // tiles published by prep_frame for the step of that parity
  int32_t ticket[2];       // dynamic tile dispenser of expand_kernel
  int32_t n_items[2];      // insert work items listed by plan_channel (expand's last tile of a channel)
  int32_t item_ticket[2];  // dynamic item dispenser of insert_kernel
  int32_t n_small[2];      // of those items, the light ones: listed from the END of items[] so that the
                           // insert workgroups take the heavy items first (longest-first keeps the tail short)
  int32_t pad[6];
};

// ---- decoder (batch of channels) ---------------------------------------------------------
// Per channel c:
//   tok[c][arena_cap]           int4 {state, cost bits, prev token (arena index, -1 root),
//                               winning arc | flags of the state}; frame f = tokens
//                               [frame_off[f], frame_off[f+1]); the newest frame is the frontier
//   frame_off[c][max_frames+2], cutoff_hist[c][max_frames+2] (cutoff of frame f's closure)
//   bucket[c][P][bucket_cap]    int4 candidate records {nextstate, cost bits, source token,
//                               arc | flags(nextstate)}, partition = top bits of hash(nextstate)
//   bucket_cnt[c][P]
//   eps table, direct mapped: eps_vals[c][K] (orderable cost << 32 | kEpsWon? | arc) and
//                               eps_toki[c][K] indexed by the epsilon-target ordinal of a state,
//                               eps_occ_list[c][...] = ordinals touched this frame (for clearing)
//   worklist[c][2][wl_cap]      epsilon-closure frontiers (slots of the eps table)
struct DecoderDev {
  GraphDev g;
  ChanCtl *ctl;
  int4 *tok;
  int32_t *frame_off;
  float *cutoff_hist;
  int4 *bucket;
  int32_t *bucket_cnt;
  int32_t prune_raw_min; // ... for channels with at least this many never-priced links (wfst_options.debug 0x800: 0, every channel)
  int32_t closure_slabs; // lattice decoders on the fused rows: workgroups per channel of a closure launch (they share the frame's epsilon links)
  int32_t link_delta;    // lattice decoders on the fused rows: a forward link's 4th word is (link cost - cost of its destination token) -- the float the
                         // back-pruning computes from the two anyway (base-inl.h:524-526), known when the link is recorded: a link is priced from the
                         // destination's extra alone; elsewhere (iterated closures: a token's cost may still improve) the link cost itself
  int32_t prune_raw;     // lattice mode: a running back-pruning pass prices its raw frames with several workgroups per channel (wfst_kernels.hip: lattice_prune_raw_*)
  int32_t *prune_par;    // [c][kPruneParInts]: lattice mode -- what a running back-pruning pass hands to its compaction launches (wfst_kernels.hip: kPrParInts)
  int32_t *emit_cnt;     // [c][32] (a line each): lattice mode on the fused rows -- entries of the channel's emitter list (the tokens of the
                         // frame being built that have epsilon arcs out: listed by the insert launch in the channel's worklist space,
                         // read and reset by the closure launch's epsilon_links)
  unsigned long long *eps_vals;
  int32_t *eps_toki;
  int32_t *eps_occ_list;        // [c][wl_cap] ordinals touched this frame
  int32_t *eps_won_list;        // [c][wl_cap] ordinals whose token an epsilon arc won this frame
  int4 *worklist;               // [c][2][wl_cap] {eps-table slot, state, cost bits, 0}
  // lattice mode (wfst_limits.lattice_links > 0): every forward link the reference would hold after
  // FinalizeDecoding is among links[c][0..link_count): {source token, destination token, arc,
  // cost bits of (source cost + acoustic) + graph}; segment k = links whose destination is on
  // frame k = [link_off[k], link_off[k+1]): the emitting links from frame k-1 first
  // [link_off[k], link_mid[k]), then the epsilon links inside frame k [link_mid[k], link_off[k+1]).
  // extra[c][token] = {orderable extra_cost (base-inl.h:482-572), cost bits of the token}, kept up to
  // date by the back-pruning passes (prune_pass: every prune_interval frames and at FinalizeDecoding),
  // which also REMOVE the dead tokens and links and move the survivors down, so that tok[] / links[]
  // hold the surviving history plus the raw frames since the last pass.  lattice_emit_kernel resolves
  // what is alive into lat_arcs[c][0..ctl.lat_arcs) and lat_toks[c][0..ctl.lat_toks) for GetRawLattice.
  int4 *links;
  int32_t *link_off, *link_mid;  // [c][max_frames+3]
  uint2 *extra;
  int32_t *remap;               // [c][arena_cap] scratch of the back-pruning passes (previous extras, then new indices)
  LatArc *lat_arcs;
  unsigned long long *lat_stats;  // [c][4] since InitDecoding: {forward links recorded, links priced by the back-pruning walks (one per
                                  // link and sweep), tokens priced by them, tokens + links scanned by the compactions | moved << 32 ... see wfst_decoder_get_lattice_stats}
  int4 *lat_toks;               // {arena index, graph state id, cost bits, frame | final << 30}
  int64_t link_cap;
  int32_t lat_arc_cap, lat_tok_cap;
  int32_t lattice;
  FrameCtl *fctl;               // [n_groups]
  TileDesc *tiles;              // [n_groups][tile_cap] tiles of the coming frame
  int32_t tile_cap;
  int32_t tok_idx_bits;         // bits of a token's backpointer that hold the arena index (31: all of them; less: a degree code above)
  int32_t degcode;              // tokens and records carry degree codes (fused rows, packed graph, arena small enough)
  int32_t *items;               // [n_groups][item_cap] channel << 16 | first partition << 8 | group size
  int32_t item_cap;
  int32_t *item_pref;           // [n_groups][item_cap][64]: entry i of an item = records in its first i + 1 buckets (plan_channel writes the
                                // prefix it has in registers; the insert workgroup gets it with the item instead of loading the counters behind it)
  const float *const *ll_base;  // [n_channels] device pointers to row 0 of each utterance matrix
  int32_t n_channels;
  int32_t stride;               // floats per log-likelihood row
  int32_t n_part, log2part;     // hash partitions per channel (power of two, <= 64)
  int32_t lds_slots, log2lds;   // LDS hash slots per partition workgroup (4096 or 8192)
  int32_t bucket_cap;           // records per bucket
  int32_t joint_max;            // insert: most records a group of partitions may hold to share one workgroup
  int32_t ecap;                 // = n_eps_targets (entries of eps_vals / eps_toki per channel)
  int32_t max_tok;              // tokens per frame
  int32_t wl_cap;
  int32_t max_frames;
  int64_t arena_cap;
  int64_t gc_reserve;           // best-path decoders: the token collection's mark lies this far below arena_cap (gc_base_mark)
  // config (LatticeFasterDecoderConfig)
  float beam, lattice_beam, beam_delta, prune_scale;
  int32_t max_active, min_active, prune_interval;
  // biglm mode (wfst_decoder_create_biglm): a token is identified by (graph row, LM pair state), the
  // reference's 64-bit PairId (my-decoder/online-decoder-mempool-base-biglm.h:77-90).
  //   pair_keys[c][pair_cap]  the channel's LM pair-state table, DiffArpaLm's _state_map/_state_vec
  //                           (newlm/diff-lm.h:92-103) as one open-addressed array: slot = pair id,
  //                           value = old-LM state | new-LM state << 32; emptied by InitDecoding
  //                           (DiffArpaLm::Reset)
  //   tok_lm[c][arena_cap]    pair id of each token;  bucket_lm[c][P][bucket_cap] of each candidate record
  //   eps_keys[c][ecap]       the epsilon table is HASHED in this mode (the state alone no longer
  //                           identifies a token): open-addressed keys (row | pair << 32) beside
  //                           eps_vals / eps_toki, ecap a power of two
  int32_t fused;  // the graph's fused closures are in use (best-path, non-biglm decoder on a graph that has them)
  // Two launches per frame (fused best-path decoders whose max_active / min_active can never bind: GetCutoff is then
  // best + beam and needs no look at the tokens): the insert workgroup that finishes a channel's last work item closes the
  // frame and prepares the next one (frame_boundary_fused), so the third launch of a frame disappears -- except on every
  // gc_stride-th frame of an advance call, which runs the classic three launches: the closure kernel there checks whether the
  // token arena wants collecting (the arena's reserve covers gc_stride + 1 frames at the per-frame limit).
  // best_row: ChanCtl::best_next carries the best token's graph ROW in its low word instead of its arena index (all the
  // seeding of next_cutoff needs, with no token read behind other workgroups' stores; ties on the best cost then go to the
  // lowest row: deterministic, where the arena order is not).
  int32_t two_launch, gc_stride, best_row;
  // seed_tiles (best_row decoders): next_cutoff's seed from the best token's arcs (base-inl.h:282-300) is computed by the
  // EXPANSION -- one extra tile per channel, listed first (TileDesc with tok_count 0: tok_begin = the best token's row, cutoff
  // = its cost) -- instead of by the frame boundary, whose serial tail it was three dependent round trips of; the tiles read
  // next_cutoff afresh every round, so the seed (and every other tile's tightening) reaches them as soon as it lands.
  int32_t seed_tiles;
  int32_t staged;   // fused (non-biglm) decoders: expand_kernel_staged (the tile's arcs staged in LDS by gather DMA) instead of expand_kernel_fused
  int32_t soft_limit;   // fused best-path decoders: max_tokens_per_frame is not a capacity but a max_active -- a frame may hold more tokens
                        // (while the arena and the candidate buckets take them); GetCutoff then tightens to the limit-th cheapest
  int32_t *degraded;    // [c] frames of the utterance on which that happened (wfst_decoder_get_degraded_frames)
  int32_t st_tile_tokens;   // staged expansion: frontier tokens per tile (wfst_options.tile_tokens; at most the kernel's 256 threads)
  int32_t best_exp; // staged best_row decoders: ChanCtl::best_next is set by the expansion (the insert launch does not look for the best token)
  int32_t ll_row;   // staged decoders: the tile's whole log-likelihood row is staged in LDS too (rows of at most 3072 columns, a multiple of
                    // four, 16-byte aligned: set by wfst_decoder_advance from the matrices it is handed); 0: one 4-byte gather per arc slot
  int32_t big;
  LmDev lm_old, lm_new;
  unsigned long long *pair_keys;
  int32_t *pair_list;           // [n_channels][pair_cap] the table slots claimed since InitDecoding, in claiming order (clear_pairs_kernel)
  int32_t pair_cap;             // power of two
  int32_t *tok_lm;
  int32_t *bucket_lm;
  unsigned long long *eps_keys;
  unsigned long long *dbg_t;  // [64] phase timers (WFST_DBG & 32): sums, maxima, counts
  int32_t dbg;  // WFST_DBG ablation bits (timing experiments only; results are wrong when set)
};

// ---- n-best (wfst_nbest.hip) ---------------------------------------------------------------
struct NbEntry {            // one partial path of a lattice state's k-best list
  float tot, lm;            // sum of (graph + acoustic), sum of graph, in path order
  unsigned long long hash;  // of the word sequence so far
  int32_t prev;             // (source lattice state << 4) | entry, -1 at the start state
  int32_t word;             // olabel of the arc that led here (0: none)
};
static_assert(sizeof(NbEntry) == 24, "NbEntry");
struct NbestDev {
  NbEntry *list;            // [c][tok_cap][K]
  int32_t *scratch;         // [c][scratch_ints]
  int32_t tok_cap, arc_cap, K;
  int64_t scratch_ints;     // 4 * arc_cap (in-arc records) + 3 * tok_cap + 1 + 3 * (max_frames + 2), rounded up to a multiple of 4
  int32_t *out_n;           // [cnt]      paths found, or -1: the lattice exceeds tok_cap / arc_cap
  int32_t *out_nwords;      // [cnt][n]
  int32_t *out_words;       // [cnt][n][max_words]
  float *out_tot, *out_lm;  // [cnt][n]
  int32_t n, max_words;
};
void launch_nbest(const DecoderDev &D, const NbestDev &N, const int32_t *chan_list_dev, int cnt, hipStream_t s);

// ---- lattice determinization (wfst_determinize.hip / wfst_determinize.h) -------------------------------
struct DetCaps;
}  // namespace wfst
#include "wfst_determinize.h"
namespace wfst {
constexpr int kPruneParInts = 64;   // ints of a channel's block of DecoderDev::prune_par
constexpr int kClSlabWord = 60;     // ... of which [60, 62), one 64-bit word: the closure launch's meeting of a channel's workgroups (finalize_frame)

struct DetDev {
  int32_t *ws;                  // [c][words_per_channel]: the lattice's CSR, then the determinizer's workspace
  int64_t words_per_channel;
  int32_t raw_states_cap, raw_arcs_cap;   // largest raw lattice taken
  DetCaps caps;
  int32_t *result;              // [cnt][4] {states, arcs, status, determinized states proper}
  int4 *out_a;                  // [cnt][out_cap] {src, dst, word, is-final-arc}
  float2 *out_w;                // [cnt][out_cap] {graph, acoustic}
  int32_t out_cap;
};
void launch_determinize(const DecoderDev &D, const DetDev &X, const int32_t *chan_list_dev, int cnt, hipStream_t s, int phase = 0);   // phase: wfst_determinize.hip
void launch_det_pack(const DetDev &X, int cnt, int4 *pack_a, float2 *pack_w, int64_t pack_cap, hipStream_t s);

// ---- n cheapest paths of a determinized / rescored lattice (wfst_nbest.hip: nbest_paths_kernel) --------------------------------
struct NbPathEntry { float cost; int32_t arc, rank, pad; };   // a partial path: its cost, the arc it arrives by (-1: the start), its rank in that arc's source list
struct NbPathsDev {
  const int4 *a;                // the lattice's arcs {src, dst, word, -} ...
  const float2 *w;              // ... and their {graph, acoustic} costs
  const int32_t *res;           // {states, arcs, status, determinized states proper} as determinize_kernel / compose2_kernel leave it
  const int32_t *fin;           // final flag per state, or null: the states from res[3] on
  int64_t in_stride, fin_stride;   // a batch: slot b's lattice sits at a / w + b * in_stride, res + 4 b, fin + b * fin_stride; its workspace,
                                // lists and outputs at b * ws_ints, b * list_cap, out + 4 b, out_off + b (n + 1), out_tot + b n, out_arcs + b out_cap
  int32_t n;                    // paths wanted (<= 4096)
  int32_t *ws;                  // scratch: 7 states + 4 max(states, arcs) + arcs + 16 ints
  int64_t ws_ints;
  NbPathEntry *lists;           // the states' lists, packed
  int64_t list_cap;
  int32_t *out;                 // {paths found, arcs on them, status (0 ok, 1 a capacity, 2 no input, 3 cyclic input), -}
  int32_t *out_off;             // [n + 1] first arc of each path in out_arcs
  float *out_tot;               // [n] cost of each path
  int32_t *out_arcs;            // arc indices (into a / w), path after path, front to back
  int32_t out_cap;
};
void launch_nbest_paths(const NbPathsDev &P, int n_slots, hipStream_t s);

// ---- second-pass LM composition on determinized lattices (wfst_compose.hip) ------------------------------------
// ComposeLattice (newfst/compose-lat-inl.h:15-130) of the determinized lattice of workspace slot 0 (DetDev::out_a / out_w, as
// determinize_kernel left it) with ComposeArpaLm(lm1), then of the result with ComposeArpaLm(lm2) -- what the service's GetLattice
// does under --use-second (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:53-78) -- each followed by Connect.
struct CmpDev {
  int32_t *ws;            // workspace (ints): see wfst_compose.hip
  int64_t ws_ints;
  int32_t pair_cap;       // composed states per pass (hash slots = 2 x)
  int32_t arc_cap;        // arcs per pass
  int32_t *result;        // {states, arcs, status (0 ok, 1 capacity), -}
  int4 *out_a;            // [arc_cap] {src, dst, olabel, final flag of dst}
  float2 *out_w;          // [arc_cap] {graph, acoustic}
  int32_t *out_fin;       // [pair_cap] final flag per state
};
void launch_compose2(const DetDev &X, const CmpDev &Y, const LmDev &lm1, const LmDev &lm2, int n_slots, hipStream_t s);

// launch wrappers (wfst_kernels.hip)
void launch_init(const DecoderDev &D, const int32_t *chan_list_dev, int n, hipStream_t s);
void launch_expand(const DecoderDev &D, int group, int par, int n_workgroups, hipStream_t s);
// boundary: 0 = the closure kernel follows (classic frame); 1 = the insert launch closes the frame and prepares the next
// (two-launch frame); 2 = closes the frame only (last frame of an advance call)
void launch_insert(const DecoderDev &D, int chan_off, int chan_cnt, const int32_t *target_dev, int boundary, int group, int par,
                   int n_workgroups, hipStream_t s);
// after_insert: the launch closes a frame an insert launch has just built (DecoderDev::closure_slabs workgroups per channel then)
void launch_closure(const DecoderDev &D, int chan_off, int chan_cnt, const int32_t *target_dev, int do_prep,
                    int group, int par, hipStream_t s, int after_insert);
// stage: -1 = the step's four launches; 0..3 = one of them (raw frames, walk, flag sweeps, moves): a profiled step times them one by one
void launch_lattice_prune_step(const DecoderDev &D, int chan_off, int chan_cnt, const int32_t *target_dev, int group, int par, hipStream_t s, int stage);
void launch_set_finalized(const DecoderDev &D, const int32_t *chan_list_dev, int n, hipStream_t s);
void launch_delay(int microseconds, hipStream_t s);
void launch_lattice_prune(const DecoderDev &D, const int32_t *chan_list_dev, int n, hipStream_t s);
void launch_lattice_emit(const DecoderDev &D, const int32_t *chan_list_dev, int n, int use_final, hipStream_t s);
void launch_best_path(const DecoderDev &D, const int32_t *chan_list_dev, int n, int use_final,
                      int cap, int32_t *ilabel, int32_t *olabel, float *graph, float *ac,
                      int32_t *n_hops, int32_t *chain_scratch, hipStream_t s);

}  // namespace wfst
#endif
