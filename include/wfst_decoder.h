/*
 * wfst_decoder.h -- C ABI of the MI355X-native batched WFST token-passing decoder.
 *
 * This is the drop-in boundary for ONE path of datemoon/ASR-decoder: the frame-synchronous
 * token-passing search of src/my-decoder (ProcessEmitting / ProcessNonemitting / beam pruning
 * over the flat HCLG of src/newfst), precomputed log-likelihoods in; best path, raw lattice and
 * n-best out.
 * Plain pointers and sizes only; no C++ or torch types.  Every entry point names the reference
 * interface it replaces (paths relative to the reference's src/).
 *
 * Threading: one wfst_decoder per host thread (like one reference decoder object per worker
 * thread, v2-asr/v2-asr-work-thread.h:66); a wfst_graph is immutable after creation and may be
 * shared by any number of decoders on the same device (like the shared read-only Fst,
 * kaldi-nnet3/kaldi-online-nnet3-my-decoder.h:121).
 *
 * Errors: every function returning int returns WFST_OK (0) or a negative WFST_E_* code;
 * wfst_last_error() gives the message of the calling thread's last failure.  There is no CPU
 * fallback: without a usable HIP device every call that needs one fails with WFST_E_DEVICE.
 */
#ifndef WFST_DECODER_H_
#define WFST_DECODER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WFST_OK 0
#define WFST_E_ARG (-1)      /* bad argument                                              */
#define WFST_E_IO (-2)       /* graph file unreadable / truncated                         */
#define WFST_E_DEVICE (-3)   /* HIP error (no device, out of memory, launch failure)      */
#define WFST_E_CAPACITY (-4) /* a per-channel limit of wfst_limits was exceeded on device */
#define WFST_E_STATE (-5)    /* call sequence violated (e.g. advance before init)         */
#define WFST_E_FORMAT (-6)   /* graph violates a layout limit (see wfst_graph_from_arrays) */

typedef struct wfst_graph wfst_graph;     /* HCLG resident in HBM; replaces `Fst` (newfst/optimize-fst.h:53-307) */
typedef struct wfst_decoder wfst_decoder; /* a batch of decoding channels; one channel replaces one
                                             OnlineLatticeDecoderMempool (my-decoder/online-decoder-mempool-base.h:77) */
typedef struct wfst_lm wfst_lm;           /* a back-off n-gram LM automaton resident in HBM; replaces `ArpaLm`
                                             (newlm/arpa2fsa.h:249-441) for the biglm decoder */

/* Field-for-field LatticeFasterDecoderConfig (my-decoder/lattice-faster-decoder-conf.h:21-44);
 * wfst_config_default() fills the reference defaults (conf.h:35-44).  hash_ratio is accepted for
 * compatibility (the device hash tables are sized by wfst_limits and wfst_options).  Best-path decoders keep
 * no forward-link lists to back-prune: prune_interval and lattice_beam there only decide which of several
 * parallel arcs GetBestPath reports, exactly as in the reference.  In LATTICE MODE
 * (wfst_limits.lattice_links > 0) the reference's PruneActiveTokens runs on the device: every prune_interval
 * frames the recorded links are pruned backwards by lattice_beam, walking back while a frame's extra costs
 * move by more than lattice_beam * prune_scale (base-inl.h:439-607, 660-661), and the token arena and link
 * store are compacted, so that memory stays bounded whatever the utterance length; FinalizeDecoding prunes
 * with delta 0 like the reference's PruneActiveTokens(0) (base-inl.h:541-607). */
typedef struct wfst_config {
  float beam;
  int32_t max_active;
  int32_t min_active;
  float lattice_beam;
  int32_t prune_interval;
  float beam_delta;
  float hash_ratio;
  float prune_scale;
} wfst_config;

/* Per-channel device capacities (0 = default).  Exceeding one makes the affected call return
 * WFST_E_CAPACITY -- with ONE exception, max_tokens_per_frame of a best-path decoder on the fused rows,
 * which degrades the search instead (below) and says so through wfst_decoder_get_degraded_frames; the
 * C++ mirror's GetBestPath(s) warn when that count is not 0.  Nothing is dropped without one of the two. */
typedef struct wfst_limits {
  int32_t max_frames;           /* frames per utterance                     (default 4096)    */
  int32_t max_tokens_per_frame; /* distinct states reached in one frame     (default 65536, or 4 x a finite max_active, at most 262144).
                                   Best-path decoders on the fused rows do not fail at it: it acts as a max_active
                                   (the frame goes on from its limit-th cheapest token; wfst_decoder_get_degraded_frames
                                   counts such frames).  Such a frame may HOLD several times the limit (the arrivals of
                                   limit expanded tokens), while the token collection is sized for limit tokens a frame:
                                   an utterance that stays over the limit frame after frame can still exhaust the arena
                                   between two collections and then fails loudly with WFST_E_CAPACITY (arena full) --
                                   size arena_tokens for it or raise the limit.  Lattice and biglm decoders return
                                   WFST_E_CAPACITY at the limit itself */
  int64_t arena_tokens;         /* token arena of one utterance, 16 bytes a token.  BEST-PATH decoders keep every
                                   token of the utterance for the traceback (there are no link lists to prune
                                   them by): an utterance of T frames with n tokens alive per frame needs about
                                   T x n IF nothing is reclaimed; when less than an eighth of the arena (a quarter: two-launch decoders) is left
                                   the decoder collects it (keeps what the frontier's backpointers reach, a percent or two,
                                   and moves it down), so that what the arena must hold is the raw tokens of the
                                   frames between two collections plus the surviving history -- a few hundred
                                   frames' worth is plenty for any utterance length; WFST_E_CAPACITY only if one
                                   collection cannot free half of it.  A collection costs ~5 ms per million
                                   tokens in the arena: size it so that collections are rare.  Arenas of at most 4194304
                                   tokens (the default) leave room in a token's backpointer for its state's degree
                                   code: the expansion then skips the row-header reads (a third of its fetches).  LATTICE-MODE decoders reclaim it every
                                   prune_interval frames (see wfst_config) and need ~3x the tokens
                                   FinalizeDecoding keeps + prune_interval frames of raw tokens.
                                   Default: max_frames x max(256, max_tokens_per_frame / 64), at least 4194304,
                                   i.e. room for the default max_frames at 1024 tokens per frame             */
  int64_t lattice_links;        /* > 0: LATTICE MODE -- record every forward link (capacity per
                                   utterance) so that FinalizeDecoding can prune by lattice_beam and
                                   GetRawLattice can be served; 0 (default): best path only        */
  int64_t lm_pairs;             /* biglm decoders: distinct (old-LM state, new-LM state) pairs one
                                   utterance may reach (default 262144; rounded up to a power of two) */
  int32_t det_raw_states;       /* GetLattice (wfst_decoder_get_determinized_lattice): the largest raw lattice the on-device
                                   determinizer takes, in states (default 65536) ...                                    */
  int32_t det_raw_arcs;         /* ... and arcs (default 2 x det_raw_states).  The determinizer's workspace is allocated by
                                   the first GetLattice call: about 1.2 KB per raw state and lattice (79 MB at the default),
                                   for as many lattices at a time as det_workspace_bytes allows                          */
  int64_t det_workspace_bytes;  /* upper bound of that workspace (default: room for every channel of the decoder, at most an
                                   eighth of the device's memory).  Lattices beyond it are determinized in further launches:
                                   a speed knob, never a refusal (at least one lattice's workspace is always allocated)   */
} wfst_limits;

/* Scheduling choices of a decoder (NULL / wfst_options_default() = the measured defaults).  None of
 * them changes a result bit; they replace what a CPU decoder has no use for.  Values out of range are
 * WFST_E_ARG.  (The library reads no environment variables.) */
typedef struct wfst_options {
  int32_t channel_groups;      /* 1..8: channel groups, each with its own stream and hipGraph; 0 = automatic
                                  (2 groups from 64 channels up, 3 from 96 up, else 1)           (0)    */
  int32_t use_hip_graph;       /* replay the frame loop of an advance call as a hipGraph        (1)    */
  int32_t log2_partitions;     /* 0..6: hash partitions (candidate buckets) per channel; -1 = by the
                                  decoder's kind: 5, lattice decoders 6                          (-1)   */
  int32_t log2_lds_slots;      /* 8..13: LDS hash slots of one insert workgroup                 (12)   */
  int32_t joint_max;           /* records a group of partitions may hold to share a workgroup   (1536) */
  int32_t expand_workgroups;   /* grid of the expansion kernel; 0 = by the decoder's kind: 2048,
                                  lattice decoders 3072                                          (0)    */
  int32_t insert_workgroups;   /* grid of the insert kernel                                     (768)  */
  int32_t upload_slice_frames; /* wfst_decoder_advance_host: frames per upload slice, 0 = copy
                                  everything before decoding                                    (48)   */
  int32_t tile_tokens;         /* 64..256: frontier tokens per tile of the staged expansion     (256)  */
  int32_t debug;               /* kernel phase timers (32 closure / 64 insert / 128 expansion, printed when the
                                  decoder is freed)                                              (0)    */
                               /* (0x1000: lattice decoders run the iterated epsilon-closure pass instead of
                                  the fused rows + flat epsilon-link pass; same results, for comparison.
                                  0x800: a running back-pruning pass prices the never-priced frames of EVERY
                                  channel on several workgroups (by default only channels with 800 k such links
                                  or more: the LDS walk is faster below); same results, for the tests.
                                  0x400: with 0x800, one workgroup of every such channel stays away from the
                                  first meeting -- the pass is abandoned after its 40 ms time-out and done over
                                  by the one-workgroup walk; same results, for the tests.
                                  0x100 / 0x200 / 0x300: a lattice decoder's closure launches run 1 / 2 / 8
                                  workgroups per channel instead of 4 (they share the frame's epsilon links);
                                  same results, for the tests.
                                  Further bits are A/B switches of timing experiments, honoured only by a
                                  library built with -DWFST_AB_SWITCHES) */
} wfst_options;

/* Graph upload choices (NULL / wfst_graph_options_default() = defaults). */
typedef struct wfst_graph_options {
  int32_t row_align_slots;     /* rows are placed so that they touch as few lines of this many 16-byte
                                  slots as possible (8 slots = the 128-byte line a random gather
                                  costs on MI355X); 1 = packed                                  (8)    */
  int32_t flatten_closures;    /* precompute each state's whole epsilon closure (<= 4 paths)     (1)    */
  int32_t fuse_closures;       /* fold the epsilon closures into the expansion (pseudo arcs behind
                                  each state's emitting arcs) where the graph allows: no epsilon
                                  cycle, closures of <= 48 paths and <= 8 hops, no negative epsilon
                                  weight; best-path and lattice decoders then run no separate closure
                                  pass (lattice mode keeps one flat pass that lists the epsilon links) (1)    */
} wfst_graph_options;

/* Original on-disk / in-memory graph records of the reference format. */
typedef struct wfst_arc { int32_t ilabel, olabel; float weight; int32_t nextstate; } wfst_arc;   /* StdArc, newfst/arc.h:17-26 */
typedef struct wfst_state_info { uint32_t num_arcs, niepsilons, noepsilons; } wfst_state_info;   /* Fst::StateInfo, newfst/optimize-fst.h:220-225 */

void wfst_config_default(wfst_config *cfg);
void wfst_options_default(wfst_options *opt);
void wfst_graph_options_default(wfst_graph_options *opt);
const char *wfst_last_error(void);
int wfst_device_count(void);

/* ---- graph ------------------------------------------------------------------------------- */

/* Fst::ReadFst(const char*) (newfst/optimize-fst.h:208-280): reads the flat format
 * {start, final_state, total_states, total_arcs, total_niepsilons, total_noepsilons} int32,
 * StateInfo x S, StdArc x A and uploads it as CSR to `device`.
 * The same call also takes the OpenFst binary files the reference ingests (detected by the OpenFst
 * magic number): a CONST fst (ConstFst<StdArc,int>::Read + Fst(ConstFst), newfst/const-fst.h:118-245,
 * newfst/optimize-fst.h:82-134 -- what the service loads with --constfst=true) and a VECTOR fst
 * (fst_format_convert_tool/read_fst.c:11-187), both with the reference's super-final construction
 * (final weight on a leading <eps>:<eps> arc to one extra state).  WFST_E_FORMAT for embedded
 * symbol tables, non-"standard" arcs or other fst types (the reference readers misread those). */
int wfst_graph_load(const char *path, int device, wfst_graph **out);
int wfst_graph_load_ex(const char *path, int device, const wfst_graph_options *opt, wfst_graph **out);

/* convert_fst IN OUT (fst_format_convert_tool/convert_fst.c:5-27): read a graph file in any format
 * wfst_graph_load takes and write the flat format.  Host only -- needs no device. */
int wfst_graph_convert_file(const char *in_path, const char *flat_out_path);

/* Same from host arrays (what Fst holds after ReadFst or after Fst(ConstFst),
 * newfst/optimize-fst.h:82-134).  Requirements, checked: every state's input-epsilon arcs precede
 * its other arcs (the reference format guarantees it, fst_format_convert_tool/read_fst.c:110-135);
 * <= 4095 input-epsilon arcs and < 2^20 emitting arcs per state (WFST_E_FORMAT otherwise). */
int wfst_graph_from_arrays(int32_t start, int32_t final_state, int32_t n_states, int32_t n_arcs,
                           const wfst_state_info *states, const wfst_arc *arcs, int device,
                           wfst_graph **out);
int wfst_graph_from_arrays_ex(int32_t start, int32_t final_state, int32_t n_states, int32_t n_arcs,
                              const wfst_state_info *states, const wfst_arc *arcs, int device,
                              const wfst_graph_options *opt, wfst_graph **out);

/* Optional transition-id -> pdf map (Kaldi DecodableMatrixScaledMapped as used by
 * kaldi-nnet3bin/kaldi-hclg-my-decoder.cc:107; hmm/transition-model.h:52-61):
 * LogLikelihood(f, ilabel) = loglikes[f][tid2pdf[ilabel]].  tid2pdf has n_tid+1 entries, entry 0
 * unused.  Without a map, column = ilabel (a matrix with NumIndices()+1 columns). */
int wfst_graph_set_tid2pdf(wfst_graph *g, const int32_t *tid2pdf, int32_t n_tid);

int wfst_graph_info(const wfst_graph *g, int32_t *start, int32_t *final_state, int32_t *n_states,
                    int32_t *n_arcs, int64_t *device_bytes);
void wfst_graph_free(wfst_graph *g);

/* ---- language models (biglm: on-the-fly LM rescoring, BASELINE configs[3]) ----------------- */

/* Records of the reference's LM automaton (newlm/arpa2fsa.h:22-40,80-91 and arpa2fsa.cc:62-67). */
typedef struct wfst_lm_state { int32_t arc_num; float backoff_prob; int32_t backoff_id; } wfst_lm_state;
typedef struct wfst_lm_arc { int32_t wordid; float weight; int32_t tostateid; } wfst_lm_arc;

/* ArpaLm::Read (newlm/arpa2fsa.h:355-397 + Fsa::Read, newlm/arpa2fsa.cc:68-176) followed by
 * ArpaLm::Rescale(scale) (arpa2fsa.cc:264-275): reads the binary LM file arpa2fsa-bin writes
 * {i32 bos, eos, unk; u64 orders; i32 count[orders]; i32 n_states; {i32 arc_num, f32 backoff, i32
 * backoff_id} x n_states; i32 n_arcs; {i32 wordid, f32 weight, i32 tostate} x n_arcs} and uploads it
 * to `device`.  The biglm CLI loads the OLD LM (the one compiled into the HCLG) with scale -1 and the
 * NEW one with scale 1 (kaldi-nnet3bin/kaldi-hclg-my-decoder-biglm.cc:55-60).  Checked, WFST_E_FORMAT
 * otherwise: counts consistent with the file size, arcs of a state sorted by word id, state 0 holding
 * arc k for word id k, destinations in range, every back-off chain ending in state 0. */
int wfst_lm_load(const char *path, float scale, int device, wfst_lm **out);
int wfst_lm_from_arrays(int32_t bos, int32_t eos, int32_t unk, int32_t n_states, const wfst_lm_state *states,
                        int32_t n_arcs, const wfst_lm_arc *arcs, float scale, int device, wfst_lm **out);
int wfst_lm_info(const wfst_lm *lm, int32_t *bos, int32_t *eos, int32_t *n_states, int32_t *n_arcs,
                 int32_t *n_words /* arcs of the empty-history state */, int64_t *device_bytes);
void wfst_lm_free(wfst_lm *lm);

/* ---- decoder ----------------------------------------------------------------------------- */

/* Decoder(Fst*, const LatticeFasterDecoderConfig&) (my-decoder/online-decoder-base.h:95,
 * base-inl.h:22-30) for n_channels independent utterance slots.  `hip_stream` is a hipStream_t
 * (NULL = a stream owned by the decoder); all work is enqueued on it. */
int wfst_decoder_create(const wfst_graph *g, const wfst_config *cfg, int32_t n_channels,
                        const wfst_limits *limits, void *hip_stream, wfst_decoder **out);
/* Same with explicit scheduling options (at most 32767 channels per decoder). */
int wfst_decoder_create_ex(const wfst_graph *g, const wfst_config *cfg, int32_t n_channels,
                           const wfst_limits *limits, const wfst_options *options, void *hip_stream,
                           wfst_decoder **out);
/* OnlineLatticeDecoderMempoolBiglm(fst, config, oldlm, newlm) (my-decoder/online-decoder-mempool-base-
 * biglm.h:21-30): every arc with an output label is rescored on the fly with cost_new(word | history) -
 * cost_old(word | history); a token is identified by (graph state, LM pair state) (:77-90).  Both LMs must
 * be on the graph's device and stay alive as long as the decoder.  Each LM is walked from its OWN state
 * (DiffArpaLm with the pair's components; the reference text hands the pair id to both LMs,
 * newlm/diff-lm.h:80,86 -- see DESIGN.md).  With limits->lattice_links > 0 it is the LATTICE decoder it is in the reference
 * (the service asks a `biglm-hclg` decoder for GetRawLattice / GetLattice / n-best, kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:
 * 58,81,97-105): forward links carry graph cost = arc weight + LM difference (:377-392, :448-458), FinalizeDecoding prunes with
 * the LM's final costs (:160-215, 469-560), and -- as in the reference's GetRawLattice (base-inl.h:930-966) -- a raw lattice only
 * FLAGS its final states: the LM's final cost of a final token is not part of the lattice.
 * WFST_E_FORMAT if the graph has an output label the LMs' empty-history state has no arc for (the
 * reference indexes that state without a bounds check, newlm/arpa2fsa.h:211-214). */
int wfst_decoder_create_biglm(const wfst_graph *g, const wfst_config *cfg, int32_t n_channels,
                              const wfst_limits *limits, const wfst_options *options, const wfst_lm *old_lm,
                              const wfst_lm *new_lm, void *hip_stream, wfst_decoder **out);
void wfst_decoder_free(wfst_decoder *d);

/* InitDecoding() (base-inl.h:40-67) for the listed channels (channels == NULL: all). */
int wfst_decoder_init(wfst_decoder *d, const int32_t *channels, int32_t n);

/* AdvanceDecoding(decodable, max_num_frames) (base-inl.h:630-668) for the listed channels at
 * once.  For channel channels[i]: loglikes[i] is a DEVICE pointer to row 0 of the utterance's
 * row-major float32 matrix [frames][stride] of already-scaled log-likelihoods (what
 * LogLikelihood(frame, index) returns, itf/decodable-itf.h:71), n_frames_ready[i] is
 * NumFramesReady(); rows [0, n_frames_ready[i]) must stay valid and unchanged until the
 * channel's next wfst_decoder_init (GetBestPath reads acoustic costs back from them).
 * max_num_frames < 0: decode everything ready.  Returns after the work is ENQUEUED on the stream;
 * use wfst_decoder_sync or any result getter to wait. */
int wfst_decoder_advance(wfst_decoder *d, const int32_t *channels, int32_t n,
                         const float *const *loglikes, const int32_t *n_frames_ready,
                         int32_t stride, int32_t max_num_frames);

/* Same with HOST matrices: rows [NumFramesDecoded, n_frames_ready) are copied into a device
 * history buffer owned by the channel (the shape a DecodableInterface-pulling caller needs).
 * PAGEABLE buffers are consumed when the call returns; a long hand-over is uploaded and decoded in
 * slices, so that the copy of one slice overlaps the search over the previous one.
 * PAGE-LOCKED buffers (wfst_host_alloc; every listed channel's): the rows go up by DMA and the call
 * returns when copies and frames are ENQUEUED, like wfst_decoder_advance -- the rows handed over must
 * stay valid and unchanged until they are decoded (wfst_decoder_sync or any result getter of the
 * channel); a host that hands over chunk after chunk keeps enqueueing while the device decodes.
 * Page-locked matrices of CONSECUTIVE listed channels that lie equally spaced in host memory (one block of utterances; slots of one
 * allocation) and hand over the same frames go up as ONE 2-D copy per run of such channels: list the channels in ascending order.
 * (The device histories of all channels are one allocation, n_channels x the longest hand-over so far, at most max_frames rows.) */
int wfst_decoder_advance_host(wfst_decoder *d, const int32_t *channels, int32_t n,
                              const float *const *loglikes_host, const int32_t *n_frames_ready,
                              int32_t stride, int32_t max_num_frames);

/* Page-locked host memory for the matrices handed to wfst_decoder_advance_host: rows in it go to the device at the link's rate
 * and without a staging copy (pageable rows work too, at a fraction of it).  NULL when the allocation fails.  The host mirror's
 * GpuLatticeDecoder keeps the rows it pulls from a DecodableInterface in such a buffer. */
void *wfst_host_alloc(size_t bytes);
void wfst_host_free(void *p);

/* FinalizeDecoding() (base-inl.h:829-847): marks the channels finalized (afterwards advance is an
 * error and get_best_path requires use_final_probs != 0, as in the reference). */
int wfst_decoder_finalize(wfst_decoder *d, const int32_t *channels, int32_t n);

/* Blocks until everything enqueued so far has run; returns WFST_E_CAPACITY / WFST_E_DEVICE if a
 * channel hit a limit or the device faulted. */
int wfst_decoder_sync(wfst_decoder *d);

/* 1 while work enqueued on the decoder's stream has not finished, 0 when it is idle (never blocks); < 0: error. */
int wfst_decoder_busy(wfst_decoder *d);

/* How many of the decoder's enqueued wfst_decoder_advance[_host] calls -- calls that brought frames; an init or finalize between two
 * of them is not counted -- have not finished yet (0 ... 16: the marks of the newest sixteen calls are looked at; never blocks);
 * < 0: error.  The depth of the device's backlog, for a batching host that hands over chunk after chunk (the host mirror's
 * GpuChannelPool): with three calls outstanding the next one would wait inside the library for the device (the staging sets of the
 * targets and row pointers are used in rotation), and while the device has work behind the call it is running, requests that arrive
 * can still join the next call for nothing; with one or none outstanding it is about to run dry. */
int wfst_decoder_calls_in_flight(wfst_decoder *d);

/* NumFramesDecoded() (my-decoder/online-decoder-base.h:139). */
int wfst_decoder_num_frames_decoded(wfst_decoder *d, int32_t channel);

/* GetBestPath(Lattice*, use_final_probs) (base-inl.h:1071-1094) for every listed channel at once.
 * The linear best-path lattice is returned hop by hop in start->final order, exactly the arcs the
 * reference's Lattice holds (first hop is the (0,0,One) arc of the root token): for channel
 * channels[i], hops are written to ilabel/olabel/graph_cost/acoustic_cost + i*cap, their number
 * to n_hops[i] (0 = the reference's `return false`: no frames decoded or no surviving token).
 * If a path is longer than cap, n_hops[i] is the needed size and WFST_E_CAPACITY is returned. */
int wfst_decoder_get_best_path(wfst_decoder *d, const int32_t *channels, int32_t n,
                               int32_t use_final_probs, int32_t cap, int32_t *ilabel,
                               int32_t *olabel, float *graph_cost, float *acoustic_cost,
                               int32_t *n_hops);

/* GetBestPath of a channel LIST in two halves, for a host that batches many decoder objects over one decoder (the C++ mirror's
 * GpuChannelPool): _enqueue puts the traceback and the copies on the decoder's results stream behind the listed channels' own
 * enqueued work and returns at once; _ready says (never blocks) whether the results have landed; _fetch waits if need be and
 * writes them out -- same outputs, for the enqueued list and capacity, as wfst_decoder_get_best_path, which for a list is the two
 * halves one behind the other.  One request may be outstanding per decoder (a second _enqueue: WFST_E_STATE); the listed channels
 * must not be advanced or initialised in between.  A device error of another channel's utterance does not fail the request. */
int wfst_decoder_best_path_enqueue(wfst_decoder *d, const int32_t *channels, int32_t n, int32_t use_final_probs, int32_t cap);
int wfst_decoder_best_path_ready(wfst_decoder *d);
int wfst_decoder_best_path_fetch(wfst_decoder *d, int32_t *ilabel, int32_t *olabel, float *graph_cost, float *acoustic_cost,
                                 int32_t *n_hops);

/* LatticeToVector (newfst/lattice-functions.cc:179-217) on one hop list: nonzero olabels ->
 * words, nonzero ilabels -> transition-ids, lm = sum graph, tot = sum (graph + acoustic), float
 * accumulation in forward order.  Host-only helper; returns the counts through n_words/n_tids. */
int wfst_lattice_to_vector(const int32_t *ilabel, const int32_t *olabel, const float *graph_cost,
                           const float *acoustic_cost, int32_t n_hops, int32_t *words,
                           int32_t max_words, int32_t *n_words, int32_t *tids, int32_t max_tids,
                           int32_t *n_tids, float *tot_score, float *lm_score);

/* The same for a batch of hop lists laid out as wfst_decoder_get_best_path returns them ([n_paths][cap] arrays, n_hops per
 * path): tot_score / lm_score of every path (the sequential float sums of LatticeToVector) and the number of words and
 * transition-ids in it.  Host-only. */
int wfst_lattice_to_vector_batch(const int32_t *ilabel, const int32_t *olabel, const float *graph_cost,
                                 const float *acoustic_cost, const int32_t *n_hops, int32_t n_paths, int32_t cap,
                                 float *tot_score, float *lm_score, int32_t *n_words, int32_t *n_tids);

/* The label half of LatticeToVector for the same batch layout: the nonzero olabels (words) and ilabels (transition-ids) of
 * every path in hop order, packed path after path -- path p's words are words[word_off[p] .. word_off[p + 1]).  words / tids
 * hold the sums of wfst_lattice_to_vector_batch's n_words / n_tids (at most n_paths * cap), the offset arrays n_paths + 1
 * entries.  Host-only. */
int wfst_lattice_labels_batch(const int32_t *ilabel, const int32_t *olabel, const int32_t *n_hops, int32_t n_paths, int32_t cap,
                              int32_t *words, int32_t *word_off, int32_t *tids, int32_t *tid_off);

/* Per-channel work counters since the last init: {frames, N tokens expanded, E emitting arcs
 * traversed, Z epsilon arcs traversed, tokens kept, peak tokens per frame, candidate records
 * bucketed, forward links recorded (lattice mode) / token collections run (best-path mode)}.  N and E follow the
 * definitions of the reference loop (base-inl.h:311-347). */
int wfst_decoder_get_stats(wfst_decoder *d, int32_t channel, int64_t stats[8]);

/* GetRawLattice(Lattice*, use_final_probs) (base-inl.h:869-975) of a channel of a decoder created in
 * lattice mode (wfst_limits.lattice_links > 0).  After FinalizeDecoding: the state-level lattice that is
 * left after its lattice_beam pruning (base-inl.h:725-847).  MID-UTTERANCE (any time after InitDecoding, as
 * the service asks for it: kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:58,81): everything alive right now
 * -- the history as the last PruneActiveTokens pass left it plus the raw frames since.  One documented
 * deviation there: the running passes judge "extra costs moved by more than lattice_beam * prune_scale" on
 * each frame's exact fixpoint, the reference sweep by sweep over its token list, so a pass may stop walking
 * back at another frame than the reference's and a mid-utterance lattice may hold a few links more or fewer
 * than the reference's at that moment (it equals the order-free oracle's, DESIGN.md section 4 deviation 7);
 * after FinalizeDecoding (delta 0 in the reference too) the lattices are identical.  States are numbered frame
 * by frame in a topological order (every arc goes to a higher id; state 0 is the start), like the
 * reference's TopSortTokens; the numbering inside a frame is implementation defined there too.
 * Per state: final flag, frame, graph state id, forward cost; per arc (sorted by source):
 * source, destination, ilabel, olabel, graph cost, acoustic cost.  If a capacity is too small the
 * needed sizes are returned in n_states/n_arcs with WFST_E_CAPACITY.  n_states == 0 with WFST_OK is
 * the reference's `return false` (no frames decoded, no token alive, or use_final_probs == 0 after
 * FinalizeDecoding). */
int wfst_decoder_get_raw_lattice(wfst_decoder *d, int32_t channel, int32_t use_final_probs,
                                 int32_t cap_states, int32_t cap_arcs, int32_t *n_states,
                                 int32_t *n_arcs, int32_t *st_final, int32_t *st_frame,
                                 int32_t *st_state, float *st_cost, int32_t *a_src, int32_t *a_dst,
                                 int32_t *a_ilabel, int32_t *a_olabel, float *a_graph, float *a_acoustic);

/* GetLattice ahead of its request: sends the finalized channels that have no determinized lattice yet to the determinizer
 * NOW, on a side stream, and returns at once (the determinizer is one lane per lattice and a launch lasts as long as its
 * largest lattice -- tens of milliseconds during which the device is all but idle).  wfst_decoder_get_best_path and
 * wfst_decoder_get_nbest served meanwhile run beside it; the first wfst_decoder_get_determinized_lattice (or batched
 * post-processing call) finds the work done or waits for it.  Results and errors are those of the calls that fetch: this
 * call changes when the work is done, not what it gives (kaldi-online-nnet3-my-decoder.cc:50-105 asks for the best path,
 * the n-best list and the lattice of a finished utterance one after the other).  One launch's worth of channels
 * (wfst_limits.det_workspace_bytes); the rest are determinized on request.  wfst_decoder_init / _advance / _finalize wait for
 * a prefetch in flight. */
int wfst_decoder_prefetch_determinized(wfst_decoder *d);

/* ... DETACHED: for a service that refills its channels at once.  The part of the determinizer that reads the channels' state
 * (a fraction of a millisecond) runs on the decoder's stream; the subset construction runs on a stream of its own, on the
 * determinizer's workspace alone, and wfst_decoder_init / _advance / _finalize do NOT wait for it: the channels decode their next
 * utterances beside it.  The lattices -- those of the utterances the channels had finalized at this call -- are kept per
 * channel and fetched with wfst_decoder_get_prefetched_lattice once harvested (by the next prefetch call, or by
 * wfst_decoder_harvest_prefetched), until the harvest after that; a channel that still holds the very utterance also serves them through
 * wfst_decoder_get_determinized_lattice.  The side stream is one more HIP stream of the process: with the runtime's default of
 * four hardware queues it can end up sharing a queue with a channel group's stream, whose launches then wait behind the
 * determinizer -- export GPU_MAX_HW_QUEUES=8 before the process's first HIP call. */
int wfst_decoder_prefetch_determinized_detached(wfst_decoder *d);
/* GetLattice AND GetNbest ahead of their requests, as the service runs them one behind the other (kaldi-nnet3/kaldi-online-nnet3-
 * my-decoder.cc:97-105: NShortestPath on the lattice DeterminizeLatticeWrapper returned): like wfst_decoder_prefetch_determinized
 * (detached = 0) / _detached (detached != 0), with the n_paths (<= 64) cheapest paths of every lattice computed right behind the
 * determinizer on its stream.  detached = 0: wfst_decoder_get_nbest_paths(channel, n_paths, 1, NULL, NULL, ...) then finds the work
 * done; detached: wfst_decoder_get_prefetched_nbest_paths once harvested (same outputs; WFST_E_STATE where the channel's lattice
 * was not covered or was larger than the prefetch's n-best takes -- 4096 states / 8192 arcs: ask for it alone then). */
int wfst_decoder_prefetch_nbest(wfst_decoder *d, int32_t n_paths, int32_t detached);
int wfst_decoder_get_prefetched_nbest_paths(wfst_decoder *d, int32_t channel, int32_t cap_paths, int32_t cap_arcs, int32_t *n_paths,
                                            int32_t *total_arcs, int32_t *path_off, float *path_tot, int32_t *a_olabel, float *a_graph,
                                            float *a_acoustic);

/* Waits for a prefetch in flight and takes its lattices over -- what the next prefetch (or any other use of the determinizer)
 * does by itself; for the last utterance of a stream of them.  wfst_decoder_get_prefetched_lattice returns the lattices of the
 * last HARVESTED detached prefetch and never waits: right behind the call that started utterance k's determinization it returns
 * utterance k - 1's. */
int wfst_decoder_harvest_prefetched(wfst_decoder *d);
int wfst_decoder_get_prefetched_lattice(wfst_decoder *d, int32_t channel, int32_t cap_states, int32_t cap_arcs,
                                        int32_t *n_states, int32_t *n_arcs, int32_t *st_final, int32_t *a_src, int32_t *a_dst,
                                        int32_t *a_ilabel, int32_t *a_olabel, float *a_graph, float *a_acoustic);

/* GetLattice(Lattice*, use_final_probs) (base-inl.h:850-866) = GetRawLattice + DeterminizeLatticeWrapper
 * (newfst/lattice-determinize-api.cc:5-21: Invert, ArcSort, LatticeDeterminizer::Determinize in the (graph,
 * acoustic) lattice semiring, OutputNoolabel, Invert): the word-level deterministic lattice, built on the
 * device from the raw lattice resident there.  Lattice-mode decoders; finalized channels (the first call
 * determinizes every finalized channel of the batch at once) or mid-utterance.  State 0 is the start; arcs
 * carry ilabel 0 and olabel = word; a final weight is an <eps>:<eps> arc into a final state of its own
 * (OutputNoolabel, lattice-determinize.h:307-377).  Arc for arc (as a multiset: labels and float costs bit for
 * bit) what the reference's determinizer makes of the same raw lattice.  n_states == 0 with WFST_OK: no
 * lattice (as wfst_decoder_get_raw_lattice).  WFST_E_CAPACITY: output larger than the given capacities (sizes
 * returned), or the subset construction outgrew its workspace (the reference's unpruned determinizer has no
 * bound either: DeterminizeLatticeOptions::_max_mem). */
int wfst_decoder_get_determinized_lattice(wfst_decoder *d, int32_t channel, int32_t use_final_probs,
                                          int32_t cap_states, int32_t cap_arcs, int32_t *n_states, int32_t *n_arcs,
                                          int32_t *st_final, int32_t *a_src, int32_t *a_dst, int32_t *a_ilabel,
                                          int32_t *a_olabel, float *a_graph, float *a_acoustic);

/* GetLattice of the service with --use-second (OnlineClgLatticeFastDecoder::GetLattice, kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:
 * 50-78): the determinized lattice composed with the OLD LM (loaded with scale -1: its scores are taken out) and then with the NEW
 * one -- ComposeLattice twice (newfst/compose-lat-inl.h:15-130: pairs (lattice state, LM state) breadth first, a word-labelled arc
 * steps ComposeArpaLm and adds its cost, an arc into a final state adds the LM's final cost and makes the composed state final),
 * each followed by Connect -- on the device, over the determinized lattice resident there and the LM automata in HBM.  Outputs as
 * wfst_decoder_get_determinized_lattice (st_final here marks the composed final states).  One channel per call. */
int wfst_decoder_get_rescored_lattice(wfst_decoder *d, int32_t channel, int32_t use_final_probs, const wfst_lm *old_lm,
                                      const wfst_lm *new_lm, int32_t cap_states, int32_t cap_arcs, int32_t *n_states,
                                      int32_t *n_arcs, int32_t *st_final, int32_t *a_src, int32_t *a_dst, int32_t *a_ilabel,
                                      int32_t *a_olabel, float *a_graph, float *a_acoustic);

/* GetNbest of the service as it is defined (OnlineClgLatticeFastDecoder::GetNbest, kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:97-105):
 * NShortestPath (newfst/lattice-to-nbest.cc:15-147) over the lattice GetLattice returns -- the determinized lattice, or with
 * old_lm / new_lm (both or neither) its second-pass rescoring -- each path with the lattice's own arcs on it, as
 * ConvertNbestToVector (:149-199) hands them out: path i = arcs path_off[i] .. path_off[i + 1] of a_olabel / a_graph /
 * a_acoustic, front to back (word or 0, and both costs, of every arc; the last arc of a path is the final weight's
 * <eps>:<eps> arc), path_tot[i] its cost (arc costs added front to back in float, as NShortestPath adds them), paths in
 * ascending cost.  On the device: a k-best dynamic program over the lattice resident there; any n up to 4096 (the reference has
 * no bound; wfst_decoder_get_nbest above is the batched short-list form on the raw lattice, n <= 16, words and totals only).
 * A determinized lattice has one path per word sequence, so the paths are distinct word sequences.  One channel per call;
 * finalized channels or mid-utterance.  *n_paths == 0 with WFST_OK: no lattice.  WFST_E_CAPACITY: more paths / arcs than the
 * given capacities (the sizes are returned), or n paths over this lattice outgrow the path workspace. */
int wfst_decoder_get_nbest_paths(wfst_decoder *d, int32_t channel, int32_t n, int32_t use_final_probs, const wfst_lm *old_lm,
                                 const wfst_lm *new_lm, int32_t cap_paths, int32_t cap_arcs, int32_t *n_paths, int32_t *total_arcs,
                                 int32_t *path_off, float *path_tot, int32_t *a_olabel, float *a_graph, float *a_acoustic);

/* The service's post-processing as a BATCH.  The reference runs GetLattice (+ the second LM pass under --use-second) and GetNbest per
 * utterance, one worker thread each, concurrently (kaldi-nnet3/kaldi-online-nnet3-my-decoder.cc:50-105, v2-asr/v2-asr-work-thread.h:66);
 * here the listed FINALIZED channels (NULL: every finalized channel) are determinized, composed with the two LMs
 * (newfst/compose-lat-inl.h:15-130, twice) and -- the second call -- searched for their n cheapest paths (newfst/lattice-to-nbest.cc)
 * in ONE launch each, a workgroup per lattice, and the results are fetched once: the per-channel calls
 * wfst_decoder_get_rescored_lattice / wfst_decoder_get_nbest_paths with the same arguments then return them without device work
 * (until the channel is initialised again).  old_lm / new_lm of the second call: both NULL = the paths of the determinized lattice.
 * WFST_E_STATE for a channel that is not finalized (mid-utterance requests: the per-channel calls). */
int wfst_decoder_rescore_lattices(wfst_decoder *d, const int32_t *channels, int32_t n, int32_t use_final_probs, const wfst_lm *old_lm,
                                  const wfst_lm *new_lm);
int wfst_decoder_nbest_paths_batch(wfst_decoder *d, const int32_t *channels, int32_t n, int32_t n_paths, int32_t use_final_probs,
                                   const wfst_lm *old_lm, const wfst_lm *new_lm);

/* The service's n-best (OnlineClgLatticeFastDecoder::GetNbest, kaldi-nnet3/kaldi-online-nnet3-my-
 * decoder.cc:50-105: GetRawLattice -> DeterminizeLatticeWrapper -> NShortestPath ->
 * ConvertNbestToVector, then LatticeToVector per path) of channels of a lattice-mode decoder, finalized
 * or mid-utterance (the service's partial n-best; what is alive now, see wfst_decoder_get_raw_lattice): the n (<= 16) lowest-cost DISTINCT word sequences of the pruned lattice, cheapest first,
 * each with tot_score = sum(graph + acoustic) and lm_score = sum(graph) of its best path.
 * Computed on the device by a k-best search over the raw lattice (no determinized lattice is
 * materialised).  Outputs for the i-th listed channel: n_paths[i]; n_words[i*n + k];
 * words[(i*n + k)*max_words ...] (first max_words ids); tot_score / lm_score [i*n + k].
 * n_paths[i] == 0: no lattice (see wfst_decoder_get_raw_lattice).  WFST_E_CAPACITY if a lattice is
 * larger than the search's working set (32768 .. 262144 states depending on the channel count;
 * the message gives the numbers). */
int wfst_decoder_get_nbest(wfst_decoder *d, const int32_t *channels, int32_t n_channels, int32_t n,
                           int32_t max_words, int32_t *n_paths, int32_t *n_words, int32_t *words,
                           float *tot_score, float *lm_score);

/* Kernel timing for the roofline report: while enabled, every expand / boundary launch of
 * wfst_decoder_advance is bracketed by HIP events recorded on the decoder's own stream.
 * wfst_decoder_get_profile waits for the stream and returns, since the last enable:
 * [0] expand kernel (ProcessEmitting inner loop -> candidate buckets), [1] insert kernel
 * (FindOrAddToken in LDS hash tables -> tokens), [2] closure kernel (ProcessNonemitting fixpoint +
 * GetCutoff + next_cutoff seed).  Leave it off in production runs. */
int wfst_decoder_set_profiling(wfst_decoder *d, int32_t enable);
int wfst_decoder_get_profile(wfst_decoder *d, double ms[3], int64_t launches[3]);
/* The time during which at least one launch of each kernel class was executing (union of the launches'
 * intervals): with several channel groups the launches of different groups overlap, and the sum of their
 * durations (wfst_decoder_get_profile) counts the shared time once per group. */
int wfst_decoder_get_profile_busy(wfst_decoder *d, double busy_ms[3]);
/* Which of the library's kernel paths this decoder runs (for benchmarks and tests: the N > 1 ranks of a sharded run must report
 * what the N = 1 run does): {staged expansion, two launches per frame, frames between two token-collection checks (gc_stride),
 * degree codes in the tokens, log-likelihood row staged in LDS (known after the first advance), best token found by the
 * expansion, the per-frame limit degrades (soft limit), channel groups}. */
int wfst_decoder_get_path_flags(wfst_decoder *d, int32_t flags[8]);
/* The number of channel groups the decoder runs with (wfst_options.channel_groups, resolved). */
int wfst_decoder_channel_groups(wfst_decoder *d);

/* Lattice-mode work counters of a channel since its last init, for the byte model of the back-pruning (bench.py): {forward links
 * recorded, links priced by the PruneActiveTokens / FinalizeDecoding walks (one per link and sweep), tokens priced by them,
 * tokens + links scanned by the compactions, tokens + links the compactions moved}. */
int wfst_decoder_get_lattice_stats(wfst_decoder *d, int32_t channel, int64_t stats[5]);

/* How long the device took to determinize the lattice of `channel` that the decoder holds (the last wfst_decoder_get_determinized_lattice
 * / prefetch of it), in milliseconds of the device's constant clock: the time of that lattice's own workgroup, not of the launch
 * (which lasts as long as its largest lattice).  Beside the reference's DeterminizeLatticeWrapper timed on a host core (bench.py).
 * WFST_E_STATE if no determinized lattice of the channel is held. */
int wfst_decoder_get_determinizer_ms(wfst_decoder *d, int32_t channel, float *ms);

/* Running back-pruning passes (PruneActiveTokens, base-inl.h:439-607) of the channel since its last init whose several-workgroup
 * pricing of the never-priced frames was ABANDONED -- a workgroup waited 40 ms for its siblings (a chip shared with other
 * processes) -- and done over by the one-workgroup walk: the same lattice, later; never an error.  -1: the several-workgroup pass is
 * off on this device (its grid would not be resident at once). */
int wfst_decoder_get_prune_raw_abandoned(wfst_decoder *d, int32_t channel, int32_t *n_passes);

/* Frames of the channel's utterance (since its last init) on which the per-frame token limit BOUND: best-path decoders on
 * the fused graph rows treat wfst_limits.max_tokens_per_frame as a max_active, not as a capacity -- a frame that reaches
 * more distinct states keeps every token (the arena permitting) and the search goes on from the limit-th cheapest, exactly
 * what the reference does at that max_active (GetCutoff, base-inl.h:188-203) where it grows its hash and pools instead
 * (base-inl.h:237-244, util/mem-pool.h:17-65).  0 = the result is the one the caller's config alone defines.  Lattice and
 * biglm decoders report 0 and keep WFST_E_CAPACITY for that limit. */
int wfst_decoder_get_degraded_frames(wfst_decoder *d, int32_t channel, int32_t *n_frames);

/* Frontier of a channel after the last decoded frame (states and costs, unordered); for tests.
 * Returns the number of tokens (may exceed cap; only cap are written). */
int wfst_decoder_get_frontier(wfst_decoder *d, int32_t channel, int32_t cap, int32_t *states,
                              float *costs);

#ifdef __cplusplus
}
#endif
#endif /* WFST_DECODER_H_ */
