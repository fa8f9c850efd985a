// determinize_kernel: the reference's DeterminizeLatticeWrapper (newfst/lattice-determinize-api.cc:5-21) on
// the pruned raw lattices resident on the device (lat_toks[] / lat_arcs[], left there by lattice_emit_kernel),
// one workgroup per channel.  The workgroup builds the inverted, arc-sorted CSR of its lattice together; the
// subset construction itself (wfst_determinize.h: a sequential algorithm by its definition, see there) runs
// on one lane, the lattices of a batch side by side.
//
// What bounds that lane (round 3, measured: tools/ubench_chase.hip and in-kernel timers on the 128 beam-15 lattices of the bench):
// a beam-15 lattice of 2.5-14 k states costs 2.2-2.7 us per string-trie node it creates (8 ms for the smallest, 69 ms for the
// largest of the batch, 17 ms on average -- the launch lasts as long as its largest lattice).  One lane pays 25 ns for a
// dependent LDS load, 55-100 ns for a dependent L1 / L2 load and 2-4 ns per dependent ALU instruction; every load that follows a
// store also waits for that store's acknowledgement (one in-order counter on gfx9).  Tried and dropped: the hot tables (first 8192
// trie nodes + a 16-bit hash, closure buffers, state index) in LDS behind low / high accessors -- slower (mean 18.8 vs 16.9 ms,
// largest 88 vs 69 ms: the accessors' instructions cost more than the loads save -- the closure's queue and element list ALONE in LDS,
// as plain pointers with a rerun in the workspace's buffers on overflow, are kept: -3 %); walking straight stretches of the raw
// lattice without queue / index traffic -- 5 % (and it changes the closure's visiting order).  What did pay: the closure's ring
// buffer without 64-bit modulo (-10 %).  Speeding this up for real needs parallelism INSIDE a lattice.
#include "wfst_determinize_wave.h"   // (wfst_determinize.h + the one-wave-per-lattice construction)
#include "wfst_device.h"

namespace wfst {

constexpr int kDetThreads = 256;

// phase 0: the whole thing.  1: the CSR only -- everything that reads the DECODER's state (the channel's control block, its resolved
// token / link lists, the arena-index scratch) -- leaving {-, -, status, raw states} in the result words and the CSR in the
// workspace; 2: the subset construction from there, which touches the workspace and the outputs alone: it may run on a side
// stream while the channel goes on to its next utterance (wfst_decoder_prefetch_determinized_detached).
__global__ __launch_bounds__(kDetThreads) void determinize_kernel(DecoderDev D, DetDev X, const int32_t *chans, int phase) {
  const int slot = blockIdx.x;
  const int c = chans ? chans[slot] : slot;
  const int tid = threadIdx.x;
  const ChanCtl *ctl = D.ctl + c;
  int32_t *res = X.result + (size_t)slot * 4;   // {states, arcs, status (0 ok, 1 workspace exceeded, 2 lattice too large), -}
  int32_t *base = X.ws + (size_t)slot * X.words_per_channel;   // workspace slots go with the launch's list, not the channel
  int32_t *off = base;                       // [raw_states_cap + 1]
  int nt, na;
  if (phase == 2) {
    nt = res[3];               // (phase 1 left the raw lattice's size here; 0: nothing to do, the result words say why)
    if (nt <= 0) return;
    na = off[nt];
  } else {
    nt = ctl->lat_toks;
    na = ctl->lat_arcs;
    if (tid == 0) { res[0] = 0; res[1] = 0; res[2] = 0; res[3] = 0; }
    if (ctl->error || nt <= 0 || ctl->n_decoded <= 0) return;
    if (nt > X.raw_states_cap || na > X.raw_arcs_cap) {
      if (tid == 0) res[2] = 2;
      return;
    }
  }
  const int4 *toks = D.lat_toks + (size_t)c * D.lat_tok_cap;
  const LatArc *larcs = D.lat_arcs + (size_t)c * D.lat_arc_cap;
  int32_t *state_of = D.remap + (size_t)c * D.arena_cap;   // arena index -> lattice state (scratch between pruning passes)
  int32_t *fin = off + X.raw_states_cap + 1; // [raw_states_cap]
  int32_t *cur = fin + X.raw_states_cap;     // [raw_states_cap]
  DetArc *arcs = reinterpret_cast<DetArc *>(cur + X.raw_states_cap);  // [raw_arcs_cap]
  int32_t *rest = reinterpret_cast<int32_t *>(arcs + X.raw_arcs_cap);
  __shared__ int s_part[kDetThreads];

  // ---- Invert + CSR + ArcSort (lattice-determinize-api.cc:8-11), workgroup-wide -------------------------
  if (phase != 2) {
  for (int i = tid; i < nt; i += kDetThreads) {
    const int4 t = toks[i];
    state_of[t.x] = i;
    fin[i] = (t.w >> 30) & 1;
    off[i] = 0;
  }
  if (tid == 0) off[nt] = 0;
  __syncthreads();
  for (int a = tid; a < na; a += kDetThreads) atomicAdd(&off[state_of[larcs[a].src_tok]], 1);
  __syncthreads();
  {  // exclusive scan of off[0..nt) (one contiguous slice per thread)
    const int per = (nt + kDetThreads - 1) / kDetThreads, b = tid * per, e = min(nt, b + per);
    int sum = 0;
    for (int i = b; i < e; ++i) sum += off[i];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int i = 0; i < kDetThreads; ++i) { const int v = s_part[i]; s_part[i] = run; run += v; }
    }
    __syncthreads();
    int run = s_part[tid];
    for (int i = b; i < e; ++i) { const int v = off[i]; off[i] = run; cur[i] = run; run += v; }
    if (tid == 0) off[nt] = na;
  }
  __syncthreads();
  for (int a = tid; a < na; a += kDetThreads) {
    const LatArc A = larcs[a];
    DetArc d;
    d.ilabel = A.olabel;   // Invert: the word is the input label now
    d.olabel = A.ilabel;
    d.w1 = A.graph;
    d.w2 = A.acoustic;
    d.to = state_of[A.dst_tok];
    arcs[atomicAdd(&cur[state_of[A.src_tok]], 1)] = d;
  }
  __syncthreads();
  for (int s = tid; s < nt; s += kDetThreads) {  // ArcSort: by input label (a state has a handful of arcs)
    const int b = off[s], e = off[s + 1];
    for (int i = b + 1; i < e; ++i) {
      const DetArc x = arcs[i];
      int j = i;
      // ties by (destination, costs): any fixed order will do -- the result does not depend on it (wfst_determinize.h)
      auto after = [](const DetArc &p, const DetArc &q) {  // p sorts after q
        if (p.ilabel != q.ilabel) return p.ilabel > q.ilabel;
        if (p.to != q.to) return p.to > q.to;
        if (p.olabel != q.olabel) return p.olabel > q.olabel;
        return p.w1 > q.w1;
      };
      while (j > b && after(arcs[j - 1], x)) {
        arcs[j] = arcs[j - 1];
        --j;
      }
      arcs[j] = x;
    }
  }
  __syncthreads();
  // the root token (arena entry 0) must be state 0: lat_toks is in arena order, so it is
  if (phase == 1) {
    if (tid == 0) res[3] = nt;   // (the CSR is complete: off[nt] = na)
    return;
  }
  }
  // ---- the subset construction: tables cleared by everyone, then one lane ---------------------------------
  __shared__ DetWs W;
  __shared__ DwShared S;   // the closure's queue, element list and state index (wfst_determinize_wave.h)
  if (tid == 0) {
    W.n_states = nt;
    W.n_arcs = na;
    W.off = off;
    W.arcs = arcs;
    W.is_final = fin;
    W.delta = 1.0f / 1024;   // kDelta (DeterminizeLatticeOptions, lattice-determinize-api.h:16-25)
    det_carve(W, rest, X.caps, nt);
  }
  for (int i = tid; i < kDwMap; i += kDetThreads) S.map[i] = 0u;
  __syncthreads();
  // The string trie's hash table is carved for the workspace's full capacity (a million slots, 4 MB); a lattice of a few thousand
  // states makes 1.5-2 nodes per raw state, so the table is first used at 16 slots per raw state (a few hundred KB: its probes
  // stay in L2) and the construction is run again over the whole table in the rare case that it outgrows that.
  __shared__ int s_err;
  const int32_t hcap_full = W.tr_hcap;
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (tid == 0) {
      int32_t h = hcap_full;
      if (attempt == 0) {
        h = 4096;
        while (h < 16 * nt && h < hcap_full) h <<= 1;
      }
      W.tr_hcap = h < hcap_full ? h : hcap_full;
    }
    __syncthreads();
    det_init(W, tid, kDetThreads);
    __syncthreads();
    {
      const int e = detw_run(W, S, nullptr);   // wave 0 runs the construction, the other waves go on to the barrier
      if (tid == 0) s_err = e;
    }
    __syncthreads();
    if (!(s_err == 1 && W.tr_hcap < hcap_full)) break;   // (1: the trie -- its node array or its hash table -- was outgrown)
  }
  if (tid == 0) {
    const int err = s_err;
    // OutputNoolabel (lattice-determinize.h:307-377) + Invert: arcs {src, dst, 0, word, graph, acoustic}; a final weight is
    // an arc <eps>:<eps> to an extra final state
    int4 *oa = X.out_a + (size_t)slot * X.out_cap;
    float2 *ow = X.out_w + (size_t)slot * X.out_cap;
    int ns = W.os_n, no = 0, over = 0;
    for (int i = 0; i < W.oa_n; ++i) {
      const DetOutArc t = W.oarcs[i];
      int dst = t.next;
      if (t.next < 0) dst = ns++;
      if (no < X.out_cap) {
        oa[no] = make_int4(t.src, dst, t.next < 0 ? 0 : t.ilabel, t.next < 0 ? 1 : 0);
        ow[no] = make_float2(t.w1, t.w2);
      } else over = 1;
      ++no;
    }
    res[0] = ns;
    res[1] = no;
    res[2] = (err || over) ? 1 : 0;
    res[3] = W.os_n;   // states below this are the determinized states proper; the rest are the final states
  }
}

// The determinized lattices of workspace slots [0, cnt) packed back to back (slot i's arcs at the sum of the arc counts of the
// slots before it; a slot that failed contributes none): the host fetches a batch's lattices with two copies instead of two per
// lattice.  One workgroup per slot.
__global__ __launch_bounds__(256) void det_pack_kernel(DetDev X, int cnt, int4 *pack_a, float2 *pack_w, int64_t pack_cap) {
  const int slot = blockIdx.x;
  int64_t off = 0;
  for (int j = 0; j < slot; ++j) off += X.result[4 * j + 2] ? 0 : min(X.result[4 * j + 1], X.out_cap);
  const int n = X.result[4 * slot + 2] ? 0 : min(X.result[4 * slot + 1], X.out_cap);
  if (off + n > pack_cap) return;   // (the host sized the buffers from the same counts: not expected)
  const int4 *a = X.out_a + (size_t)slot * X.out_cap;
  const float2 *w = X.out_w + (size_t)slot * X.out_cap;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { pack_a[off + i] = a[i]; pack_w[off + i] = w[i]; }
}
void launch_det_pack(const DetDev &X, int cnt, int4 *pack_a, float2 *pack_w, int64_t pack_cap, hipStream_t s) {
  hipLaunchKernelGGL(det_pack_kernel, dim3(cnt), dim3(256), 0, s, X, cnt, pack_a, pack_w, pack_cap);
}

void launch_determinize(const DecoderDev &D, const DetDev &X, const int32_t *chans, int cnt, hipStream_t s, int phase) {
  hipLaunchKernelGGL(determinize_kernel, dim3(cnt), dim3(kDetThreads), 0, s, D, X, chans, phase);
}

}  // namespace wfst
